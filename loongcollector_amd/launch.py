"""One process per GPU: the rank launcher behind `bench.py --gpus N` when no launcher (torch.distributed.run) has set the rank
environment, and the check that the world the process finds is the world it was asked for.

The path is an embarrassingly parallel line shard: ranks share nothing but the final all-gather of their
counters, so a launch that silently degrades to one rank would still print a plausible line.  It must not:
  * WORLD_SIZE set (a launcher started us)  -> it has to equal --gpus, else the process exits with an error;
  * WORLD_SIZE unset and --gpus N > 1       -> this process becomes the launcher: it checks that N devices are visible, starts N
    copies of its own command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, waits for all of
    them and exits with the first non-zero status (the other ranks are terminated by PID).
"""
import os
import socket
import subprocess
import sys
import time


def _free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def ensure_ranks(gpus, need_devices=True, argv=None, device_count=None):
    """Returns when this process is a rank of a world of `gpus` ranks (or the only process, gpus == 1).  Otherwise it launches the
    ranks and exits with their status -- it never returns in the launcher."""
    if gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != gpus:
            raise SystemExit("--gpus %d but the launcher's WORLD_SIZE is %s: refusing to run a different world than asked for"
                             % (gpus, env_world))
        if "RANK" not in os.environ or "LOCAL_RANK" not in os.environ:
            raise SystemExit("WORLD_SIZE is set but RANK / LOCAL_RANK are not: incomplete launcher environment")
        return
    if gpus == 1:
        return
    if need_devices:
        if device_count is None:
            import torch
            device_count = torch.cuda.device_count()
        if device_count < gpus:
            raise SystemExit("--gpus %d but only %d HIP device(s) are visible" % (gpus, device_count))
    argv = list(sys.argv if argv is None else argv)
    port = _free_port()
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_WORLD_SIZE=str(gpus))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable] + argv, env=env))
    status = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0 and status == 0:
                status = rc
                for q in alive:  # a rank failed: the others would wait in a collective for ever
                    q.terminate()
        time.sleep(0.05)
    sys.exit(status if status >= 0 else 1)
