"""`bench.py --gpus N` must mean N ranks or fail: the launch path (loongcollector_amd/launch.py) on a box without a GPU.
The ranks rendezvous over gloo on 127.0.0.1 and run the job's only collective (shard.gather_job); no device work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def _run(argv, env, timeout=300):
    return subprocess.run([sys.executable, BENCH] + argv, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_gpus_2_without_a_launcher_starts_two_ranks():
    out = _run(["--gpus", "2", "--steps", "7", "--warmup", "2", "--launch-check"], _clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                          # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 7 and d["warmup"] == 2
    assert [g["rank"] for g in d["per_gpu"]] == [0, 1] and [g["local_rank"] for g in d["per_gpu"]] == [0, 1]


def test_gpus_1_is_one_process():
    out = _run(["--gpus", "1", "--launch-check"], _clean_env())
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and len(d["per_gpu"]) == 1


def test_world_size_of_the_launcher_must_equal_gpus():
    out = _run(["--gpus", "8", "--launch-check"], _clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29417"))
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr and not out.stdout.strip()


def test_more_gpus_than_devices_fails_loudly():
    # (this container has no HIP device; on the GPU box: one) -- asking for 64 ranks of the real bench must not print a line
    out = _run(["--gpus", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e", "--no-configs"], _clean_env())
    assert out.returncode != 0 and "HIP device" in out.stderr and not out.stdout.strip()


def test_under_torch_distributed_run():
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29431", BENCH, "--gpus", "2", "--launch-check"],
                         env=_clean_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_bounded_window_in_agent_leg_round_logic_without_a_device():
    """bench.py measure_in_agent_window (the in-agent leg with a bounded number of groups alive): rounds, barriers, accounting and
    the failure path, driven with stand-ins for the ctypes wrappers -- every group is processed exactly once by exactly one thread, the
    first round is not timed, a failing process() call surfaces instead of hanging the barrier."""
    import threading
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench

    class Group:
        alive = 0
        peak = 0
        lock = threading.Lock()

        def __init__(self, lines):
            self.lines, self.done, self.closed = lines, 0, False
            with Group.lock:
                Group.alive += 1
                Group.peak = max(Group.peak, Group.alive)

        @classmethod
        def from_lines(cls, data, off, length):
            return cls(len(off))

        def contents(self):
            return [[("k", "v")]]

        def close(self):
            assert not self.closed and self.done == 1
            self.closed = True
            with Group.lock:
                Group.alive -= 1

    class Proc:
        fail_after = None

        def __init__(self, cfg):
            self.n = 0
            self.lock = threading.Lock()

        def process(self, g):
            with self.lock:
                self.n += g.lines
                g.done += 1
                if Proc.fail_after is not None and self.n > Proc.fail_after:
                    raise RuntimeError("device lost")

        def counters(self):
            return {"out_successful_events_total": self.n, "out_failed_events_total": 0}

    n_groups, group_lines = 70, 10
    length = np.full(n_groups * group_lines, 512, dtype=np.uint32)
    off = np.arange(n_groups * group_lines, dtype=np.uint32) * 513
    for threads in (1, 4):
        Group.alive = Group.peak = 0
        mbps, first = bench.measure_in_agent_window(b"x", ["k"], None, off, length, group_lines, n_groups, threads, window=4, _types=(Group, Proc))
        assert mbps > 0 and first == [("k", "v")]
        assert Group.alive == 0 and Group.peak <= 4 * threads + threads   # a window per thread (+ the warm-up groups), never the whole run
    Proc.fail_after = 300
    try:
        bench.measure_in_agent_window(b"x", ["k"], None, off, length, group_lines, n_groups, 2, window=4, _types=(Group, Proc))
        raise AssertionError("the failing call was swallowed")
    except RuntimeError as e:
        assert "device lost" in str(e)
    finally:
        Proc.fail_after = None
