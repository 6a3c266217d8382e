"""Thread -> device placement on the device (include/lc_regex_gpu.h lc_runtime_*): threads that enter the
processor are bound by the policy, results do not depend on the binding, and a device pointer of another GPU is refused.  The box the
driver tests on has ONE GPU: the multi-device branches run where more are visible and say so when they cannot."""
import subprocess
import sys
import os
import threading

import numpy as np
import pytest

from loongcollector_amd import binding as B, corpus
from loongcollector_amd.processor import EventGroup, Processor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _group(n=300, seed=5):
    data, off, length = corpus.apache_batch(n, "A", seed=seed, poison_every=11)
    raw = data.tobytes()
    return {"events": [{"contents": {"content": raw[int(off[i]):int(off[i]) + int(length[i])].decode("latin-1")}} for i in range(n)]}


def _cfg():
    return {"SourceKey": "content", "Regex": corpus.REGEX_A,
            "Keys": ["ip", "time", "method", "url", "request_time", "request_length", "status", "length", "ref_url", "browser"],
            "KeepingSourceWhenParseFail": True}


def test_results_do_not_depend_on_the_binding():
    """the same group through the same processor from: a thread bound by the default policy, a thread pinned explicitly to every
    visible device, a thread that asked for LC_BIND_INHERIT -- identical contents and counters' deltas"""
    ndev = B.device_count()
    assert ndev >= 1
    p = Processor(_cfg())
    spec = _group()
    results = {}

    def run(tag, setup):
        setup()
        g = EventGroup(spec)
        p.process(g)
        results[tag] = (g.contents(), B.thread_device())

    jobs = [("policy", lambda: None), ("inherit", lambda: B.bind_thread(B.LC_BIND_INHERIT))]
    jobs += [("dev%d" % d, (lambda d=d: B.set_thread_device(d))) for d in range(ndev)]
    for tag, setup in jobs:
        t = threading.Thread(target=run, args=(tag, setup))
        t.start()
        t.join()
    assert set(results) == {tag for tag, _ in jobs}
    want = results["policy"][0]
    assert sum(1 for ev in want if any(k == "ip" for k, _ in ev)) > 200
    for tag, (got, dev) in results.items():
        assert got == want, tag
        if tag.startswith("dev"):
            assert dev == int(tag[3:])
    assert 0 <= results["policy"][1] < ndev
    assert results["inherit"][1] == -1


def test_default_policy_deals_threads_round_robin():
    """a fresh process, default policy: the k-th thread that enters a host entry point lands on device k % ndev"""
    code = r"""
import threading, numpy as np
from loongcollector_amd import binding as B
n = B.device_count()
rx = B.GpuRegex(r"(\w+)\t(\w+).*")
data = np.frombuffer(b"a\tb", dtype=np.uint8)
devs = []
def run():
    caps, status = rx.match_host(data, np.array([0], np.uint32), np.array([3], np.uint32))
    assert status[0] == 1 and list(caps[0]) == [0, 1, 2, 3]
    devs.append(B.thread_device())
for k in range(2 * n + 1):
    t = threading.Thread(target=run); t.start(); t.join()
assert devs == [k % n for k in range(2 * n + 1)], devs
print("ok", n)
"""
    env = dict(os.environ)
    env.pop("LC_BIND_POLICY", None)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_a_pointer_of_another_device_is_refused():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible GPU: no other device to take a pointer from")
    rx = B.GpuRegex(corpus.REGEX_A)
    n = 64
    data, off, length = corpus.apache_batch(n, "A")
    with torch.cuda.device(1):
        d_data = torch.from_numpy(data).to("cuda:1")
    torch.cuda.set_device(0)
    d_off = torch.from_numpy(off.astype(np.int32)).to("cuda:0")
    d_caps = torch.empty((n, 2 * rx.groups), dtype=torch.int32, device="cuda:0")
    d_status = torch.empty((n,), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(RuntimeError, match="belongs to device 1"):
        rx.match_device(d_data, d_off, None, n, d_caps, d_status, sep_bytes=1)
