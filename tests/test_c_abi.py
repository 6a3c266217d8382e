"""The C-ABI shared library loads and exports every symbol include/*.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from loongcollector_amd import binding as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    syms = set()
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(inc, fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b(lc_[a-z0-9_]+)\s*\(", text):
            syms.add(m.group(1))
        for m in re.finditer(r"^[ \t]*extern[ \t]+(?!\"C\")[^;(\n]*?\b(\w+)[ \t]*;", text, flags=re.M):
            syms.add(m.group(1))
    return syms


def test_library_exports_every_declared_symbol():
    lib = B.load()   # (maps the HIP runtime torch ships first, see binding._share_torch_hip_runtime)
    syms = _declared_symbols()
    assert {"lc_regex_compile", "lc_regex_match_device", "lc_regex_match_host"} <= syms
    missing = [s for s in sorted(syms) if not hasattr(lib, s)]
    assert not missing, missing


def test_match_calls_fail_loudly_without_a_device():
    if B.device_count() > 0:
        pytest.skip("a HIP device is present")
    rx = B.GpuRegex(r"(\w+)\t(\w+).*")
    data = np.frombuffer(b"a\tb", dtype=np.uint8)
    with pytest.raises(B.GpuUnavailableError):
        rx.match_host(data, np.array([0], np.uint32), np.array([3], np.uint32))


def test_thread_to_device_rule_and_policy_calls():
    """runner thread -> GPU as threadNo % nGPU (core/runner/ProcessorRunner.h:40 is the index the reference
    uses for its per-thread regex copies).  The rule itself is a pure function; the policy calls validate their arguments and, on
    a box without a device, an explicit binding fails loudly instead of pretending."""
    for n in (1, 2, 4, 8):
        got = [B.device_for_ordinal(t, n) for t in range(3 * n)]
        assert got == [t % n for t in range(3 * n)]
        # every device gets the same number of the first k*n threads
        assert sorted(got) == sorted(list(range(n)) * 3)
    assert B.device_for_ordinal(5, 0) == -1 and B.device_for_ordinal(5, -3) == -1
    assert B.device_for_ordinal(2 ** 32 - 1, 8) == (2 ** 32 - 1) % 8
    L = B.load()
    assert L.lc_runtime_set_bind_policy(7, 0) == B.LC_ERR_ARG
    assert L.lc_runtime_set_bind_policy(B.LC_BIND_FIXED, -1) == B.LC_ERR_ARG
    import subprocess
    import sys
    code = r"""
import threading
from loongcollector_amd import binding as B
assert B.bind_policy() == B.LC_BIND_ROUND_ROBIN          # the library's default
assert B.thread_device() == -1                           # nobody is bound before a host entry
B.set_bind_policy(B.LC_BIND_FIXED, 0)
assert B.bind_policy() == B.LC_BIND_FIXED
n = B.device_count()
L = B.load()
if n == 0:
    assert L.lc_runtime_set_thread_device(0) == B.LC_ERR_NO_DEVICE
    assert L.lc_runtime_bind_thread(-1) == -B.LC_ERR_NO_DEVICE
else:
    assert L.lc_runtime_set_thread_device(n) == B.LC_ERR_ARG
    seen = []
    def run():
        seen.append((B.bind_thread(), B.thread_device()))
    ts = [threading.Thread(target=run) for _ in range(4)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert seen == [(0, 0)] * 4
print("ok")
"""
    env = dict(os.environ)
    env.pop("LC_BIND_POLICY", None)
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
    env["LC_BIND_POLICY"] = "inherit"
    out = subprocess.run([sys.executable, "-c", "from loongcollector_amd import binding as B; print(B.bind_policy())"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.stdout.strip() == str(B.LC_BIND_INHERIT), out.stdout + out.stderr


def test_info_reports_engine_and_sizes():
    info = B.GpuRegex(r"(\w+)\t(\w+).*").info()
    assert info["engine"] == B.LC_ENGINE_TDFA and info["mark_count"] == 2 and info["states"] > 2


def test_config_json_nesting_is_bounded():
    """jsoncpp (the reference's parser) stops at 1000 levels; the plugins' JSON reader must not run out of stack either"""
    from loongcollector_amd.multiline import Multiline
    L = B.load()
    for deep in (b'{"a":' + b"[" * 200000 + b"]" * 200000 + b"}", b'{"a":' * 100000 + b"1" + b"}" * 100000):
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(256)
        Multiline(StartPattern="x")  # binds the argtypes
        assert L.lc_multiline_create(deep, len(deep), ctypes.byref(h), err, 256) != 0
        assert b"nesting too deep" in err.value


def test_loading_the_library_leaves_the_process_environment_alone():
    """Round 3 shipped a constructor that set GPU_MAX_HW_QUEUES at load time (and a second setenv in the Grok processor's Init that
    could then never take effect).  The variable is process-wide and belongs to the host: a fresh interpreter loads the library,
    creates a Grok processor and a parse processor, and must see its environment unchanged; the explicit request
    (lc_runtime_prefer_hw_queues) sets it once and keeps a value that is already there."""
    import subprocess
    import sys
    code = r"""
import os, ctypes, json
os.environ.pop("GPU_MAX_HW_QUEUES", None)
before = dict(os.environ)
libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; libc.getenv.argtypes = [ctypes.c_char_p]
from loongcollector_amd import binding as B, grok
L = B.load()
assert libc.getenv(b"GPU_MAX_HW_QUEUES") is None, "library load changed the environment"
g = grok.Grok(Match=["%{WORD:w} %{NUMBER:n}"], SourceKey="content")
rx = B.GpuRegex(r"(\w+) (\d+)")
assert libc.getenv(b"GPU_MAX_HW_QUEUES") is None, "creating processors changed the environment"
L.lc_runtime_prefer_hw_queues.restype = ctypes.c_int; L.lc_runtime_prefer_hw_queues.argtypes = [ctypes.c_int]
assert L.lc_runtime_prefer_hw_queues(0) != 0 and L.lc_runtime_prefer_hw_queues(1000) != 0
assert libc.getenv(b"GPU_MAX_HW_QUEUES") is None
assert L.lc_runtime_prefer_hw_queues(16) == 0 and libc.getenv(b"GPU_MAX_HW_QUEUES") == b"16"
assert L.lc_runtime_prefer_hw_queues(8) == 0 and libc.getenv(b"GPU_MAX_HW_QUEUES") == b"16"   # an existing value is kept
print("ok")
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         cwd=os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.stdout[-500:], out.stderr[-1500:])
