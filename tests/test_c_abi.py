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
