#!/usr/bin/env python3
"""Host share of one in-agent group (gather + stitch + policy + compaction + size sums; the match call answered from a table computed
beforehand), on this machine's CPU, for both event models the host code can be built against:
  standin    csrc/event_model.hpp (ArenaVector, chunk pool, one-call stitch)             -- what the standalone library ships
  refshape   csrc/event_model.hpp built with LC_REFERENCE_SHAPED_EVENT_MODEL (std::vector contents, no pool, per-key stitch)
  reference  the reference's own core/models/*.cpp compiled from /root/reference (oracle/_ref) -- the per-key stitch an agent build takes
Uses tests/native/host_double.cpp (test infrastructure: the five device calls answered by the oracle).  Usage: stitch_bench.py [groups] [repeats]"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    groups = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    import test_processor_host_double as T
    from loongcollector_amd import corpus
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    data, off, length = corpus.apache_batch(1000, "A")
    data = np.ascontiguousarray(data)
    off = np.ascontiguousarray(off, dtype=np.uint32)
    length = np.ascontiguousarray(length, dtype=np.uint32)
    res = {}
    for name, build in (("standin", T._build_standin), ("refshape", lambda d: T._build_standin(d, refshape=True)), ("reference", T._build_reference)):
        if name == "reference" and not os.path.isdir(T.REF):
            continue
        L = ctypes.CDLL(build(out_dir))
        vp, cp = ctypes.c_void_p, ctypes.c_char_p
        L.hd_create.restype = vp
        L.hd_create.argtypes = [cp, cp, ctypes.c_size_t]
        L.hd_bench_stitch.restype = ctypes.c_double
        L.hd_bench_stitch.argtypes = [vp, vp, vp, vp, ctypes.c_uint32, ctypes.c_uint32, cp, ctypes.c_uint32,
                                      ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.hd_last_minor_faults_per_group.restype = ctypes.c_double
        err = ctypes.create_string_buffer(512)
        h = L.hd_create(json.dumps({"SourceKey": "content", "Regex": corpus.REGEX_A, "Keys": corpus.KEYS_A}).encode(), err, 512)
        assert h, err.value
        build_us, size_us = ctypes.c_double(), ctypes.c_double()
        us = L.hd_bench_stitch(h, data.ctypes.data, off.ctypes.data, length.ctypes.data, 1000, groups, b"content", repeats,
                               ctypes.byref(build_us), ctypes.byref(size_us))
        res[name] = {"host_us_per_1000_event_group": round(us, 1), "group_build_us": round(build_us.value, 1),
                     "one_DataSize_walk_us": round(size_us.value, 1), "minor_faults_per_group": round(L.hd_last_minor_faults_per_group(), 1)}
    print(json.dumps({"what": "host share of lc_processor_process per 1000-event group, no device in the way", "groups_alive": groups,
                      "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t") if os.path.exists("/proc/cpuinfo") else "?",
                      **res}))


if __name__ == "__main__":
    main()
