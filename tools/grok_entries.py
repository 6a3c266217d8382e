#!/usr/bin/env python3
"""How each Match entry of configs[2] is run (engines, automaton sizes, where their tables live).  CPU only (compiles the list)."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loongcollector_amd import binding  # noqa: E402
from loongcollector_amd.grok import Grok  # noqa: E402

cfg = json.load(open(os.path.join(ROOT, "tests", "golden", "grok_config3.json"), encoding="utf-8"))
t0 = time.time()
g = Grok(Match=cfg["match"], CustomPatterns=cfg["custom_patterns"])
t1 = time.time()
g.wait_ready()
print("# create %.1f s, anchored warm-up %.1f s" % (t1 - t0, time.time() - t1))
L = binding.load()
L.lc_grok_entry_info.restype = ctypes.c_int
L.lc_grok_entry_info.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32)]
print("%2s %-46s %-4s %7s %3s %6s %6s | %3s %7s %3s %9s %4s %4s" % ("i", "entry", "eng", "states", "mem", "pscr", "rscr", "anc", "states", "mem", "bytes", "regs", "cls"))
where = {0: "-", 1: "LDS", 2: "L2", 3: "NFA"}
for i, m in enumerate(cfg["match"]):
    o = (ctypes.c_uint32 * 12)()
    assert L.lc_grok_entry_info(g._h, i, o) == 0
    print("%2d %-46s %-4s %7d %3s %6d %6d | %3d %7d %3s %9d %4d %4d" % (i, m[:46], {1: "tdfa", 2: "nfa"}[o[0]], o[1], where[o[2]], o[3], o[4], o[5], o[6], where[o[7]], o[8], o[9], o[10]))
