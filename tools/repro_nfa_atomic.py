#!/usr/bin/env python3
"""tools/repro_nfa_atomic.py -- the one difference the forced-engine device fuzz of round 6 found (profiles/round6_bt_fuzz_gpu.txt, D):
'(?>(?:(c)|(?:a)+?).)' as a full match on 'aa1' with the thread-list engine FORCED (LC_ENGINE_NFA).  Prints device against oracle for the
value alone, repeated, and among neighbours.  The oracle is the checker."""
import os, sys, random, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from loongcollector_amd import binding as B
from oracle.oracle import OracleRegex
from test_gpu_parity import pack, run_device
pats = [b'(?>(?:(c)|(?:a)+?).)', b'(?>a+?.)', b'(?>(?:c|a+?).)', b'(?>(?:(c)|a+?).)']
rng = random.Random(5)
sets = {"alone": [b'aa1'], "x64": [b'aa1'] * 64, "short": [b'aa1', b'aa', b'a1', b'aaa', b'', b'a', b'ca', b'c1', b'aaaa1'],
        "mixed": [bytes(rng.choice(b'abc1 ') for _ in range(rng.randint(0, 40))) for _ in range(47)] + [b'aa1']}
for p in pats:
    o = OracleRegex(p)
    for eng in (B.LC_ENGINE_NFA, B.LC_ENGINE_AUTO, B.LC_ENGINE_TDFA, B.LC_ENGINE_BT):
        try: rx = B.GpuRegex(p, engine=eng)
        except Exception as e:
            print(p, eng, "refused", e); continue
        for name, subs in sets.items():
            data, off, length = pack(subs)
            caps, status = run_device(torch, rx, data, off, length)
            bad = []
            for i, s in enumerate(subs):
                w = o.fullmatch(s)
                exp = B.LC_NOMATCH if w is None else B.LC_MATCH
                if int(status[i]) != exp: bad.append((s, int(status[i]), exp))
            print(p, "engine", eng, name, "bad", bad[:6], flush=True)
