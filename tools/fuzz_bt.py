#!/usr/bin/env python3
"""tools/fuzz_bt.py SEED N -- generated patterns (back-references, look-arounds of both directions, conditionals, atomic groups, greedy and
lazy repeats, nested) compiled for the device backtracking engine (LC_ENGINE_BT) as full matches and as searches; every program is walked
by the kernel's own routine compiled for the host (csrc/bt_vm.hpp btRun through tests/native/bt_host_check.cpp -- build it by running
tests/test_backref.py once) over short subjects and compared with the oracle (oracle/bt_regex.c): result and every capture offset.
The oracle is the checker; nothing here is a product path.  tests/test_backref.py runs a bounded round of it (seed 1, 400 patterns);
the round's runs: seeds 1-9, 450 000 host checks, and 3.96 million checks through the kernel on the MI355X, no difference
(profiles/round6_bt_fuzz_gpu.txt)."""
import ctypes, os, random, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loongcollector_amd import binding as B
from oracle.oracle import OracleRegex
L = ctypes.CDLL(os.path.join(ROOT, 'tests', '_build', 'libbt_host_check.so'))
L.bt_host_run.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
def gen(rng, groups, d=0):
    r = rng.random()
    if r < 0.28 or d > 3: return rng.choice(['a', 'b', 'c', '[ab]', '[^a]', '.', r'\d', r'\w', ' '])
    if r < 0.36 and groups[0] > 0: return '\\%d' % rng.randint(1, groups[0])
    if r < 0.50: return gen(rng, groups, d + 1) + gen(rng, groups, d + 1)
    if r < 0.58: return '(?:' + gen(rng, groups, d + 1) + '|' + gen(rng, groups, d + 1) + ')'
    if r < 0.70:
        inner = gen(rng, groups, d + 1); groups[0] += 1; return '(' + inner + ')'
    if r < 0.74: return rng.choice([r'\b', '^', '$', r'\B'])
    if r < 0.80: return '(?' + rng.choice(['=', '!']) + gen(rng, groups, d + 1) + ')'
    if r < 0.84: return '(?<' + rng.choice(['=', '!']) + rng.choice(['a', 'ab', '[ab]c', r'\d\d', 'a|b', 'ab|cd']) + ')'
    if r < 0.87: return '(?>' + gen(rng, groups, d + 1) + ')'
    if r < 0.90 and groups[0] > 0: return '(?(%d)%s|%s)' % (rng.randint(1, groups[0]), gen(rng, groups, d + 1), gen(rng, groups, d + 1))
    q = rng.choice(['*', '+', '?', '{1,2}', '*?', '+?', '??', '{2}', '{0,3}?'])
    return '(?:' + gen(rng, groups, d + 1) + ')' + q
def main(seed, n):
    rng = random.Random(seed); bad = []; checked = comp = gave = 0
    for _ in range(n):
        groups = [0]; p = gen(rng, groups).encode()
        for flags, search in ((0, False), (B.LC_SYNTAX_SEARCH, True)):
            try: o = OracleRegex(p)
            except ValueError: continue
            try: rx = B.GpuRegex(p, syntax_flags=flags, engine=B.LC_ENGINE_BT)
            except (B.RegexUnsupportedError, B.RegexSyntaxError): continue
            comp += 1
            blob = rx.table(B.LC_TABLE_BT_BLOB, np.uint32); nc = int(blob[2])
            for _ in range(6):
                s = bytes(rng.choice(b'abc1 ') for _ in range(rng.randint(0, 9)))
                caps = np.full(nc, -9, np.int32)
                r = L.bt_host_run(blob.ctypes.data, s, len(s), 0, caps.ctypes.data, nc, 16384, 1 << 20)
                try: w = o.search(s) if search else o.fullmatch(s)
                except Exception: continue
                checked += 1
                if r < 0: gave += 1; continue
                wf = None if w is None else [v for ab in w for v in ab]
                got = None if r == 0 else (list(caps[2:]) if search else list(caps[2:]))
                exp = None if wf is None else (wf if search else wf[2:])
                if got != exp: bad.append((p, search, s, got, exp))
    return checked, comp, gave, bad
if __name__ == '__main__':
    c, comp, gave, bad = main(int(sys.argv[1]), int(sys.argv[2]))
    print('checked', c, 'compiled', comp, 'gave up', gave, 'bad', len(bad))
    for b in bad[:12]: print(b)
