#!/usr/bin/env python3
"""Longer differential fuzz of the host compilers than the CPU suite affords: fresh random patterns (plain and atomic / possessive /
look-around ones) in full-match, search and anchored-search mode; the logical tagged-DFA tables and the packed blobs (LDS kernels or the
global-memory kernel), i.e. after dead-store elimination, state minimisation and the pack-time fold, against the oracle.
    python tools/fuzz_tables.py FIRST_SEED LAST_SEED      (40 seeds: 14 000 (pattern, mode) pairs, 300 000 checks, ~5 min)"""
import sys, random, importlib.util, os, time
sys.path.insert(0, "/root/repo")
from loongcollector_amd import binding as B
from oracle.oracle import OracleRegex
from tests.helpers.table_interp import NfaInterp, TdfaInterp, packed_tdfa_interp
spec = importlib.util.spec_from_file_location("g", "/root/repo/tests/golden/gen_regex_golden.py"); gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
aspec = importlib.util.spec_from_file_location("a", "/root/repo/tests/golden/gen_atomic_golden.py"); agen = importlib.util.module_from_spec(aspec); aspec.loader.exec_module(agen)
t0 = time.time(); checked = pats = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = random.Random(5000 + seed); g = gen.Gen(rng)
    for k in range(120):
        if k % 3 == 2: p, smp = agen.gen(rng), None
        else: p, _, smp = g.alt(0)
        try: orx = OracleRegex(p)
        except ValueError: continue
        for flags, fn in ((0, orx.fullmatch), (B.LC_SYNTAX_SEARCH, orx.search), (B.LC_SYNTAX_SEARCH | B.LC_SYNTAX_PREFIX, None)):
            try: rx = B.GpuRegex(p, syntax_flags=flags)
            except (B.RegexUnsupportedError, B.RegexSyntaxError): continue
            if rx.info()["engine"] != B.LC_ENGINE_TDFA: continue
            pats += 1
            its = [TdfaInterp(rx), packed_tdfa_interp(rx)]
            subs = [gen.rand_subject(rng) for _ in range(4)] + [bytes(rng.choice(b"abc1 ") for _ in range(rng.randint(0, 12))) for _ in range(4)]
            if smp is not None: subs += [gen.mutate(rng, smp()) for _ in range(4)]
            for s in subs:
              try:
                if fn is None:
                    e = orx.search(s); want = None if e is None or e[0][0] != 0 else [v for ab in e for v in ab]
                elif flags: 
                    e = fn(s); want = None if e is None else [v for ab in e for v in ab]
                else:
                    e = fn(s); want = None if e is None else [v for ab in e[1:] for v in ab]
              except RuntimeError:
                continue
              if True:
                for it in its:
                    checked += 1
                    got = it.fullmatch(s)
                    assert got == want, (p, s, flags, type(it).__name__, got, want)
print("ok: %d patterns x modes, %d checks, %.0f s" % (pats, checked, time.time() - t0))
