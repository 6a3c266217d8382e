#!/usr/bin/env python3
"""Differential fuzz of the NFA tables (packNfaBlob: steady masks, doomed-spawn rows, suffix flag) and of the atomic-elision pass: fresh
random patterns, plain and atomic / possessive, full match and search, walked by tests/helpers NfaInterp / AtomicNfaInterp (the kernels' walk
restated) against the oracle.    python tools/fuzz_nfa.py FIRST_SEED LAST_SEED     (250 seeds: ~60 000 pattern x modes, ~1 min)"""
import sys, random, importlib.util, time
sys.path.insert(0, "/root/repo")
from loongcollector_amd import binding as B
from oracle.oracle import OracleRegex
from tests.helpers.table_interp import NfaInterp
from tests.helpers.nfa_atomic_interp import AtomicNfaInterp
spec = importlib.util.spec_from_file_location("g", "/root/repo/tests/golden/gen_regex_golden.py"); gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
aspec = importlib.util.spec_from_file_location("a", "/root/repo/tests/golden/gen_atomic_golden.py"); agen = importlib.util.module_from_spec(aspec); aspec.loader.exec_module(agen)
t0=time.time(); checked=pats=quasi=elided=0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = random.Random(9000 + seed); g = gen.Gen(rng)
    for k in range(120):
        if k % 3 == 2: p, smp = agen.gen(rng), None
        else: p, _, smp = g.alt(0)
        try: orx = OracleRegex(p)
        except ValueError: continue
        for flags, fn in ((0, orx.fullmatch), (B.LC_SYNTAX_SEARCH, orx.search)):
            try: rx = B.GpuRegex(p, syntax_flags=flags, engine=B.LC_ENGINE_NFA)
            except (B.RegexUnsupportedError, B.RegexSyntaxError): continue
            if not rx.has_nfa_program(): continue
            kept, el = rx.atomic_groups(); elided += el
            it = AtomicNfaInterp(rx) if kept else NfaInterp(rx)
            if not kept: quasi += it.quasi_rows is not None
            pats += 1
            subs = [gen.rand_subject(rng) for _ in range(4)] + [bytes(rng.choice(b"abc1 ") for _ in range(rng.randint(0, 12))) for _ in range(4)]
            if smp is not None: subs += [gen.mutate(rng, smp()) for _ in range(4)]
            for s in subs:
                try:
                    e = fn(s)
                except RuntimeError:
                    continue
                want = None if e is None else ([v for ab in e for v in ab] if flags else [v for ab in e[1:] for v in ab])
                got = it.fullmatch(s)
                if got == "overflow": continue
                checked += 1
                assert got == want, (p, s, flags, type(it).__name__, got, want)
print("ok: %d patterns x modes (%d with quasi rows, %d groups elided), %d checks, %.0f s" % (pats, quasi, elided, checked, time.time()-t0))
