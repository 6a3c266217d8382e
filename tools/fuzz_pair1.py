#!/usr/bin/env python3
"""Differential fuzz of the ONE-STAMP byte-pair tables (LC_TDFA_PAIR=2: regex_handle.cpp planTdfaDerive + packTdfaBlob; the walk of
tdfaStreamPair1Chunk restated in tests/helpers/table_interp.py TdfaPair1Interp): fresh random patterns, full-match and search mode,
random alignments of the line in memory, against the oracle.  Reports how many tables derived registers and settled DOUBLEs.
    python tools/fuzz_pair1.py FIRST_SEED LAST_SEED"""
import os
os.environ["LC_TDFA_PAIR"] = "2"
os.environ["LC_TDFA_COMPACT"] = "512"
import sys, random, importlib.util, time
sys.path.insert(0, "/root/repo")
import numpy as np
from loongcollector_amd import binding as B
from oracle.oracle import OracleRegex
from tests.helpers.table_interp import TdfaPair1Interp
spec = importlib.util.spec_from_file_location("g", "/root/repo/tests/golden/gen_regex_golden.py"); gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
t0 = time.time(); checked = pats = derived = doubles = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = random.Random(9000 + seed); g = gen.Gen(rng)
    for k in range(120):
        p, _, smp = g.alt(0)
        try: orx = OracleRegex(p)
        except ValueError: continue
        for flags, fn in ((0, orx.fullmatch), (B.LC_SYNTAX_SEARCH, orx.search)):
            try: rx = B.GpuRegex(p, syntax_flags=flags)
            except (B.RegexUnsupportedError, B.RegexSyntaxError): continue
            if rx.info()["engine"] != B.LC_ENGINE_TDFA: continue
            blob = rx.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32)
            if blob is None or not int(blob[7]) or int(blob[int(blob[7]) // 4 + 4]) != 1: continue
            it = TdfaPair1Interp(rx)
            pats += 1; derived += bool(it.derive)
            subs = [gen.rand_subject(rng) for _ in range(4)] + [bytes(rng.choice(b"abc1 ") for _ in range(rng.randint(0, 12))) for _ in range(4)]
            if smp is not None: subs += [gen.mutate(rng, smp()) for _ in range(6)]
            for s in subs:
                try:
                    e = fn(s)
                except RuntimeError:
                    continue
                want = None if e is None else [v for ab in (e if flags else e[1:]) for v in ab]
                for head in (rng.randrange(16), rng.randrange(16)):
                    got = it.fullmatch_pair1(s, head=head)
                    checked += 1; doubles += it.doubles
                    assert got == want, (p, s, flags, head, got, want, it.derive)
print("ok: %d pair tables (%d with derived registers), %d checks, %d doubles settled, %.0f s" % (pats, derived, checked, doubles, time.time() - t0))
