#!/usr/bin/env python3
"""Differential fuzz of the LAZY / partial tagged DFAs (csrc/tdfa.cpp buildTdfaLazy, lc_regex_lazy_train): random patterns (plain and
atomic / possessive / look-around ones) in full-match, search and anchored-search mode, compiled for the thread-list engine; the partial
automaton is built along HALF of the subjects, in two training calls, and every subject -- trained or not -- walks it both ways
(tdfa_l2_kernel's walk and tdfa_wave_kernel's): a trained subject must be DECIDED, any subject is either a MISS (the kernels leave those
to the thread-list engine) or decided exactly as the oracle decides it.  Forced misses come by themselves: the other half of the subjects.
    python tools/fuzz_lazy.py FIRST_SEED LAST_SEED       (20 seeds: ~5 000 (pattern, mode) pairs, ~100 000 checks, ~4 min)"""
import importlib.util
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loongcollector_amd import binding as B  # noqa: E402
from oracle.oracle import OracleRegex  # noqa: E402
from tests.helpers.table_interp import TdfaL2BlobInterp  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run(first, last, per_seed=120, verbose=False):
    gen = _load("g", os.path.join(ROOT, "tests", "golden", "gen_regex_golden.py"))
    agen = _load("a", os.path.join(ROOT, "tests", "golden", "gen_atomic_golden.py"))
    stats = {"patterns": 0, "checks": 0, "decided": 0, "missed": 0, "not_in_use": 0}
    for seed in range(first, last):
        rng = random.Random(7000 + seed)
        g = gen.Gen(rng)
        for k in range(per_seed):
            if k % 3 == 2:
                p, smp = agen.gen(rng), None
            else:
                p, _, smp = g.alt(0)
            try:
                orx = OracleRegex(p)
            except ValueError:
                continue
            for flags in (0, B.LC_SYNTAX_SEARCH, B.LC_SYNTAX_SEARCH | B.LC_SYNTAX_PREFIX):
                try:
                    rx = B.GpuRegex(p, syntax_flags=flags, engine=B.LC_ENGINE_NFA)
                except (B.RegexUnsupportedError, B.RegexSyntaxError):
                    continue
                subs = [gen.rand_subject(rng) for _ in range(6)] + [bytes(rng.choice(b"abc1 ") for _ in range(rng.randint(0, 12))) for _ in range(6)]
                if smp is not None:
                    subs += [gen.mutate(rng, smp()) for _ in range(6)]
                trained = subs[::2]
                r = rx.lazy_train(trained[:len(trained) // 2])
                r = rx.lazy_train(trained[len(trained) // 2:])          # (a second call: the sample grows, the tables are rebuilt)
                if not r["in_use"]:
                    stats["not_in_use"] += 1
                    continue
                stats["patterns"] += 1
                it = TdfaL2BlobInterp(rx, B.LC_TABLE_LAZY_TDFA_BLOB)
                assert it.miss != 0
                for idx, s in enumerate(subs):
                    try:
                        if flags == 0:
                            e = orx.fullmatch(s)
                            want = None if e is None else [v for ab in e[1:] for v in ab]
                        elif flags == B.LC_SYNTAX_SEARCH:
                            e = orx.search(s)
                            want = None if e is None else [v for ab in e for v in ab]
                        else:
                            e = orx.search(s)
                            want = None if e is None or e[0][0] != 0 else [v for ab in e for v in ab]
                    except RuntimeError:
                        continue
                    for walk in (it.fullmatch, it.fullmatch_wave):
                        got = walk(s)
                        stats["checks"] += 1
                        if got == it.MISS:
                            stats["missed"] += 1
                            assert idx % 2 == 1, ("a TRAINED subject misses", p, flags, s)
                            continue
                        stats["decided"] += 1
                        assert got == want, (p, s, flags, walk.__name__, got, want)
        if verbose:
            print("seed", seed, stats, flush=True)
    return stats


if __name__ == "__main__":
    t0 = time.time()
    st = run(int(sys.argv[1]), int(sys.argv[2]), verbose=True)
    print("ok: %(patterns)d patterns x modes with a lazy automaton (%(not_in_use)d without), %(checks)d checks: %(decided)d decided, %(missed)d missed" % st,
          "%.0f s" % (time.time() - t0))
