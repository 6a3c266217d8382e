"""Differential fuzz on the CPU: fresh random patterns (not the committed golden set) through the host compilers'
tables (TDFA + NFA interpreters) against the oracle -- full-match and search modes.  Catches compiler regressions
without a GPU; the GPU suite runs the same kernels against the golden vectors."""
import importlib.util
import os
import random

import pytest

from loongcollector_amd import binding as B
from oracle.oracle import OracleRegex
from tests.helpers.table_interp import NfaInterp, TdfaInterp

_spec = importlib.util.spec_from_file_location(
    "gen_regex_golden", os.path.join(os.path.dirname(__file__), "golden", "gen_regex_golden.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_patterns_tables_vs_oracle(seed):
    rng = random.Random(1000 + seed)
    g = gen.Gen(rng)
    checked = unsupported = 0
    for _ in range(150):
        p, _, smp = g.alt(0)
        try:
            orx = OracleRegex(p)
        except ValueError:
            continue
        for flags, oracle_fn in ((0, orx.fullmatch), (B.LC_SYNTAX_SEARCH, orx.search)):
            try:
                rx = B.GpuRegex(p, syntax_flags=flags)
            except B.RegexUnsupportedError:
                unsupported += 1
                continue
            interps = ([NfaInterp(rx)] if rx.has_nfa_program() else []) + (
                [TdfaInterp(rx)] if rx.info()["engine"] == B.LC_ENGINE_TDFA else [])
            subjects = [gen.mutate(rng, smp()) for _ in range(5)] + [gen.rand_subject(rng) for _ in range(3)]
            if flags:
                subjects = [gen.rand_subject(rng)[:3] + s + gen.rand_subject(rng)[:3] for s in subjects]
            for s in subjects:
                exp = oracle_fn(s)
                if flags:
                    want = None if exp is None else [v for ab in exp for v in ab]
                else:
                    want = None if exp is None else [v for ab in exp[1:] for v in ab]
                for it in interps:
                    checked += 1
                    assert it.fullmatch(s) == want, (p, s, flags)
    assert checked > 2000 and unsupported <= 4  # (a few random monsters exceed the 64-path follow-list limit)
