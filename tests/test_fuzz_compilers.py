"""Differential fuzz on the CPU: fresh random patterns (not the committed golden set) through the host compilers'
tables (TDFA + NFA interpreters) against the oracle -- full-match and search modes.  Catches compiler regressions
without a GPU; the GPU suite runs the same kernels against the golden vectors."""
import importlib.util
import os
import random

import pytest

from loongcollector_amd import binding as B
from oracle.oracle import OracleRegex
from tests.helpers.nfa_atomic_interp import AtomicNfaInterp
from tests.helpers.nfa_dfs_interp import DfsNfaInterp
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


_aspec = importlib.util.spec_from_file_location(
    "gen_atomic_golden", os.path.join(os.path.dirname(__file__), "golden", "gen_atomic_golden.py"))


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_atomic_patterns_tdfa_tables_vs_oracle(seed):
    """Fresh random patterns full of (?>X) / possessive quantifiers / look assertions: the TDFA builder's segment
    lineage (tdfa.cpp commitAtomic) against the oracle's backtracking commit, full-match and search."""
    pytest.importorskip("regex")  # the generator module imports it; the check itself is oracle-only
    agen = importlib.util.module_from_spec(_aspec)
    _aspec.loader.exec_module(agen)
    rng = random.Random(7000 + seed)
    checked = 0
    for _ in range(250):
        p = agen.gen(rng)
        try:
            orx = OracleRegex(p)
        except ValueError:
            continue
        for flags, oracle_fn in ((0, orx.fullmatch), (B.LC_SYNTAX_SEARCH, orx.search)):
            try:
                rx = B.GpuRegex(p, syntax_flags=flags)
            except B.RegexUnsupportedError as e:
                assert "can match the empty string" in str(e) or "too many" in str(e) or "limit" in str(e), (p, str(e))
                continue
            its = ([TdfaInterp(rx)] if rx.info()["engine"] == B.LC_ENGINE_TDFA else []) + (
                [AtomicNfaInterp(rx)] if rx.has_nfa_program() else [])
            for _ in range(8):
                s = bytes(rng.choice(b'abc1 ') for _ in range(rng.randint(0, 8)))
                exp = oracle_fn(s)
                want = None if exp is None else [v for ab in (exp if flags else exp[1:]) for v in ab]
                for it in its:
                    checked += 1
                    got = it.fullmatch(s)
                    assert got == want or got == "overflow", (p, s, flags, type(it).__name__)
                if rx.has_nfa_program():   # the depth-first decide walk: never undecided, with and without its memo
                    dfs = DfsNfaInterp(rx)
                    for memo in (True, False):
                        checked += 1
                        assert dfs.fullmatch(s, memo=memo, budget=3_000_000) == want, (p, s, flags, "dfs", memo)
    assert checked > 5000


def test_random_patterns_with_run_captures_and_packed_blobs():
    """The same differential check with a capture-only look-ahead "(?=(S*))" spliced into the random pattern, and with the
    PACKED tables as the kernels address them (standard blob and the compact 16-bit-register blob) next to the logical
    ones."""
    import numpy as np

    from tests.helpers.table_interp import TdfaBlobInterp, packed_tdfa_interp
    runs = [r"(?=(.*))", r"(?=([a-c]*))", r"(?=(?:([^ ]*)))", r"(?=(\w*))"]
    rng = random.Random(77003)
    g = gen.Gen(rng)
    checked = with_runs = 0
    for _ in range(120):
        p, _, smp = g.alt(0)
        if rng.random() < 0.6:
            r = rng.choice(runs)
            k = rng.randrange(len(p) + 1)
            cand = p[:k] + r + p[k:]
            try:
                OracleRegex(cand)
                p = cand
            except ValueError:
                p = r + p
        try:
            orx = OracleRegex(p)
        except ValueError:
            continue
        for flags, oracle_fn in ((0, orx.fullmatch), (B.LC_SYNTAX_SEARCH, orx.search)):
            try:
                rx = B.GpuRegex(p, syntax_flags=flags)
            except (B.RegexUnsupportedError, B.RegexSyntaxError):
                continue
            with_runs += bool(rx.run_captures())
            interps = [AtomicNfaInterp(rx)] if rx.has_nfa_program() else []
            if rx.info()["engine"] == B.LC_ENGINE_TDFA:
                interps += [TdfaInterp(rx), packed_tdfa_interp(rx)]
                if rx.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32) is not None:
                    interps.append(TdfaBlobInterp(rx, compact=True))
            subjects = [gen.mutate(rng, smp()) for _ in range(4)] + [gen.rand_subject(rng) for _ in range(3)]
            if flags:
                subjects = [gen.rand_subject(rng)[:3] + s + gen.rand_subject(rng)[:3] for s in subjects]
            for s in subjects:
                try:
                    exp = oracle_fn(s)
                except RuntimeError:      # the backtracking oracle gave up (complexity guard)
                    continue
                want = None if exp is None else [v for ab in (exp if flags else exp[1:]) for v in ab]
                for it in interps:
                    got = it.fullmatch(s)
                    if got == "overflow":
                        continue
                    checked += 1
                    assert got == want, (p, s, flags, type(it).__name__)
    assert checked > 3000 and with_runs > 40


@pytest.mark.parametrize("budget", ["1", "3", "8", "96"])
def test_relaxed_screens_never_reject_a_value_the_pattern_matches(monkeypatch, budget):
    """lcCompileRelaxedScreen + screen_dfa.cpp on fresh random patterns (plain ones and ones full of atomic groups, possessive
    quantifiers and look assertions), relaxed down to budgets at which every alternation and counter collapses: whatever the
    oracle's search matches, the screen accepts.  (A screen that rejected a matching value would change results, not speed.)"""
    monkeypatch.setenv("LC_RELAX_BUDGET", budget)
    agen = importlib.util.module_from_spec(_aspec)
    _aspec.loader.exec_module(agen)
    rng = random.Random(777 + int(budget))
    g = gen.Gen(rng)
    screens = checked = hits = rejected = 0
    for k in range(700):
        if k % 2:
            p, smp = agen.gen(rng), None
        else:
            p, _, smp = g.alt(0)
        try:
            orx = OracleRegex(p)
        except ValueError:
            continue
        scr = B.GpuRegex.compile_screen(p, max_states=20000, max_table_bytes=2 << 20, relaxed=True)
        if scr is None:           # (accepts the empty string: screens nothing)
            continue
        screens += 1
        it = TdfaInterp(scr)
        subjects = [gen.rand_subject(rng) for _ in range(6)] + [bytes(rng.choice(b"abc1 ") for _ in range(rng.randint(0, 10))) for _ in range(6)]
        if smp is not None:
            subjects += [gen.mutate(rng, smp()) for _ in range(5)]
        subjects += [gen.rand_subject(rng)[:3] + s + gen.rand_subject(rng)[:3] for s in subjects[:6]]
        for s in subjects:
            hit = orx.search(s) is not None
            ok = it.fullmatch(s) is not None
            checked += 1
            hits += hit
            rejected += not ok
            assert ok or not hit, (p, s, budget)
    assert screens > 60 and checked > 1500 and hits > 200 and rejected > 50, (screens, checked, hits, rejected)


def test_nfa_tables_with_shortcuts_on_fresh_random_patterns():
    """tools/fuzz_nfa.py, two seeds of it: fresh random patterns (plain and atomic / possessive), full match and search; the NFA walk
    WITH the kernel's shortcuts (steady masks, doomed-spawn rows, suffix exit) and after the atomic-elision pass, against the oracle.
    (The tool was run over 260 seeds when the shortcuts went in: 58 688 pattern x modes, 626 846 checks.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_nfa.py"), "1000", "1003"], capture_output=True, text=True, timeout=600,
                         cwd=root)
    assert out.returncode == 0 and out.stdout.startswith("ok:"), (out.stdout[-400:], out.stderr[-1500:])
    words = out.stdout.split()
    assert int(words[1]) > 400 and int(out.stdout.split("(")[1].split()[0]) > 100      # patterns x modes; of them with doomed-spawn rows
