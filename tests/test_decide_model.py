"""The depth-first DECIDE walk (csrc/nfa_decide_kernel.hpp, modelled in tests/helpers/nfa_dfs_interp.py) against the golden
vectors and against the breadth-first interpreters, on the CPU: it must decide every line the thread-list kernels decide,
identically, and the lines they cannot (more live threads than they hold)."""
import json
import os
import random

import pytest

from loongcollector_amd import binding as B
from oracle.oracle import OracleRegex
from tests.helpers.nfa_atomic_interp import AtomicNfaInterp
from tests.helpers.nfa_dfs_interp import DfsNfaInterp


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_dfs_walk_reproduces_every_golden_vector(golden_dir):
    d = _load(golden_dir, "regex_golden.json")
    bad, n = [], 0
    for c in d["cases"]:
        rx = B.GpuRegex(c["p"].encode("latin-1"))
        if not rx.has_nfa_program():
            continue
        it = DfsNfaInterp(rx)
        for subj, flat in c["subs"]:
            exp = None if flat is None else flat[2:]
            for memo in (True, False):
                n += 1
                got = it.fullmatch(subj.encode("latin-1"), memo=memo, budget=2_000_000)
                if got != exp:
                    bad.append((c["p"], subj, memo, got, exp))
    assert n > 8000
    assert not bad, bad[:5]


def test_dfs_walk_reproduces_every_search_vector(golden_dir):
    d = _load(golden_dir, "regex_search_golden.json")
    bad, n = [], 0
    for c in d["cases"]:
        try:
            rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=B.LC_SYNTAX_SEARCH)
        except B.RegexUnsupportedError:
            continue
        if not rx.has_nfa_program():
            continue
        it = DfsNfaInterp(rx)
        for subj, flat in c["subs"]:
            n += 1
            got = it.fullmatch(subj.encode("latin-1"))
            if got != flat:
                bad.append((c["p"], subj, got, flat))
    assert n > 1000
    assert not bad, bad[:5]


def test_dfs_walk_atomic_groups_with_and_without_memo(golden_dir):
    """The cut rules (and their replay from the memo) on PCRE1 ^ `regex`-module vectors, full match and search."""
    d = _load(golden_dir, "regex_atomic_golden.json")
    bad, n = [], 0
    for kind, flags in (("full", 0), ("search", B.LC_SYNTAX_SEARCH)):
        for c in d[kind]:
            try:
                rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=flags)
            except B.RegexUnsupportedError:
                continue
            if not rx.has_nfa_program():
                continue
            it = DfsNfaInterp(rx)
            for subj, flat in c["subs"]:
                exp = flat if kind == "search" or flat is None else flat[2:]
                for memo in (True, False):
                    n += 1
                    got = it.fullmatch(subj.encode("latin-1"), memo=memo, budget=5_000_000)
                    if got != exp:
                        bad.append((kind, c["p"], subj, memo, got, exp))
    assert n > 8000, n
    assert not bad, bad[:5]


@pytest.mark.parametrize("pattern,subject", [
    (r"(.*)a(.{70})", "a" * 90 + "x" * 70),                  # > 64 live threads: the one-thread-per-lane kernel gives up
    (r"(.*)a(.{140})", "a" * 200 + "y" * 140),               # > 128: the two-threads-per-lane kernel gives up too
    (r"(?>a+|b)*(.*)a(.{70})c", "a" * 100 + "q" * 70 + "c"),  # atomic group + > 64 threads
    (r"(?>(?>(?>(?>(?>(?>(?>(a+))b?)c?)d?)e?)f?)g?)(h)", "aaabcdh"),  # 7 nested memberships: the kernels carry 6
])
def test_dfs_walk_decides_lines_the_thread_list_kernels_cannot(pattern, subject):
    rx = B.GpuRegex(pattern.encode(), engine=B.LC_ENGINE_NFA)
    s = subject.encode()
    exp = OracleRegex(pattern).fullmatch(s)
    assert exp is not None
    flat = [v for ab in exp[1:] for v in ab]
    assert AtomicNfaInterp(rx).fullmatch(s) == "overflow"
    it = DfsNfaInterp(rx)
    assert it.fullmatch(s) == flat
    assert it.steps < 40 * len(s) * (rx.info()["positions"] + 1)        # memoised: linear in positions x length
    assert DfsNfaInterp(rx).fullmatch(s[:-1] + b"!") == (None if OracleRegex(pattern).fullmatch(s[:-1] + b"!") is None else flat)


def test_dfs_walk_resumed_search_equals_breadth_first(golden_dir):
    rx = B.GpuRegex(rb"(\d+)-(?>[a-z]+)(\w?)", syntax_flags=B.LC_SYNTAX_SEARCH, engine=B.LC_ENGINE_NFA)
    s = b"xx 12-abc 345-zz9 end 7-q"
    bfs, dfs = AtomicNfaInterp(rx), DfsNfaInterp(rx)
    start = 0
    seen = 0
    while True:
        a, b = bfs.fullmatch(s, start=start), dfs.fullmatch(s, start=start)
        assert a == b
        if a is None:
            break
        seen += 1
        start = a[1]
    assert seen == 3


def test_dfs_walk_random_atomic_patterns_vs_oracle():
    rnd = random.Random(20260921)
    atoms = ["a", "b", "ab", "a+", "b*", "(?>a+)", "(?>a|ab)", "(?>b*)", "a++", "(?:a|b)*+", "(a|b)", "(?>(?>a)b|a)", "[ab]?"]
    n = 0
    for _ in range(300):
        pat = "".join(rnd.choice(atoms) for _ in range(rnd.randint(2, 5)))
        try:
            rx = B.GpuRegex(pat.encode(), engine=B.LC_ENGINE_NFA)
        except (B.RegexUnsupportedError, B.RegexSyntaxError):
            continue
        orx = OracleRegex(pat)
        it = DfsNfaInterp(rx)
        for _ in range(12):
            s = "".join(rnd.choice("ab") for _ in range(rnd.randint(0, 9))).encode()
            exp = orx.fullmatch(s)
            flat = None if exp is None else [v for ab in exp[1:] for v in ab]
            for memo in (True, False):
                n += 1
                assert it.fullmatch(s, memo=memo) == flat, (pat, s, memo)
    assert n > 3000
