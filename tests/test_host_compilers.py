"""Host-side compilers (regex_parse -> follow_nfa -> tdfa) checked on the CPU by interpreting the very tables the
kernels execute (tests/helpers/table_interp.py) against the golden vectors and against the oracle.
No compute call goes through the C ABI here -- the product has no CPU path."""
import json
import os

import numpy as np
import pytest

from loongcollector_amd import binding as B
from oracle.oracle import OracleRegex
from tests.helpers.nfa_atomic_interp import AtomicNfaInterp
from tests.helpers.table_interp import NfaInterp, TdfaInterp


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "regex_golden.json")) as f:
        return json.load(f)


def test_tdfa_and_nfa_tables_reproduce_every_golden_vector(golden):
    bad = []
    n = 0
    for c in golden["cases"]:
        rx = B.GpuRegex(c["p"].encode("latin-1"))
        assert rx.groups == c["g"]
        interps = [("nfa", NfaInterp(rx))] if rx.has_nfa_program() else []
        if rx.info()["engine"] == B.LC_ENGINE_TDFA:
            interps.append(("tdfa", TdfaInterp(rx)))
        assert interps
        for subj, flat in c["subs"]:
            s = subj.encode("latin-1")
            exp = None if flat is None else flat[2:]
            for name, it in interps:
                n += 1
                got = it.fullmatch(s)
                if got != exp:
                    bad.append((name, c["p"], subj, got, exp))
    assert n > 8000
    assert not bad, bad[:5]


@pytest.mark.parametrize("kind", ["A", "B"])
def test_tables_vs_oracle_on_bench_corpus(kind):
    from loongcollector_amd import corpus
    pattern = corpus.REGEX_A if kind == "A" else corpus.REGEX_B
    data, off, length = corpus.apache_batch(96, kind, pool_lines=96, poison_every=7)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:-1], length)
    rx = B.GpuRegex(pattern)
    assert rx.info()["engine"] == B.LC_ENGINE_TDFA
    td, nf = TdfaInterp(rx), NfaInterp(rx)
    raw = data.tobytes()
    for i in range(96):
        s = raw[off[i]:off[i] + length[i]]
        for it in (td, nf):
            got = it.fullmatch(s)
            if exp_status[i]:
                assert got == list(exp_caps[i])
            else:
                assert got is None


def test_bench_regex_tables_are_small_enough_for_lds():
    from loongcollector_amd import corpus
    for p in (corpus.REGEX_A, corpus.REGEX_B):
        info = B.GpuRegex(p).info()
        assert info["engine"] == B.LC_ENGINE_TDFA
        # (round 4: the standard tables carry the one-stamp byte-pair table -- states x (classes + 1)^2 words -- beside the 2 KB of
        # single-byte rows)
        assert info["table_bytes"] < 26 * 1024 and info["states"] < 64 and info["registers"] <= 2 * info["mark_count"] + 1


@pytest.mark.parametrize("pat,code", [
    (r"(a", B.RegexSyntaxError), (r"a)", B.RegexSyntaxError), (r"[a", B.RegexSyntaxError), (r"a**", B.RegexSyntaxError),
    (r"*a", B.RegexSyntaxError), (r"a{3,1}", B.RegexSyntaxError),
    # (round 5: "(?=ab)a" compiles -- a fixed-length look-ahead is a window, follow_nfa.cpp; a body of variable length is not regular
    # bookkeeping the automata do, and a look-behind needs fixed-width text in front of it)
    # (round 6: "(a)\1" compiles -- back-references run on the device backtracking engine, tests/test_backref.py; a reference to a
    # group the pattern does not have is boost's error_backref)
    # ... and so do general look-arounds "(?=a+b)a", "(?<!ab)c", "\w+(?<=ab)c" and "(a*)*", which the position automaton cannot express;
    # a look-behind of variable length is refused by boost as well)
    (r"(a)\2", B.RegexSyntaxError), (r"(?<=a+)b", B.RegexUnsupportedError), (r"(?<!a+b)c", B.RegexUnsupportedError),
    # (a conditional on a group runs there too; one that asks about a group the pattern does not have is an invalid reference)
    (r"(?R)b", B.RegexUnsupportedError), (r"(?(?=a)b|c)", B.RegexUnsupportedError), (r"(?(1)a|b)", B.RegexSyntaxError),
])
def test_invalid_and_unsupported_patterns_fail_loudly(pat, code):
    # reference: IsRegexValid false -> Init fails (ParamExtractor.cpp:199-209, ProcessorParseRegexNative.cpp:53-63)
    with pytest.raises(code):
        B.GpuRegex(pat)


def test_named_groups_and_named_only_mode():
    rx = B.GpuRegex(r"(?<ip>\d+)\.(\d+) (?P<rest>.*)")
    assert rx.groups == 3 and rx.group_name(1) == "ip" and rx.group_name(2) is None and rx.group_name(3) == "rest"
    rx2 = B.GpuRegex(r"(?<ip>\d+)\.(\d+) (?P<rest>.*)", syntax_flags=B.LC_SYNTAX_NAMED_ONLY)
    assert rx2.groups == 2 and rx2.group_name(2) == "rest"
    assert TdfaInterp(rx2).fullmatch(b"10.2 xyz") == [0, 2, 5, 8]


def test_syntax_flags():
    assert TdfaInterp(B.GpuRegex(r"(get) (.)", syntax_flags=B.LC_SYNTAX_ICASE)).fullmatch(b"GeT x") == [0, 3, 4, 5]
    assert TdfaInterp(B.GpuRegex(r"a.b", syntax_flags=B.LC_SYNTAX_NO_DOTALL)).fullmatch(b"a\nb") is None
    assert TdfaInterp(B.GpuRegex(r"a.b")).fullmatch(b"a\nb") == []
    assert TdfaInterp(B.GpuRegex(r"a\n^b")).fullmatch(b"a\nb") == []
    assert TdfaInterp(B.GpuRegex(r"a\n^b", syntax_flags=B.LC_SYNTAX_NO_MULTILINE)).fullmatch(b"a\nb") is None


def test_forced_engines_and_tdfa_limit_fallback():
    # a pattern whose determinisation explodes must fall back to the NFA engine under AUTO and fail under TDFA
    blowup = r"(.*)a" + "." * 14 + r"(.*)"
    with pytest.raises(B.RegexUnsupportedError):
        B.GpuRegex(blowup, engine=B.LC_ENGINE_TDFA)
    rx = B.GpuRegex(blowup)
    assert rx.info()["engine"] == B.LC_ENGINE_NFA
    s = b"xxa" + b"y" * 14 + b"zz"
    exp = OracleRegex(blowup).fullmatch(s)
    assert NfaInterp(rx).fullmatch(s) == [v for ab in exp[1:] for v in ab]


def test_search_mode_tables_reproduce_every_search_vector(golden_dir):
    """LC_SYNTAX_SEARCH compiles (?s:.*?)(re)(?s:.*): group 1 = whole match, own groups shifted by one."""
    with open(os.path.join(golden_dir, "regex_search_golden.json")) as f:
        d = json.load(f)
    bad, n, unsupported = [], 0, 0
    for c in d["cases"]:
        try:
            rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=B.LC_SYNTAX_SEARCH)
        except B.RegexUnsupportedError:
            unsupported += 1
            continue
        assert rx.groups == c["g"] + 1
        interps = ([NfaInterp(rx)] if rx.has_nfa_program() else []) + (
            [TdfaInterp(rx)] if rx.info()["engine"] == B.LC_ENGINE_TDFA else [])
        for subj, flat in c["subs"]:
            for it in interps:
                n += 1
                got = it.fullmatch(subj.encode("latin-1"))
                if got != flat:
                    bad.append((c["p"], subj, got, flat))
    assert n > 2000 and unsupported <= 2
    assert not bad, bad[:5]


def test_resumed_searches_on_the_logical_and_the_packed_tables(golden_dir):
    """lc_regex_match_device_from: a search RESUMED inside the line starts in the state "only the wrapper's prefix thread is
    alive and the previous byte had class c" (startAfter).  The logical tables, the packed blobs (LDS kernels or the
    global-memory kernel) and the NFA program -- after dead-store elimination and state minimisation -- against the oracle's
    search from that offset (look-behinds see the byte before it)."""
    from oracle.oracle import OracleRegex
    from tests.helpers.table_interp import packed_tdfa_interp
    with open(os.path.join(golden_dir, "regex_search_golden.json")) as f:
        d = json.load(f)
    rng = __import__("random").Random(11)
    n = hits = 0
    for c in d["cases"][::3]:
        p = c["p"].encode("latin-1")
        try:
            rx = B.GpuRegex(p, syntax_flags=B.LC_SYNTAX_SEARCH)
            o = OracleRegex(p)
        except (B.RegexUnsupportedError, ValueError):
            continue
        interps = [NfaInterp(rx)] if rx.has_nfa_program() else []
        if rx.info()["engine"] == B.LC_ENGINE_TDFA:
            interps += [TdfaInterp(rx), packed_tdfa_interp(rx)]
        for subj, _ in c["subs"][:6]:
            s = subj.encode("latin-1")
            for start in {0, 1, len(s) // 2, max(0, len(s) - 1), len(s)} if s else {0}:
                if start > len(s):
                    continue
                exp = o.search(s, start)
                want = None if exp is None else [v for ab in exp for v in ab]
                hits += exp is not None
                for it in interps:
                    n += 1
                    assert it.fullmatch(s, start=start) == want, (c["p"], s, start, type(it).__name__)
    assert n > 3000 and hits > 400, (n, hits)


def test_anchored_search_is_the_search_whose_match_starts_at_the_first_byte(golden_dir):
    """LC_SYNTAX_SEARCH | LC_SYNTAX_PREFIX compiles (re)(?s:.*) with the search's group layout.  On every search vector: when the
    search's match starts at byte 0 the anchored search reports exactly that match -- same captures --, otherwise (later start, or
    no match) it reports none.  (The Grok matcher tries the anchored automaton first: far smaller, no "anywhere" prefix.)"""
    with open(os.path.join(golden_dir, "regex_search_golden.json")) as f:
        d = json.load(f)
    bad, at0, later, none = [], 0, 0, 0
    for c in d["cases"]:
        try:
            rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=B.LC_SYNTAX_SEARCH | B.LC_SYNTAX_PREFIX)
        except B.RegexUnsupportedError:
            continue
        assert rx.groups == c["g"] + 1
        interps = ([NfaInterp(rx)] if rx.has_nfa_program() else []) + (
            [TdfaInterp(rx)] if rx.info()["engine"] == B.LC_ENGINE_TDFA else [])
        for subj, flat in c["subs"]:
            want = flat if flat is not None and flat[0] == 0 else None
            # a search that starts later may still have an anchored match of lower priority... no: leftmost wins, so a match
            # at byte 0 would have been the search's answer
            at0 += want is not None
            later += flat is not None and want is None
            none += flat is None
            for it in interps:
                got = it.fullmatch(subj.encode("latin-1"))
                if got != want:
                    bad.append((c["p"], subj, got, want))
    assert at0 > 300 and later > 100 and none > 100, (at0, later, none)
    assert not bad, bad[:5]


def test_atomic_groups_and_possessive_quantifiers_on_both_engines_tables(golden_dir):
    """Atomic groups: segment lineage in the TDFA builder (tdfa.cpp commitAtomic), and the same commit rules applied per
    step by the NFA engine."""
    with open(os.path.join(golden_dir, "regex_atomic_golden.json")) as f:
        d = json.load(f)
    bad, n, unsupported = [], 0, 0
    for kind, flags in (("full", 0), ("search", B.LC_SYNTAX_SEARCH)):
        for c in d[kind]:
            try:
                rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=flags)
            except B.RegexUnsupportedError as e:
                # the one construct the device engines refuse here: a loop whose body can match the empty string
                assert "unbounded repeat of a sub-expression that can match the empty string" in str(e), (c["p"], str(e))
                unsupported += 1
                continue
            if rx.info()["engine"] == B.LC_ENGINE_BT:
                # (round 6) a loop whose body can match the empty string has no position automaton: the handle runs the device
                # backtracking engine, whose programs are walked over this same golden set in tests/test_backref.py
                unsupported += 1
                continue
            interps = [TdfaInterp(rx)] if rx.info()["engine"] == B.LC_ENGINE_TDFA else []
            if rx.has_nfa_program():   # the NFA engine's ordered commit pass (nfa_kernel.hpp nfaAtomicStep), in Python
                interps.append(AtomicNfaInterp(rx))
            assert interps
            for subj, flat in c["subs"]:
                exp = flat if kind == "search" or flat is None else flat[2:]
                for it in interps:
                    n += 1
                    got = it.fullmatch(subj.encode("latin-1"))
                    if got != exp:
                        bad.append((kind, c["p"], subj, got, exp))
    assert n > 8000 and unsupported <= 40, (n, unsupported)
    assert not bad, bad[:5]


def test_fixed_length_lookarounds_on_both_engines_tables(golden_dir):
    """Multi-byte look-arounds with a fixed-length body (round 5): a look-ahead the rest of the pattern does not decide becomes a
    product of the follow NFA with the body's chain (follow_nfa.cpp applyWindows), a look-behind is decided from the fixed-width text
    in front of it (regex_parse.cpp) -- or the pattern is refused, loudly.  Vectors: `regex` module and PCRE1 agree
    (tests/golden/gen_lookaround_golden.py).  Both engines' tables, full match and search, against them."""
    with open(os.path.join(golden_dir, "regex_lookaround_golden.json")) as f:
        d = json.load(f)
    bad, n, refused = [], 0, []
    for kind, flags in (("full", 0), ("search", B.LC_SYNTAX_SEARCH)):
        for c in d[kind]:
            try:
                rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=flags)
            except B.RegexUnsupportedError as e:
                # what stays refused: a loop whose body can match the empty string (as everywhere), and a look-behind that the text in
                # front of it does not decide
                assert ("can match the empty string" in str(e) or "look-behind that the preceding sub-expression does not decide" in str(e)), (c["p"], str(e))
                refused.append(c["p"])
                continue
            if rx.info()["engine"] == B.LC_ENGINE_BT:
                # (round 6) what used to be refused here -- a loop whose body can match the empty string, a look-behind the text in
                # front of it does not decide -- runs on the device backtracking engine: programs walked in tests/test_backref.py
                refused.append(c["p"])
                continue
            interps = [TdfaInterp(rx)] if rx.info()["engine"] == B.LC_ENGINE_TDFA else []
            if rx.has_nfa_program():
                interps.append(NfaInterp(rx))
            assert interps
            for subj, flat in c["subs"]:
                exp = flat if kind == "search" or flat is None else flat[2:]
                for it in interps:
                    n += 1
                    got = it.fullmatch(subj.encode("latin-1"))
                    if got != exp:
                        bad.append((kind, c["p"], subj, got, exp))
    assert n > 6000 and len(refused) <= 40, (n, len(refused))
    assert not bad, bad[:5]
    # the library's own two users compile and find what the backtracking engines find
    mongo = rb'\{ (?<={ ).*(?= } ntoreturn:) \}'
    for flags, s, want in ((0, b"{ a: { b: 2 } } ntoreturn:", None), (B.LC_SYNTAX_SEARCH, b"query: { a: { b: 2 } } ntoreturn:5", [7, 22])):
        rx = B.GpuRegex(mongo, syntax_flags=flags)
        for it in ([TdfaInterp(rx)] if rx.info()["engine"] == B.LC_ENGINE_TDFA else []) + ([NfaInterp(rx)] if rx.has_nfa_program() else []):
            assert it.fullmatch(s) == want, (flags, it.fullmatch(s))
    # where the automaton would need the history of the input -- a look-behind behind a field of variable width -- the pattern goes to
    # the device backtracking engine (round 6; also under Grok's dialect -- regexp2 backtracks --, refused under the Go regex plugin's: RE2)
    assert B.GpuRegex(rb"\w+(?<!ab)c").info()["engine"] == B.LC_ENGINE_BT
    assert B.GpuRegex(rb"\w+(?<!ab)c", syntax_flags=B.LC_SYNTAX_NAMED_ONLY | B.LC_SYNTAX_REGEXP2).info()["engine"] == B.LC_ENGINE_BT
    with pytest.raises(B.RegexUnsupportedError, match="look-behind that the preceding sub-expression does not decide"):
        B.GpuRegex(rb"\w+(?<!ab)c", syntax_flags=B.LC_SYNTAX_REGEXP2)
    # ... and a window in a pattern that keeps atomic groups: no automaton, the backtracking engine
    with pytest.raises(B.RegexUnsupportedError, match="look-ahead in a pattern with atomic groups"):
        B.GpuRegex(rb"(?>a+|ab)(?=bc)\w+", engine=B.LC_ENGINE_TDFA)
    assert B.GpuRegex(rb"(?>a+|ab)(?=bc)\w+").info()["engine"] == B.LC_ENGINE_BT


def test_every_pattern_of_the_example_library_compiles():
    """example_config/processor_grok_patterns/*: every pattern of every file, as the one Match entry of a Grok processor (expansion,
    parse, follow NFA -- AnchoredFirst off: no tagged-DFA construction, that is a matter of speed).  Through round 4 MONGO_QUERY and
    MONGO_SLOWQUERY (mongodb:2-3) failed Init on their multi-byte look-arounds.  The reference tree is not on the GPU box: skipped there."""
    from loongcollector_amd.grok import Grok
    lib = "/root/reference/example_config/processor_grok_patterns"
    if not os.path.isdir(lib):
        pytest.skip("needs the reference tree (/root/reference)")
    names = []
    for fn in sorted(os.listdir(lib)):
        with open(os.path.join(lib, fn), encoding="utf-8") as f:
            for line in f:
                line = line.strip()
                if line and not line.startswith("#") and len(line.split(None, 1)) == 2:
                    names.append((fn, line.split(None, 1)[0]))
    assert len(names) > 400 and ("mongodb", "MONGO_SLOWQUERY") in names
    failed = []
    for fn, name in names:
        try:
            Grok(Match=["%{" + name + "}"], CustomPatternDir=[lib], AnchoredFirst=False)
        except Exception as e:   # noqa: BLE001 -- every failure is reported with its reason
            failed.append((fn, name, str(e)[:160]))
    assert not failed, failed


def test_prefix_mode_is_regex_search_match_continuous(golden_dir):
    """LC_SYNTAX_PREFIX = boost::regex_search(match_continuous) (StringTools.cpp:263-289), the per-line question of the
    multiline splitter (ProcessorSplitMultilineLogStringNative.cpp:184-272): tables vs the oracle's anchored search, on the
    search corpus with the patterns' own groups."""
    with open(os.path.join(golden_dir, "regex_search_golden.json")) as f:
        d = json.load(f)
    from oracle.oracle import OracleRegex
    n = hits = 0
    for c in d["cases"][:150]:
        p = c["p"].encode("latin-1")
        try:
            rx = B.GpuRegex(p, syntax_flags=B.LC_SYNTAX_PREFIX)
        except B.RegexUnsupportedError:
            continue
        o = OracleRegex(p)
        its = ([NfaInterp(rx)] if rx.has_nfa_program() else []) + (
            [TdfaInterp(rx)] if rx.info()["engine"] == B.LC_ENGINE_TDFA else [])
        for subj, _ in c["subs"]:
            s = subj.encode("latin-1")
            for cut in (0, 3):
                t = s[cut:]
                exp = o.prefixmatch(t)
                want = None if exp is None else [v for ab in exp[1:] for v in ab]
                hits += exp is not None
                for it in its:
                    n += 1
                    assert it.fullmatch(t) == want, (c["p"], t)
    assert n > 1500 and hits > 200
    # the multiline idiom of the reference docs: a log starts with a timestamp
    rx = B.GpuRegex(rb"\d{4}-\d{2}-\d{2} \d{2}:\d{2}:\d{2}", syntax_flags=B.LC_SYNTAX_PREFIX)
    it = TdfaInterp(rx)
    assert it.fullmatch(b"2024-01-04 14:36:10 ERROR boom") is not None
    assert it.fullmatch(b"    at com.example.Foo.bar(Foo.java:42)") is None
    assert it.fullmatch(b" 2024-01-04 14:36:10 leading space") is None
    # together with LC_SYNTAX_SEARCH: the anchored search (group 1 = the whole match), see the test below
    assert B.GpuRegex(b"(a)", syntax_flags=B.LC_SYNTAX_PREFIX | B.LC_SYNTAX_SEARCH).groups == 2


def test_run_captures_on_both_table_formats():
    """"(?=(S*))": a look-ahead that always holds and only captures (regex_ast.hpp Node::runCapture).  The tables stamp the
    begin slot; the reader fills in the end (lc_regex_run_captures).  Both table formats against the oracle, which runs
    the look-ahead for real."""
    from oracle.oracle import OracleRegex
    from tests.helpers.table_interp import NfaInterp, TdfaInterp
    from tests.helpers.wide_patterns import RUN_CAPTURE_PATTERNS, RUN_CAPTURE_SUBJECTS
    runs = 0
    for pat in RUN_CAPTURE_PATTERNS:
        o = OracleRegex(pat)
        for eng, interp in ((B.LC_ENGINE_TDFA, TdfaInterp), (B.LC_ENGINE_NFA, NfaInterp)):
            rx = B.GpuRegex(pat, engine=eng)
            runs += len(rx.run_captures())
            it = interp(rx)
            for s in RUN_CAPTURE_SUBJECTS:
                want = o.fullmatch(s)
                want = None if want is None else [v for be in want for v in be][2:]
                assert it.fullmatch(s) == want, (pat, s)
    assert runs == 2 * (len(RUN_CAPTURE_PATTERNS) - 1)     # "(?=.*)abc" has nothing to capture
    # anything else inside a look-ahead is no run capture but a general look-around: refused by the automata, taken by the device
    # backtracking engine when the engine is left to the library (round 6, tests/test_backref.py)
    for pat in (rb"(?=(a+))a*", rb"(?=(.*?))a", rb"(?!(.*))a", rb"(?=(a*)b)a*b"):
        assert B.GpuRegex(pat).info()["engine"] == B.LC_ENGINE_BT
        with pytest.raises(B.RegexUnsupportedError):
            B.GpuRegex(pat, engine=B.LC_ENGINE_TDFA)


@pytest.mark.parametrize("compact", ["256", "1024"])
def test_packed_tdfa_blobs_walk_like_the_logical_tables(golden_dir, monkeypatch, compact):
    """device_tables.h as the kernels address it: the standard blob (class-indexed rows, 32-bit registers) and the tables of
    the opt-in COMPACT kernel variants (16-bit registers; 1024 = byte-indexed rows, small automata only) against the
    logical-table interpreter."""
    from tests.helpers.table_interp import TdfaBlobInterp, packed_tdfa_interp
    monkeypatch.setenv("LC_TDFA_COMPACT", compact)
    with open(os.path.join(golden_dir, "regex_golden.json")) as f:
        golden = json.load(f)
    wide = checked = 0
    for c in golden["cases"][::5]:
        try:
            rx = B.GpuRegex(c["p"].encode("latin-1"), engine=B.LC_ENGINE_TDFA)
        except B.RegexUnsupportedError:
            continue
        ref = TdfaInterp(rx)
        interps = [packed_tdfa_interp(rx)]
        if rx.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32) is not None:
            wide += 1
            interps.append(TdfaBlobInterp(rx, compact=True))
            assert interps[-1].block == int(compact) and interps[-1].wide == (compact == "1024")
        for s, _ in c["subs"]:
            s = s.encode("latin-1")
            want = ref.fullmatch(s)
            checked += 1
            for it in interps:
                assert it.fullmatch(s) == want, (c["p"], s, it.compact)
    assert checked > 500 and wide > 50


def test_multi_stamp_programs_are_folded_into_set_registers():
    """A capture group that can match "" stamps its begin and end on ONE transition.  Packed, such a program becomes a stamp
    of the set's own register (device_tables.h TD_NREGS, fold words) and the table has no general program left; the captures
    stay what the logical tables give -- empty fields first, last, repeated, next to non-empty ones.  A pattern whose programs
    copy registers keeps them."""
    from tests.helpers.table_interp import TdfaBlobInterp
    from loongcollector_amd import corpus
    cases = [
        (corpus.REGEX_A, [b'1.2.3.4 - - [10/Oct/2000:13:55:36 -0700] "GET /a HTTP/1.0" 200 2326 "" "curl/8"',
                          b'1.2.3.4 - - [10/Oct/2000:13:55:36 -0700] "GET /a HTTP/1.0" 200 2326 "http://x" ""',
                          b'1.2.3.4 - - [10/Oct/2000:13:55:36 -0700] "GET /a HTTP/1.0" 200 2326 "" ""',
                          b'1.2.3.4 - - [10/Oct/2000:13:55:36 -0700] "GET /a HTTP/1.0" 200 2326 "r" "u"']),
        (rb'(\S*) (\S*) (\S*)', [b"  ", b"a  ", b" b ", b"  c", b"aa bb cc", b"a b", b" "]),
        (rb'"([^"]*)"(?:,"([^"]*)")*', [b'""', b'"",""', b'"a","","b"', b'"a","b",""', b'"","",""']),
        (rb'(a*)(b*)(a*)', [b"", b"a", b"b", b"ab", b"ba", b"aba", b"aabbaa"]),
    ]
    for pat, lines in cases:
        rx = B.GpuRegex(pat, engine=B.LC_ENGINE_TDFA)
        ref, packed = TdfaInterp(rx), TdfaBlobInterp(rx)
        assert packed.fold and packed.no_general, pat
        interps = [packed]
        if rx.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32) is not None:
            interps.append(TdfaBlobInterp(rx, compact=True))
        for s in lines:
            want = ref.fullmatch(s)
            for it in interps:
                assert it.fullmatch(s) == want, (pat, s)
    assert any(ref.fullmatch(s) is not None for s in lines)
    # programs that copy registers stay general programs, and such a table folds nothing
    for pat in (rb"((a)|(b))+", rb"(.+)=(.+)"):
        it = TdfaBlobInterp(B.GpuRegex(pat, engine=B.LC_ENGINE_TDFA))
        assert not it.no_general and it.fold is None, pat


def test_hostile_patterns_fail_fast_or_compile_fast():
    """A pattern comes from a configuration file: whatever it is, Init must answer quickly -- compile, or refuse with a
    reason.  ("(a?){200}a{200}" used to spend two minutes in the tagged-DFA construction before being refused.)"""
    import time
    pats = [
        r"(?:a{1000}){1000}", r"((a*)*)*b", r"(a|aa)+$", "(" * 5000 + "a" + ")" * 5000, "|".join("w%d" % i for i in range(20000)),
        r"(?:[a-z]{1,64}\.){1,64}[a-z]{2,63}", r"(a?){200}a{200}", r"(?>(?>(?>a+)+)+)+b", r"[\x00-\xff]{65535}", r"a{0,100000}",
        r"(x+x+)+y", r"(?:(?:(?:(?:(?:a|b)*c)*d)*e)*f)*g", "(?=(" + "a" * 100000 + "*))b", r"\b" * 10000 + "a",
        "(?i)" + "abcdefghijklmnopqrstuvwxyz" * 200, "(" * 70 + "a" + ")" * 70, "[" + "a-z" * 50000 + "]",
    ]
    outcomes = {"ok": 0, "refused": 0}
    for p in pats:
        for flags in (0, B.LC_SYNTAX_SEARCH):
            t0 = time.time()
            try:
                B.GpuRegex(p.encode(), syntax_flags=flags)
                outcomes["ok"] += 1
            except (B.RegexUnsupportedError, B.RegexSyntaxError) as e:
                assert str(e)
                outcomes["refused"] += 1
            assert time.time() - t0 < 10, (p[:40], time.time() - t0)
    assert outcomes["ok"] >= 10 and outcomes["refused"] >= 10


ELIDE_CASES = [
    # (pattern, atomic instances kept, groups elided)
    (r"(?>\d\d){1,2}", 0, 1),                                                   # Grok YEAR: fixed length
    (r"""(?>(?<!\\)(?>"(?>\\.|[^\\"]+)+"|""|(?>'(?>\\.|[^\\']+)+')|''|(?>`(?>\\.|[^\\`]+)+`)|``))""", 0, 1),   # Grok QUOTEDSTRING
    (r"(?>[A-Za-z]+:|\\)(?:\\[^\\?*]*)+", 0, 1),                                # Grok WINPATH: delimiter-terminated
    (r"(?<![0-9.+-])(?>[+-]?(?:(?:[0-9]+(?:\.[0-9]+)?)|(?:\.[0-9]+)))", 1, 0),  # Grok BASE10NUM: "12" is a prefix of "12.5" -> kept
    (r"(?>a|ab)c", 1, 0),                                                       # the textbook case: kept
    (r"(?>a+)b", 1, 0), (r"[^,]*+,", 1, 0),                                     # possessive runs are not prefix-free: kept
    (r"(?>ab|cd)e", 0, 1), (r"x(?>[0-9]+;)y", 0, 1),
    (r"(?>(\d)\d)", 1, 0),                                                      # captures inside: never touched
    (r"(?>a(?=b))b", 1, 0),                                                     # look-ahead inside: never touched
]


def test_redundant_atomic_groups_become_plain_groups():
    """csrc/atomic_elide.cpp: (?>X) with a prefix-free language that the plain form shares, nothing captured or asserted inside, is
    compiled as (?:X) -- no ordered commit pass in the NFA kernel, no commit bookkeeping in the determinisation.  What is elided and
    what is kept; and, for every case, the tables against the oracle (which implements atomic groups natively) on subjects made of
    the patterns' own alphabet, full match and search."""
    import random
    rng = random.Random(11)
    checked = 0
    for pat, kept, elided in ELIDE_CASES:
        rx = B.GpuRegex(pat)
        assert rx.atomic_groups() == (kept, elided), (pat, rx.atomic_groups())
        alphabet = sorted(set(c for c in pat if c.isalnum() or c in "\"'`\\:;,.+- ")) + ["7", "q", " "]
        subs = ["".join(rng.choice(alphabet) for _ in range(rng.randint(0, 9))) for _ in range(250)]
        subs += ["12", "12.5", "1999", "19", "\"a\\\"b\"", "\"\"", "'x y'", "`z`", "c:\\dir\\f", "abc", "ac", "aab", "a,b,", "x12;y", "abe", "cde"]
        for flags in (0, B.LC_SYNTAX_SEARCH):
            rxf = B.GpuRegex(pat, syntax_flags=flags)
            interps = [TdfaInterp(rxf)] if rxf.info()["engine"] == B.LC_ENGINE_TDFA else []
            if rxf.has_nfa_program():
                interps.append(AtomicNfaInterp(rxf) if rxf.atomic_groups()[0] else NfaInterp(rxf))
            o = OracleRegex(pat)
            for s in subs:
                sb = s.encode("latin-1")
                want = o.search(sb) if flags else o.fullmatch(sb)
                exp = None if want is None else [v for be in want for v in be][0 if flags else 2:]
                for it in interps:
                    checked += 1
                    assert it.fullmatch(sb) == exp, (pat, flags, s, type(it).__name__)
    assert checked > 8000
    # the Grok library: every time stamp carries YEAR, the access-log formats QUOTEDSTRING
    from loongcollector_amd.grok import Grok
    g = Grok(Match=["%{TIMESTAMP_ISO8601:ts} %{QS:q}"])
    rx = B.GpuRegex(g.expanded(0).encode(), syntax_flags=B.LC_SYNTAX_SEARCH | B.LC_SYNTAX_NAMED_ONLY | B.LC_SYNTAX_REGEXP2)
    kept, elided = rx.atomic_groups()
    assert kept == 0 and elided >= 2


def test_wave_walk_of_global_memory_automata_equals_the_byte_walk(golden_dir):
    """tdfa_wave_kernel (one value per wavefront) crosses quiet runs without table reads and stops in the absorbing state: its walk,
    restated in tests/helpers/table_interp.py TdfaL2BlobInterp.fullmatch_wave, against the plain byte walk of the same blob and the
    oracle -- on an automaton of 1 000+ states, on an anchored Grok search with a GREEDYDATA tail, and on the quiet-mask table itself
    (bit c of state s  <=>  class c keeps s and runs no program)."""
    import random
    from tests.helpers.table_interp import TdfaL2BlobInterp
    rng = random.Random(3)
    cases = [(rb"(?:a|b)*a(?:a|b){12}(c+)(d*)", 0, b"abcdx"),
             # quiet runs in the middle of the walk ([^x]* keeps its state without a program), behind and in front of captures
             (rb"([^x]*)x(?:a|b)*a(?:a|b){11}([^y]*)y(.*)", 0, b"abxy q"),
             (rb"(?:a|b|c)*c(?:a|b|c){12}(\d+)-([a-c]*)", 0, b"abc1-")]
    from loongcollector_amd.grok import Grok
    with open(os.path.join(golden_dir, "grok_config3.json"), encoding="utf-8") as f:
        cfg3 = json.load(f)
    g = Grok(Match=["%{SYSLOG5424LINE}"], CustomPatterns=cfg3["custom_patterns"])
    grok_flags = (B.LC_SYNTAX_SEARCH | B.LC_SYNTAX_NAMED_ONLY | B.LC_SYNTAX_NO_DOTALL | B.LC_SYNTAX_NO_MULTILINE | B.LC_SYNTAX_REGEXP2)
    cases.append((g.expanded(0).encode(), grok_flags | B.LC_SYNTAX_PREFIX, None))
    checked = runs = 0
    for pat, flags, alphabet in cases:
        rx = B.GpuRegex(pat, syntax_flags=flags, engine=B.LC_ENGINE_TDFA)
        assert rx.table(B.LC_TABLE_TDFA_L2_BLOB, np.uint32) is not None, pat[:40]
        it = TdfaL2BlobInterp(rx)
        for st in range(1, it.nstates):                       # the quiet table says what the transition table says
            for c in range(min(it.ncls, 64)):
                assert bool((int(it.quiet[st]) >> c) & 1) == (int(it.trans[st * it.ncls + c]) == st), (st, c)
        if it.absorb:
            assert all(int(it.trans[it.absorb * it.ncls + c]) == it.absorb for c in range(it.ncls)) and int(it.final_id[it.absorb]) != 0xFFFF
        if alphabet:
            subs = [bytes(rng.choice(alphabet) for _ in range(rng.randint(0, 60))) for _ in range(400)]
            subs += [b"ab" * 9 + b"a" + b"b" * 12 + b"c" * k + b"d" * (k // 2) for k in (1, 2, 255, 256, 257, 600)]
            subs += [b"q" * k + b"x" + b"ab" * 5 + b"a" + b"b" * 11 + b" " * j + b"y" + b"tail" * 70 for k in (0, 3, 254, 255, 256, 700) for j in (0, 257)]
            subs += [b"ab" * 4 + b"c" + b"abcabcabcabc" + b"7" * k + b"-" + b"abc" * k for k in (1, 85, 86, 300)]
        else:
            subs = [b"<34>1 - host%d app 1 ID%d - " % (k, k) + b" ".join(rng.choice([b"alpha", b"beta7", b"x=1"]) for _ in range(rng.randint(0, 200)))
                    for k in range(60)]
            subs += [b"<34>1 2014-10-11T22:14:15.003Z h a 1 I [x y=\"1\"] tail\nmore", b"no match here", b""]
        o = OracleRegex(pat, 0) if not flags else None
        for s in subs:
            a, b = it.fullmatch(s), it.fullmatch_wave(s)
            assert a == b, (pat[:40], s[:80])
            checked += 1
            runs += a is not None
            if o is not None:
                want = o.fullmatch(s)
                assert a == (None if want is None else [v for be in want for v in be][2:]), s
    assert checked > 1300 and runs > 80


QUASI_PATTERNS = [
    r"(\w+): An (\w+) (.*) SA \(SPI= (.*?)\) between (\d+) and (\d+)",      # GREEDYDATA in the middle, a literal behind it
    r"\((?:Primary|Secondary)\) Monitoring on [Ii]nterface (.*) waiting",
    r"Group = (.*), IP = (\d+), NAT",
    r"a(.*)bcd(.*)bce",
    r"(?:denied|discarded|dropped) (\w+) src (.*?):(\d+) dst",                 # lazy field, several alternatives at the start
    r"x([^,]*),([^,]*),y",
]


def test_doomed_spawns_are_skipped_without_changing_a_result():
    """NF_OFF_QUASI (regex_handle.cpp packNfaBlob): a byte on which a thread's only lasting move is its own clean self loop -- every
    other path that passes leads to a position no follow path of which takes the NEXT byte -- is as steady as a self-loop-only byte.
    The NFA walk with the shortcut (what nfa_match_kernel does) against the walk without any shortcut and the oracle, full match,
    search and anchored search, on subjects built from the patterns' own words; and the tables exist for these shapes."""
    import random
    rng = random.Random(23)
    words = ["IPSEC:", "An", "outbound", "SA", "S", "SP", "(SPI=", "0x1f)", "between", "12", "and", "34", " ", "(Primary)", "(Secondary)", "Monitoring",
             "on", "interface", "Interface", "waiting", "wait", "w", "Group", "=", ",", "IP", "NAT", "a", "b", "bc", "bcd", "bce", "denied", "den",
             "discarded", "dropped", "tcp", "src", "dst", "in:", "7", "x", "y", "x1,2,y", ",y"]
    rows = checked = 0
    for pat in QUASI_PATTERNS:
        for flags in (0, B.LC_SYNTAX_SEARCH, B.LC_SYNTAX_SEARCH | B.LC_SYNTAX_PREFIX):
            rx = B.GpuRegex(pat, syntax_flags=flags, engine=B.LC_ENGINE_NFA)
            it = NfaInterp(rx)
            rows += 0 if it.quasi_rows is None else len(it.quasi_rows)
            o = OracleRegex(pat)
            for _ in range(120):
                s = " ".join(rng.choice(words) for _ in range(rng.randint(1, 40))).encode()
                if rng.random() < 0.5:   # half of the subjects contain a real match
                    s = rng.choice([b"", b"junk "]) + {0: b"IPSEC: An outbound tunnel mode SA (SPI= 0x7) between 1 and 2", 1: b"(Primary) Monitoring on interface " + s + b" waiting",
                                                        2: b"Group = " + s + b", IP = 9, NAT", 3: b"a" + s + b"bcd" + s + b"bce", 4: b"denied tcp src " + s + b":80 dst",
                                                        5: b"x1,2,y"}[QUASI_PATTERNS.index(pat)] + rng.choice([b"", b" tail " + s])
                fast, slow = it.fullmatch(s), it.fullmatch(s, steady=False)
                assert fast == slow, (pat, flags, s)
                if flags == 0:
                    want = o.fullmatch(s)
                    exp = None if want is None else [v for be in want for v in be][2:]
                elif flags == B.LC_SYNTAX_SEARCH:
                    want = o.search(s)
                    exp = None if want is None else [v for be in want for v in be]
                else:
                    exp = fast   # (anchored search: the two walks against each other)
                assert fast == exp, (pat, flags, s, fast, exp)
                checked += 1
    assert rows >= 10 and checked > 2000


def test_a_construction_that_failed_on_its_limits_is_not_repeated(golden_dir):
    """An anchored Grok format that does not determinise within the table format's 65 535 states costs seconds of a core to find
    out; the verdict is remembered for the life of the process (lc_regex_compile), so a pipeline reload does not pay again.  Same
    return code and message both times; a handle that builds is not affected."""
    import time
    from loongcollector_amd.grok import Grok
    with open(os.path.join(golden_dir, "grok_config3.json"), encoding="utf-8") as f:
        cfg3 = json.load(f)
    g = Grok(Match=["%{CISCOFW302020_302021}", "%{CISCOFW106021}"], CustomPatterns=cfg3["custom_patterns"], AnchoredFirst=False)
    flags = (B.LC_SYNTAX_SEARCH | B.LC_SYNTAX_NAMED_ONLY | B.LC_SYNTAX_NO_DOTALL | B.LC_SYNTAX_NO_MULTILINE | B.LC_SYNTAX_REGEXP2 |
             B.LC_SYNTAX_PREFIX)
    took, said = [], []
    hopeless = g.expanded(0).encode() + b"(?:qz7-memo-test)?"   # (a pattern no other test of this process has compiled before)
    for _ in range(2):
        t0 = time.perf_counter()
        with pytest.raises(B.RegexUnsupportedError) as e:
            B.GpuRegex(hopeless, syntax_flags=flags, engine=B.LC_ENGINE_TDFA)
        took.append(time.perf_counter() - t0)
        said.append(str(e.value))
    assert said[0] == said[1] and "with its tables in global memory: tdfa: state limit exceeded" in said[0]
    assert took[1] * 5 < took[0], took
    sizes = []
    for _ in range(2):
        rx = B.GpuRegex(g.expanded(1).encode(), syntax_flags=flags, engine=B.LC_ENGINE_TDFA)
        sizes.append(rx.table(B.LC_TABLE_TDFA_L2_BLOB, np.uint32).tobytes())
    assert sizes[0] == sizes[1] and len(sizes[0]) > 100000


def test_follow_lists_by_byte_class_are_the_follow_lists_filtered(golden_dir, monkeypatch):
    """device_tables.h NF_OFF_CSTART / NF_OFF_CPATHS (round 5, opt-in LC_NFA_CLASS_LISTS=1: measured, no gain on configs[2]): for every
    position p and byte class c the class list is exactly the sub-list of p's follow paths whose target takes c -- same order (the order
    is the priority), MATCH paths left out -- and the part of the blob in front of the lists (NF_STAGE_BYTES: what an LDS kernel stages)
    holds every other table.  Without the variable no lists are packed."""
    from loongcollector_amd import corpus
    plain = B.GpuRegex(rb"(a|ab)*c", engine=B.LC_ENGINE_NFA).table(B.LC_TABLE_NFA_BLOB, np.uint32)
    assert int(plain[24]) == 0 and int(plain[26]) == len(plain) * 4
    monkeypatch.setenv("LC_NFA_CLASS_LISTS", "1")
    with open(os.path.join(golden_dir, "regex_golden.json")) as f:
        golden = json.load(f)
    patterns = [c["p"].encode("latin-1") for c in golden["cases"][::9]]
    patterns += [corpus.REGEX_A.encode() if isinstance(corpus.REGEX_A, str) else corpus.REGEX_A,
                 rb"(?:(?:[0-9a-f]{1,4}:){7}[0-9a-f]{1,4}|(?:\d{1,3}\.){3}\d{1,3}) (\w+)=(\S*)", rb"(a|ab|abc)*(?>x+)y"]
    checked = 0
    for pat in patterns:
        for flags in (0, B.LC_SYNTAX_SEARCH):
            try:
                rx = B.GpuRegex(pat, syntax_flags=flags, engine=B.LC_ENGINE_NFA)
            except Exception:
                continue
            blob = rx.table(B.LC_TABLE_NFA_BLOB, np.uint32)
            it = NfaInterp(rx)
            off_cs, off_cp, stage = int(blob[24]), int(blob[25]), int(blob[26])
            assert off_cs and off_cp and stage % 16 == 0 and stage <= off_cs < off_cp
            assert all(int(blob[k]) < stage for k in (4, 5, 6, 7, 11, 12, 13, 17))      # every other table lies in the staged part
            fs = blob[int(blob[6]) // 4:int(blob[6]) // 4 + it.npos + 2]
            cstart = blob[off_cs // 4:off_cs // 4 + (it.npos + 1) * it.ncls + 1]
            cpaths = blob[off_cp // 4:]
            for p in range(it.npos + 1):
                for c in range(it.ncls):
                    want = [int(fs[p]) + k for k, (tgt, _, _) in enumerate(it.follow[p]) if tgt >= 0 and (it.posmask[tgt] >> c) & 1]
                    lo, hi = int(cstart[p * it.ncls + c]), int(cstart[p * it.ncls + c + 1])
                    assert [int(x) for x in cpaths[lo:hi]] == want, (pat, p, c)
                    checked += 1
    assert checked > 20000


def test_a_spawn_that_leaves_an_atomic_group_is_not_a_doomed_spawn():
    """Round 6's forced-engine device fuzz (profiles/round6_bt_fuzz_gpu.txt D): '(?>a+?.)' on "aa1" matched on the thread-list kernel
    because the doomed-spawn row of the loop position skipped the step on which the spawned '.' thread leaves -- commits -- the atomic
    group and ends the loop's thread.  The rows (regex_handle.cpp packNfaBlob, NF_OFF_QUASI) must not cover such a spawn; the plain form
    keeps its row.  The device side of this is tests/test_gpu_parity.py::test_atomic_lazy_loop_commits_on_the_thread_list_engine."""
    for p in (b'(?>a+?.)', b'(?>(?:(c)|(?:a)+?).)', b'(?>(?:c|a+?).)', b'x(?>b*?[ab])y'):
        rx = B.GpuRegex(p, engine=B.LC_ENGINE_NFA)
        assert rx.atomic_groups()[0] >= 1
        assert NfaInterp(rx).quasi_rows is None, p
    assert NfaInterp(B.GpuRegex(b'a+?.', engine=B.LC_ENGINE_NFA)).quasi_rows is not None
