"""The CPU oracle (oracle/bt_regex.c) against the committed golden vectors.

regex_golden.json holds full-match capture offsets on which CPython `re` and PCRE1 8.45 agree
(generator: tests/golden/gen_regex_golden.py).  The oracle restates boost::regex_match as called at
core/common/StringTools.cpp:183-211; boost itself is unavailable offline (parity unpinned against boost).
"""
import json
import os

import pytest

from oracle.oracle import OracleRegex


def _load(golden_dir):
    with open(os.path.join(golden_dir, "regex_golden.json")) as f:
        return json.load(f)


def test_oracle_matches_every_golden_vector(golden_dir):
    d = _load(golden_dir)
    assert d["n_cases"] > 4000 and d["n_matched"] > 1000
    bad = []
    for c in d["cases"]:
        rx = OracleRegex(c["p"].encode("latin-1"))
        assert rx.groups == c["g"], c["p"]
        for subj, flat in c["subs"]:
            got = rx.fullmatch(subj.encode("latin-1"))
            got_flat = None if got is None else [v for ab in got for v in ab]
            if got_flat != flat:
                bad.append((c["p"], subj, got_flat, flat))
    assert not bad, bad[:5]


def test_survey_known_answer_vectors():
    # SURVEY.md appendix B probes (PCRE1 8.45 + Python re.fullmatch agree)
    ra = OracleRegex(r'([\d\.]+) \S+ \S+ \[(\S+) \S+\] \"(\w+) ([^\\"]*)\" ([\d\.]+) (\d+) (\d+) (\d+|-) \"([^\\"]*)\" \"([^\\"]*)\"')
    line = (b'127.0.0.1 - - [07/Jul/2022:10:43:30 +0800] "POST /PutData?Category=YunOsAccountOpLog" '
            b'0.024 18204 200 37 "-" "aliyun-sdk-java"')
    assert ra.fullmatch(line)[1:] == [(0, 9), (15, 35), (44, 48), (49, 84), (86, 91), (92, 97), (98, 101), (102, 104),
                                      (106, 107), (110, 125)]
    rb = OracleRegex(r'^([^ ]*) ([^ ]*) ([^ ]*) \[([^\]]*)\] "(\S+) ([^\"]*) (\S*)" ([^ ]*) ([^ ]*) "([^\"]*)" "([^\"]*)"')
    l2 = (b'203.0.113.45 - - [25/Jun/2024:23:59:59 +0000] "GET /wp-admin/admin-ajax.php?action=x HTTP/1.1" 200 1847 '
          b'"https://www.google.com/" "Mozilla/5.0 (Windows NT 10.0; Win64; x64)"')
    assert rb.fullmatch(l2)[1:] == [(0, 12), (13, 14), (15, 16), (18, 44), (47, 50), (51, 84), (85, 93), (95, 98),
                                    (99, 103), (105, 128), (131, 172)]


@pytest.mark.parametrize("pat", [r"(a)\2", r"(?<!a+b)c", r"(?R)b", r"(?(1)a|b)", r"(", r"a)", r"[a", r"a**", r"\p{L}"])
def test_unsupported_or_invalid_patterns_fail_to_compile(pat):
    with pytest.raises(ValueError):
        OracleRegex(pat)


def test_boost_line_separator_rules():
    # restated from Boost.Regex perl_matcher::match_start_line/match_end_line (knowledge, unpinned):
    # separators are \n \r \f; no match between \r and \n.
    assert OracleRegex(r"a$\r\nb").fullmatch(b"a\r\nb") is not None
    assert OracleRegex(r"a\r$\nb").fullmatch(b"a\r\nb") is None
    assert OracleRegex(r"a\r\n^b").fullmatch(b"a\r\nb") is not None
    assert OracleRegex(r"a\r^\nb").fullmatch(b"a\r\nb") is None
    assert OracleRegex(r"a\f^b").fullmatch(b"a\fb") is not None


def test_complexity_budget_reports_failure():
    # boost throws std::runtime_error on pathological backtracking; StringTools.cpp:200-205 -> parse failure
    rx = OracleRegex(r"(a|aa)+(a|aa)+(a|aa)+(a|aa)+(a|aa)+(a|aa)+b")
    with pytest.raises(RuntimeError):
        rx.fullmatch(b"a" * 64)


def test_oracle_search_matches_every_search_vector(golden_dir):
    """leftmost-first search (Go processor_regex without FullMatch, plugins/processor/regex/regex.go:105-129)"""
    with open(os.path.join(golden_dir, "regex_search_golden.json")) as f:
        d = json.load(f)
    assert d["n_cases"] > 1000
    bad = []
    for c in d["cases"]:
        rx = OracleRegex(c["p"].encode("latin-1"))
        for subj, flat in c["subs"]:
            got = rx.search(subj.encode("latin-1"))
            got_flat = None if got is None else [v for ab in got for v in ab]
            if got_flat != flat:
                bad.append((c["p"], subj, got_flat, flat))
    assert not bad, bad[:5]


def test_oracle_matches_atomic_and_possessive_vectors(golden_dir):
    """(?>X), X*+ ... : vectors on which the `regex` module and PCRE1 agree (tests/golden/gen_atomic_golden.py)."""
    with open(os.path.join(golden_dir, "regex_atomic_golden.json")) as f:
        d = json.load(f)
    assert d["n_full"] > 2500 and d["n_search"] > 1000
    bad = []
    for kind in ("full", "search"):
        for c in d[kind]:
            rx = OracleRegex(c["p"].encode("latin-1"))
            assert rx.groups == c["g"], c["p"]
            for subj, flat in c["subs"]:
                s = subj.encode("latin-1")
                got = rx.fullmatch(s) if kind == "full" else rx.search(s)
                got_flat = None if got is None else [v for ab in got for v in ab]
                if got_flat != flat:
                    bad.append((kind, c["p"], subj, got_flat, flat))
    assert not bad, bad[:5]


def test_oracle_matches_fixed_length_lookaround_vectors(golden_dir):
    """(?=lit) (?!lit) (?<=lit) (?<!lit) with a fixed-length body: vectors on which the `regex` module and PCRE1 agree
    (tests/golden/gen_lookaround_golden.py; the library's MONGO_QUERY shape among them).  The look-behind form is new in round 5
    (bt_regex.c I_BACK: the body runs k bytes back, as the backtracking engines do it)."""
    with open(os.path.join(golden_dir, "regex_lookaround_golden.json")) as f:
        d = json.load(f)
    assert d["n_full"] > 2000 and d["n_search"] > 1000 and d["dropped_disagreements"] == 0
    bad = []
    for kind in ("full", "search"):
        for c in d[kind]:
            rx = OracleRegex(c["p"].encode("latin-1"))
            assert rx.groups == c["g"], c["p"]
            for subj, flat in c["subs"]:
                s = subj.encode("latin-1")
                got = rx.fullmatch(s) if kind == "full" else rx.search(s)
                got_flat = None if got is None else [v for ab in got for v in ab]
                if got_flat != flat:
                    bad.append((kind, c["p"], subj, got_flat, flat))
    assert not bad, bad[:5]


def test_reference_boost_regex_search_vectors(golden_dir):
    """The reference's own boost-boundary vectors (core/unittest/common/StringToolsUnittest.cpp:128-209, TestBoostRegexSearch:
    regex_search with match_continuous): the oracle's prefix match, and -- the same question, compiled as LC_SYNTAX_PREFIX --
    the device tables walked by the CPU interpreters of the tests."""
    from loongcollector_amd import binding as B
    from tests.helpers.table_interp import NfaInterp, TdfaInterp
    with open(os.path.join(golden_dir, "boost_search_vectors.json")) as f:
        d = json.load(f)
    n = 0
    for c in d["cases"]:
        o = OracleRegex(c["p"].encode("latin-1"))
        rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=B.LC_SYNTAX_PREFIX)
        its = ([NfaInterp(rx)] if rx.has_nfa_program() else []) + ([TdfaInterp(rx)] if rx.info()["engine"] == B.LC_ENGINE_TDFA else [])
        assert its
        for subj, want in c["subs"]:
            s = subj.encode("latin-1")
            assert (o.prefixmatch(s) is not None) == want, (c["cite"], subj)
            for it in its:
                n += 1
                assert (it.fullmatch(s) is not None) == want, (c["cite"], subj, type(it).__name__)
    assert n >= 16


def test_headline_corpus_is_generated_line_by_line_and_every_line_matches():
    """std::mt19937_64, seed 20260921, every line generated on its own, exactly 512 bytes, every line a full
    match of its regex (tools/corpus_gen.cpp behind corpus.apache_lines) -- and reproducible: a prefix of a longer run is the shorter run."""
    from loongcollector_amd import corpus
    for kind, rx, groups in (("A", corpus.REGEX_A, 10), ("B", corpus.REGEX_B, 11)):
        data, off, length = corpus.apache_lines(20000, kind)
        assert data.shape == (20000 * 513,) and (length == 512).all() and int(off[-1]) == 20000 * 513
        raw = data.tobytes()
        lines = [raw[i * 513:i * 513 + 512] for i in range(20000)]
        assert all(raw[i * 513 + 512] == 10 for i in range(0, 20000, 97))
        assert len(set(lines)) == 20000                                   # no line occurs twice
        caps, status = OracleRegex(rx).fullmatch_batch(data, off[:-1], length)
        assert status.all() and caps.shape == (20000, 2 * groups)
        short, _, _ = corpus.apache_lines(500, kind)
        assert short.tobytes() == raw[:500 * 513]
        methods = [l.split(b'"')[1].split(b" ")[0] for l in lines]
        share = {m: methods.count(m) / 20000 for m in (b"GET", b"POST", b"PUT", b"DELETE")}
        assert abs(share[b"GET"] - 0.70) < 0.02 and abs(share[b"POST"] - 0.20) < 0.02 and abs(share[b"PUT"] - 0.05) < 0.01
    poisoned, off, length = corpus.apache_lines(100, "A", poison_every=7)
    _, status = OracleRegex(corpus.REGEX_A).fullmatch_batch(poisoned, off[:-1], length)
    assert [int(s) for s in status] == [0 if i % 7 == 0 else 1 for i in range(100)]
