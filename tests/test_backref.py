"""Back-references (\\1 .. \\N) -- the device BACKTRACKING engine (csrc/bt_vm.hpp, LC_ENGINE_BT; round 6).

What the reference does: boost::regex_match backtracks, so a pattern with back-references is just another pattern
(core/common/StringTools.cpp:183-211; Init accepts whatever boost compiles, core/common/ParamExtractor.cpp:199-209).

CPU tests (no GPU): the oracle against the vectors three independent engines agree on (tests/golden/gen_backref_golden.py), and the
product's PROGRAMS (lc_regex_table LC_TABLE_BT_BLOB) walked by the very routine the kernel runs per lane, compiled for the host
(tests/native/bt_host_check.cpp) -- on the back-reference vectors, and, with the engine asked for explicitly, on every regular golden
set the automata are pinned on.  GPU tests: the kernel through the C ABI on the same vectors, against the oracle on generated values,
and through the processor.
"""
import ctypes
import json
import os
import random
import subprocess

import numpy as np
import pytest

from loongcollector_amd import binding as B
from oracle.oracle import OracleRegex

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vectors(golden_dir):
    with open(os.path.join(golden_dir, "backref_vectors.json")) as f:
        d = json.load(f)
    # (the 7 dropped vectors: PCRE1 8.45 takes "(?=.*\\d)..." for an anchored pattern and misses two searches; the `regex` module keeps a
    # capture made inside a NEGATIVE look-ahead where CPython and PCRE1 -- and the oracle -- undo it)
    assert d["n_full"] > 3500 and d["n_search"] > 1500 and d["n_named_full"] >= 29 and d["dropped_disagreements"] <= 7
    assert d["n_look_full"] >= 60 and d["n_look_search"] >= 120 and d["n_cond_full"] >= 25
    return d


@pytest.fixture(scope="module")
def host_vm():
    """tests/native/bt_host_check.cpp -> tests/_build/libbt_host_check.so (g++, host only)"""
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(ROOT, "tests", "native", "bt_host_check.cpp")
    hdr = os.path.join(ROOT, "loongcollector_amd", "csrc", "bt_vm.hpp")
    lib = os.path.join(out_dir, "libbt_host_check.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", lib, src])
    L = ctypes.CDLL(lib)
    L.bt_host_run.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32,
                              ctypes.c_uint32, ctypes.c_uint32]
    L.bt_host_run.restype = ctypes.c_int

    def run(rx, subject, budget=1 << 22, scratch_words=16384):
        blob = rx.table(B.LC_TABLE_BT_BLOB, np.uint32)
        assert blob is not None and len(blob) > 8
        ncaps = int(blob[2])
        caps = np.full(ncaps, -9, dtype=np.int32)
        r = L.bt_host_run(blob.ctypes.data, subject, len(subject), 0, caps.ctypes.data, ncaps, scratch_words, budget)
        return r, caps.tolist()
    return run


def _flat(got):
    return None if got is None else [v for ab in got for v in ab]


# ------------------------------------------------------------------------------------------------ the oracle is pinned first
def test_oracle_matches_the_backreference_vectors(vectors):
    bad = []
    for kind in ("full", "search", "icase_full", "named_full", "look_full", "look_search", "cond_full", "cond_search"):
        for c in vectors[kind]:
            rx = OracleRegex(c["p"].encode("latin-1"), flags=(1 if kind == "icase_full" else 0))   # ORX_ICASE
            assert rx.groups == c["g"], c["p"]
            for subj, flat in c["subs"]:
                s = subj.encode("latin-1")
                got = rx.search(s) if kind.endswith("search") else rx.fullmatch(s)
                if _flat(got) != flat:
                    bad.append((kind, c["p"], subj, _flat(got), flat))
    assert not bad, bad[:5]


# ------------------------------------------------------------------------------------------------ the product's compile half
def test_patterns_with_backreferences_compile_to_the_backtracking_engine():
    rx = B.GpuRegex(rb'(\w+) \1')
    assert rx.info()["engine"] == B.LC_ENGINE_BT and rx.groups == 1
    assert not rx.has_nfa_program()
    # a search shifts the references with the groups: group 1 is the whole match, \1 now names group 2
    rs = B.GpuRegex(rb'(a)\1', syntax_flags=B.LC_SYNTAX_SEARCH)
    assert rs.info()["engine"] == B.LC_ENGINE_BT and rs.groups == 2
    # boost: error_backref at compile time -> Init fails as for any invalid regex
    with pytest.raises(B.RegexSyntaxError):
        B.GpuRegex(rb'(a)\2')
    # an automaton cannot run it: asking for one is refused, never approximated
    with pytest.raises(B.RegexUnsupportedError):
        B.GpuRegex(rb'(a)\1', engine=B.LC_ENGINE_TDFA)
    # by name and relative (boost Perl syntax): resolved to the group's number at compile time
    assert B.GpuRegex(rb'(?<w>\w+) \k<w>').info()["engine"] == B.LC_ENGINE_BT
    with pytest.raises(B.RegexUnsupportedError):
        B.GpuRegex(rb'\k<later>(?<later>a)')
    # Grok's dialect (named-only numbering here, regexp2 numbers unnamed groups first): by NAME only, and one group per name
    with pytest.raises(B.RegexUnsupportedError):
        B.GpuRegex(rb'(?<w>\w+) \1', syntax_flags=B.LC_SYNTAX_NAMED_ONLY | B.LC_SYNTAX_REGEXP2)
    assert B.GpuRegex(rb'(?<w>\w+) \k<w>', syntax_flags=B.LC_SYNTAX_NAMED_ONLY | B.LC_SYNTAX_REGEXP2).info()["engine"] == B.LC_ENGINE_BT
    with pytest.raises(B.RegexUnsupportedError):
        B.GpuRegex(rb'(?<w>a)|(?<w>b)\k<w>', syntax_flags=B.LC_SYNTAX_NAMED_ONLY | B.LC_SYNTAX_REGEXP2)
    # the Go regex plugin compiles with Go's regexp (RE2): no back-references at all
    with pytest.raises(B.RegexUnsupportedError):
        B.GpuRegex(rb'(?<w>\w+) \k<w>', syntax_flags=B.LC_SYNTAX_REGEXP2)
    # regular patterns still get their automata
    assert B.GpuRegex(rb'(a)b').info()["engine"] == B.LC_ENGINE_TDFA


def test_backtracking_programs_on_the_backreference_vectors(vectors, host_vm):
    bad, checked = [], 0
    for kind, flags in (("full", 0), ("search", B.LC_SYNTAX_SEARCH), ("icase_full", B.LC_SYNTAX_ICASE), ("named_full", 0), ("look_full", 0),
                        ("look_search", B.LC_SYNTAX_SEARCH), ("cond_full", 0), ("cond_search", B.LC_SYNTAX_SEARCH)):
        for c in vectors[kind]:
            # (a look-around the automata can run -- a window of byte classes -- keeps its automaton: here the engine is asked for)
            rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=flags, engine=B.LC_ENGINE_AUTO if "look" not in kind else B.LC_ENGINE_BT)
            assert rx.info()["engine"] == B.LC_ENGINE_BT, c["p"]
            assert rx.groups == c["g"] + (1 if kind.endswith("search") else 0)
            for subj, flat in c["subs"]:
                s = subj.encode("latin-1")
                r, caps = host_vm(rx, s)
                checked += 1
                if flat is None:
                    ok = r == 0
                else:  # slots 0/1 are the wrapped whole-value match; a search's group 1 is the golden group 0
                    ok = r == 1 and caps[:2] == [0, len(s)] and caps[2:] == (flat if kind.endswith("search") else flat[2:])
                if not ok:
                    bad.append((kind, c["p"], subj, r, caps, flat))
    assert checked > 5000
    assert not bad, bad[:5]


@pytest.mark.parametrize("name,key", [("regex_golden.json", "cases"), ("regex_atomic_golden.json", "full"), ("regex_atomic_golden.json", "search"),
                                      ("regex_search_golden.json", "cases"), ("regex_lookaround_golden.json", "full"),
                                      ("regex_lookaround_golden.json", "search")])
def test_backtracking_programs_agree_with_the_goldens_of_the_automata(golden_dir, host_vm, name, key):
    """LC_ENGINE_BT asked for explicitly: the same vectors the tagged DFA and the thread-list engine are pinned on (CPython re ∧ PCRE1,
    regex ∧ PCRE1).  Patterns the engine does not run (multi-byte look-around windows) are refused at compile time and skipped."""
    with open(os.path.join(golden_dir, name)) as f:
        d = json.load(f)
    search = key == "search" or "search" in name
    bad, checked, refused, gave_up = [], 0, 0, 0
    for c in d[key]:
        try:
            rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=B.LC_SYNTAX_SEARCH if search else 0, engine=B.LC_ENGINE_BT)
        except B.RegexUnsupportedError:
            refused += 1
            continue
        for subj, flat in c["subs"]:
            s = subj.encode("latin-1")
            r, caps = host_vm(rx, s)
            checked += 1
            if r == -1:  # (the fuzzed sets hold patterns that backtrack exponentially: out of the device's step budget = reported)
                gave_up += 1
                continue
            ok = (r == 0) if flat is None else (r == 1 and caps[2:] == (flat if search else flat[2:]))
            if not ok:
                bad.append((c["p"], subj, r, caps, flat))
    assert checked > 1000 and refused * 10 < len(d[key]) and gave_up * 200 < checked, (checked, refused, gave_up)
    assert not bad, bad[:5]


def test_budget_and_stack_are_reported_never_guessed(host_vm):
    rx = B.GpuRegex(rb'(a|aa)+\1b')
    s = b'a' * 40 + b'c'
    assert host_vm(rx, s, budget=1 << 14)[0] == -1          # exponential: out of steps -> gave up
    assert host_vm(B.GpuRegex(rb'(?:(a)|b)*\1'), b'ab' * 600 + b'a', scratch_words=256)[0] == -2   # out of stack: a second pass, then gave up
    r, caps = host_vm(B.GpuRegex(rb'(?:(a)|b)*\1'), b'ab' * 600 + b'a')   # (with the whole slice it is decided)
    assert r == 1 and caps == [0, 1201, 1198, 1199]


# ------------------------------------------------------------------------------------------------ the kernel
@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    torch.cuda.set_device(0)
    return torch


@pytest.mark.gpu
def test_backreference_vectors_through_the_c_abi(torch_dev, vectors):
    from test_gpu_parity import pack, run_device
    bad, checked = [], 0
    for kind, flags in (("full", 0), ("search", B.LC_SYNTAX_SEARCH), ("icase_full", B.LC_SYNTAX_ICASE), ("named_full", 0), ("look_full", 0),
                        ("look_search", B.LC_SYNTAX_SEARCH), ("cond_full", 0), ("cond_search", B.LC_SYNTAX_SEARCH)):
        for c in vectors[kind]:
            rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=flags)
            subs = [s.encode("latin-1") for s, _ in c["subs"]]
            data, off, length = pack(subs)
            caps, status = run_device(torch_dev, rx, data, off, length)
            for i, (_, flat) in enumerate(c["subs"]):
                checked += 1
                exp = flat if kind.endswith("search") or flat is None else flat[2:]
                ok = (status[i] == B.LC_NOMATCH and (caps[i] == -1).all()) if exp is None else (
                    status[i] == B.LC_MATCH and list(caps[i]) == exp)
                if not ok:
                    bad.append((kind, c["p"], subs[i], int(status[i]), list(caps[i]), exp))
    assert checked > 5000
    assert not bad, bad[:5]


@pytest.mark.gpu
def test_backtracking_kernel_against_the_oracle_on_generated_lines(torch_dev):
    """20 000 lines of 40-600 B per pattern: key=value records whose values repeat or do not; bit-exact status and offsets; both input
    forms (offsets + separator, offsets + lengths); a batch beyond the 8 192 lanes of a launch (grid-stride)."""
    from test_gpu_parity import run_device
    rng = random.Random(7)
    words = [b"alpha", b"beta", b"gamma", b"x1", b"req-42", b"GET", b"POST", b"a.b.c", b"0a:1b"]
    for pattern in (rb'(\S+) (\S+) \[(\w+)\] "(\S+) (.*?)" (\d+) \1 (.*)', rb'(\w+)=(\w+);(?:\w+=\w+;)*?\1=\2;.*',
                    rb'(["\'])(.*?)\1 (\S+) \3( .*)?'):
        lines = []
        for _ in range(20000):
            a, b, c = rng.choice(words), rng.choice(words), rng.choice(words)
            pad = b" ".join(rng.choice(words) for _ in range(rng.randint(0, 80)))
            shape = rng.randint(0, 5)
            if shape == 0:
                lines.append(a + b" " + b + b" [" + c.replace(b".", b"").replace(b"-", b"").replace(b":", b"") + b"] \"" + a + b" /p?" + pad + b"\" 200 " +
                             (a if rng.random() < 0.7 else b) + b" " + pad)
            elif shape == 1:
                k, v = rng.choice([b"k", b"key", b"id"]), rng.choice([b"v", b"7", b"val"])
                mid = b"".join(rng.choice([b"p=q;", b"id=8;", b"k=v;"]) for _ in range(rng.randint(0, 6)))
                lines.append(k + b"=" + v + b";" + mid + (k + b"=" + v if rng.random() < 0.6 else b"z=z") + b";" + pad)
            elif shape == 2:
                q = rng.choice([b'"', b"'"])
                lines.append(q + pad[:rng.randint(0, 40)] + (q if rng.random() < 0.8 else b"`") + b" " + a + b" " + (a if rng.random() < 0.7 else b) +
                             (b" " + pad if rng.random() < 0.5 else b""))
            else:
                lines.append(pad)
        data = np.frombuffer(b"\n".join(lines) + b"\n", dtype=np.uint8)
        length = np.array([len(x) for x in lines], dtype=np.uint32)
        off = np.zeros(len(lines) + 1, dtype=np.uint32)
        off[1:] = np.cumsum(length + 1)
        exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:-1], length)
        rx = B.GpuRegex(pattern)
        assert rx.info()["engine"] == B.LC_ENGINE_BT
        caps, status = run_device(torch_dev, rx, data, off, None, sep=1)
        assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps), pattern
        caps2, status2 = run_device(torch_dev, rx, data, off[:-1], length)
        assert np.array_equal(status2, exp_status) and np.array_equal(caps2, exp_caps), pattern
        assert 1000 < int((exp_status == 1).sum()) < 19000


@pytest.mark.gpu
def test_the_backtracking_engine_agrees_with_the_automata_on_the_bench_corpus(torch_dev):
    """Regex A of the headline on 20 000 corpus lines (every seventh poisoned): LC_ENGINE_BT asked for explicitly against the oracle."""
    from loongcollector_amd import corpus
    from test_gpu_parity import run_device
    data, off, length = corpus.apache_batch(20000, "A", poison_every=7)
    exp_caps, exp_status = OracleRegex(corpus.REGEX_A).fullmatch_batch(data, off[:-1], length)
    rx = B.GpuRegex(corpus.REGEX_A, engine=B.LC_ENGINE_BT)
    caps, status = run_device(torch_dev, rx, data, off, None, sep=1)
    assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)


@pytest.mark.gpu
def test_a_value_that_exhausts_the_budget_is_reported_gave_up(torch_dev, monkeypatch):
    from test_gpu_parity import pack, run_device
    monkeypatch.setenv("LC_BT_BUDGET", str(1 << 14))
    rx = B.GpuRegex(rb'(a|aa)+\1b')
    subs = [b'a' * 40 + b'c', b'aab', b'aaab', b'c']
    data, off, length = pack(subs)
    caps, status = run_device(torch_dev, rx, data, off, length)
    assert list(status) == [B.LC_GAVE_UP, B.LC_MATCH, B.LC_MATCH, B.LC_NOMATCH]
    assert (caps[0] == -1).all() and (caps[3] == -1).all()
    o = OracleRegex(rb'(a|aa)+\1b')
    assert list(caps[1]) == list(o.fullmatch(subs[1])[1]) and list(caps[2]) == list(o.fullmatch(subs[2])[1])


@pytest.mark.gpu
def test_the_parse_processor_with_a_backreference_regex_against_the_processor_oracle():
    """ProcessorParseRegexNative::Init accepts what boost compiles (ProcessorParseRegexNative.cpp:64-67): a Regex with \\1 initialises, and
    1 000-event groups go through gather -> bt_match_kernel -> stitch with the reference's key / policy behaviour (the processor oracle)."""
    from loongcollector_amd.processor import EventGroup, Processor
    from oracle.processor_oracle import LogEventModel, ProcessorOracle
    rng = random.Random(3)
    cfg = {"SourceKey": "content", "Regex": r'(\w+)=(["\'])(.*?)\2 (\w+) \1:(\d+)(?: .*)?', "Keys": ["key", "quote", "value", "verb", "n"],
           "KeepingSourceWhenParseFail": True, "KeepingSourceWhenParseSucceed": False, "CopingRawLog": False, "RenamedSourceKey": "rawLog"}
    p = Processor(cfg)
    po = ProcessorOracle(cfg)
    total_ok = 0
    for _ in range(3):
        events, models = [], []
        for _ in range(1000):
            k = rng.choice(["id", "user", "k9"])
            q = rng.choice(['"', "'"])
            v = "".join(rng.choice("abc \"'=") for _ in range(rng.randint(0, 30)))
            line = "%s=%s%s%s %s %s:%d%s" % (k, q, v, q if rng.random() < 0.85 else "`", rng.choice(["GET", "PUT"]),
                                             k if rng.random() < 0.8 else "zz", rng.randint(0, 999), " tail " * rng.randint(0, 3))
            contents = [["content", line]]
            events.append({"contents": contents, "timestamp": 1, "type": 1})
            models.append(LogEventModel([(a, b.encode()) for a, b in contents]))
        g = EventGroup({"events": events})
        p.process(g)
        out = po.process_group(models)
        want = [None if ev is None else [(a, b.decode()) for a, b in ev.live()] for ev in out]
        assert g.contents() == want
        total_ok += sum(1 for ev in want if ev and any(a == "verb" for a, _ in ev))
    assert 500 < total_ok < 2900


@pytest.mark.gpu
def test_multiline_split_with_backreference_patterns_against_the_oracle():
    """ProcessorSplitMultilineLogStringNative's patterns are boost regexes like any other (MultilineOptions.cpp:203-266): a StartPattern /
    EndPattern with a back-reference flags the lines through bt_match_kernel (prefix match, LC_SYNTAX_PREFIX) and the records are the
    multiline oracle's."""
    from loongcollector_amd.multiline import Multiline
    from oracle.multiline_oracle import MultilineOracle
    rng = random.Random(5)
    pool = [b"<a> open", b"</a>", b"</b>", b"<bb> x", b"</bb> tail", b"11-11 start", b"12-11 not", b"  cont", b"", b"77-77", b"x=1;x=1", b"x=1;y=1"]
    for config in ({"StartPattern": r"(\d\d)-\1.*"}, {"StartPattern": r"<(\w+)>.*", "EndPattern": r"</(\w)\1?>.*"},
                   {"EndPattern": r"(\w)=(\d);\1=\2", "UnmatchedContentTreatment": "discard"}):
        o, m = MultilineOracle(**config), Multiline(**config)
        for _ in range(150):
            val = b"\n".join(rng.choice(pool) for _ in range(rng.randint(0, 14)))
            assert m.split(val) == o.split(val), (config, val)


@pytest.mark.gpu
def test_values_that_fill_their_first_slice_are_decided_by_the_second_pass(torch_dev):
    """(?:(a)|b)*\\1 keeps an alternative and an undo record per byte: 3 000-byte values overflow the 8 KB slices of pass 1, are left
    pending and decided by pass 2 (64 KB slices); a 40 000-byte value overflows those too and is reported LC_GAVE_UP."""
    from test_gpu_parity import pack, run_device
    rx = B.GpuRegex(rb'(?:(a)|b)*\1')
    o = OracleRegex(rb'(?:(a)|b)*\1')
    subs = [b'ab' * 1500 + b'a', b'ab' * 1500 + b'b', b'aa', b'ab' * 20000 + b'a', b'b' * 3000, b'']
    data, off, length = pack(subs)
    caps, status = run_device(torch_dev, rx, data, off, length)
    for i in (0, 1, 2, 4, 5):
        want = o.fullmatch(subs[i])
        assert (status[i] == B.LC_MATCH and list(caps[i]) == list(want[1])) if want else (status[i] == B.LC_NOMATCH and (caps[i] == -1).all()), i
    assert status[3] == B.LC_GAVE_UP


@pytest.mark.gpu
def test_the_filter_with_backreference_and_lookaround_leaves_against_the_filter_oracle():
    """processor_filter_regex_native's leaves are boost regexes too (ProcessorFilterNative.cpp): leaves the automata run and leaves only
    the backtracking engine runs share one group trip (lc_regex_match_device_multi falls back per job)."""
    from loongcollector_amd.processor import EventGroup, Filter
    from oracle.filter_oracle import FilterOracle
    rng = random.Random(23)
    config = {"ConditionExp": {"operator": "and", "operands": [
        {"type": "regex", "key": "path", "exp": r"/(\w+)/\1(?:/.*)?"},
        {"operator": "or", "operands": [{"type": "regex", "key": "ua", "exp": r"(?=.*\d)[A-Za-z/.\d]+"},
                                        {"type": "regex", "key": "status", "exp": r"[23]\d\d"}]}]}}
    fields = {"path": ["/a/a", "/api/api/x", "/a/b", "/x/x/", "/xy/x", ""], "ua": ["curl/8.1", "Mozilla", "bot", "Go-http/2"],
              "status": ["200", "404", "301", "500"]}
    events = []
    for _ in range(2000):
        events.append({k: rng.choice(v) for k, v in fields.items() if rng.random() < 0.9})
    want = FilterOracle(config).process([{k: v.encode("utf-8") for k, v in e.items()} for e in events])
    f = Filter(config)
    g = EventGroup({"events": [{"contents": e, "timestamp": 1, "type": 1} for e in events]})
    f.process(g)
    d = g.to_dict()
    got = [e["contents"] for e in d["events"]] if d else []
    assert 100 < len(want) < len(events)
    assert got == [{k: v.decode("utf-8") for k, v in e.items()} for e in want]


@pytest.mark.gpu
@pytest.mark.parametrize("match", [
    [r"%{WORD:a} \k<a>( %{GREEDYDATA:rest})?", r"(?P<k>\w+)=(?P<v>\S*)", "%{INT:n}"],                # a back-reference by name in front
    [r"%{LOGLEVEL:level}:? %{GREEDYDATA:msg}", r"(?P<q>[\"'])(?P<body>.*?)\k<q>", r"%{NOTSPACE:first}"],  # ... and behind a plain entry
    [r"%{WORD:w}(?= \d+x)", r"%{WORD:other}"],                                                          # a general look-ahead
])
def test_grok_lists_with_entries_on_the_backtracking_engine_against_the_grok_oracle(torch_dev, match):
    """regexp2 backtracks (plugins/processor/grok/processor_grok.go:148-194, :343): a Match entry with \\k<name> or a general look-around is
    an entry like any other.  Such an entry runs bt_match_kernel; the handle walks its list entry by entry (the speculative plan is built
    from automata).  Fields, first-match-wins order and the FindNextMatch rounds against the Grok oracle."""
    from loongcollector_amd.grok import Grok
    from oracle.grok_oracle import GrokOracle
    rng = random.Random(77)
    words = [b"GET", b"GET", b"err", b"err", b"x=1", b"k=", b"'q s'", b"\"a\"", b"'mixed\"", b"abc_9", b"7", b"12x", b"ERROR", b"info:", b""]
    values = [rng.choice([b" ", b"  ", b","]).join(rng.choice(words) for _ in range(rng.randint(0, 8))) for _ in range(3000)]
    g = Grok(Match=match)
    assert B.LC_ENGINE_BT in [g.engine(i) for i in range(len(match))]
    o = GrokOracle(match)
    pattern, fields = g.match_host(values)
    won = 0
    for v, p, f in zip(values, pattern, fields):
        res, want = o.process_value(v)
        assert f == want, (match, v)
        assert (p >= 0) == (res == 0)
        won += p >= 0
    assert 300 < won


@pytest.mark.gpu
def test_concurrent_runner_threads_on_one_backtracking_handle():
    """Eight runner threads call Process on ONE instance whose Regex runs on the backtracking engine (the contract of
    core/collection_pipeline/queue/ProcessQueueManager.cpp:167-205): the program on the device is shared, every thread's launches have
    a scratch block of their own (gpu_runtime.hip launchBt: cached per thread and stream) -- a stack shared by two launches in flight
    would show as wrong captures here.  Groups of different sizes, so that a thread's block grows while others are in flight."""
    import threading
    from loongcollector_amd import corpus
    from loongcollector_amd.processor import EventGroup, Processor
    from oracle.processor_oracle import LogEventModel, ProcessorOracle
    n_threads, groups_per_thread = 8, 6
    sizes = [300, 1500, 700, 64, 2000, 900]
    per_thread = sum(sizes)
    data, off, length = corpus.apache_batch(n_threads * per_thread, "B", poison_every=23)
    raw = data.tobytes()
    lines = [raw[off[i]:off[i] + length[i]].decode("latin-1") for i in range(len(length))]
    regex = corpus.REGEX_B + r"(?:\1)?"       # (an optional back-reference behind regex B: the whole pattern on LC_ENGINE_BT)
    cfg = {"SourceKey": "content", "Regex": regex, "Keys": corpus.KEYS_B}
    p = Processor(cfg)
    groups, at = [], 0
    for t in range(n_threads):
        mine = []
        for g in range(groups_per_thread):
            mine.append((at, EventGroup({"events": [{"contents": {"content": s}, "timestamp": 1, "type": 1} for s in lines[at:at + sizes[g]]]})))
            at += sizes[g]
        groups.append(mine)
    errors = []

    def run(t):
        try:
            for _, g in groups[t]:
                p.process(g)
        except Exception as e:  # noqa
            errors.append(e)

    threads = [threading.Thread(target=run, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors
    po = ProcessorOracle(cfg)
    for t in range(n_threads):
        for g, (start, grp) in enumerate(groups[t]):
            out = po.process_group([LogEventModel([("content", s.encode("latin-1"))]) for s in lines[start:start + sizes[g]]])
            assert grp.contents() == [[(k, v.decode("latin-1")) for k, v in ev.live()] for ev in out], (t, g)
    c = p.counters()
    assert c["in_events_total"] == len(lines) and c["out_failed_events_total"] == po.counters["out_failed"]


def test_generated_patterns_on_the_backtracking_engine_against_the_oracle(host_vm):
    """A bounded round of tools/fuzz_bt.py: 400 generated patterns (back-references, look-arounds, conditionals, atomic groups, lazy and
    greedy repeats), full match and search, six subjects each -- the product's program walked by the kernel's routine against the oracle."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_bt", os.path.join(ROOT, "tools", "fuzz_bt.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    checked, compiled, gave_up, bad = fz.main(1, 400)
    assert checked > 4000 and compiled > 700 and gave_up * 100 < checked
    assert not bad, bad[:5]


@pytest.mark.gpu
def test_generated_patterns_through_the_kernel_against_the_oracle(torch_dev):
    """The generator of tools/fuzz_bt.py on the device: 250 patterns, full match and search, 48 subjects each (up to 40 bytes, so that the
    eight-byte loads of the counted repeats and every alignment of a value's first byte are exercised), bt_match_kernel against the oracle."""
    import importlib.util
    from test_gpu_parity import pack, run_device
    spec = importlib.util.spec_from_file_location("fuzz_bt", os.path.join(ROOT, "tools", "fuzz_bt.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    # (LC_FUZZ_BT_SEED / LC_FUZZ_BT_N: longer runs on other seeds, recorded in profiles/round6_bt_fuzz_gpu.txt; LC_FUZZ_BT_ENGINE=0
    # lets the handle pick the engine -- tagged DFA, thread list or this one -- for the same generated patterns)
    rng = random.Random(int(os.environ.get("LC_FUZZ_BT_SEED", "11")))
    checked, bad, gave_up = 0, [], 0
    for _ in range(int(os.environ.get("LC_FUZZ_BT_N", "250"))):
        p = fz.gen(rng, [0]).encode()
        for flags, search in ((0, False), (B.LC_SYNTAX_SEARCH, True)):
            try:
                o = OracleRegex(p)
                rx = B.GpuRegex(p, syntax_flags=flags, engine=int(os.environ.get("LC_FUZZ_BT_ENGINE", B.LC_ENGINE_BT)))
            except (ValueError, B.RegexUnsupportedError, B.RegexSyntaxError):
                continue
            subs = [bytes(rng.choice(b'abc1 ') for _ in range(rng.randint(0, 40))) for _ in range(48)]
            data, off, length = pack(subs)
            caps, status = run_device(torch_dev, rx, data, off, length)
            for i, s in enumerate(subs):
                checked += 1
                if status[i] == B.LC_GAVE_UP:
                    gave_up += 1
                    continue
                try:
                    w = o.search(s) if search else o.fullmatch(s)
                except Exception:   # (the oracle's own complexity bound)
                    continue
                exp = None if w is None else [v for ab in w for v in ab][0 if search else 2:]
                ok = (status[i] == B.LC_NOMATCH and (caps[i] == -1).all()) if exp is None else (status[i] == B.LC_MATCH and list(caps[i]) == exp)
                if not ok:
                    bad.append((p, search, s, int(status[i]), list(caps[i]), exp))
    print("device fuzz: checked %d gave up %d bad %d" % (checked, gave_up, len(bad)))
    assert checked > 15000 and gave_up * 50 < checked, (checked, gave_up)
    assert not bad, bad[:5]
