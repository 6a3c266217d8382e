"""The depth-first decide kernel (csrc/nfa_decide_kernel.hpp) on the device, through the C ABI.

It settles the lines the thread-list kernels leave LC_OVERFLOW, so that no status other than LC_MATCH / LC_NOMATCH (or
the explicit LC_GAVE_UP, boost's complexity exception) ever reaches a processor.  LC_ENGINE_DECIDE sends EVERY line
through it, which lets the whole golden corpus cross-check it against the oracle and the other two engines."""
import json
import os
import random

import numpy as np
import pytest

from loongcollector_amd import binding as B, corpus
from oracle.oracle import OracleRegex
from tests.test_gpu_parity import pack, run_device, torch_dev  # noqa: F401

pytestmark = pytest.mark.gpu


def _check(cases, flags, kind, torch, strip_whole):
    bad, n = [], 0
    for c in cases:
        try:
            rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=flags)
        except B.RegexUnsupportedError:
            continue
        if not rx.has_nfa_program():
            continue
        subs = [s.encode("latin-1") for s, _ in c["subs"]]
        data, off, length = pack(subs)
        caps, status = run_device(torch, rx, data, off, length, engine=B.LC_ENGINE_DECIDE)
        for i, (_, flat) in enumerate(c["subs"]):
            n += 1
            exp = flat if (flat is None or not strip_whole) else flat[2:]
            if exp is None:
                ok = status[i] == B.LC_NOMATCH and (caps[i] == -1).all()
            else:
                ok = status[i] == B.LC_MATCH and list(caps[i]) == exp
            if not ok:
                bad.append((kind, c["p"], subs[i], int(status[i]), list(caps[i]), exp))
    return n, bad


def test_decide_kernel_reproduces_every_golden_vector(torch_dev, golden_dir):
    with open(os.path.join(golden_dir, "regex_golden.json")) as f:
        n, bad = _check(json.load(f)["cases"], 0, "full", torch_dev, True)
    assert n > 4000
    assert not bad, bad[:5]


def test_decide_kernel_search_and_atomic_golden_vectors(torch_dev, golden_dir):
    with open(os.path.join(golden_dir, "regex_search_golden.json")) as f:
        n1, bad = _check(json.load(f)["cases"], B.LC_SYNTAX_SEARCH, "search", torch_dev, False)
    with open(os.path.join(golden_dir, "regex_atomic_golden.json")) as f:
        d = json.load(f)
    n2, bad2 = _check(d["full"], 0, "atomic-full", torch_dev, True)
    n3, bad3 = _check(d["search"], B.LC_SYNTAX_SEARCH, "atomic-search", torch_dev, False)
    assert n1 > 1000 and n2 + n3 > 4000, (n1, n2, n3)
    assert not (bad + bad2 + bad3), (bad + bad2 + bad3)[:5]


OVERFLOWING = [
    (r"(.*)a(.{70})", [b"a" * 90 + b"x" * 70, b"a" * 100, b"b" * 10, b"xa" * 80]),
    (r"(.*)a(.{140})", [b"a" * 200 + b"y" * 140, b"a" * 200, b"a" + b"b" * 140, b"b" * 300, b""]),
    (r"(?>a+|b)*(.*)a(.{70})c", [b"a" * 100 + b"q" * 70 + b"c", b"a" * 100 + b"q" * 70, b"ab" * 50 + b"a" + b"q" * 70 + b"c"]),
    (r"(?>(?>(?>(?>(?>(?>(?>(a+))b?)c?)d?)e?)f?)g?)(h)", [b"aaabcdh", b"aaabcd", b"ah", b"aabbh"]),
]


@pytest.mark.parametrize("pattern,subs", OVERFLOWING)
def test_lines_the_thread_list_kernels_overflow_on_are_decided(torch_dev, pattern, subs):
    """More than 128 live threads, more than 64 with atomic groups, more than 6 nested memberships: all settled, bit-exact
    with the oracle, through the ordinary NFA entry (thread-list kernels first, decide kernel behind them)."""
    rx = B.GpuRegex(pattern.encode(), engine=B.LC_ENGINE_NFA)
    data, off, length = pack(subs)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off, length)
    assert exp_status[0] == 1
    for eng in (B.LC_ENGINE_NFA, B.LC_ENGINE_DECIDE):
        caps, status = run_device(torch_dev, rx, data, off, length, engine=eng)
        assert np.array_equal(status, exp_status), (eng, status, exp_status)
        assert np.array_equal(caps, exp_caps), eng
    L = B.load()
    import ctypes
    stats = (ctypes.c_uint64 * 2)()
    assert L.lc_decide_stats(stats) == 0
    assert stats[0] == len(subs) and stats[1] == 0          # the last call (LC_ENGINE_DECIDE) settled every line


def test_decide_kernel_on_many_long_lines_with_resumed_searches(torch_dev):
    """4 KiB values, a search pattern with > 64 threads, resumed inside the line: the worker pool takes the lines off the
    list, the memo keeps each walk linear."""
    rnd = random.Random(5)
    pat = r"a(.{70})b"
    srx = B.GpuRegex(pat, syntax_flags=B.LC_SYNTAX_SEARCH, engine=B.LC_ENGINE_NFA)
    lines = []
    for k in range(300):
        s = bytearray(rnd.choice(b"ab") for _ in range(rnd.randint(200, 4096)))
        lines.append(bytes(s))
    data, off, length = pack(lines)
    o = OracleRegex(pat)
    t = torch_dev
    d_data = t.from_numpy(np.concatenate([data, np.zeros(16, np.uint8)])).cuda()
    d_off = t.from_numpy(off.view(np.int32)).cuda()
    d_len = t.from_numpy(length.view(np.int32)).cuda()
    frm = np.array([rnd.randint(0, 150) for _ in lines], np.uint32)
    d_from = t.from_numpy(frm.view(np.int32)).cuda()
    for eng in (B.LC_ENGINE_NFA, B.LC_ENGINE_DECIDE):
        d_caps = t.full((len(lines), 2 * srx.groups), -7, dtype=t.int32, device="cuda")
        d_status = t.full((len(lines),), 9, dtype=t.uint8, device="cuda")
        srx.match_device_from(d_data, d_off, d_len, len(lines), d_caps, d_status, d_from=d_from, engine=eng)
        t.cuda.synchronize()
        caps, status = d_caps.cpu().numpy(), d_status.cpu().numpy()
        for i, s in enumerate(lines):
            m = o.search(s, int(frm[i]))
            if m is None:
                assert status[i] == B.LC_NOMATCH, (eng, i)
            else:
                assert status[i] == B.LC_MATCH and caps[i].tolist() == [v for be in m for v in be], (eng, i)


def test_random_atomic_patterns_decide_engine_vs_oracle(torch_dev):
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "gen_atomic_golden", os.path.join(os.path.dirname(__file__), "golden", "gen_atomic_golden.py"))
    pytest.importorskip("regex")
    agen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(agen)
    rng = random.Random(99)
    checked = 0
    for _ in range(150):
        p = agen.gen(rng)
        try:
            o = OracleRegex(p)
        except ValueError:
            continue
        subs = [bytes(rng.choice(b"abc1 ") for _ in range(rng.randint(0, 12))) for _ in range(12)]
        data, off, length = pack(subs)
        for flags, fn in ((0, o.fullmatch), (B.LC_SYNTAX_SEARCH, o.search)):
            try:
                rx = B.GpuRegex(p, syntax_flags=flags)
            except B.RegexUnsupportedError:
                continue
            if not rx.has_nfa_program():
                continue
            caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_DECIDE)
            for i, s in enumerate(subs):
                try:
                    exp = fn(s)
                except RuntimeError:
                    continue
                want = None if exp is None else [v for ab in (exp if flags else exp[1:]) for v in ab]
                got = None if status[i] == B.LC_NOMATCH else list(caps[i])
                checked += 1
                assert status[i] in (B.LC_MATCH, B.LC_NOMATCH) and got == want, (p, s, flags, got, want)
    assert checked > 2500


def test_processors_never_see_an_undecided_line():
    """Processor level (LC_OVERFLOW used to be folded into "parse failed"): the parse processor and the filter
    on values that overflow the thread-list kernels give exactly the oracle's answers."""
    from loongcollector_amd.processor import EventGroup, Filter, Processor
    from oracle.filter_oracle import FilterOracle
    from oracle.processor_oracle import LogEventModel, ProcessorOracle
    pattern = r"(.*)a(.{140})"
    vals = ["a" * 200 + "y" * 140, "a" * 150, "b" * 20, "za" + "q" * 140]
    for keep_fail in (False, True):
        cfg = {"SourceKey": "content", "Regex": pattern, "Keys": ["head", "tail"], "KeepingSourceWhenParseFail": keep_fail,
               "_Engine": "nfa"}
        fixture = {"events": [{"contents": [["content", v]], "timestamp": 1, "type": 1} for v in vals]}
        p, g = Processor(cfg), EventGroup(fixture)
        p.process(g)
        po = ProcessorOracle({k: v for k, v in cfg.items() if k != "_Engine"})
        out = po.process_group([LogEventModel([("content", v.encode())]) for v in vals])
        assert g.contents() == [[(k, v.decode()) for k, v in ev.live()] for ev in out]
        c = p.counters()
        assert (c["discarded_events_total"], c["out_failed_events_total"], c["out_successful_events_total"]) == (
            po.counters["discarded"], po.counters["out_failed"], po.counters["out_successful"])
        assert c["out_failed_events_total"] == 1
        assert c["undecided_events_total"] == 0 and c["complexity_exceeded_events_total"] == 0
    # the filter: a NOT over an overflowing leaf (an undecided value used to come out as "keep")
    fcfg = {"ConditionExp": {"operator": "not", "operands": [{"key": "content", "exp": pattern, "type": "regex"}]}}
    f = Filter(fcfg)
    g2 = EventGroup({"events": [{"contents": [["content", v]], "timestamp": 1, "type": 1} for v in vals]})
    f.process(g2)
    kept = [dict(ev)["content"] for ev in g2.contents()]
    assert kept == [vals[2]]
    assert kept == [c["content"].decode() for c in FilterOracle(fcfg).process([{"content": v.encode()} for v in vals])]


def test_depth_first_first_pass_gives_the_same_results(torch_dev):
    """lc_nfa_set_dfs(1): the NFA engine settles lines with a lane-per-line backtracking walk first and hands the rest (tiny
    step budget here, so that a good part of the lines IS handed on) to the thread-list kernels: status and capture offsets
    equal the default path's and the oracle's, for a plain, a search and an atomic-group pattern."""
    import ctypes
    import torch
    L = B.load()
    L.lc_nfa_set_dfs.argtypes = [ctypes.c_int]
    L.lc_dfs_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    cases = [(corpus.REGEX_B, 0), (r"(\d+)-(\w+)", B.LC_SYNTAX_SEARCH), (r"((?>a+)b|a+c)(x*)", 0)]
    try:
        for pattern, flags in cases:
            if pattern == corpus.REGEX_B:
                data, off, length = corpus.mixed_batch(3000, min_len=1, max_len=900)
                off = off[:-1]
            else:
                lines = []
                for _ in range(2000):
                    lines.append(("".join(rng.choice(list("ab cx-19_"), size=int(rng.integers(0, 40))))).encode())
                length = np.array([len(s) for s in lines], dtype=np.uint32)
                off = np.zeros(len(lines), dtype=np.uint32)
                off[1:] = np.cumsum(length[:-1])
                data = np.frombuffer(b"".join(lines) + b"\0" * 16, dtype=np.uint8).copy()
            n = len(off)
            rx = B.GpuRegex(pattern, syntax_flags=flags, engine=B.LC_ENGINE_NFA)
            d_data = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).to(dev)
            d_off = torch.from_numpy(off.view(np.int32).copy()).to(dev)
            d_len = torch.from_numpy(length.view(np.int32).copy()).to(dev)
            res = []
            for mode in (0, 1):
                L.lc_nfa_set_dfs(mode)
                d_caps = torch.full((n, 2 * rx.groups), 7, dtype=torch.int32, device=dev)
                d_status = torch.full((n,), 9, dtype=torch.uint8, device=dev)
                B.launched_kernels()
                rx.match_device(d_data, d_off, d_len, n, d_caps, d_status, engine=B.LC_ENGINE_NFA)
                torch.cuda.synchronize()
                names = B.launched_kernels()
                assert ("nfa_dfs_kernel" in names) == bool(mode)
                res.append((d_status.cpu().numpy(), d_caps.cpu().numpy()))
            stats = (ctypes.c_uint64 * 2)()
            L.lc_dfs_stats(stats)
            assert stats[0] > 0                                     # the walk did carve frame stacks
            assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
            assert set(np.unique(res[1][0])) <= {0, 1}              # nothing pending / undecided is left behind
    finally:
        L.lc_nfa_set_dfs(-1)
