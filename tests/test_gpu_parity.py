"""Parity tests proper: the HIP path, called through the C ABI, against the oracle and the golden vectors.
Bit-exact bar: status bytes and int32 capture offsets must be identical."""
import json
import os
import random

import numpy as np
import pytest

from loongcollector_amd import binding as B
from loongcollector_amd import corpus
from oracle.oracle import OracleRegex

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    assert B.device_count() >= 1
    torch.cuda.set_device(0)
    return torch


def run_device(torch, rx, data, off, length=None, sep=0, ngroups=None, engine=B.LC_ENGINE_AUTO):
    """off: n entries when `length` is given, n+1 entries otherwise"""
    G = rx.groups if ngroups is None else ngroups
    n = len(off) if length is not None else len(off) - 1
    dev = torch.device("cuda:0")
    pad = np.zeros(max(1, len(data)), dtype=np.uint8)
    pad[:len(data)] = data
    d_data = torch.from_numpy(pad).to(dev)
    d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint32).view(np.int32)).to(dev)
    d_len = None if length is None else torch.from_numpy(np.ascontiguousarray(length, dtype=np.uint32).view(np.int32)).to(dev)
    d_caps = torch.full((max(n, 1), max(2 * G, 1)), -7, dtype=torch.int32, device=dev)
    d_status = torch.full((max(n, 1),), 9, dtype=torch.uint8, device=dev)
    rx.match_device(d_data, d_off, d_len, n, d_caps, d_status, ngroups=G, sep_bytes=sep,
                    stream=torch.cuda.current_stream().cuda_stream, engine=engine)
    torch.cuda.synchronize()
    return d_caps.cpu().numpy()[:n, :2 * G], d_status.cpu().numpy()[:n]


def pack(subjects):
    off, length, chunks, at = [], [], [], 0
    for s in subjects:
        off.append(at)
        length.append(len(s))
        chunks.append(s)
        at += len(s)
    data = np.frombuffer(b"".join(chunks), dtype=np.uint8) if at else np.zeros(0, np.uint8)
    return data, np.array(off, np.uint32), np.array(length, np.uint32)


def test_golden_vectors_through_the_c_abi(torch_dev, golden_dir):
    with open(os.path.join(golden_dir, "regex_golden.json")) as f:
        golden = json.load(f)
    bad = []
    checked = {B.LC_ENGINE_TDFA: 0, B.LC_ENGINE_NFA: 0}
    for c in golden["cases"]:
        rx = B.GpuRegex(c["p"].encode("latin-1"))
        subs = [s.encode("latin-1") for s, _ in c["subs"]]
        data, off, length = pack(subs)
        engines = [B.LC_ENGINE_NFA] if rx.has_nfa_program() else []
        if rx.info()["engine"] == B.LC_ENGINE_TDFA:
            engines.append(B.LC_ENGINE_TDFA)
        for eng in engines:
            caps, status = run_device(torch_dev, rx, data, off, length, engine=eng)
            for i, (_, flat) in enumerate(c["subs"]):
                checked[eng] += 1
                if flat is None:
                    ok = status[i] == B.LC_NOMATCH and (caps[i] == -1).all()
                else:
                    ok = status[i] == B.LC_MATCH and list(caps[i]) == flat[2:]
                if not ok:
                    bad.append((eng, c["p"], subs[i], int(status[i]), list(caps[i]), flat))
    assert checked[B.LC_ENGINE_TDFA] > 4000 and checked[B.LC_ENGINE_NFA] > 4000
    assert not bad, bad[:5]


@pytest.mark.parametrize("kind", ["A", "B"])
def test_bench_corpus_bit_exact_vs_oracle(torch_dev, kind):
    pattern = corpus.REGEX_A if kind == "A" else corpus.REGEX_B
    n = 20000
    data, off, length = corpus.apache_batch(n, kind, poison_every=13)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:-1], length)
    rx = B.GpuRegex(pattern)
    caps, status = run_device(torch_dev, rx, data, off, None, sep=1)   # offsets[n+1] + separator form
    assert np.array_equal(status, exp_status)
    assert np.array_equal(caps, exp_caps)
    caps2, status2 = run_device(torch_dev, rx, data, off[:-1], length)  # (off,len) form
    assert np.array_equal(status2, exp_status) and np.array_equal(caps2, exp_caps)


@pytest.mark.parametrize("n", [20000, 70000])
@pytest.mark.parametrize("kind", ["A", "B"])
def test_empty_captured_fields_take_the_folded_set_registers(torch_dev, kind, n):
    """Every third line has an empty referrer: its group's begin and end are stamped by ONE transition, which the packed
    tables turn into a stamp of the set's own register (regex_handle.cpp planTdfaFold); the kernels (32-bit registers below
    64 Ki lines, 16-bit above) settle member = max(member, set) before the captures go out.  Against the oracle."""
    pattern = corpus.REGEX_A if kind == "A" else corpus.REGEX_B
    data, off, length = corpus.apache_batch(n, kind, poison_every=17, empty_every=3)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:-1], length)
    assert (exp_caps[exp_status == 1, 2::2] == exp_caps[exp_status == 1, 3::2]).any()     # some group IS empty
    rx = B.GpuRegex(pattern)
    B.launched_kernels()                                   # (forget what earlier tests launched)
    caps, status = run_device(torch_dev, rx, data, off, None, sep=1)
    assert np.array_equal(status, exp_status)
    bad = np.nonzero((caps != exp_caps).any(axis=1))[0]
    assert bad.size == 0, (bad[:8].tolist(), caps[bad[0]].tolist(), exp_caps[bad[0]].tolist())
    assert "nogeneral" in B.launched_kernels()


@pytest.mark.parametrize("kind", ["A", "B"])
def test_nfa_kernel_bit_exact_on_bench_corpus(torch_dev, kind):
    """The wave-per-line NFA kernel (the engine AUTO falls back to) against the oracle AND against the TDFA kernel."""
    pattern = corpus.REGEX_A if kind == "A" else corpus.REGEX_B
    n = 6000
    data, off, length = corpus.apache_batch(n, kind, poison_every=11)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:-1], length)
    rx = B.GpuRegex(pattern)
    caps_n, status_n = run_device(torch_dev, rx, data, off, None, sep=1, engine=B.LC_ENGINE_NFA)
    caps_t, status_t = run_device(torch_dev, rx, data, off, None, sep=1, engine=B.LC_ENGINE_TDFA)
    assert np.array_equal(status_n, exp_status) and np.array_equal(caps_n, exp_caps)
    assert np.array_equal(status_t, status_n) and np.array_equal(caps_t, caps_n)


def test_auto_falls_back_to_nfa_kernel_when_tdfa_explodes(torch_dev):
    blowup = r"(.*)a" + "." * 14 + r"(.*)"
    rx = B.GpuRegex(blowup)
    assert rx.info()["engine"] == B.LC_ENGINE_NFA
    rng = np.random.default_rng(3)
    subs = [bytes(rng.choice(list(b"ab"), size=int(rng.integers(0, 60))).astype(np.uint8)) for _ in range(500)]
    data, off, length = pack(subs)
    exp_caps, exp_status = OracleRegex(blowup).fullmatch_batch(data, off, length)
    caps, status = run_device(torch_dev, rx, data, off, length)
    assert 0 < exp_status.sum() < len(subs)
    assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)


def test_nfa_thread_overflow_second_chance_and_report(torch_dev):
    """More than 64 live threads: the NFA kernel reports LC_OVERFLOW and raises its flag, the two-threads-per-lane kernel
    behind it (nfa_wide_kernel.hpp) decides the line; more than 128 goes to the decide kernel."""
    # on "aaaa..." every one of the 70 '.' positions is alive at once: more than 64 simultaneous threads
    pattern = r"(.*)a(.{70})"
    subs = [b"a" * 100, b"b" * 10, b"a" + b"b" * 70, b"xa" * 80, b"a" * 71]
    data, off, length = pack(subs)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off, length)
    rx = B.GpuRegex(pattern, engine=B.LC_ENGINE_NFA)
    caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_NFA)
    assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)
    assert list(exp_status) == [1, 0, 1, 1, 1]
    # a search pattern, resumed inside the line
    srx = B.GpuRegex(r"a(.{70})b", syntax_flags=B.LC_SYNTAX_SEARCH, engine=B.LC_ENGINE_NFA)
    line = b"b" + b"a" * 90 + b"b" + b"a" * 80 + b"b"
    d_data = torch_dev.from_numpy(np.frombuffer(line, np.uint8).copy()).cuda()
    d_off = torch_dev.zeros(1, dtype=torch_dev.int32, device="cuda")
    d_len = torch_dev.tensor([len(line)], dtype=torch_dev.int32, device="cuda")
    for frm in (0, 21):
        d_from = torch_dev.tensor([frm], dtype=torch_dev.int32, device="cuda")
        d_caps = torch_dev.full((1, 2 * srx.groups), -7, dtype=torch_dev.int32, device="cuda")
        d_status = torch_dev.full((1,), 9, dtype=torch_dev.uint8, device="cuda")
        srx.match_device_from(d_data, d_off, d_len, 1, d_caps, d_status, d_from=d_from, engine=B.LC_ENGINE_NFA)
        torch_dev.cuda.synchronize()
        o = OracleRegex(r"a(.{70})b").search(line, frm)
        assert int(d_status[0]) == 1 and d_caps.cpu().numpy()[0].tolist() == [v for be in o for v in be], (frm, o)
    # beyond 128 threads: settled by the depth-first decide kernel behind the two (tests/test_gpu_decide.py)
    rx2 = B.GpuRegex(r"(.*)a.{140}", engine=B.LC_ENGINE_NFA)
    data, off, length = pack([b"a" * 200, b"a" + b"b" * 140, b"a" * 140])
    exp_caps, exp_status = OracleRegex(r"(.*)a.{140}").fullmatch_batch(data, off, length)
    caps, status = run_device(torch_dev, rx2, data, off, length, engine=B.LC_ENGINE_NFA)
    assert list(status) == [B.LC_MATCH, B.LC_MATCH, B.LC_NOMATCH] == list(exp_status)
    assert np.array_equal(caps, exp_caps)


def test_ragged_mixed_corpus_with_failures(torch_dev):
    data, off, length = corpus.mixed_batch(6000)
    exp_caps, exp_status = OracleRegex(corpus.REGEX_B).fullmatch_batch(data, off[:-1], length)
    assert 0.2 < 1 - exp_status.mean() < 0.4  # JSON lines must fail
    caps, status = run_device(torch_dev, B.GpuRegex(corpus.REGEX_B), data, off, None, sep=1)
    assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)


def test_edge_cases_empty_and_unaligned_and_long_lines(torch_dev):
    pattern = r"(\w*)\t?(\w*)(.*)"
    rng = np.random.default_rng(5)
    subs = [b"", b"a", b"\t", b"a\tb", b"x" * 15, b"x" * 16, b"x" * 17, b"ab\tcd" + b" tail" * 3000, b""]
    for _ in range(300):
        L = int(rng.integers(0, 70))
        subs.append(bytes(rng.choice(list(b"ab\t _"), size=L).astype(np.uint8)))
    data, off, length = pack(subs)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off, length)
    caps, status = run_device(torch_dev, B.GpuRegex(pattern), data, off, length)
    assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)


def test_zero_lines_and_extra_groups(torch_dev):
    rx = B.GpuRegex(r"(\d+) (\d+)")
    data, off, length = pack([b"12 34", b"x"])
    caps, status = run_device(torch_dev, rx, data, off, length, ngroups=4)
    assert list(status) == [1, 0]
    assert list(caps[0]) == [0, 2, 3, 5, -1, -1, -1, -1] and (caps[1] == -1).all()
    caps0, status0 = run_device(torch_dev, rx, np.zeros(0, np.uint8), np.zeros(0, np.uint32), np.zeros(0, np.uint32))
    assert caps0.shape[0] == 0 and status0.shape[0] == 0


def test_match_host_pipeline_equals_device_path(torch_dev):
    n = 300000  # > one pipelined chunk, so both staging slots and the drain path are exercised
    data, off, length = corpus.apache_batch(n, "A", poison_every=1001)
    rx = B.GpuRegex(corpus.REGEX_A)
    caps_h, status_h = rx.match_host(data, off[:-1], length)
    exp_caps, exp_status = OracleRegex(corpus.REGEX_A).fullmatch_batch(data, off[:-1], length)
    assert np.array_equal(status_h, exp_status) and np.array_equal(caps_h, exp_caps)
    # scattered (non-contiguous, out-of-order) views take the gather path
    perm = np.random.default_rng(1).permutation(5000)
    caps_p, status_p = rx.match_host(data, off[:-1][perm], length[perm])
    assert np.array_equal(status_p, exp_status[perm]) and np.array_equal(caps_p, exp_caps[perm])


def test_full_size_batch_properties(torch_dev):
    """BASELINE config 2 size (1 Mi lines): size-independent properties instead of a full oracle pass --
    (1) every line matches, (2) captures tile the line in order, (3) the checksum of the capture table equals the
    checksum of the pool's oracle captures gathered by the same indices."""
    n = 1 << 20
    pool_lines = 8192
    data, off, length = corpus.apache_batch(n, "A", pool_lines=pool_lines)
    rx = B.GpuRegex(corpus.REGEX_A)
    caps, status = run_device(torch_dev, rx, data, off, None, sep=1)
    assert status.min() == 1
    assert (caps[:, 0] == 0).all() and (np.diff(caps, axis=1) >= 0).all() and (caps[:, -1] == 511).all()
    pool = corpus.apache_pool("A", pool_lines)
    p_off = (np.arange(pool_lines) * 512).astype(np.uint32)
    p_caps, _ = OracleRegex(corpus.REGEX_A).fullmatch_batch(pool.reshape(-1), p_off, np.full(pool_lines, 512, np.uint32))
    idx = np.random.Generator(np.random.MT19937(corpus.SEED + 7919)).integers(0, pool_lines, size=n)
    assert np.array_equal(caps, p_caps[idx])


def test_search_mode_golden_vectors_on_both_kernels(torch_dev, golden_dir):
    with open(os.path.join(golden_dir, "regex_search_golden.json")) as f:
        golden = json.load(f)
    bad, checked = [], 0
    for c in golden["cases"]:
        rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=B.LC_SYNTAX_SEARCH)
        subs = [s.encode("latin-1") for s, _ in c["subs"]]
        data, off, length = pack(subs)
        engines = ([B.LC_ENGINE_NFA] if rx.has_nfa_program() else []) + (
            [B.LC_ENGINE_TDFA] if rx.info()["engine"] == B.LC_ENGINE_TDFA else [])
        for eng in engines:
            caps, status = run_device(torch_dev, rx, data, off, length, engine=eng)
            for i, (_, flat) in enumerate(c["subs"]):
                checked += 1
                ok = (status[i] == B.LC_NOMATCH and (caps[i] == -1).all()) if flat is None else (
                    status[i] == B.LC_MATCH and list(caps[i]) == flat)
                if not ok:
                    bad.append((eng, c["p"], subs[i], int(status[i]), list(caps[i]), flat))
    assert checked > 2000
    assert not bad, bad[:5]


def test_config4_multi_tenant_64_pipelines_round_robin(torch_dev):
    """BASELINE config 4: 64 concurrent pipelines, each with its own regex, batched round-robin onto one GPU the way
    ProcessQueueManager::PopItem hands out groups (one ~512 KB group per pipeline per turn).  Every pipeline's tables
    are resident in HBM once and are staged into LDS per launch (the 'per-pipeline NFA switch')."""
    import torch
    rng = np.random.default_rng(20260923)
    delims = [" ", "|", "\t"]
    pipelines = []
    for p in range(64):
        ngroups = int(rng.integers(6, 15))
        d = delims[p % 3]
        cls = {" ": r"[^ ]", "|": r"[^|]", "\t": r"[^\t]"}[d]
        pattern = re_escape(d).join("(%s*)" % cls for _ in range(ngroups))
        lines = []
        for _ in range(1000):
            fields = ["".join(chr(int(c)) for c in rng.integers(97, 123, size=int(rng.integers(0, 12))))
                      for _ in range(ngroups + (1 if rng.integers(0, 10) == 0 else 0))]  # 10 %: one field too many
            lines.append(d.join(fields).encode())
        pipelines.append((B.GpuRegex(pattern), OracleRegex(pattern), lines))
    dev = torch.device("cuda:0")
    streams = [torch.cuda.Stream() for _ in range(4)]
    results = []
    for turn in range(2):                       # two round-robin turns over all 64 pipelines
        for p, (rx, _, lines) in enumerate(pipelines):
            part = lines[turn * 500:(turn + 1) * 500]
            data, off, length = pack(part)
            s = streams[p % 4]
            with torch.cuda.stream(s):
                d_data = torch.from_numpy(data).to(dev, non_blocking=True)
                d_off = torch.from_numpy(off.view(np.int32)).to(dev, non_blocking=True)
                d_len = torch.from_numpy(length.view(np.int32)).to(dev, non_blocking=True)
                d_caps = torch.empty((len(part), 2 * rx.groups), dtype=torch.int32, device=dev)
                d_status = torch.empty((len(part),), dtype=torch.uint8, device=dev)
                rx.match_device(d_data, d_off, d_len, len(part), d_caps, d_status, stream=s.cuda_stream)
            results.append((p, turn, d_caps, d_status, (d_data, d_off, d_len)))
    torch.cuda.synchronize()
    for p, turn, d_caps, d_status, _ in results:
        _, orx, lines = pipelines[p]
        part = lines[turn * 500:(turn + 1) * 500]
        data, off, length = pack(part)
        exp_caps, exp_status = orx.fullmatch_batch(data, off, length)
        assert np.array_equal(d_status.cpu().numpy(), exp_status), p
        assert np.array_equal(d_caps.cpu().numpy(), exp_caps), p
        assert 0.8 < exp_status.mean() < 0.97


def test_multi_job_launch_equals_one_call_per_job(torch_dev):
    """lc_regex_match_device_multi (BASELINE configs[3], one launch for many pipelines): jobs with different regexes, sizes
    (including 1 line, a size that is not a multiple of the workgroup, an empty job), with and without a length table, one
    pattern that runs on the NFA engine -- results bit-identical to the oracle, job by job."""
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(77)
    specs = [(r"([^ ]*) ([^ ]*) ([^ ]*)", " ", 1000), (r"([^|]*)\|([^|]*)", "|", 1), (r"(\w+)\t(\d+)\t(.*)", "\t", 257),
             (r"([a-z]+) (\d+)", " ", 0), (r"(.*)a(.{70})", "", 40), (corpus.REGEX_B, None, 300), (r"([^ ]*) ([^ ]*)", " ", 513)]
    jobs, keep, expect = [], [], []
    for k, (pattern, d, n) in enumerate(specs):
        if d is None:
            data, off, length = corpus.apache_batch(n, "B", pool_lines=64)
            off = off[:-1]
        else:
            lines = []
            for _ in range(n):
                if d:
                    nf = int(rng.integers(1, 5))
                    lines.append(d.join("".join(chr(int(c)) for c in rng.integers(97, 123, size=int(rng.integers(0, 9))))
                                        if rng.integers(0, 3) else str(int(rng.integers(0, 999))) for _ in range(nf)).encode())
                else:
                    lines.append(("b" * int(rng.integers(0, 30)) + "a" * int(rng.integers(0, 3)) + "c" * int(rng.integers(60, 80))).encode())
            data, off, length = pack(lines) if n else (np.zeros(16, np.uint8), np.zeros(1, np.uint32), np.zeros(0, np.uint32))
            off = off[:n]
        rx = B.GpuRegex(pattern)
        use_len = k % 2 == 0
        d_data = torch.from_numpy(np.concatenate([data, np.zeros(64, np.uint8)])).to(dev)
        full_off = np.concatenate([off, [len(data)]]).astype(np.uint32) if not use_len else off.astype(np.uint32)
        d_off = torch.from_numpy((full_off if len(full_off) else np.zeros(1, np.uint32)).view(np.int32)).to(dev)
        d_len = torch.from_numpy(length.view(np.int32).copy()).to(dev) if use_len and n else None
        d_caps = torch.full((max(n, 1), 2 * rx.groups), 7, dtype=torch.int32, device=dev)
        d_status = torch.full((max(n, 1),), 9, dtype=torch.uint8, device=dev)
        sep = 0 if (use_len and n) or d is not None else 1  # pack() puts the lines back to back; apache_batch separates them
        jobs.append((rx, d_data, d_off, d_len, n, d_caps, d_status, sep))
        keep.append((d_data, d_off, d_len))
        if n:
            expect.append(OracleRegex(pattern).fullmatch_batch(data, off, length))
        else:
            expect.append(None)
    arr = B.make_jobs(jobs)
    B.launched_kernels()
    B.match_device_multi(arr, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    names = B.launched_kernels()
    assert "tdfa_stream_multi_kernel" in names                      # the TDFA jobs went out as one packed launch
    engines = [j[0].info()["engine"] for j in jobs]
    assert B.LC_ENGINE_NFA in engines and "nfa" in names            # ... and the NFA job on its own kernels
    for (rx, _, _, _, n, d_caps, d_status, _), exp in zip(jobs, expect):
        if not n:
            assert int(d_status[0]) == 9                            # an empty job touches nothing
            continue
        assert np.array_equal(d_status.cpu().numpy()[:n], exp[1]), rx
        assert np.array_equal(d_caps.cpu().numpy()[:n], exp[0]), rx


def re_escape(d):
    return {"|": r"\|", " ": " ", "\t": r"\t"}[d]


@pytest.mark.parametrize("engine", [B.LC_ENGINE_TDFA, B.LC_ENGINE_NFA])
def test_length_scheduled_ragged_path_is_bit_exact(torch_dev, engine):
    """lc_regex_match_device_ragged: same results, at the ORIGINAL line indices, as the plain path and the oracle."""
    import torch
    dev = torch.device("cuda:0")
    n = 7001 if engine == B.LC_ENGINE_TDFA else 1500
    data, off, length = corpus.mixed_batch(n, min_len=1, max_len=5000)
    length[::97] = 0                       # empty lines in the mix (views of zero length)
    exp_caps, exp_status = OracleRegex(corpus.REGEX_B).fullmatch_batch(data, off[:-1], length)
    rx = B.GpuRegex(corpus.REGEX_B)
    G = rx.groups
    d_data = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(off[:-1].view(np.int32).copy()).to(dev)
    d_len = torch.from_numpy(length.view(np.int32).copy()).to(dev)
    d_caps = torch.full((n, 2 * G), -7, dtype=torch.int32, device=dev)
    d_status = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    d_scratch = torch.empty((B.sched_scratch_bytes(n) // 4 + 1,), dtype=torch.int32, device=dev)
    rx.match_device_ragged(d_data, d_off, d_len, n, d_caps, d_status, d_scratch, engine=engine,
                           stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_status.cpu().numpy(), exp_status)
    assert np.array_equal(d_caps.cpu().numpy(), exp_caps)
    order = d_scratch.cpu().numpy().view(np.uint32)[512:512 + n]
    assert sorted(order.tolist()) == list(range(n))                      # a permutation
    assert (np.diff((length[order] >> 5).astype(np.int64)) <= 0).all()    # longest bucket first


def test_atomic_groups_and_possessive_quantifiers_on_both_kernels(torch_dev, golden_dir):
    """(?>X) / X*+ vectors (regex module ∧ PCRE1, tests/golden/gen_atomic_golden.py) through the C ABI, full and search."""
    with open(os.path.join(golden_dir, "regex_atomic_golden.json")) as f:
        golden = json.load(f)
    bad, checked = [], 0
    for kind, flags in (("full", 0), ("search", B.LC_SYNTAX_SEARCH)):
        for c in golden[kind]:
            try:
                rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=flags)
            except B.RegexUnsupportedError:
                continue
            subs = [s.encode("latin-1") for s, _ in c["subs"]]
            data, off, length = pack(subs)
            engines = ([B.LC_ENGINE_TDFA] if rx.info()["engine"] == B.LC_ENGINE_TDFA else []) + (
                [B.LC_ENGINE_NFA] if rx.has_nfa_program() else [])
            for eng in engines:
                caps, status = run_device(torch_dev, rx, data, off, length, engine=eng)
                for i, (_, flat) in enumerate(c["subs"]):
                    checked += 1
                    exp = flat if kind == "search" or flat is None else flat[2:]
                    ok = (status[i] == B.LC_NOMATCH and (caps[i] == -1).all()) if exp is None else (
                        status[i] == B.LC_MATCH and list(caps[i]) == exp)
                    if not ok:
                        bad.append((kind, eng, c["p"], subs[i], int(status[i]), list(caps[i]), exp))
    assert checked > 7000
    assert not bad, bad[:5]


def test_fixed_length_lookarounds_on_both_kernels(torch_dev, golden_dir):
    """(?=lit) (?!lit) (?<=lit) (?<!lit) vectors (regex module ∧ PCRE1, tests/golden/gen_lookaround_golden.py) through the C ABI, full
    match and search, on every engine the handle has: the product positions of a look-ahead window are ordinary positions to the kernels."""
    with open(os.path.join(golden_dir, "regex_lookaround_golden.json")) as f:
        golden = json.load(f)
    bad, checked = [], 0
    for kind, flags in (("full", 0), ("search", B.LC_SYNTAX_SEARCH)):
        for c in golden[kind]:
            try:
                rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=flags)
            except B.RegexUnsupportedError:
                continue
            subs = [s.encode("latin-1") for s, _ in c["subs"]]
            data, off, length = pack(subs)
            engines = ([B.LC_ENGINE_TDFA] if rx.info()["engine"] == B.LC_ENGINE_TDFA else []) + (
                [B.LC_ENGINE_NFA] if rx.has_nfa_program() else [])
            for eng in engines:
                caps, status = run_device(torch_dev, rx, data, off, length, engine=eng)
                for i, (_, flat) in enumerate(c["subs"]):
                    checked += 1
                    exp = flat if kind == "search" or flat is None else flat[2:]
                    ok = (status[i] == B.LC_NOMATCH and (caps[i] == -1).all()) if exp is None else (
                        status[i] == B.LC_MATCH and list(caps[i]) == exp)
                    if not ok:
                        bad.append((kind, eng, c["p"], subs[i], int(status[i]), list(caps[i]), exp))
    assert checked > 6000
    assert not bad, bad[:5]


def test_prefix_mode_on_both_kernels(torch_dev, golden_dir):
    """LC_SYNTAX_PREFIX (regex_search with match_continuous, the multiline splitter's per-line question) vs the oracle"""
    from oracle.oracle import OracleRegex
    with open(os.path.join(golden_dir, "regex_search_golden.json")) as f:
        golden = json.load(f)
    bad, checked = [], 0
    for c in golden["cases"][:120]:
        p = c["p"].encode("latin-1")
        try:
            rx = B.GpuRegex(p, syntax_flags=B.LC_SYNTAX_PREFIX)
        except B.RegexUnsupportedError:
            continue
        o = OracleRegex(p)
        subs = [s.encode("latin-1")[cut:] for s, _ in c["subs"] for cut in (0, 3)]
        data, off, length = pack(subs)
        engines = ([B.LC_ENGINE_NFA] if rx.has_nfa_program() else []) + (
            [B.LC_ENGINE_TDFA] if rx.info()["engine"] == B.LC_ENGINE_TDFA else [])
        for eng in engines:
            caps, status = run_device(torch_dev, rx, data, off, length, engine=eng)
            for i, s in enumerate(subs):
                checked += 1
                exp = o.prefixmatch(s)
                ok = (status[i] == B.LC_NOMATCH) if exp is None else (
                    status[i] == B.LC_MATCH and list(caps[i]) == [v for ab in exp[1:] for v in ab])
                if not ok:
                    bad.append((eng, c["p"], s, int(status[i]), list(caps[i]), exp))
    assert checked > 1500
    assert not bad, bad[:5]


@pytest.mark.parametrize("compact", ["0", "256", "512"])
def test_byte_pair_transition_tables_are_bit_exact(torch_dev, golden_dir, monkeypatch, compact):
    """Opt-in byte-pair stepping of the TDFA kernel (LC_TDFA_PAIR=1 at compile time of the pattern): one dependent LDS
    lookup per two bytes; same results as the single-byte table on the golden vectors, the bench corpus and resumed
    searches -- on the 32-bit kernel and on the compact (16-bit register) variants."""
    monkeypatch.setenv("LC_TDFA_PAIR", "1")
    monkeypatch.setenv("LC_TDFA_COMPACT", compact)
    with open(os.path.join(golden_dir, "regex_golden.json")) as f:
        golden = json.load(f)
    bad, checked, paired = [], 0, 0
    for c in golden["cases"][:400]:
        rx = B.GpuRegex(c["p"].encode("latin-1"), engine=B.LC_ENGINE_AUTO)
        if rx.info()["engine"] != B.LC_ENGINE_TDFA:
            continue
        paired += rx.info()["table_bytes"] > 4096
        subs = [s.encode("latin-1") for s, _ in c["subs"]]
        data, off, length = pack(subs)
        caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_TDFA)
        for i, (_, flat) in enumerate(c["subs"]):
            checked += 1
            ok = (status[i] == B.LC_NOMATCH and (caps[i] == -1).all()) if flat is None else (
                status[i] == B.LC_MATCH and list(caps[i]) == flat[2:])
            if not ok:
                bad.append((c["p"], subs[i], int(status[i]), list(caps[i]), flat))
    assert checked > 2000 and paired > 50
    assert not bad, bad[:5]
    from loongcollector_amd import corpus
    from oracle.oracle import OracleRegex
    rx = B.GpuRegex(corpus.REGEX_A)
    data, off, length = corpus.mixed_batch(4096, seed=5) if hasattr(corpus, "mixed_batch") else corpus.apache_batch(4096, "A", 512)
    caps, status = run_device(torch_dev, rx, data, off[:len(length)], length)
    exp_caps, exp_status = OracleRegex(corpus.REGEX_A).fullmatch_batch(data, off[:len(length)], length)
    assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)


def _golden_on_tdfa(torch_dev, golden_dir, name, flags, skip, limit):
    """golden vectors of `name` through the TDFA engine; -> (checked, patterns whose launch was the one-stamp kernel, bad)"""
    with open(os.path.join(golden_dir, name)) as f:
        golden = json.load(f)
    bad, checked, pair1 = [], 0, 0
    for c in golden["cases"][:limit]:
        rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=flags)
        if rx.info()["engine"] != B.LC_ENGINE_TDFA:
            continue
        subs = [s.encode("latin-1") for s, _ in c["subs"]]
        data, off, length = pack(subs)
        B.launched_kernels()
        caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_TDFA)
        pair1 += "pair1" in B.launched_kernels()
        for i, (_, flat) in enumerate(c["subs"]):
            checked += 1
            ok = (status[i] == B.LC_NOMATCH and (caps[i] == -1).all()) if flat is None else (
                status[i] == B.LC_MATCH and list(caps[i]) == flat[skip:])
            if not ok:
                bad.append((c["p"], subs[i], int(status[i]), list(caps[i]), flat))
    return checked, pair1, bad


@pytest.mark.parametrize("compact", [None, "512"])
def test_one_stamp_pair_tables_forced_on_every_table_format(torch_dev, golden_dir, monkeypatch, compact):
    """The ONE-STAMP byte-pair tables (LC_TDFA_PAIR=2, device_tables.h TP_FORMAT 1; the default of large regex-A batches) forced
    onto every pattern that can carry them: the STANDARD 32-bit tables (compact unset: launches below 64 Ki lines) and the
    COMPACT 512-lane tables used for every batch (LC_TDFA_COMPACT=512).  Golden vectors (full match and search), the bench
    corpora A and B with failing lines and empty fields, ragged JSON/nginx lines, unaligned / empty / 64 KiB+ lines -- each
    against the oracle or the third-party goldens, each asserting that the one-stamp kernel is what ran.
    Semantics protected: core/common/StringTools.cpp:183-211 (regex_match) / :213-236 (regex_search)."""
    monkeypatch.setenv("LC_TDFA_PAIR", "2")
    if compact is None:
        monkeypatch.delenv("LC_TDFA_COMPACT", raising=False)
    else:
        monkeypatch.setenv("LC_TDFA_COMPACT", compact)
    checked, pair1, bad = _golden_on_tdfa(torch_dev, golden_dir, "regex_golden.json", 0, 2, 100000)
    assert not bad, bad[:5]
    assert checked > 3000 and pair1 > 100, (checked, pair1)
    checked, pair1, bad = _golden_on_tdfa(torch_dev, golden_dir, "regex_search_golden.json", B.LC_SYNTAX_SEARCH, 0, 100000)
    assert not bad, bad[:5]
    assert checked > 1000 and pair1 > 30, (checked, pair1)

    def on_device(pattern, data, off, length, sep, label):
        rx = B.GpuRegex(pattern)
        B.launched_kernels()
        caps, status = run_device(torch_dev, rx, data, off, length, sep=sep)
        names = B.launched_kernels()
        wide = compact is not None or len(status) >= 65536
        blob = rx.table(B.LC_TABLE_TDFA_WIDE_BLOB if wide else B.LC_TABLE_TDFA_BLOB, np.uint32)
        one_stamp = int(blob[7]) != 0 and int(blob[int(blob[7]) // 4 + 4]) == 1      # TD_OFF_PAIR, TP_FORMAT
        assert one_stamp, label
        assert ("pair1" in names) == one_stamp, (label, names)
        assert ("compact" in names) == wide, (label, names)
        ran.append((label, names))
        return caps, status

    ran = []

    for kind, pattern in (("A", corpus.REGEX_A), ("B", corpus.REGEX_B)):
        for n, kw in ((20000, dict(poison_every=13)), (20000, dict(poison_every=17, empty_every=3)), (70000, dict(poison_every=11, empty_every=5))):
            data, off, length = corpus.apache_batch(n, kind, **kw)
            exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:-1], length)
            caps, status = on_device(pattern, data, off, None, 1, (kind, n, kw))
            bad = np.nonzero((status != exp_status) | (caps != exp_caps).any(axis=1))[0]
            assert bad.size == 0, (kind, n, kw, bad[:8].tolist(), caps[bad[0]].tolist(), exp_caps[bad[0]].tolist())
            caps, status = on_device(pattern, data, off[:-1], length, 0, (kind, n, kw, "off+len"))
            assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)
    # ragged nginx / JSON lines (regex B; the JSON lines fail)
    data, off, length = corpus.mixed_batch(6000)
    exp_caps, exp_status = OracleRegex(corpus.REGEX_B).fullmatch_batch(data, off[:-1], length)
    caps, status = on_device(corpus.REGEX_B, data, off, None, 1, ("B", "mixed"))
    assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)
    # empty, one-byte, 15/16/17-byte, random short and long lines at unaligned offsets
    pattern = r"(\w*)\t?(\w*)(.*)"
    rng = np.random.default_rng(5)
    subs = [b"", b"a", b"\t", b"a\tb", b"x" * 15, b"x" * 16, b"x" * 17, b"ab\tcd" + b" tail" * 3000, b""]
    for _ in range(300):
        subs.append(bytes(rng.choice(list(b"ab\t _"), size=int(rng.integers(0, 70))).astype(np.uint8)))
    data, off, length = pack(subs)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off, length)
    caps, status = on_device(pattern, data, off, length, 0, "edge")
    assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)
    # lines around and beyond 64 KiB: the compact kernel hands them to the 32-bit tables' launch (which carries pair tables too)
    pattern = rb"(\w+) (\d+) (.*)\|(\w*)"
    subs = []
    for n in [0, 1, 20, 65534, 65535, 65536, 65537, 70001, 131072, 300]:
        body = b"key 12345 " + b"x" * max(0, n - 14) + b"|end"
        subs.append(body[:n] if n < 14 else body)
    subs += [b"key 1 " + b"y" * 65600, b"no match " * 8000]
    data, off, length = pack(subs)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off, length)
    caps, status = on_device(pattern, data, off, length, 0, "long")
    assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)
    assert exp_status.sum() >= 7


def test_long_lines_behind_the_first_256_blocks_of_a_large_batch(torch_dev):
    """Round 6: the second launch behind the COMPACT kernel (the lines of 64 KiB and more) comes with at most 256 workgroups that take
    the line blocks in turn.  A batch of 80 000 lines (313 blocks of 256: the COMPACT kernel is what a batch of 64 Ki lines and more
    gets) with such lines in the first block, beyond the 256th, and in the last one; and a batch without any."""
    from oracle.oracle import OracleRegex
    pattern = rb"(\w+) (\d+) (.*)\|(\w*)"
    rx = B.GpuRegex(pattern)
    assert rx.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32) is not None
    n = 80000
    subs = [b"k%d %d some text|e%d" % (i, i * 7, i % 10) if i % 9 else b"no separator here %d" % i for i in range(n)]
    for where, size in ((3, 65536), (66000, 70001), (79999, 65600), (70123, 131072)):
        subs[where] = b"key 12345 " + b"x" * (size - 14) + b"|end"
    subs[66001] = b"nothing " * 9000  # 72 000 bytes that do not match
    for lines in (subs, [s for s in subs if len(s) < 60000]):
        data, off, length = pack(lines)
        exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off, length)
        B.launched_kernels()
        caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_TDFA)
        assert "compact" in str(B.launched_kernels())
        assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)
    assert exp_status.sum() > n // 2


def test_persistent_wavefronts_of_the_one_stamp_kernel(torch_dev, monkeypatch):
    """Round 6, measured and left off (LC_TDFA_PERSIST=1): the COMPACT one-stamp kernel launched with as many workgroups as the chip
    holds, every wavefront going on with its 64 lines of the next block.  600 000 ragged lines (the launch needs two rounds of resident
    workgroups to take this form), failing lines, empty lines and lines of 64 KiB and more among them, against the oracle."""
    from oracle.oracle import OracleRegex
    monkeypatch.setenv("LC_TDFA_PERSIST", "1")
    pattern = rb"(\w+) (\d+) (.*)\|(\w*)"
    rx = B.GpuRegex(pattern)
    n = 600000
    rng = random.Random(5)
    subs = []
    for i in range(n):
        k = rng.randrange(12)
        if k == 0: subs.append(b"")
        elif k == 1: subs.append(b"no separator %d" % i)
        else: subs.append(b"k%d %d %s|e%d" % (i, i * 3, b"x" * rng.randrange(0, 90), i % 7))
    for where, size in ((5, 65536), (300001, 70001), (599999, 66000)):
        subs[where] = b"key 12345 " + b"y" * (size - 14) + b"|end"
    data, off, length = pack(subs)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off, length)
    B.launched_kernels()
    caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_TDFA)
    assert "persist" in str(B.launched_kernels())
    assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)
    assert exp_status.sum() > n // 2


def test_resumed_searches_on_long_lines_both_kernels(torch_dev):
    """lc_regex_match_device_from: a subset of the lines, each search resumed at its own offset (also beyond the first
    256-byte chunk of the NFA kernel and across the TDFA kernel's 64-byte stages), against the oracle's search(start)."""
    from oracle.oracle import OracleRegex
    torch = torch_dev
    dev = torch.device("cuda:0")
    rng = random.Random(77)
    words = [b"alpha", b"10.2.3.4", b"x=42", b"2024-01-04", b"beta99", b"--", b"k=", b"7"]
    lines = []
    for _ in range(700):
        n = rng.choice([3, 20, 60, 150, 400])
        lines.append(b" ".join(rng.choice(words) for _ in range(n)))
    data, off, length = pack(lines)
    n = len(lines)
    pad = np.zeros(len(data) + 16, dtype=np.uint8)
    pad[:len(data)] = data
    d_data = torch.from_numpy(pad).to(dev)
    d_off = torch.from_numpy(off.view(np.int32).copy()).to(dev)
    d_len = torch.from_numpy(length.view(np.int32).copy()).to(dev)
    subset = np.array(sorted(rng.sample(range(n), 500)), dtype=np.uint32)
    start = np.zeros(n, dtype=np.uint32)
    for i in subset:
        L = int(length[i])
        start[i] = rng.choice([0, 1, L // 2, max(0, L - 3), L, min(L, 255), min(L, 256), min(L, 257), min(L, 700)])
    d_sub = torch.from_numpy(subset.view(np.int32).copy()).to(dev)
    d_from = torch.from_numpy(start.view(np.int32).copy()).to(dev)
    for pattern in (rb"(\w+)=(\d*)", rb"(?<![0-9.])(\d+)\.(\d+)\.(\d+)\.(\d+)(?![0-9])", rb"\b(\d{4})-(\d\d)-(\d\d)\b", rb"(?>[a-z]+)(\d\d)"):
        rx = B.GpuRegex(pattern, syntax_flags=B.LC_SYNTAX_SEARCH)
        o = OracleRegex(pattern)
        G = rx.groups
        engines = ([B.LC_ENGINE_TDFA] if rx.info()["engine"] == B.LC_ENGINE_TDFA else []) + (
            [B.LC_ENGINE_NFA] if rx.has_nfa_program() else [])
        assert engines
        for eng in engines:
            d_caps = torch.full((n, 2 * G), -7, dtype=torch.int32, device=dev)
            d_status = torch.full((n,), 9, dtype=torch.uint8, device=dev)
            rx.match_device_from(d_data, d_off, d_len, len(subset), d_caps, d_status, d_lines=d_sub, d_from=d_from, engine=eng)
            torch.cuda.synchronize()
            caps, status = d_caps.cpu().numpy(), d_status.cpu().numpy()
            touched = np.zeros(n, dtype=bool)
            touched[subset] = True
            assert (status[~touched] == 9).all() and (caps[~touched] == -7).all()   # lines not listed are not written
            for i in subset:
                exp = o.search(lines[i], int(start[i]))
                if exp is None:
                    assert status[i] == B.LC_NOMATCH, (pattern, eng, i, int(start[i]))
                else:
                    assert status[i] == B.LC_MATCH and list(caps[i]) == [v for ab in exp for v in ab], (
                        pattern, eng, lines[i][:80], int(start[i]), list(caps[i]), exp)


@pytest.mark.parametrize("wave_max", ["0", "65536"])
def test_automata_too_large_for_lds_run_from_global_memory(torch_dev, monkeypatch, wave_max):
    """tdfa_l2_kernel / tdfa_wave_kernel: a tagged DFA of 1 000+ states (beyond the 64 KiB LDS window of the LDS kernels) keeps its
    tables in global memory and still runs as a DFA instead of falling to the NFA engine -- one value per LANE (large batches;
    LC_TDFA_WAVE_MAX=0 forces it here) or one value per WAVEFRONT (batches up to 64 Ki values: wave-uniform state, quiet runs
    crossed chunk-wise).  Full match in both input forms, values with long quiet runs and runs that end at chunk boundaries, and
    a search with a listed subset of lines and resume offsets, against the oracle."""
    monkeypatch.setenv("LC_TDFA_WAVE_MAX", wave_max)
    rng = random.Random(99)
    full = rb"(?:a|b)*a(?:a|b){12}(c+)(d*)"
    rx = B.GpuRegex(full)
    info = rx.info()
    assert info["engine"] == B.LC_ENGINE_TDFA and info["states"] > 1000 and rx.table(B.LC_TABLE_TDFA_BLOB, np.uint32) is None
    assert rx.table(B.LC_TABLE_TDFA_L2_BLOB, np.uint32) is not None
    def subject():
        head = bytes(rng.choice(b"ab") for _ in range(rng.randint(0, 300)))
        return head + rng.choice([b"", b"c", b"ccc", b"cd", b"ccddd", b"x", b"cdc"])
    subs = [subject() for _ in range(3000)] + [b"", b"a" * 10 + b"c", b"ab" * 2000 + b"a" + b"b" * 12 + b"cccd"]
    # quiet runs: "c+" and "d*" stay in their states -- runs of 1 .. 1100 bytes that end on, before and behind 256-byte chunk borders
    for run in (1, 2, 3, 5, 200, 254, 255, 256, 257, 258, 511, 512, 513, 1100):
        for lead in (0, 1, 2, 3, 13):
            subs.append(b"b" * lead + b"a" + b"b" * 12 + b"c" * run + b"d" * (run // 3))
            subs.append(b"b" * lead + b"a" + b"b" * 12 + b"c" * run + b"x")
    data, off, length = pack(subs)
    o = OracleRegex(full)
    exp_caps, exp_status = o.fullmatch_batch(data, off, length)
    B.launched_kernels()
    caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_TDFA)
    assert ("tdfa_l2_kernel:wave" in B.launched_kernels()) == (wave_max != "0")
    assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps) and 100 < exp_status.sum() < len(subs) - 100
    sep_data = np.frombuffer(b"\n".join(subs) + b"\n", dtype=np.uint8)
    sep_off = np.zeros(len(subs) + 1, dtype=np.uint32)
    sep_off[1:] = np.cumsum(length.astype(np.uint64) + 1).astype(np.uint32)
    caps2, status2 = run_device(torch_dev, rx, sep_data, sep_off, None, sep=1)
    assert np.array_equal(status2, exp_status) and np.array_equal(caps2, exp_caps)
    # search, subset of lines, resumed inside the line
    spat = rb"a(?:a|b){11}(c+)"
    srx = B.GpuRegex(spat, syntax_flags=B.LC_SYNTAX_SEARCH)
    assert srx.info()["engine"] == B.LC_ENGINE_TDFA and srx.table(B.LC_TABLE_TDFA_L2_BLOB, np.uint32) is not None
    so = OracleRegex(spat)
    torch = torch_dev
    dev = torch.device("cuda:0")
    n = len(subs)
    pad = np.zeros(len(data) + 16, dtype=np.uint8)
    pad[:len(data)] = data
    d_data = torch.from_numpy(pad).to(dev)
    d_off = torch.from_numpy(off.view(np.int32).copy()).to(dev)
    d_len = torch.from_numpy(length.view(np.int32).copy()).to(dev)
    subset = np.array(sorted(rng.sample(range(n), 700)), dtype=np.uint32)
    start = np.zeros(n, dtype=np.uint32)
    for i in subset:
        L = int(length[i])
        start[i] = rng.choice([0, 0, 1, L // 2, max(0, L - 3), L, min(L, 17)])
    d_sub = torch.from_numpy(subset.view(np.int32).copy()).to(dev)
    d_from = torch.from_numpy(start.view(np.int32).copy()).to(dev)
    G = srx.groups
    d_caps = torch.full((n, 2 * G), -7, dtype=torch.int32, device=dev)
    d_status = torch.full((n,), 9, dtype=torch.uint8, device=dev)
    srx.match_device_from(d_data, d_off, d_len, len(subset), d_caps, d_status, d_lines=d_sub, d_from=d_from, engine=B.LC_ENGINE_TDFA)
    torch.cuda.synchronize()
    caps, status = d_caps.cpu().numpy(), d_status.cpu().numpy()
    touched = np.zeros(n, dtype=bool)
    touched[subset] = True
    assert (status[~touched] == 9).all() and (caps[~touched] == -7).all()
    hits = 0
    for i in subset:
        exp = so.search(subs[i], int(start[i]))
        if exp is None:
            assert status[i] == B.LC_NOMATCH, (i, int(start[i]))
        else:
            hits += 1
            assert status[i] == B.LC_MATCH and list(caps[i]) == [v for ab in exp for v in ab], (subs[i][-40:], int(start[i]))
    assert hits > 50


def test_random_atomic_patterns_on_both_kernels(torch_dev):
    """Fresh random patterns with atomic groups / possessive quantifiers / look assertions (not the committed golden set),
    full-match and search, TDFA and NFA kernels against the oracle."""
    import importlib.util
    pytest.importorskip("regex")
    from oracle.oracle import OracleRegex
    spec = importlib.util.spec_from_file_location(
        "gen_atomic_golden", os.path.join(os.path.dirname(__file__), "golden", "gen_atomic_golden.py"))
    agen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(agen)
    rng = random.Random(4242)
    checked = overflow = 0
    for _ in range(220):
        p = agen.gen(rng)
        try:
            o = OracleRegex(p)
        except ValueError:
            continue
        subs = [bytes(rng.choice(b"abc1 ") for _ in range(rng.randint(0, 10))) for _ in range(10)]
        data, off, length = pack(subs)
        for flags, fn in ((0, o.fullmatch), (B.LC_SYNTAX_SEARCH, o.search)):
            try:
                rx = B.GpuRegex(p, syntax_flags=flags)
            except B.RegexUnsupportedError:
                continue
            engines = ([B.LC_ENGINE_TDFA] if rx.info()["engine"] == B.LC_ENGINE_TDFA else []) + (
                [B.LC_ENGINE_NFA] if rx.has_nfa_program() else [])
            for eng in engines:
                caps, status = run_device(torch_dev, rx, data, off, length, engine=eng)
                for i, s in enumerate(subs):
                    if status[i] == B.LC_OVERFLOW:
                        overflow += 1
                        continue
                    exp = fn(s)
                    want = None if exp is None else [v for ab in (exp if flags else exp[1:]) for v in ab]
                    checked += 1
                    got = None if status[i] == B.LC_NOMATCH else list(caps[i])
                    assert got == want, (p, s, flags, eng, got, want)
    assert checked > 5000 and overflow < checked // 50


def test_run_captures_on_both_kernels(torch_dev):
    """"(?=(S*))" groups: begin stamped by the automaton, end filled in by run_capture_kernel (gpu_runtime.hip)"""
    from tests.helpers.wide_patterns import RUN_CAPTURE_PATTERNS, RUN_CAPTURE_SUBJECTS
    data, off, length = pack(RUN_CAPTURE_SUBJECTS)
    checked = 0
    for pat in RUN_CAPTURE_PATTERNS:
        o = OracleRegex(pat)
        for eng in (B.LC_ENGINE_TDFA, B.LC_ENGINE_NFA):
            rx = B.GpuRegex(pat, engine=eng)
            caps, status = run_device(torch_dev, rx, data, off, length, engine=eng)
            for i, s in enumerate(RUN_CAPTURE_SUBJECTS):
                want = o.fullmatch(s)
                checked += 1
                if want is None:
                    assert status[i] == B.LC_NOMATCH and (caps[i] == -1).all(), (pat, s)
                else:
                    assert status[i] == B.LC_MATCH and list(caps[i]) == [v for be in want for v in be][2:], (pat, s)
    assert checked == 2 * len(RUN_CAPTURE_PATTERNS) * len(RUN_CAPTURE_SUBJECTS)


@pytest.mark.parametrize("compact", ["256", "512", "1024"])
def test_compact_kernels_and_their_long_line_second_pass(torch_dev, monkeypatch, compact):
    """Opt-in COMPACT kernel variants (LC_TDFA_COMPACT when the pattern is compiled): 16-bit offsets, swizzled staging,
    1024 = byte-indexed rows; lines of 64 KiB and more are left to a second launch of the 32-bit kernel.  Lengths around
    the boundary, captures at both ends, resumed searches, and the bench corpus."""
    monkeypatch.setenv("LC_TDFA_COMPACT", compact)
    pattern = rb"(\w+) (\d+) (.*)\|(\w*)"
    rx = B.GpuRegex(pattern)
    assert rx.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32) is not None
    lens = [0, 1, 20, 65534, 65535, 65536, 65537, 70001, 131072, 300]
    subs = []
    for i, n in enumerate(lens):
        body = b"key 12345 " + b"x" * max(0, n - 14) + b"|end"
        subs.append(body[:n] if n < 14 else body)
    subs += [b"key 1 " + b"y" * 65600, b"no match " * 8000]
    data, off, length = pack(subs)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off, length)
    caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_TDFA)
    assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)
    assert exp_status.sum() >= 7
    # search pattern resumed deep inside a long line: offsets are relative to the resume point inside the kernel
    srx = B.GpuRegex(rb"(\d+)=(\w+)", syntax_flags=B.LC_SYNTAX_SEARCH)
    assert srx.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32) is not None
    line = b"-" * 70000 + b" 77=ab " + b"-" * 70000 + b" 8=c"
    d_data = torch_dev.from_numpy(np.frombuffer(line, np.uint8).copy()).cuda()
    d_off = torch_dev.zeros(1, dtype=torch_dev.int32, device="cuda")
    d_len = torch_dev.tensor([len(line)], dtype=torch_dev.int32, device="cuda")
    got = []
    for frm in (0, 69990, 70007, 140000):
        d_from = torch_dev.tensor([frm], dtype=torch_dev.int32, device="cuda")
        d_caps = torch_dev.full((1, 2 * srx.groups), -7, dtype=torch_dev.int32, device="cuda")
        d_status = torch_dev.full((1,), 9, dtype=torch_dev.uint8, device="cuda")
        srx.match_device_from(d_data, d_off, d_len, 1, d_caps, d_status, d_from=d_from, engine=B.LC_ENGINE_TDFA)
        torch_dev.cuda.synchronize()
        got.append((int(d_status[0]), d_caps.cpu().numpy()[0].tolist()))
    a, b = 70001, 140008
    assert got == [(1, [a, a + 5, a, a + 2, a + 3, a + 5]), (1, [a, a + 5, a, a + 2, a + 3, a + 5]),
                   (1, [b, b + 3, b, b + 1, b + 2, b + 3]), (1, [b, b + 3, b, b + 1, b + 2, b + 3])], got
    for kind, regex in (("A", corpus.REGEX_A), ("B", corpus.REGEX_B)):
        data, off, length = corpus.apache_batch(5000, kind, poison_every=13)
        brx = B.GpuRegex(regex)
        assert brx.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32) is not None
        exp_caps, exp_status = OracleRegex(regex).fullmatch_batch(data, off[:-1], length)
        caps, status = run_device(torch_dev, brx, data, off, None, sep=1, engine=B.LC_ENGINE_TDFA)
        assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)


@pytest.mark.parametrize("kind", ["A", "B"])
def test_large_batch_takes_the_compact_kernel_and_stays_bit_exact(torch_dev, kind):
    """>= 64 Ki lines: the launcher picks the 16-bit-register kernel + the 32-bit mop-up launch (gpu_runtime.hip launchTdfa).
    Matching and poisoned lines against the oracle, both input forms."""
    pattern = corpus.REGEX_A if kind == "A" else corpus.REGEX_B
    rx = B.GpuRegex(pattern)
    assert rx.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32) is not None
    n = 70000
    data, off, length = corpus.apache_batch(n, kind, poison_every=11)
    exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:-1], length)
    def same(caps, status):
        bad = np.nonzero((status != exp_status) | (caps != exp_caps).any(axis=1))[0]
        assert bad.size == 0, (bad.size, bad[:8].tolist(), status[bad[:4]].tolist(), exp_status[bad[:4]].tolist(),
                               caps[bad[0]].tolist(), exp_caps[bad[0]].tolist(), int(length[bad[0]]))
    same(*run_device(torch_dev, rx, data, off, None, sep=1))
    same(*run_device(torch_dev, rx, data, off[:-1], length))
    assert kind == "B" or 0 < exp_status.sum() < n      # the poisoned lines still satisfy the permissive regex B


def test_reference_boost_regex_search_vectors_on_the_device(torch_dev, golden_dir):
    """core/unittest/common/StringToolsUnittest.cpp:128-209 (TestBoostRegexSearch, regex_search + match_continuous): the
    reference's own expectations, asked of every device engine that can run the pattern in LC_SYNTAX_PREFIX mode."""
    with open(os.path.join(golden_dir, "boost_search_vectors.json")) as f:
        d = json.load(f)
    checked = 0
    for c in d["cases"]:
        rx = B.GpuRegex(c["p"].encode("latin-1"), syntax_flags=B.LC_SYNTAX_PREFIX)
        subs = [s.encode("latin-1") for s, _ in c["subs"]]
        data, off, length = pack(subs)
        engines = ([B.LC_ENGINE_NFA] if rx.has_nfa_program() else []) + (
            [B.LC_ENGINE_TDFA] if rx.info()["engine"] == B.LC_ENGINE_TDFA else [])
        assert engines
        for eng in engines:
            caps, status = run_device(torch_dev, rx, data, off, length, engine=eng)
            for i, (_, want) in enumerate(c["subs"]):
                checked += 1
                assert (status[i] == B.LC_MATCH) == want, (c["cite"], c["subs"][i][0], eng)
    assert checked >= 16


def test_doomed_spawn_skipping_on_the_device(torch_dev):
    """nfa_match_kernel / nfa_wide_kernel cross bytes whose spawned threads are gone behind the next byte (NF_OFF_QUASI) -- free text
    in the middle of a format, long values, matches and near-misses, full match / search / anchored search on the NFA engine, against
    the oracle.  The CPU twin (tests/test_host_compilers.py) checks the same shapes on the table interpreter."""
    from tests.test_host_compilers import QUASI_PATTERNS
    rng = random.Random(31)
    words = [b"IPSEC:", b"An", b"outbound", b"SA", b"S", b"SP", b"(SPI=", b"0x1f)", b"between", b"12", b"and", b"34", b"(Primary)", b"Monitoring",
             b"on", b"interface", b"waiting", b"wait", b"w", b"Group", b"=", b",", b"IP", b"NAT", b"a", b"b", b"bc", b"bcd", b"bce", b"denied", b"den",
             b"dropped", b"tcp", b"src", b"dst", b"7", b"x", b"y", b"x1,2,y", b",y"]
    heads = {0: b"IPSEC: An outbound %s SA (SPI= 0x7) between 1 and 2", 1: b"(Primary) Monitoring on interface %s waiting", 2: b"Group = %s, IP = 9, NAT",
             3: b"a%sbcd%sbce", 4: b"denied tcp src %s:80 dst", 5: b"x1,%s,y"}
    checked = 0
    for k, pat in enumerate(QUASI_PATTERNS):
        subs = []
        for _ in range(300):
            filler = b" ".join(rng.choice(words) for _ in range(rng.choice([1, 5, 40, 300, 800])))
            s = heads[k].replace(b"%s", filler) if rng.random() < 0.6 else filler
            subs.append(rng.choice([b"", b"junk "]) + s + rng.choice([b"", b" tail"]))
        data, off, length = pack(subs)
        o = OracleRegex(pat)
        for flags in (0, B.LC_SYNTAX_SEARCH):
            rx = B.GpuRegex(pat, syntax_flags=flags, engine=B.LC_ENGINE_NFA)
            assert rx.table(B.LC_TABLE_NFA_BLOB, np.uint32)[23] != 0 or k == 5      # NF_OFF_QUASI: the tables exist for these shapes
            caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_NFA)
            for i, s in enumerate(subs):
                want = o.search(s) if flags else o.fullmatch(s)
                checked += 1
                if want is None:
                    assert status[i] == B.LC_NOMATCH, (pat, flags, s[:80])
                else:
                    exp = [v for be in want for v in be]
                    assert status[i] == B.LC_MATCH and list(caps[i]) == (exp if flags else exp[2:]), (pat, flags, s[:80])
    assert checked == 2 * 300 * len(QUASI_PATTERNS)


def test_wide_kernel_as_the_first_chance(torch_dev, golden_dir, monkeypatch):
    """Round 5, "wide first" (nfa_wide_kernel.hpp `first`): the two-threads-per-lane kernel walks EVERY line instead of only the lines
    the one-thread-per-lane kernel gave up on.  LC_NFA_WIDE_FIRST=1 sends every whole chain that way: the golden vectors, the bench
    corpus with poisoned lines, lines beyond 64 and beyond 128 threads, a resumed search -- all equal to the oracle, and to what the
    usual order of the kernels gives."""
    monkeypatch.setenv("LC_NFA_WIDE_FIRST", "1")
    with open(os.path.join(golden_dir, "regex_golden.json")) as f:
        golden = json.load(f)
    bad, checked = [], 0
    for c in golden["cases"][::3]:
        rx = B.GpuRegex(c["p"].encode("latin-1"))
        if not rx.has_nfa_program():
            continue
        subs = [s.encode("latin-1") for s, _ in c["subs"]]
        data, off, length = pack(subs)
        caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_NFA)
        for i, (_, flat) in enumerate(c["subs"]):
            checked += 1
            if flat is None:
                ok = status[i] == B.LC_NOMATCH and (caps[i] == -1).all()
            else:
                ok = status[i] == B.LC_MATCH and list(caps[i]) == flat[2:]
            if not ok:
                bad.append((c["p"], subs[i], int(status[i]), list(caps[i]), flat))
    assert checked > 1300 and not bad, bad[:5]
    B.launched_kernels()
    for kind in ("A", "B"):
        pattern = corpus.REGEX_A if kind == "A" else corpus.REGEX_B
        data, off, length = corpus.apache_batch(3000, kind, poison_every=11)
        exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:-1], length)
        rx = B.GpuRegex(pattern)
        caps_n, status_n = run_device(torch_dev, rx, data, off, None, sep=1, engine=B.LC_ENGINE_NFA)
        assert np.array_equal(status_n, exp_status) and np.array_equal(caps_n, exp_caps)
    launched = B.launched_kernels()
    assert "nfa_wide_kernel" in launched and "nfa_match_kernel" not in launched, launched
    # more than 64 threads (decided here), more than 128 (goes on to the decide kernels: this kernel raises the flag itself)
    for pattern, subs in ((r"(.*)a(.{70})", [b"a" * 100, b"b" * 10, b"a" + b"b" * 70, b"xa" * 80, b"a" * 71, b""]),
                          (r"(.*)a.{140}", [b"a" * 200, b"a" + b"b" * 140, b"a" * 140, b"b"])):
        data, off, length = pack(subs)
        exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off, length)
        rx = B.GpuRegex(pattern, engine=B.LC_ENGINE_NFA)
        caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_NFA)
        assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps), pattern
    srx = B.GpuRegex(r"a(.{70})b", syntax_flags=B.LC_SYNTAX_SEARCH, engine=B.LC_ENGINE_NFA)
    line = b"b" + b"a" * 90 + b"b" + b"a" * 80 + b"b"
    d_data = torch_dev.from_numpy(np.frombuffer(line + b"\0" * 16, np.uint8).copy()).cuda()
    d_off = torch_dev.zeros(1, dtype=torch_dev.int32, device="cuda")
    d_len = torch_dev.tensor([len(line)], dtype=torch_dev.int32, device="cuda")
    for frm in (0, 21):
        d_from = torch_dev.tensor([frm], dtype=torch_dev.int32, device="cuda")
        d_caps = torch_dev.full((1, 2 * srx.groups), -7, dtype=torch_dev.int32, device="cuda")
        d_status = torch_dev.full((1,), 9, dtype=torch_dev.uint8, device="cuda")
        srx.match_device_from(d_data, d_off, d_len, 1, d_caps, d_status, d_from=d_from, engine=B.LC_ENGINE_NFA)
        torch_dev.cuda.synchronize()
        o = OracleRegex(r"a(.{70})b").search(line, frm)
        assert int(d_status[0]) == 1 and d_caps.cpu().numpy()[0].tolist() == [v for be in o for v in be], (frm, o)


@pytest.mark.parametrize("lds", ["0", "1"])
def test_small_automata_on_the_wave_kernel(torch_dev, golden_dir, monkeypatch, lds):
    """Round 5: a handle that asks for the wave-per-value kernel (lc_regex_prefer_wave_tdfa: the Grok matcher's entries) -- the walk
    with its state in SGPRs and the classes of a chunk looked up once; and, LC_TDFA_WAVE_LDS_TRANS=1 (measured, left off), transition
    table + register programs staged into LDS when they fit 48 KB (tdfa_wave_kernel<LT>).  Same results as the oracle on the bench
    corpora (poisoned lines included), on the golden full-match vectors, and on resumed searches."""
    monkeypatch.setenv("LC_TDFA_WAVE_LDS_TRANS", lds)
    for kind in ("A", "B"):
        pattern = corpus.REGEX_A if kind == "A" else corpus.REGEX_B
        rx = B.GpuRegex(pattern)
        assert rx.prefer_wave_tdfa()
        data, off, length = corpus.apache_batch(3000, kind, poison_every=11)
        exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:-1], length)
        B.launched_kernels()
        caps, status = run_device(torch_dev, rx, data, off, None, sep=1, engine=B.LC_ENGINE_TDFA)
        assert ("tdfa_l2_kernel:wave:lds" in B.launched_kernels()) == (lds == "1")
        assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps)
    with open(os.path.join(golden_dir, "regex_golden.json")) as f:
        golden = json.load(f)
    bad, checked = [], 0
    for c in golden["cases"][::4]:
        rx = B.GpuRegex(c["p"].encode("latin-1"))
        if rx.info()["engine"] != B.LC_ENGINE_TDFA or not rx.prefer_wave_tdfa():
            continue
        subs = [s.encode("latin-1") for s, _ in c["subs"]]
        data, off, length = pack(subs)
        caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_TDFA)
        for i, (_, flat) in enumerate(c["subs"]):
            checked += 1
            ok = (status[i] == B.LC_NOMATCH and (caps[i] == -1).all()) if flat is None else \
                 (status[i] == B.LC_MATCH and list(caps[i]) == flat[2:])
            if not ok:
                bad.append((c["p"], subs[i], int(status[i]), list(caps[i]), flat))
    assert checked > 800 and not bad, bad[:5]
    srx = B.GpuRegex(r"(\d+)-([a-z]+)", syntax_flags=B.LC_SYNTAX_SEARCH)
    assert srx.prefer_wave_tdfa()
    line = b"xx 12-ab 345-cde " + b"y" * 600 + b" 6-f"
    d_data = torch_dev.from_numpy(np.frombuffer(line + b"\0" * 16, np.uint8).copy()).cuda()
    d_off = torch_dev.zeros(1, dtype=torch_dev.int32, device="cuda")
    d_len = torch_dev.tensor([len(line)], dtype=torch_dev.int32, device="cuda")
    for frm in (0, 4, 9, 20, len(line) - 3):
        d_from = torch_dev.tensor([frm], dtype=torch_dev.int32, device="cuda")
        d_caps = torch_dev.full((1, 2 * srx.groups), -7, dtype=torch_dev.int32, device="cuda")
        d_status = torch_dev.full((1,), 9, dtype=torch_dev.uint8, device="cuda")
        srx.match_device_from(d_data, d_off, d_len, 1, d_caps, d_status, d_from=d_from, engine=B.LC_ENGINE_TDFA)
        torch_dev.cuda.synchronize()
        o = OracleRegex(r"(\d+)-([a-z]+)").search(line, frm)
        if o is None:
            assert int(d_status[0]) == B.LC_NOMATCH, frm
        else:
            assert int(d_status[0]) == 1 and d_caps.cpu().numpy()[0].tolist() == [v for be in o for v in be], (frm, o)


def test_thread_list_kernels_on_follow_lists_by_byte_class(torch_dev, golden_dir, monkeypatch):
    """LC_NFA_CLASS_LISTS=1 (device_tables.h NF_OFF_CSTART: measured on configs[2], no gain, left off): a step's candidates come from
    the lists of the paths whose target takes the byte instead of the whole follow lists -- the narrow kernel, the wide kernel
    (second chance and first chance), atomic groups; all equal to the oracle."""
    monkeypatch.setenv("LC_NFA_CLASS_LISTS", "1")
    with open(os.path.join(golden_dir, "regex_golden.json")) as f:
        golden = json.load(f)
    bad, checked = [], 0
    for c in golden["cases"][::4]:
        rx = B.GpuRegex(c["p"].encode("latin-1"))
        if not rx.has_nfa_program():
            continue
        assert int(rx.table(B.LC_TABLE_NFA_BLOB, np.uint32)[24]) != 0
        subs = [s.encode("latin-1") for s, _ in c["subs"]]
        data, off, length = pack(subs)
        caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_NFA)
        for i, (_, flat) in enumerate(c["subs"]):
            checked += 1
            ok = (status[i] == B.LC_NOMATCH and (caps[i] == -1).all()) if flat is None else \
                 (status[i] == B.LC_MATCH and list(caps[i]) == flat[2:])
            if not ok:
                bad.append((c["p"], subs[i], int(status[i]), list(caps[i]), flat))
    assert checked > 1000 and not bad, bad[:5]
    for wide_first in ("0", "1"):
        monkeypatch.setenv("LC_NFA_WIDE_FIRST", wide_first)
        for pattern, subs in ((r"(.*)a(.{70})", [b"a" * 100, b"b" * 10, b"a" + b"b" * 70, b"xa" * 80, b"a" * 71, b""]),
                              (r"(.*)a.{140}", [b"a" * 200, b"a" + b"b" * 140, b"a" * 140, b"b"]),
                              (r"(?>a+)(b|bc)+d", [b"aaabbcd", b"abcbcd", b"aab", b"d"])):
            data, off, length = pack(subs)
            exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off, length)
            rx = B.GpuRegex(pattern, engine=B.LC_ENGINE_NFA)
            caps, status = run_device(torch_dev, rx, data, off, length, engine=B.LC_ENGINE_NFA)
            assert np.array_equal(status, exp_status) and np.array_equal(caps, exp_caps), (pattern, wide_first)
        data, off, length = corpus.apache_batch(2000, "A", poison_every=11)
        exp_caps, exp_status = OracleRegex(corpus.REGEX_A).fullmatch_batch(data, off[:-1], length)
        caps_n, status_n = run_device(torch_dev, B.GpuRegex(corpus.REGEX_A), data, off, None, sep=1, engine=B.LC_ENGINE_NFA)
        assert np.array_equal(status_n, exp_status) and np.array_equal(caps_n, exp_caps)


def test_atomic_lazy_loop_commits_on_the_thread_list_engine(torch_dev):
    """A lazy loop inside an atomic group whose exit leaves the group: '(?>a+?.)' takes "a" + one byte and is committed -- "aa1" is no
    full match.  The thread-list kernel's doomed-spawn rows once skipped the step that commits (found by the forced-engine device fuzz of
    round 6, profiles/round6_bt_fuzz_gpu.txt D).  Every engine the handle can be asked for, against the oracle."""
    subs = [b'aa1', b'aa', b'a1', b'aaa', b'', b'a', b'ca', b'c1', b'aaaa1', b'xby', b'xbby', b'xaby', b'xbay', b'xy'] * 5
    for p in (b'(?>a+?.)', b'(?>(?:(c)|(?:a)+?).)', b'(?>(?:c|a+?).)', b'x(?>b*?[ab])y'):
        o = OracleRegex(p)
        for eng in (B.LC_ENGINE_NFA, B.LC_ENGINE_AUTO, B.LC_ENGINE_TDFA, B.LC_ENGINE_BT):
            for flags, search in ((0, False), (B.LC_SYNTAX_SEARCH, True)):
                rx = B.GpuRegex(p, syntax_flags=flags, engine=eng)
                data, off, length = pack(subs)
                caps, status = run_device(torch_dev, rx, data, off, length)
                for i, s in enumerate(subs):
                    w = o.search(s) if search else o.fullmatch(s)
                    exp = None if w is None else [v for ab in w for v in ab][0 if search else 2:]
                    if exp is None:
                        assert status[i] == B.LC_NOMATCH and (caps[i] == -1).all(), (p, eng, search, s)
                    else:
                        assert status[i] == B.LC_MATCH and list(caps[i]) == exp, (p, eng, search, s, list(caps[i]), exp)
