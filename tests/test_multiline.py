"""Multiline splitter: the record oracle against the reference's own unit-test cases and the
Init rules on the CPU; the device-flag path against the oracle and the same cases with -m gpu.
Reference: core/plugin/processor/inner/ProcessorSplitMultilineLogStringNative.cpp, core/file_server/MultilineOptions.cpp."""
import json
import os
import random

import pytest

from loongcollector_amd import binding as B
from loongcollector_amd.multiline import Multiline, MultilineInitError
from oracle.multiline_oracle import MultilineOracle


@pytest.fixture(scope="module")
def vectors(golden_dir):
    with open(os.path.join(golden_dir, "multiline_vectors.json"), encoding="utf-8") as f:
        return json.load(f)


def _value(vectors, tokens):
    return "\n".join(vectors["lines"][t] for t in tokens).encode("utf-8")


def test_oracle_reproduces_the_reference_unit_test_cases(vectors):
    assert len(vectors["cases"]) >= 40
    for c in vectors["cases"]:
        val = _value(vectors, c["in"])
        recs, counters = MultilineOracle(**c["config"]).split(val)
        got = [val[b:b + l].decode("utf-8").split("\n") for b, l, _ in recs]
        assert got == [[vectors["lines"][t] for t in ev] for ev in c["out"]], c["cite"]
        assert counters[0] == len(c["in"])


def test_init_rules():
    """ProcessorSplitMultilineLogStringNative.cpp:66-76 / .h:68-70 (patterns as written), MultilineOptions.cpp:203-205,250-266"""
    m = Multiline(StartPattern="Exception.*", ContinuePattern=r"\s+at\s.*", EndPattern=r"\s*\.\.\.\d+ more")
    assert m.patterns == {"start": True, "continue": True, "end": True} and m.is_multiline   # the processor keeps all three
    # ContinuePattern alone / no pattern: the reference's ProcessEvent has no defined behaviour (it indexes an empty regex
    # vector, .cpp:176-184) and the input plugin never builds the processor for it -- refused
    for cfg in ({"ContinuePattern": r"\s+at\s.*"}, {}, {"StartPattern": "("}):
        with pytest.raises(MultilineInitError):
            Multiline(**cfg)
        with pytest.raises(ValueError):
            MultilineOracle(**cfg)
    m = Multiline(StartPattern=".*")
    assert m.patterns["start"] is True and not m.is_multiline    # present for the processor, "not multiline" for the input
    assert Multiline(EndPattern="x$").patterns["end"] is True
    # an invalid regex is ignored with a warning (MultilineOptions.cpp:109-118); one the device cannot run fails Init
    m = Multiline(StartPattern="(", EndPattern="x")
    assert m.patterns == {"start": False, "continue": False, "end": True} and "StartPattern is not a valid regex" in m.warnings
    assert MultilineOracle(StartPattern="(", EndPattern="x").start is None
    with pytest.raises(MultilineInitError):
        Multiline(StartPattern=r"(?<=a+)b")
    # (round 6: back-references run on the device backtracking engine -- tests/test_backref.py has the split on the device)
    assert Multiline(StartPattern=r"(a)\1").patterns["start"] is True
    o = MultilineOracle(StartPattern="a", ContinuePattern="b", EndPattern="c")
    assert o.cont is not None and o.is_multiline
    # "END$" as written: the line has to end there (the stripped form would also accept "END7x")
    recs, _ = MultilineOracle(StartPattern="BEGIN", EndPattern=r"END\d*$").split(b"BEGIN\nEND7x\nEND7")
    assert recs == [(0, 16, 1)]


def test_no_cpu_path():
    if B.load().lc_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(B.GpuUnavailableError):
        Multiline(StartPattern="Exception").split(b"Exception\n  at x")


@pytest.mark.gpu
def test_reference_cases_on_the_device(vectors):
    for c in vectors["cases"]:
        val = _value(vectors, c["in"])
        recs, counters = Multiline(**c["config"]).split(val)
        got = [val[b:b + l].decode("utf-8").split("\n") for b, l, _ in recs]
        assert got == [[vectors["lines"][t] for t in ev] for ev in c["out"]], c["cite"]
        assert (recs, counters) == MultilineOracle(**c["config"]).split(val)


@pytest.mark.gpu
@pytest.mark.parametrize("config", [
    {"StartPattern": r"\d{4}-\d{2}-\d{2} .*"},
    {"StartPattern": r"\[\w+\]", "ContinuePattern": r"\s+at\s.*", "UnmatchedContentTreatment": "discard"},
    {"StartPattern": "BEGIN", "EndPattern": r"END\d*$"},
    {"ContinuePattern": r"\s+.*", "EndPattern": r"\}"},
    {"EndPattern": ";$", "UnmatchedContentTreatment": "discard"},
    {"StartPattern": r"\[\w+\].*", "ContinuePattern": r"\s+at\s.*", "EndPattern": r"\}$"},   # all three stay in use
    {"StartPattern": ".*"},                                                                  # every line starts a record
])
def test_random_buffers_against_the_oracle(config):
    rng = random.Random(41)
    pool = [b"2024-01-04 boom", b"  at com.example.A.b(A.java:1)", b"[ERROR] x", b"BEGIN tx", b"END7", b"END", b"END7x", b"}x", b"}", b"{",
            b"stmt;", b"noise", b"", b"\tcontinued", b"2024-13-99 not checked"]
    o = MultilineOracle(**config)
    m = Multiline(**config)
    for _ in range(200):
        val = b"\n".join(rng.choice(pool) for _ in range(rng.randint(0, 12)))
        if rng.random() < 0.2:
            val += b"\n"
        assert m.split(val) == o.split(val), (config, val)


# ---------------------------------------------------------------------------------------------- the processors on whole event groups
def _group_of_one(value, meta=None, position=None):
    from loongcollector_amd.processor import EventGroup
    ev = {"contents": {"content": value.decode("utf-8")}, "timestamp": 12345678901, "timestampNanosecond": 0, "type": 1}
    if position:
        ev["fileOffset"], ev["rawSize"] = position
    g = {"events": [ev]}
    if meta:
        g["metadata"] = meta
    return EventGroup(g)


@pytest.mark.gpu
def test_split_processor_builds_the_events_of_the_reference_cases(vectors):
    """lc_multiline_process_group = ProcessorSplitMultilineLogStringNative::Process: for every `// case:` block of the reference's
    unit test, the events that come out -- contents, order, timestamps, type -- are the ones the reference asserts."""
    total_in = total_unmatched = 0
    for c in vectors["cases"]:
        m = Multiline(**c["config"])
        g = _group_of_one(_value(vectors, c["in"]))
        m.process(g)
        out = g.to_dict().get("events", []) if g.to_json() != "null" else []
        want = ["\n".join(vectors["lines"][t] for t in ev) for ev in c["out"]]
        assert [e["contents"]["content"] for e in out] == want, c["cite"]
        assert all(e["type"] == 1 and e["timestamp"] == 12345678901 for e in out), c["cite"]
        matched_lines, unmatched_lines, matched_events = m.counters()
        assert matched_lines + unmatched_lines == len(c["in"]), c["cite"]
        total_in += len(c["in"])
        total_unmatched += unmatched_lines
    assert total_in > 100 and total_unmatched > 10


@pytest.mark.gpu
def test_split_processor_raw_content_positions_and_passthrough(vectors):
    from loongcollector_amd.processor import EventGroup
    B_, C_, U_ = (vectors["lines"][t] for t in "BCU")
    # TestEnableRawEvent (ProcessorSplitMultilineLogStringNativeUnittest.cpp:1182-1246): unmatched + begin, discard -> ONE raw event
    m = Multiline(StartPattern=vectors["patterns"]["LOG_BEGIN_REGEX"], UnmatchedContentTreatment="discard", EnableRawContent=True)
    g = _group_of_one((U_ + "\n" + B_).encode())
    m.process(g)
    assert g.to_dict()["events"] == [{"content": B_, "timestamp": 12345678901, "timestampNanosecond": 0, "type": 4}]
    # CreateNewEvent :325-337: position = source offset + delta; length = line + 1, or the rest of the source event for the record
    # emitted with isLastLog; the LOG_FILE_OFFSET_KEY content when the group carries that metadata
    m = Multiline(StartPattern=vectors["patterns"]["LOG_BEGIN_REGEX"], ContinuePattern=vectors["patterns"]["LOG_CONTINUE_REGEX"])
    value = "\n".join([U_, B_, C_, B_]).encode()
    g = _group_of_one(value, meta={"log.file.offset": "__file_offset__"}, position=(1000, len(value) + 1))
    m.process(g)
    ev = g.to_dict()["events"]
    assert [e["contents"]["content"] for e in ev] == [U_, B_ + "\n" + C_, B_]
    o1, o2 = 1000 + len(U_) + 1, 1000 + len(U_) + 1 + len(B_) + 1 + len(C_) + 1
    # (the B+C log is emitted while the LAST line is being looked at -- it is that line's isLastLog the reference passes on
    # (:253-262), so the log's length runs to the end of the source event, exactly as in the reference)
    assert [(e["fileOffset"], e["rawSize"]) for e in ev] == [(1000, len(U_) + 1), (o1, len(value) + 1 - (o1 - 1000)),
                                                             (o2, len(value) + 1 - (o2 - 1000))]
    assert [e["contents"]["__file_offset__"] for e in ev] == ["1000", str(o1), str(o2)]
    # ProcessEvent :133-156: an event with two contents, or without the source key, passes through untouched
    g = EventGroup({"events": [{"contents": {"content": B_, "other": "x"}, "timestamp": 1, "type": 1},
                               {"contents": {"msg": B_}, "timestamp": 2, "type": 1},
                               {"contents": {"content": B_ + "\n" + C_}, "timestamp": 3, "type": 1}]})
    m.process(g)
    ev = g.to_dict()["events"]
    assert [sorted(e["contents"]) for e in ev] == [["content", "other"], ["msg"], ["content"]]
    assert ev[2]["contents"]["content"] == B_ + "\n" + C_ and ev[2]["timestamp"] == 3


@pytest.mark.gpu
def test_merge_processor_regex_mode_on_the_reference_cases(golden_dir):
    """lc_merge_multiline_process_group, MergeType regex = ProcessorMergeMultilineLogNative::MergeLogsByRegex: the `// case:` blocks
    of ProcessorMergeMultilineLogNativeUnittest.cpp.  Input: one event per line, lying back to back in the group's buffer (what
    ProcessorSplitLogStringNative leaves); merged values are extended in place, a line feed written back between the lines."""
    import numpy as np
    from loongcollector_amd.multiline import MergeMultiline
    from loongcollector_amd.processor import EventGroup
    with open(os.path.join(golden_dir, "multiline_merge_vectors.json"), encoding="utf-8") as f:
        mv = json.load(f)
    assert len(mv["cases"]) >= 40
    merged_total = 0
    for c in mv["cases"]:
        lines = [mv["lines"][t].encode("utf-8") for t in c["in"]]
        length = np.array([len(s) for s in lines], dtype=np.uint32)
        off = np.zeros(len(lines), dtype=np.uint32)
        off[1:] = np.cumsum(length[:-1] + 1)
        data = np.frombuffer(b"\n".join(lines) + b"\n", dtype=np.uint8)
        g = EventGroup.from_lines(data, off, length)
        p = MergeMultiline(MergeType="regex", **c["config"])
        p.process(g)
        got = [dict(ev)["content"] for ev in g.contents()]
        assert got == ["\n".join(mv["lines"][t] for t in ev) for ev in c["out"]], c["cite"]
        merged, unmatched = p.counters()
        assert merged + unmatched == len(lines), c["cite"]
        merged_total += merged
    assert merged_total > 40


def _merge_pattern_cases(golden_dir):
    with open(os.path.join(golden_dir, "multiline_merge_pattern_vectors.json"), encoding="utf-8") as f:
        mv = json.load(f)
    assert len(mv["cases"]) >= 400
    return mv


@pytest.mark.gpu
def test_merge_processor_reads_the_patterns_as_the_reference_s_merge_processor_does(golden_dir):
    """The merge processor matches with MultilineOptions' OWN regexes (Get*PatternReg(), ProcessorMergeMultilineLogNative.cpp:219-224,
    244-262): what ParseRegex (MultilineOptions.cpp:250-266) leaves after stripping a trailing '$' and ".*"s, and without ContinuePattern
    when all three are given (:185-200) -- the splitter compiles the strings as written.  tests/golden/multiline_merge_pattern_vectors.json
    is the OUTPUT of the reference's merge processor (compiled from source in the build container) on 408 line groups under 17 configs;
    the product's, on the device: the same events, contents and counters.  (tests/test_multiline_host_double.py runs the product's host
    code beside the reference's processor itself.)"""
    from loongcollector_amd.multiline import MergeMultiline
    mv = _merge_pattern_cases(golden_dir)
    procs = {}
    for c in mv["cases"]:
        key = json.dumps(c["config"], sort_keys=True)
        if key not in procs:
            procs[key] = MergeMultiline(MergeType="regex", **c["config"])
        p = procs[key]
        before = p.counters()
        g, n = _events_of_lines("\n".join(mv["lines"][t] for t in c["in"]).encode("utf-8"))
        assert n == len(c["in"])
        p.process(g)
        events = g.to_dict().get("events", []) if g.to_json() != "null" else []
        # (the vectors also name the event kept by its timestamp; lc_group_from_lines stamps every event alike)
        assert [e["contents"]["content"] for e in events] == [content for _, content in c["out"]], (c["config"], c["in"])
        now = p.counters()
        assert [now[0] - before[0], now[1] - before[1]] == c["counters"], (c["config"], c["in"])


@pytest.mark.gpu
def test_merge_processor_flag_mode():
    """MergeLogsByFlag :113-159: events that carry the "P" content are partial logs (container runtimes split long lines); runs of
    them are joined WITHOUT line feeds with the first event that has no flag; only groups with the HAS_PART_LOG metadata are touched."""
    from loongcollector_amd.multiline import MergeMultiline
    from loongcollector_amd.processor import EventGroup

    import merge_fixtures as mf

    def group(with_meta):
        return EventGroup(mf.flag_group(with_meta))

    p = MergeMultiline(MergeType="flag")
    g = group(False)
    p.process(g)
    assert len(g) == 6                                             # no HAS_PART_LOG metadata: nothing happens
    # the fixture copies every value separately, so the in-place merge is checked through the lengths and the events kept
    g = group(True)
    p.process(g)
    out = g.to_dict()
    assert "metadata" not in out or "has.part.log" not in out.get("metadata", {})
    ev = out["events"]
    # (the same group goes through the reference's own merge processor in tests/test_reference_neighbours.py)
    assert [e["timestamp"] for e in ev] == mf.FLAG_TIMESTAMPS      # first event of every merged run survives
    assert all("P" not in e["contents"] for e in ev[:2])
    assert [len(e["contents"]["content"]) for e in ev] == [len(c) for c in mf.FLAG_CONTENTS]
    assert p.counters() == mf.FLAG_COUNTERS
    with pytest.raises(MultilineInitError):
        MergeMultiline(MergeType="nope")
    with pytest.raises(MultilineInitError):
        MergeMultiline()


# ---------------------------------------------------------------------------------------------- the record scan (CPU model of the kernel)
from loongcollector_amd import multiline as ML  # noqa: E402


def _flags_and_table(o, val):
    """what the device computes before the scan: the split kernels' line table and the three answers per line (here: the oracle's
    regexes, line by line)"""
    lines = val.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()          # a trailing line feed does not open an empty last line (GetNextLine :382-392)
    off, at = [], 0
    for ln in lines:
        off.append(at)
        at += len(ln) + 1
    off.append(at)           # len[i] = off[i+1] - off[i] - 1, also for an unterminated last line
    hit = lambda rx, ln: 1 if (rx is not None and rx.prefixmatch(ln) is not None) else 0
    flags = [hit(o.start, ln) | hit(o.cont, ln) << 1 | hit(o.end, ln) << 2 | (8 if not ln else 0) for ln in lines]
    mode = (ML.ML_HAS_START if o.start else 0) | (ML.ML_HAS_CONT if o.cont else 0) | (ML.ML_HAS_END if o.end else 0) | \
        (ML.ML_DISCARD if o.discard else 0) | ML.ML_FLUSH
    return mode, flags, off


def _scan_equals_the_walk(config, val):
    o = MultilineOracle(**config)
    mode, flags, off = _flags_and_table(o, val)
    recs, counts = ML.bounds_model(mode, flags, off, len(val))
    exp_recs, exp_counters = o.split(val)
    assert [(b, l, f & 1) for b, l, f in recs] == exp_recs, (config, val)
    assert tuple(counts[:3]) == exp_counters, (config, val)
    # the same scan with item records: the same ranges, named by item
    if b"\n\n" in b"\n" + val + b"\n":
        return recs          # (an empty line at the end of an unmatched run is a line-splitter rule, not one of the event walk)
    items, _ = ML.bounds_model(mode, flags)
    assert [(off[b], exp_recs[k][1] if (f & ML.ML_LAST and f & 1 and k == len(items) - 1) else off[b + c] - 1 - off[b], f & 1)
            for k, (b, c, f) in enumerate(items)] == exp_recs
    return recs


def test_scan_model_reproduces_the_reference_unit_test_cases(vectors):
    for c in vectors["cases"]:
        _scan_equals_the_walk(c["config"], _value(vectors, c["in"]))


SCAN_CONFIGS = [
    {"StartPattern": r"\d{4}-\d{2}-\d{2} .*"},
    {"StartPattern": r"\[\w+\]", "ContinuePattern": r"\s+at\s.*", "UnmatchedContentTreatment": "discard"},
    {"StartPattern": r"\[\w+\]", "ContinuePattern": r"\s+at\s.*"},
    {"StartPattern": "BEGIN", "EndPattern": r"END\d*$"},
    {"StartPattern": "BEGIN", "EndPattern": r"END\d*$", "UnmatchedContentTreatment": "discard"},
    {"ContinuePattern": r"\s+.*", "EndPattern": r"\}"},
    {"ContinuePattern": r"\s+.*", "EndPattern": r"\}", "UnmatchedContentTreatment": "discard"},
    {"EndPattern": ";$", "UnmatchedContentTreatment": "discard"},
    {"EndPattern": ";$"},
    {"StartPattern": r"\[\w+\].*", "ContinuePattern": r"\s+at\s.*", "EndPattern": r"\}$"},
    {"StartPattern": ".*"},
]
POOL = [b"2024-01-04 boom", b"  at com.example.A.b(A.java:1)", b"[ERROR] x", b"BEGIN tx", b"END7", b"END", b"END7x", b"}x", b"}", b"{",
        b"stmt;", b"noise", b"", b"\tcontinued", b"2024-13-99 not checked"]


@pytest.mark.parametrize("config", SCAN_CONFIGS)
def test_scan_model_on_random_buffers_of_many_slices(config):
    """buffers long enough for many slices (16 items each at least), with runs that cross slice borders: long continuations, long
    unmatched runs handed to HandleUnmatchLogs in one call, logs that never end (the flush)"""
    rng = random.Random(1234)
    for trial in range(120):
        n = rng.choice([0, 1, 2, 15, 16, 17, 31, 33, 100, 700, 5000]) if trial < 60 else rng.randint(0, 400)
        bias = rng.choice([None, 1, 11, 13, 12])           # some buffers are dominated by one kind of line
        val = b"\n".join(POOL[bias] if bias is not None and rng.random() < 0.9 else rng.choice(POOL) for _ in range(n))
        if rng.random() < 0.3:
            val += b"\n"
        _scan_equals_the_walk(config, val)


def test_scan_model_empty_lines_at_the_end_of_unmatched_ranges():
    """HandleUnmatchLogs' loop (:350) stops before an empty LAST line of the range it is handed; the flush hands over the rest of
    the value, where an empty last line still has its line feed behind it"""
    for val in (b"BEGIN\n\n", b"BEGIN\nx\n\n", b"BEGIN\n", b"a\n\n", b"\n", b"\n\n", b"a\n\nb", b" a\n\nEND\n\n\n"):
        for cfg in ({"StartPattern": "BEGIN", "EndPattern": "END"}, {"EndPattern": "END"}, {"StartPattern": "BEGIN"},
                    {"ContinuePattern": r"\s+.*", "EndPattern": "END"}, {"StartPattern": "BEGIN", "ContinuePattern": r"\s+.*"}):
            _scan_equals_the_walk(cfg, val)


def test_scan_model_marks_last_and_runs():
    # continue + end: a log that fails its end pattern goes to HandleUnmatchLogs as ONE run; the flush marks LAST
    o = {"ContinuePattern": r"\s+.*", "EndPattern": r"\}"}
    val = b" a\n b\nnoise\n c\n d"
    recs = _scan_equals_the_walk(o, val)
    assert [(f >> 1) & 1 for _, _, f in recs] == [0, 1, 1, 0, 1]           # [" a", " b", "noise"] one run, [" c", " d"] the flush
    assert [bool(f & ML.ML_LAST) for _, _, f in recs] == [False, False, False, True, True]
    # start only: the record emitted while the LAST line is processed carries isLastLog (:174, :327-329), and so does the flush
    recs = _scan_equals_the_walk({"StartPattern": "S"}, b"S1\nx\nS2")
    assert [bool(f & ML.ML_LAST) for _, _, f in recs] == [True, True]
    recs = _scan_equals_the_walk({"StartPattern": "S"}, b"S1\nx\nS2\n")
    assert [bool(f & ML.ML_LAST) for _, _, f in recs] == [False, True]


# ---------------------------------------------------------------------------------------------- the device scan at size
@pytest.mark.gpu
@pytest.mark.parametrize("config", SCAN_CONFIGS)
def test_device_trip_on_buffers_of_many_slices(config):
    """the whole trip (upload, split kernels, status launches, flags, scan) on buffers of up to 5000 lines: many slices, runs that
    cross slice borders, the flush; records, LAST / RUN bits and counters equal the oracle's walk and the scan's host model"""
    rng = random.Random(77)
    o = MultilineOracle(**config)
    m = Multiline(**config)
    for trial in range(30):
        n = rng.choice([1, 16, 17, 33, 100, 700, 2500, 5000])
        bias = rng.choice([None, 1, 11, 13, 12])
        val = b"\n".join(POOL[bias] if bias is not None and rng.random() < 0.9 else rng.choice(POOL) for _ in range(n))
        if rng.random() < 0.3:
            val += b"\n"
        assert m.split(val) == o.split(val), (config, n, bias)
        mode, flags, off = _flags_and_table(o, val)
        model, _ = ML.bounds_model(mode, flags, off, len(val))
        assert m.split_raw(val) == model, (config, n, bias)


@pytest.mark.gpu
def test_device_trip_on_a_stack_trace_corpus():
    from loongcollector_amd import corpus
    for head in (0, 7):
        val = corpus.multiline_buffer(512 << 10, unmatched_head=head)
        cfg = {"StartPattern": corpus.MULTILINE_START}
        recs, counters = Multiline(**cfg).split(val)
        assert (recs, counters) == MultilineOracle(**cfg).split(val)
        assert counters[1] == head and counters[2] > 500
        assert all(recs[k + 1][0] == recs[k][0] + recs[k][1] + 1 for k in range(len(recs) - 1))  # the records tile the buffer
        assert recs[0][0] == 0 and recs[-1][0] + recs[-1][1] == len(val)                          # (the flush keeps the last line feed)


@pytest.mark.gpu
def test_merge_processor_counts_empty_events_like_the_reference():
    """HandleUnmatchLogs of the merge processor counts and moves EVENTS [begin, cur] (:360-392): events without any content, which the
    walk skips (:190), are inside such a range when a log fails its end pattern, and behind the last item at the flush"""
    from loongcollector_amd.multiline import MergeMultiline
    from loongcollector_amd.processor import EventGroup

    import merge_fixtures as mf
    # (the same groups go through the reference's own merge processor in tests/test_reference_neighbours.py)
    for events, config, timestamps, counters in mf.EMPTY_EVENT_CASES:
        g = EventGroup(mf.empty_event_group(events))
        p = MergeMultiline(MergeType="regex", **config)
        p.process(g)
        left = [] if g.to_json() == "null" else [e.get("timestamp") for e in (g.to_dict().get("events") or [])]
        assert left == timestamps, (events, config)
        assert p.counters() == counters, (events, config)


def _events_of_lines(val):
    """one event per line of `val`, lying back to back in the group's buffer (what ProcessorSplitLogStringNative leaves)"""
    import numpy as np
    from loongcollector_amd.processor import EventGroup
    lines = val.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    length = np.array([len(s) for s in lines], dtype=np.uint32)
    off = np.zeros(len(lines), dtype=np.uint32)
    off[1:] = np.cumsum(length[:-1].astype(np.uint64) + 1).astype(np.uint32)
    data = np.frombuffer(b"\n".join(lines) + b"\n", dtype=np.uint8)
    return EventGroup.from_lines(data, off, length), len(lines)


@pytest.mark.gpu
@pytest.mark.parametrize("treatment", ["single_line", "discard"])
def test_merge_processor_on_thousands_of_adjacent_events(treatment):
    """Thousands of events that lie back to back (split -> merge, the usual chain): the views trip copies a run of adjacent values
    WITH the byte between them, so its staging is sized from the bytes it copies, not from the sum of the lengths (round 3 sized it
    from the sum and overran into its own offset table beyond a few dozen joins).  Against the oracle's records of the same buffer.
    Reference: ProcessorMergeMultilineLogNative.cpp:226-298 (MergeLogsByRegex)."""
    from loongcollector_amd import corpus
    from loongcollector_amd.multiline import MergeMultiline
    val = corpus.multiline_buffer(384 << 10, unmatched_head=5)
    g, n = _events_of_lines(val)
    assert n > 3000
    cfg = {"StartPattern": corpus.MULTILINE_START, "UnmatchedContentTreatment": treatment}
    p = MergeMultiline(MergeType="regex", **cfg)
    p.process(g)
    got = [dict(ev)["content"] for ev in g.contents()]
    recs, counters = MultilineOracle(**cfg).split(val)
    want = [val[b:b + l].rstrip(b"\n").decode("utf-8") if k + 1 == len(recs) else val[b:b + l].decode("utf-8")
            for k, (b, l, *_rest) in enumerate(recs)]
    assert got == want
    merged, unmatched = p.counters()
    assert unmatched == 5 and merged + unmatched == n


@pytest.mark.gpu
def test_merge_processor_shared_by_threads_with_discard():
    """Runner threads share a processor instance (ProcessQueueManager.cpp:167-205).  UnmatchedContentTreatment = discard is applied by
    the host walk (it counts EVENTS); the device trip is told so by a parameter -- round 3 flipped the shared flag around the trip, and
    a second thread could read it in between."""
    import threading
    from loongcollector_amd import corpus
    from loongcollector_amd.multiline import MergeMultiline
    cfg = {"StartPattern": corpus.MULTILINE_START, "UnmatchedContentTreatment": "discard"}
    vals = [corpus.multiline_buffer(48 << 10, unmatched_head=3 + t) for t in range(4)]
    p = MergeMultiline(MergeType="regex", **cfg)
    want = []
    for v in vals:
        recs, _ = MultilineOracle(**cfg).split(v)
        want.append([v[b:b + l].rstrip(b"\n").decode() if k + 1 == len(recs) else v[b:b + l].decode() for k, (b, l, *_r) in enumerate(recs)])
    bad = []

    def run(t):
        for _ in range(25):
            g, _n = _events_of_lines(vals[t])
            p.process(g)
            got = [dict(ev)["content"] for ev in g.contents()]
            if got != want[t]:
                bad.append((t, len(got), len(want[t])))
    threads = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not bad, bad[:4]


def test_the_patterns_the_merge_processor_compiles_walk_like_the_oracle(golden_dir):
    """The forms the merge processor hands to lc_regex_compile (csrc/multiline_gpu.cpp parseRegexStripped: the string as written, or --
    when it ends in '$' -- what MultilineOptions::ParseRegex leaves of it) for every config of the committed merge vectors: the product's
    tables (tagged DFA and thread-list program, walked by tests/helpers/table_interp.py as the kernels walk them) answer the prefix question
    as the oracle does, on every line of the vectors."""
    from oracle.oracle import OracleRegex
    from tests.helpers.table_interp import NfaInterp, TdfaInterp
    mv = _merge_pattern_cases(golden_dir)
    lines = [ln.encode("utf-8") for ln in mv["lines"]]

    def compiled_form(pattern):
        t = pattern[:-1] if pattern.endswith("$") else pattern
        while t.endswith(".*"):
            t = t[:-2]
        return None if not t else (t if pattern.endswith("$") else pattern)

    seen, walked = set(), 0
    for c in mv["cases"]:
        for k in ("StartPattern", "ContinuePattern", "EndPattern"):
            form = compiled_form(c["config"].get(k, ""))
            if not form or form in seen:
                continue
            seen.add(form)
            rx = B.GpuRegex(form.encode(), syntax_flags=B.LC_SYNTAX_PREFIX)
            o = OracleRegex(form.encode())
            its = ([NfaInterp(rx)] if rx.has_nfa_program() else []) + ([TdfaInterp(rx)] if rx.info()["engine"] == B.LC_ENGINE_TDFA else [])
            assert its, form
            for ln in lines:
                want = o.prefixmatch(ln) is not None
                for it in its:
                    walked += 1
                    assert (it.fullmatch(ln) is not None) == want, (form, ln)
    assert len(seen) >= 12 and walked > 300
    assert ";" in seen and r"\}" in seen and r"\[\w+\]" in seen and "END" in seen   # (what is left of ";$", "\}$", "\[\w+\]$", "END.*$")
