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
        Multiline(StartPattern=r"(a)\1")
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


@pytest.mark.gpu
def test_merge_processor_flag_mode():
    """MergeLogsByFlag :113-159: events that carry the "P" content are partial logs (container runtimes split long lines); runs of
    them are joined WITHOUT line feeds with the first event that has no flag; only groups with the HAS_PART_LOG metadata are touched."""
    from loongcollector_amd.multiline import MergeMultiline
    from loongcollector_amd.processor import EventGroup

    def group(with_meta):
        evs = [{"contents": [["content", "aaa"], ["P", ""]], "timestamp": 1, "type": 1},
               {"contents": [["content", "bbb"], ["P", ""]], "timestamp": 2, "type": 1},
               {"contents": [["content", "ccc"]], "timestamp": 3, "type": 1},
               {"contents": [["content", "single"]], "timestamp": 4, "type": 1},
               {"contents": [["content", "tail1"], ["P", ""]], "timestamp": 5, "type": 1},
               {"contents": [["content", "tail2"], ["P", ""]], "timestamp": 6, "type": 1}]
        g = {"events": evs}
        if with_meta:
            g["metadata"] = {"has.part.log": "P"}
        return EventGroup(g)

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
    assert [e["timestamp"] for e in ev] == [1, 4, 5]               # first event of every merged run survives
    assert all("P" not in e["contents"] for e in ev[:2])
    assert [len(e["contents"]["content"]) for e in ev] == [9, 6, 10]
    assert p.counters() == (6, 0)
    with pytest.raises(MultilineInitError):
        MergeMultiline(MergeType="nope")
    with pytest.raises(MultilineInitError):
        MergeMultiline()
