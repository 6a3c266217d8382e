"""The neighbours of the parser -- line splitter, multiline splitter, regex filter -- as the REFERENCE's own
code, compiled from /root/reference into oracle/_ref/libref_processor.so (oracle/ref_processor: ProcessorSplitLogStringNative.cpp,
ProcessorSplitMultilineLogStringNative.cpp + MultilineOptions.cpp, ProcessorFilterNative.cpp against the reference's real headers;
boost::regex answered by the oracle's matcher), beside the oracles the GPU tests compare the product with (oracle/split_oracle.py,
multiline_oracle.py, filter_oracle.py): the same lines, records, surviving events and counters.  CPU only; skipped where the reference
tree is not present (the GPU box)."""
import ctypes
import json
import os
import random
import subprocess

import pytest

from oracle.filter_oracle import FilterOracle
from oracle.multiline_oracle import MultilineOracle
from oracle.split_oracle import split_lines

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/core"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (/root/reference): its processors are compiled from there")


class RefPlugin:
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            for d in ("oracle", os.path.join("oracle", "ref_models"), os.path.join("oracle", "ref_processor")):
                subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, d)])
            L = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_processor.so"))
            vp, cp = ctypes.c_void_p, ctypes.c_char_p
            L.refp_create_kind.restype = vp
            L.refp_create_kind.argtypes = [cp, cp, cp, ctypes.c_size_t]
            L.refp_destroy.argtypes = [vp]
            L.refp_process_json.restype = vp
            L.refp_process_json.argtypes = [vp, cp, cp, ctypes.c_size_t]
            L.refp_counters_json.restype = vp
            L.refp_counters_json.argtypes = [vp]
            L.refp_take_alarms.restype = vp
            L.refp_free.argtypes = [vp]
            cls._lib = L
        return cls._lib

    def __init__(self, kind, config):
        self.L = self.lib()
        err = ctypes.create_string_buffer(512)
        self.h = self.L.refp_create_kind(kind.encode(), json.dumps(config).encode(), err, 512)
        self.L.refp_free(self.L.refp_take_alarms())
        if not self.h:
            raise ValueError(err.value.decode())

    def process(self, fixture):
        """-> the group's events after Process, as the fixture writer prints them"""
        err = ctypes.create_string_buffer(512)
        p = self.L.refp_process_json(self.h, json.dumps(fixture).encode(), err, 512)
        assert p, err.value
        try:
            d = json.loads(ctypes.string_at(p).decode("utf-8"), object_pairs_hook=list)
        finally:
            self.L.refp_free(p)
        return [dict(ev) for ev in dict(d or []).get("events", [])]

    def counters(self):
        p = self.L.refp_counters_json(self.h)
        try:
            return json.loads(ctypes.string_at(p).decode())
        finally:
            self.L.refp_free(p)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.refp_destroy(self.h)
            self.h = None


def _one_event(value, **extra):
    ev = {"contents": {"content": value}, "timestamp": 12345678901, "timestampNanosecond": 0, "type": 1}
    ev.update(extra)
    return {"events": [ev]}


def test_line_splitter():
    """ProcessorSplitLogStringNative.cpp:101-174 beside oracle/split_oracle.py: lines = the SplitChar-delimited segments; an unterminated
    tail is a line, empty segments are lines, a trailing SplitChar opens no new line; an event that does not hold exactly the source key, and a
    non-log event, pass as they are."""
    rng = random.Random(7)
    pieces = ["", "a", "line two", "x" * 40, " ", "tab\tsep", "semi;colon", "\r"]
    for split_char in (10, ord(";"), 9):
        p = RefPlugin("processor_split_string_native", {"SplitChar": split_char})
        sep = chr(split_char)
        for _ in range(120):
            value = sep.join(rng.choice(pieces) for _ in range(rng.randint(0, 9)))
            if rng.random() < 0.3:
                value += sep
            if not value:
                continue
            buf = value.encode()
            want = [buf[b:b + l].decode() for b, l in split_lines(buf, split_char)]
            got = p.process(_one_event(value))
            assert [dict(ev["contents"])["content"] for ev in got] == want, (split_char, value)
    p = RefPlugin("processor_split_string_native", {})
    # a non-log event passes; so does a log event that does not hold exactly the source key (:110-127, with an alarm)
    got = p.process({"events": [{"content": "raw bytes", "timestamp": 1, "type": 4},
                                {"contents": {"content": "a\nb", "other": "kept"}, "timestamp": 5, "type": 1},
                                {"contents": {"other": "x\ny"}, "timestamp": 5, "type": 1},
                                {"contents": {"content": "a\nb"}, "timestamp": 5, "type": 1}]})
    assert [ev.get("type") for ev in got] == [4, 1, 1, 1, 1]
    assert [dict(ev["contents"]) for ev in got[1:]] == [{"content": "a\nb", "other": "kept"}, {"other": "x\ny"}, {"content": "a"}, {"content": "b"}]


def _records_text(val, recs):
    return [val[b:b + l].decode("utf-8") for b, l, _ in recs]


def test_multiline_splitter_on_the_reference_unit_test_cases_and_random_buffers(golden_dir):
    """ProcessorSplitMultilineLogStringNative.cpp:126-392 + MultilineOptions.cpp beside oracle/multiline_oracle.py: the records of the 46
    imported unit-test cases and of random buffers under eight pattern configurations, and the three counters."""
    with open(os.path.join(golden_dir, "multiline_vectors.json"), encoding="utf-8") as f:
        vectors = json.load(f)
    ran = 0
    for c in vectors["cases"]:
        val = "\n".join(vectors["lines"][t] for t in c["in"])
        if not val:
            continue
        p = RefPlugin("processor_split_multiline_log_string_native", c["config"])
        got = [dict(ev["contents"])["content"] for ev in p.process(_one_event(val))]
        recs, counters = MultilineOracle(**c["config"]).split(val.encode("utf-8"))
        assert got == _records_text(val.encode("utf-8"), recs), c["cite"]
        assert [g.split("\n") for g in got] == [[vectors["lines"][t] for t in ev] for ev in c["out"]], c["cite"]
        cnt = p.counters()
        assert cnt["unmatched_lines_total"] == counters[1] and cnt["matched_events_total"] == counters[2], (c["cite"], cnt, counters)
        ran += 1
    assert ran >= 40
    rng = random.Random(41)
    pool = ["2024-01-04 boom", "  at com.example.A.b(A.java:1)", "[ERROR] x", "BEGIN tx", "END7", "END", "END7x", "}x", "}", "{",
            "stmt;", "noise", "", "\tcontinued", "2024-13-99 not checked"]
    configs = [
        {"StartPattern": r"\d{4}-\d{2}-\d{2} .*"},
        {"StartPattern": r"\d{4}-\d{2}-\d{2} .*", "UnmatchedContentTreatment": "discard"},
        {"StartPattern": r"\[\w+\].*", "ContinuePattern": r"\s+at\s.*"},
        {"StartPattern": "BEGIN.*", "EndPattern": r"END\d*"},
        {"ContinuePattern": r"\s+at\s.*", "EndPattern": r"\}"},
        {"EndPattern": ";$", "UnmatchedContentTreatment": "discard"},
        {"StartPattern": r"\[\w+\].*", "ContinuePattern": r"\s+at\s.*", "EndPattern": r"\}$"},
        {"StartPattern": ".*"},
    ]
    for config in configs:
        o = MultilineOracle(**config)
        p = RefPlugin("processor_split_multiline_log_string_native", config)
        before = {"unmatched_lines_total": 0, "matched_events_total": 0}
        for _ in range(150):
            val = "\n".join(rng.choice(pool) for _ in range(rng.randint(1, 12)))
            if rng.random() < 0.2:
                val += "\n"
            if not val:
                continue
            got = [dict(ev["contents"])["content"] for ev in p.process(_one_event(val))]
            recs, counters = o.split(val.encode())
            assert got == _records_text(val.encode(), recs), (config, val)
            cnt = p.counters()
            assert cnt["unmatched_lines_total"] - before["unmatched_lines_total"] == counters[1], (config, val)
            assert cnt["matched_events_total"] - before["matched_events_total"] == counters[2], (config, val)
            before = cnt


def test_regex_filter_on_the_reference_unit_test_vectors_and_random_groups(golden_dir):
    """ProcessorFilterNative.cpp:30-486 beside oracle/filter_oracle.py: Init's precedence and refusals, the imported unit-test vectors,
    and 2 000 random events under Include, FilterKey / FilterRegex and a ConditionExp tree -- the surviving events, in order."""
    with open(os.path.join(golden_dir, "filter_vectors.json"), encoding="utf-8") as f:
        vectors = json.load(f)
    for c in vectors["cases"]:
        if not c["in"] or any(not e for e in c["in"]):
            continue  # (the fixture reader cannot express an event group without events / an event without contents the way the test builds them)
        p = RefPlugin("processor_filter_regex_native", c["config"])
        got = p.process({"events": [{"contents": e, "timestamp": 1, "type": 1} for e in c["in"]]})
        assert [dict(ev["contents"]) for ev in got] == c["out"], c["cite"]
    for c in vectors["init_fail"]:
        with pytest.raises(ValueError):
            RefPlugin("processor_filter_regex_native", c["config"])
        with pytest.raises(ValueError):
            FilterOracle(c["config"])
    for c in vectors["init_ok"]:
        RefPlugin("processor_filter_regex_native", c["config"])
        FilterOracle(c["config"])
    rng = random.Random(17)
    fields = {
        "status": ["200", "204", "301", "404", "500", "2000", ""],
        "method": ["GET", "HEAD", "POST", "GETX"],
        "path": ["/api/v1/x", "/admin", "/admin/users", "/index.html", "/apix", "/café"],
        "ua": ["curl/8.1", "Mozilla/5.0", "Googlebot/2.1", "bot", ""],
        "ip": ["10.0.0.1", "192.168.1.1", "10.1.2.3.4", "10.x.0.1"],
    }
    events = []
    for _ in range(2000):
        e = {k: rng.choice(vals) for k, vals in fields.items() if rng.random() < 0.85}
        if e:
            events.append(e)
    for config in ({"Include": {"status": "2\\d\\d|30[14]", "method": "GET|HEAD"}},
                   {"FilterKey": ["path", "ua"], "FilterRegex": ["/api/.*", ".*(?:bot|curl).*"]},
                   {"ConditionExp": {"operator": "or", "operands": [
                       {"operator": "not", "operands": [{"type": "regex", "key": "status", "exp": "[23]\\d\\d"}]},
                       {"operator": "and", "operands": [{"type": "regex", "key": "path", "exp": "/admin(?:/.*)?"},
                                                        {"type": "regex", "key": "ip", "exp": "10\\.\\d+\\.\\d+\\.\\d+"}]}]},
                    "DiscardingNonUTF8": True}):
        want = FilterOracle(config).process([{k: v.encode("utf-8") for k, v in e.items()} for e in events])
        p = RefPlugin("processor_filter_regex_native", config)
        got = p.process({"events": [{"contents": [[k, v] for k, v in e.items()], "timestamp": 1, "type": 1} for e in events]})
        assert 0 < len(want) < len(events)
        assert [dict(ev["contents"]) for ev in got] == [{k: v.decode("utf-8") for k, v in e.items()} for e in want], config


def test_merge_processor_reproduces_the_imported_unit_test_cases(golden_dir):
    """ProcessorMergeMultilineLogNative.cpp (MergeType regex) behind the reference's own line splitter, on the cases imported from its
    unit test into tests/golden/multiline_merge_vectors.json -- the vectors the product's merge processor is compared with on the GPU
    are what the reference's code produces."""
    L = RefPlugin.lib()
    L.refp_process_chain_json.restype = ctypes.c_void_p
    L.refp_process_chain_json.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    with open(os.path.join(golden_dir, "multiline_merge_vectors.json"), encoding="utf-8") as f:
        mv = json.load(f)
    split = RefPlugin("processor_split_string_native", {})
    ran = 0
    for c in mv["cases"]:
        if not c["in"]:
            continue
        value = "\n".join(mv["lines"][t] for t in c["in"])
        if not value:
            continue
        merge = RefPlugin("processor_merge_multiline_log_native", dict(c["config"], MergeType="regex"))
        err = ctypes.create_string_buffer(512)
        p = L.refp_process_chain_json(split.h, merge.h, json.dumps(_one_event(value)).encode(), err, 512)
        assert p, err.value
        try:
            d = json.loads(ctypes.string_at(p).decode("utf-8"), object_pairs_hook=list)
        finally:
            L.refp_free(p)
        got = [dict(dict(ev)["contents"])["content"] for ev in dict(d or []).get("events", [])]
        assert got == ["\n".join(mv["lines"][t] for t in ev) for ev in c["out"]], c["cite"]
        ran += 1
    assert ran >= 40


def test_merge_processor_on_the_groups_the_product_s_gpu_tests_use():
    """The hand-written groups of tests/merge_fixtures.py -- partial-log flags (MergeLogsByFlag :113-159), events without contents inside
    and behind unmatched ranges (HandleUnmatchLogs :360-392) -- through the reference's own merge processor: the events left, the joined
    values and the two counters the product's GPU tests (tests/test_multiline.py) expect are what the reference's code leaves."""
    import merge_fixtures as mf
    p = RefPlugin("processor_merge_multiline_log_native", {"MergeType": "flag"})
    untouched = p.process(mf.flag_group(False))                       # no HAS_PART_LOG metadata: nothing happens
    assert [e["timestamp"] for e in untouched] == [1, 2, 3, 4, 5, 6] and p.counters()["merged_events_total"] == 0
    out = p.process(mf.flag_group(True))
    assert [e["timestamp"] for e in out] == mf.FLAG_TIMESTAMPS
    # (the fixture reader copies every key and value separately, so the in-place memmove runs over dead keys of the merged-away events:
    # the value under "content" and the absence of the flag are what carries over)
    assert [dict(e["contents"])["content"] for e in out] == mf.FLAG_CONTENTS
    assert all("P" not in dict(e["contents"]) for e in out)
    cnt = p.counters()
    assert (cnt["merged_events_total"], cnt["unmatched_events_total"]) == mf.FLAG_COUNTERS
    for kind, config in (("nope", {"MergeType": "nope"}), ("none", {})):
        with pytest.raises(ValueError):
            RefPlugin("processor_merge_multiline_log_native", config)
    for events, config, timestamps, counters in mf.EMPTY_EVENT_CASES:
        p = RefPlugin("processor_merge_multiline_log_native", dict(config, MergeType="regex"))
        out = p.process(mf.empty_event_group(events))
        assert [e.get("timestamp") for e in out] == timestamps, (events, config)
        cnt = p.counters()
        assert (cnt["merged_events_total"], cnt["unmatched_events_total"]) == counters, (events, config)


def _split_then_merge(L, split, merge, value):
    err = ctypes.create_string_buffer(512)
    p = L.refp_process_chain_json(split.h, merge.h, json.dumps(_one_event(value)).encode(), err, 512)
    assert p, err.value
    try:
        d = json.loads(ctypes.string_at(p).decode("utf-8"), object_pairs_hook=list)
    finally:
        L.refp_free(p)
    return [dict(dict(ev)["contents"])["content"] for ev in dict(d or []).get("events", [])]


def test_merge_behind_the_splitter_leaves_the_multiline_oracle_s_records():
    """The product's merge processor is checked on the GPU against oracle/multiline_oracle.py's records of the same buffer
    (test_merge_processor_on_thousands_of_adjacent_events): here the REFERENCE's splitter + merge processor on random buffers under six
    pattern sets (those the two processors read alike), and on the corpus buffer of that GPU test, leave exactly those records (the last one without its final line feed) and
    count merged + unmatched = the lines."""
    from loongcollector_amd import corpus
    L = RefPlugin.lib()
    L.refp_process_chain_json.restype = ctypes.c_void_p
    L.refp_process_chain_json.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    split = RefPlugin("processor_split_string_native", {})

    def want(config, val):
        recs, counters = MultilineOracle(**config).split(val)
        texts = [val[b:b + l].decode("utf-8") for b, l, *_ in recs]
        if texts and texts[-1].endswith("\n"):
            texts[-1] = texts[-1][:-1]
        return texts, counters

    rng = random.Random(43)
    pool = ["2024-01-04 boom", "  at com.example.A.b(A.java:1)", "[ERROR] x", "BEGIN tx", "END7", "END", "END7x", "}x", "}", "{",
            "stmt;", "noise", "\tcontinued", "2024-13-99 not checked"]
    configs = [
        {"StartPattern": r"\d{4}-\d{2}-\d{2} .*"},
        {"StartPattern": r"\d{4}-\d{2}-\d{2} .*", "UnmatchedContentTreatment": "discard"},
        {"StartPattern": r"\[\w+\].*", "ContinuePattern": r"\s+at\s.*"},
        {"StartPattern": "BEGIN.*", "EndPattern": r"END\d*"},
        {"ContinuePattern": r"\s+at\s.*", "EndPattern": r"\}"},
        {"EndPattern": r"END\d", "UnmatchedContentTreatment": "discard"},
    ]
    # (configs whose patterns end in '$', consist of ".*" only, or give all three patterns are NOT here: on those the reference's merge
    # processor -- MultilineOptions' stripped regexes -- and its splitter -- the strings as written -- part ways, and the oracle follows the
    # splitter.  tests/test_multiline_host_double.py and tests/golden/multiline_merge_pattern_vectors.json hold the merge processor's reading.)
    for config in configs:
        merge = RefPlugin("processor_merge_multiline_log_native", dict(config, MergeType="regex"))
        before = merge.counters()
        for _ in range(120):
            lines = [rng.choice(pool) for _ in range(rng.randint(1, 14))]
            val = "\n".join(lines).encode()
            got = _split_then_merge(L, split, merge, val.decode())
            texts, counters = want(config, val)
            assert got == texts, (config, val)
            cnt = merge.counters()
            assert (cnt["merged_events_total"] - before["merged_events_total"]) + (cnt["unmatched_events_total"] - before["unmatched_events_total"]) == len(lines), (config, val)
            assert cnt["unmatched_events_total"] - before["unmatched_events_total"] == counters[1], (config, val)
            before = cnt
    for treatment in ("single_line", "discard"):                       # the GPU test's own buffer
        val = corpus.multiline_buffer(384 << 10, unmatched_head=5)
        config = {"StartPattern": corpus.MULTILINE_START, "UnmatchedContentTreatment": treatment}
        merge = RefPlugin("processor_merge_multiline_log_native", dict(config, MergeType="regex"))
        got = _split_then_merge(L, split, merge, val.decode("utf-8"))
        texts, counters = want(config, val)
        assert got == texts and len(got) > 100
        n = len(val.rstrip(b"\n").split(b"\n"))
        cnt = merge.counters()
        assert cnt["unmatched_events_total"] == 5 and cnt["merged_events_total"] + cnt["unmatched_events_total"] == n
