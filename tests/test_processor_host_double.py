"""The HOST translation units of processor_parse_regex_gpu (csrc/processor_parse_regex_gpu.cpp + csrc/event_model.cpp: gather -> one
match call -> stitch + policy + compaction, after ProcessorParseRegexNative.cpp:108-253) on a machine without a GPU.

The processor reaches the device through five C-ABI calls only; tests/native/host_double.cpp answers those five from the CPU oracle
and is linked with the product's two source files into tests/_build/libhost_double.so -- test infrastructure, built here, never
part of loongcollector_amd/lib (which has no such path: tests/test_processor_host.py asserts the loud failure).  What the GPU suite
checks through the real library (tests/test_gpu_processor.py) is checked here for the host code alone, so that a change to the
stitch or the policy is caught where it is made:
  * every case of the reference's unit test (tests/golden/reference_unittest_vectors.json);
  * the policy matrix (every combination of the four CommonParserOptions x key collisions x event shapes) against the processor
    oracle -- contents, order and counters;
  * the paths the device decides: LC_GAVE_UP (a parse failure that is counted), LC_OVERFLOW (event untouched), a failed device call
    (group untouched, counted, loud);
  * the one-call stitch (LogEvent::AppendCapturesNoCopy) against K x SetContentNoCopy + DelContent on the same events."""
import ctypes
import itertools
import json
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import oracle as _oracle  # builds oracle/liboracle.so
from oracle.processor_oracle import LogEventModel, ProcessorOracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LC_NOMATCH, LC_MATCH, LC_OVERFLOW, LC_GAVE_UP = 0, 1, 2, 3
LC_ERR_HIP = 4


REF = "/root/reference/core"


def _build_standin(out_dir, refshape=False):
    """refshape: csrc/event_model.hpp built with LC_REFERENCE_SHAPED_EVENT_MODEL (heap std::vector contents reserved to 16, no chunk
    pool, per-key stitch) -- the shape bench.py's in_agent_reference_shape_MBps leg measures on the GPU box"""
    so = os.path.join(out_dir, "libhost_double_refshape.so" if refshape else "libhost_double.so")
    csrc = os.path.join(ROOT, "loongcollector_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "native", "host_double.cpp"), os.path.join(csrc, "processor_parse_regex_gpu.cpp"),
            os.path.join(csrc, "event_model.cpp")]
    deps = srcs + [os.path.join(csrc, h) for h in ("event_model.hpp", "processor_parse_regex_gpu.hpp", "json_min.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-o", so] +
                              (["-DLC_REFERENCE_SHAPED_EVENT_MODEL"] if refshape else []) + srcs +
                              ["-L" + os.path.join(ROOT, "oracle"), "-loracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return so


def _build_reference(out_dir):
    """the product's processor_parse_regex_gpu.cpp compiled against the REFERENCE's event-model headers and linked with
    oracle/_ref/libref_models.so = core/models/*.cpp of /root/reference compiled by oracle/ref_models/Makefile (the per-key stitch
    an agent build takes, on the real LogEvent::SetContentNoCopy / DelContent / SourceBuffer)"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_models")])
    so = os.path.join(out_dir, "libhost_double_ref.so")
    csrc = os.path.join(ROOT, "loongcollector_amd", "csrc")
    srcs = [os.path.join(ROOT, "tests", "native", "host_double.cpp"), os.path.join(ROOT, "tests", "native", "ref_group_io.cpp"),
            os.path.join(csrc, "processor_parse_regex_gpu.cpp")]
    deps = srcs + [os.path.join(csrc, h) for h in ("processor_parse_regex_gpu.hpp", "json_min.hpp")] + [
        os.path.join(ROOT, "oracle", "_ref", "libref_models.so")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        ref_lib = os.path.join(ROOT, "oracle", "_ref")
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-w", "-DLC_USE_REFERENCE_HEADERS", "-DLC_REFERENCE_MODELS_ONLY",
                               "-I", os.path.join(ROOT, "oracle", "ref_models", "stubs"), "-I", os.path.join(ROOT, "tests", "refhdr"),
                               "-I", REF, "-I", os.path.join(REF, "config"), "-I", os.path.join(ROOT, "include"), "-I", csrc, "-o", so] + srcs +
                              ["-L" + os.path.join(ROOT, "oracle"), "-loracle", "-L" + ref_lib, "-lref_models",
                               "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath," + ref_lib, "-Wl,-z,defs"])
    return so


@pytest.fixture(scope="module", params=["standin", "refshape", "reference"])
def double(request):
    """standin: the product's host code on csrc/event_model.hpp (what the standalone library ships).  reference: the same product
    source on the reference's OWN LogEvent / PipelineEventGroup / SourceBuffer, compiled from /root/reference (oracle/_ref)."""
    _oracle.build() if hasattr(_oracle, "build") else None
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    if request.param == "reference":
        if not os.path.isdir(REF):
            pytest.skip("needs the reference tree (/root/reference): its event model is compiled from there")
        so = _build_reference(out_dir)
    else:
        so = _build_standin(out_dir, refshape=request.param == "refshape")
    L = ctypes.CDLL(so)
    vp, cp = ctypes.c_void_p, ctypes.c_char_p
    L.hd_create.restype = vp
    L.hd_create.argtypes = [cp, cp, ctypes.c_size_t]
    L.hd_destroy.argtypes = [vp]
    L.hd_process_json.restype = vp
    L.hd_process_json.argtypes = [vp, cp, cp, ctypes.c_size_t]
    L.hd_free.argtypes = [vp]
    L.hd_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.hd_want_alarms.argtypes = [vp]
    L.hd_take_alarms.restype = vp
    L.hd_take_alarms.argtypes = [vp]
    L.hd_force.argtypes = [ctypes.c_int, ctypes.c_int]
    L.hd_last_sizes.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.hd_bench_stitch.restype = ctypes.c_double
    L.hd_bench_stitch.argtypes = [vp, vp, vp, vp, ctypes.c_uint32, ctypes.c_uint32, cp, ctypes.c_uint32,
                                  ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.hd_event_model_is_reference.restype = ctypes.c_int
    assert L.hd_event_model_is_reference() == {"standin": 0, "reference": 1, "refshape": 2}[request.param]
    yield L
    L.hd_force(-1, 0)


class HostProcessor:
    COUNTERS = ["discarded", "out_failed", "out_key_not_found", "out_successful", "complexity_exceeded", "undecided", "device_failed"]

    def __init__(self, L, config):
        self.L = L
        err = ctypes.create_string_buffer(512)
        self.h = L.hd_create(json.dumps(config).encode(), err, 512)
        if not self.h:
            raise ValueError(err.value.decode())

    def process(self, fixture):
        """-> per event: ordered (key, value) list, None for a non-log event"""
        err = ctypes.create_string_buffer(512)
        p = self.L.hd_process_json(self.h, json.dumps(fixture).encode(), err, 512)
        assert p, err.value
        try:
            d = json.loads(ctypes.string_at(p).decode(), object_pairs_hook=list)
        finally:
            self.L.hd_free(p)
        # in/out_size_bytes: the processor's running sums are PipelineEventGroup::DataSize() before and after (ProcessorInstance.cpp:46-63)
        sizes = (ctypes.c_uint64 * 4)()
        self.L.hd_last_sizes(self.h, sizes)
        assert (sizes[2], sizes[3]) == (sizes[0], sizes[1]), list(sizes)
        self.last_sizes = (int(sizes[0]), int(sizes[1]))
        out = []
        for ev in dict(d).get("events", []):
            ev = dict(ev)
            out.append(list(ev.get("contents", [])) if ev.get("type") == 1 else None)
        return out

    def counters(self):
        c = (ctypes.c_uint64 * 7)()
        self.L.hd_counters(self.h, c)
        return dict(zip(self.COUNTERS, [int(x) for x in c]))

    def alarms(self):
        p = self.L.hd_take_alarms(self.h)
        try:
            text = ctypes.string_at(p)
        finally:
            self.L.hd_free(p)
        # ("kind<TAB>message<NL>" per alarm; a message may hold line feeds of its own: a line that does not begin "digits<TAB>" goes on)
        import re
        out = []
        for l in text.split(b"\n")[:-1] if text.endswith(b"\n") else text.split(b"\n"):
            if re.match(rb"^\d+\t", l):
                out.append((int(l.split(b"\t", 1)[0]), l.split(b"\t", 1)[1]))
            elif out:
                out[-1] = (out[-1][0], out[-1][1] + b"\n" + l)
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hd_destroy(self.h)
            self.h = None


def test_every_reference_unit_test_case_through_the_host_code(double, golden_dir):
    with open(os.path.join(golden_dir, "reference_unittest_vectors.json")) as f:
        cases = json.load(f)["cases"]
    ran = 0
    for case in cases:
        if not case["events"]:
            continue
        p = HostProcessor(double, case["config"])
        got = p.process({"events": case["events"]})
        if "expect_contents" in case:
            assert [dict(c) for c in got] == case["expect_contents"], case["name"]
        c = p.counters()
        names = {"discarded_events_total": "discarded", "out_failed_events_total": "out_failed",
                 "out_key_not_found_events_total": "out_key_not_found", "out_successful_events_total": "out_successful"}
        for name, want in case.get("expect_counters", {}).items():
            if name in names:
                assert c[names[name]] == want, (case["name"], name, c)
        po = ProcessorOracle(case["config"])
        out = po.process_group([LogEventModel([(k, v.encode()) for k, v in sorted(e["contents"].items())]) for e in case["events"]])
        assert got == [[(k, v.decode()) for k, v in ev.live()] for ev in out], case["name"]
        ran += 1
    assert ran >= 7


def _matrix_group(rng, lines):
    events, models = [], []
    for _ in range(16):
        kind = rng.integers(0, 10)
        if kind == 0:
            events.append({"content": "raw", "timestamp": 1, "type": 4})
            models.append(None)
            continue
        contents = []
        if rng.integers(0, 2):
            contents.append(["__file_offset__", "123"])
        if kind != 1:
            contents.append(["content", lines[int(rng.integers(0, len(lines)))]])
        if rng.integers(0, 3) == 0:
            contents.append(["key2", "preexisting"])
        if rng.integers(0, 8) == 0:
            contents.append(["_time_", "t"])
            contents.append(["_source_", "stdout"])
        events.append({"contents": contents, "timestamp": 1, "type": 1})
        models.append(LogEventModel([(k, v.encode()) for k, v in contents]))
    return events, models


def test_policy_matrix_against_the_oracle_through_the_host_code(double):
    rng = np.random.default_rng(23)
    lines = ["v1\tv2", "value3\tvalue4 tail", "nomatch", "", "a\tb\nc", "x\t", "\ty", "k\tv"]
    combos = 0
    for keep_fail, keep_ok, coping, renamed, keys in itertools.product(
            [False, True], [False, True], [False, True], ["", "rawLog", "content", "key2"],
            [["key1", "key2"], ["content", "key2"], ["key1", "key2", "key3"], ["rawLog", "x"], ["key1", "key1"]]):
        cfg = {"SourceKey": "content", "Regex": r"(\w+)\t(\w*).*", "Keys": keys, "KeepingSourceWhenParseFail": keep_fail,
               "KeepingSourceWhenParseSucceed": keep_ok, "CopingRawLog": coping, "RenamedSourceKey": renamed}
        events, models = _matrix_group(rng, lines)
        p = HostProcessor(double, cfg)
        got = p.process({"events": events, "metadata": {"log.file.offset": "__file_offset__"}})
        po = ProcessorOracle(cfg)
        out = po.process_group(models, file_offset_key="__file_offset__")
        assert got == [None if ev is None else [(k, v.decode()) for k, v in ev.live()] for ev in out], cfg
        c = p.counters()
        assert (c["discarded"], c["out_failed"], c["out_key_not_found"], c["out_successful"]) == (
            po.counters["discarded"], po.counters["out_failed"], po.counters["out_key_not_found"], po.counters["out_successful"]), cfg
        combos += 1
    assert combos == 2 * 2 * 2 * 4 * 5


def test_unmatched_optional_group_is_an_empty_value_at_the_end_of_the_line(double):
    p = HostProcessor(double, {"SourceKey": "content", "Regex": r"(\d+)(?: (\w+))?", "Keys": ["num", "word"]})
    got = p.process({"events": [{"contents": {"content": "12"}, "timestamp": 1, "type": 1},
                               {"contents": {"content": "12 ab"}, "timestamp": 1, "type": 1}]})
    assert got == [[("num", "12"), ("word", "")], [("num", "12"), ("word", "ab")]]


def test_alarm_texts_and_order(double):
    lines = ["value1\tvalue2", "nomatch", "a\tb tail", "", "x"]
    events = [{"timestamp": 1, "type": 1, "contents": {"content": l}} for l in lines]
    p = HostProcessor(double, {"SourceKey": "content", "Regex": r"(\w+)\t(\w+).*", "Keys": ["key1", "key2"], "KeepingSourceWhenParseFail": True})
    double.hd_want_alarms(p.h)
    p.process({"events": events})
    assert p.alarms() == [(0, b"errorlog:" + l.encode()) for l in lines if "\t" not in l]
    assert p.counters()["out_failed"] == 3
    p3 = HostProcessor(double, {"SourceKey": "content", "Regex": r"(\w+)\t(\w+).*", "Keys": ["key1", "key2", "key3"]})
    double.hd_want_alarms(p3.h)
    p3.process({"events": [events[0], events[2]]})
    assert p3.alarms() == [(2, b"parse key count not match3errorlog:" + l.encode()) for l in (lines[0], lines[2])]
    assert p3.counters()["out_failed"] == 0


def test_what_the_device_can_answer_besides_match_and_no_match(double):
    cfg = {"SourceKey": "content", "Regex": r"(\w+) (\w+)", "Keys": ["a", "b"], "KeepingSourceWhenParseFail": True}
    events = [{"timestamp": 1, "type": 1, "contents": {"content": "x y"}}, {"timestamp": 1, "type": 1, "contents": {"content": "p q"}}]
    try:
        # the matcher gave up on the line: boost's complexity exception -> a parse failure (StringTools.cpp:200-205), counted twice over
        double.hd_force(LC_GAVE_UP, 0)
        p = HostProcessor(double, cfg)
        double.hd_want_alarms(p.h)
        assert p.process({"events": events}) == [[("content", "x y")], [("content", "p q")]]
        c = p.counters()
        assert (c["out_failed"], c["complexity_exceeded"], c["out_successful"]) == (2, 2, 2)
        assert [k for k, _ in p.alarms()] == [1, 1]
        # undecided (only with the decide pass switched off): the event goes on untouched under its own counter
        double.hd_force(LC_OVERFLOW, 0)
        p = HostProcessor(double, cfg)
        assert p.process({"events": events}) == [[("content", "x y")], [("content", "p q")]]
        c = p.counters()
        assert (c["undecided"], c["out_failed"], c["out_successful"], c["discarded"]) == (2, 0, 0, 0)
        # the device call failed: nothing is lost, nothing is parsed, the events are counted -- and the failure is reported through the
        # alarm sink (kind 3), not as a line on stderr per group
        double.hd_force(-1, LC_ERR_HIP)
        p = HostProcessor(double, dict(cfg, KeepingSourceWhenParseFail=False))
        double.hd_want_alarms(p.h)
        assert p.process({"events": events}) == [[("content", "x y")], [("content", "p q")]]
        assert p.alarms() == [(3, b"GPU match failed (rc=4: forced failure of the test double); 2 events left unparsed")]
        c = p.counters()
        assert (c["device_failed"], c["out_failed"], c["out_successful"], c["discarded"]) == (2, 0, 0, 0)
    finally:
        double.hd_force(-1, 0)


def test_one_call_stitch_equals_the_per_key_form_on_a_large_group(double):
    """3000 Apache lines, 2 % of them poisoned: events that hold only the source take the one-call stitch
    (LogEvent::AppendCapturesNoCopy with the source dropped in the same call); the same lines in events that carry a second content
    take K x SetContentNoCopy + DelContent.  Both must leave what the oracle's captures say, in Keys order."""
    from loongcollector_amd import corpus
    from oracle.oracle import OracleRegex
    n = 3000
    data, off, length = corpus.apache_batch(n, "A", poison_every=50)
    raw = data.tobytes()
    lines = [raw[off[i]:off[i] + length[i]].decode("latin-1") for i in range(n)]
    caps, status = OracleRegex(corpus.REGEX_A).fullmatch_batch(data, off[:-1], length)
    for extra, keep_ok in itertools.product([False, True], [False, True]):
        p = HostProcessor(double, {"SourceKey": "content", "Regex": corpus.REGEX_A, "Keys": corpus.KEYS_A, "KeepingSourceWhenParseFail": True,
                                   "KeepingSourceWhenParseSucceed": keep_ok, "RenamedSourceKey": "__raw__"})
        events = [{"contents": ([["tag", "t"]] if extra else []) + [["content", s]], "timestamp": 1, "type": 1} for s in lines]
        got = p.process({"events": events})
        head = [("tag", "t")] if extra else []
        for i in range(n):
            if status[i]:
                want = head + [(k, lines[i][caps[i][2 * j]:caps[i][2 * j + 1]]) for j, k in enumerate(corpus.KEYS_A)]
                if keep_ok:
                    want.append(("__raw__", lines[i]))
            else:
                want = head + [("__raw__", lines[i])]
            assert got[i] == want, (extra, keep_ok, i)
        c = p.counters()
        assert c["out_failed"] == n // 50 and c["out_successful"] == n


def test_the_stitch_bench_entry_runs(double):
    """hd_bench_stitch is what DESIGN.md's host-share figures come from (gather + stitch + policy per 1000-event group, the match call
    answered from a table computed beforehand); here only that it runs and parses every event."""
    from loongcollector_amd import corpus
    data, off, length = corpus.apache_batch(1000, "A")
    data = np.ascontiguousarray(data)
    off = np.ascontiguousarray(off, dtype=np.uint32)
    length = np.ascontiguousarray(length, dtype=np.uint32)
    p = HostProcessor(double, {"SourceKey": "content", "Regex": corpus.REGEX_A, "Keys": corpus.KEYS_A})
    build, size = ctypes.c_double(), ctypes.c_double()
    us = double.hd_bench_stitch(p.h, data.ctypes.data, off.ctypes.data, length.ctypes.data, 1000, 4, b"content", 2, ctypes.byref(build),
                                ctypes.byref(size))
    assert us > 0 and build.value > 0
    assert p.counters()["out_successful"] == 2 * 4 * 1000


def test_arena_chunks_go_round_through_the_pool(double):
    """csrc/event_model.hpp ArenaChunkPool: a dead group's full-size arena chunks serve the next group's stitch (no first-touch page
    faults in an agent's steady state); smaller chunks and big blocks are not pooled; the pool is bounded."""
    if double.hd_event_model_is_reference():
        pytest.skip("the chunk pool belongs to the stand-in event model")
    double.hd_arena_pool_check.restype = ctypes.c_int
    assert double.hd_arena_pool_check() == 0


def test_contents_container_in_the_arena_and_on_the_heap(double):
    """csrc/event_model.hpp ArenaVector behind LogEvent: overwrite, tombstone, growth and the one-call stitch keep contents, order
    and size accounting -- for an event of a group (arena) and for one made outside any group (heap)."""
    if double.hd_event_model_is_reference():
        pytest.skip("ArenaVector belongs to the stand-in event model")
    double.hd_contents_container_check.restype = ctypes.c_int
    assert double.hd_contents_container_check() == 0


def test_source_key_and_erase_policy_against_the_reference_s_own_common_parser_options():
    """Round 5 (late): core/plugin/processor/CommonParserOptions.cpp of the reference -- ShouldAddSourceContent,
    ShouldAddLegacyUnmatchedRawLog, ShouldEraseEvent (:91-117) -- is compiled from where it lies into oracle/_ref/libref_models.so (the
    two container keys' values read out of the reference's ProcessorParseContainerLogNative.cpp by the Makefile) and compared with the
    product's restatement on the reference's own LogEvent: 16 option/outcome sets x with and without the file-offset metadata x ten
    event shapes (empty, only the offset key, the container pair in both orders, one / two other keys, ...) x the three functions."""
    if not os.path.isdir(REF):
        pytest.skip("needs the reference tree (/root/reference)")
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    L = ctypes.CDLL(_build_reference(out_dir))
    L.hd_policy_matrix_vs_reference.restype = ctypes.c_int
    L.hd_policy_matrix_vs_reference.argtypes = [ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_size_t]
    bad = ctypes.c_int(-1)
    first = ctypes.create_string_buffer(256)
    cases = L.hd_policy_matrix_vs_reference(ctypes.byref(bad), first, 256)
    assert cases == 16 * 2 * 10 * 3 + 1
    assert bad.value == 0, first.value.decode()


# ---------------------------------------------------------------------------------------------------------------------------------
# The REFERENCE's own processor_parse_regex_native, compiled from /root/reference (oracle/ref_processor: ProcessorParseRegexNative.cpp,
# CommonParserOptions.cpp, ParamExtractor.cpp, Processor.cpp against the reference's real headers; boost::regex answered by the
# oracle's matcher, the agent around the plugin by shims) -- and the product's host code beside it, group by group.
class RefProcessor:
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_models")])
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "ref_processor")])
            L = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_processor.so"))
            vp, cp = ctypes.c_void_p, ctypes.c_char_p
            L.refp_create.restype = vp
            L.refp_create.argtypes = [cp, cp, ctypes.c_size_t]
            L.refp_destroy.argtypes = [vp]
            L.refp_process_json.restype = vp
            L.refp_process_json.argtypes = [vp, cp, cp, ctypes.c_size_t]
            L.refp_counters.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
            L.refp_take_alarms.restype = vp
            L.refp_free.argtypes = [vp]
            cls._lib = L
        return cls._lib

    def __init__(self, config):
        self.L = self.lib()
        err = ctypes.create_string_buffer(512)
        self.h = self.L.refp_create(json.dumps(config).encode(), err, 512)
        self.take_alarms()          # (a failed Init reports through the alarm manager: not part of a group's alarms)
        if not self.h:
            raise ValueError(err.value.decode())

    def process(self, fixture):
        err = ctypes.create_string_buffer(512)
        p = self.L.refp_process_json(self.h, json.dumps(fixture).encode(), err, 512)
        assert p, err.value
        try:
            d = json.loads(ctypes.string_at(p).decode(), object_pairs_hook=list)
        finally:
            self.L.refp_free(p)
        return [list(dict(ev).get("contents", [])) if dict(ev).get("type") == 1 else None for ev in dict(d).get("events", [])]

    def counters(self):
        c = (ctypes.c_uint64 * 4)()
        self.L.refp_counters(self.h, c)
        return dict(zip(["discarded", "out_failed", "out_key_not_found", "out_successful"], [int(x) for x in c]))

    def take_alarms(self):
        p = self.L.refp_take_alarms()
        try:
            return [(t, m.encode("latin-1")) for t, _, m in json.loads(ctypes.string_at(p).decode("latin-1"))]
        finally:
            self.L.refp_free(p)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.refp_destroy(self.h)
            self.h = None


def _same_as_reference(double, cfg, fixture, what):
    ref, mine = RefProcessor(cfg), HostProcessor(double, cfg)
    double.hd_want_alarms(mine.h)
    want, got = ref.process(fixture), mine.process(fixture)
    assert got == want, (what, cfg)
    c = mine.counters()
    assert {k: c[k] for k in ("discarded", "out_failed", "out_key_not_found", "out_successful")} == ref.counters(), (what, cfg)
    assert [m for _, m in mine.alarms()] == [m for _, m in ref.take_alarms()], (what, cfg)


def test_the_reference_s_own_processor_beside_the_product(double, golden_dir):
    """oracle/_ref/libref_processor.so is the reference's ProcessorParseRegexNative -- Init, Process, ProcessEvent, RegexLogLineParser,
    AddLog, CommonParserOptions and the parameter readers -- compiled from its sources; only the regex match itself is answered by the
    oracle (on both sides: the product's test double asks the oracle too).  Same events, contents, order, counters and alarm texts on:
    every case of the reference's unit test; the option matrix (2 x 2 x 2 x 4 option values x 5 key lists) on random event shapes
    incl. non-log events, missing source keys, pre-existing keys, the file-offset key and the container pair; the alarm cases; and
    the configs either side must refuse."""
    if not os.path.isdir(REF):
        pytest.skip("needs the reference tree (/root/reference): its processor is compiled from there")
    with open(os.path.join(golden_dir, "reference_unittest_vectors.json")) as f:
        cases = json.load(f)["cases"]
    ran = 0
    for case in cases:
        if case["events"]:
            _same_as_reference(double, case["config"], {"events": case["events"]}, case["name"])
            ran += 1
    assert ran >= 7
    rng = np.random.default_rng(29)
    lines = ["v1\tv2", "value3\tvalue4 tail", "nomatch", "", "a\tb\nc", "x\t", "\ty", "k\tv"]
    combos = 0
    for keep_fail, keep_ok, coping, renamed, keys in itertools.product(
            [False, True], [False, True], [False, True], ["", "rawLog", "content", "key2"],
            [["key1", "key2"], ["content", "key2"], ["key1", "key2", "key3"], ["rawLog", "x"], ["key1", "key1"]]):
        cfg = {"SourceKey": "content", "Regex": r"(\w+)\t(\w*).*", "Keys": keys, "KeepingSourceWhenParseFail": keep_fail,
               "KeepingSourceWhenParseSucceed": keep_ok, "CopingRawLog": coping, "RenamedSourceKey": renamed}
        events, _ = _matrix_group(rng, lines)
        for meta in ({"log.file.offset": "__file_offset__"}, {}):
            _same_as_reference(double, cfg, {"events": events, "metadata": meta}, "matrix")
        combos += 1
    assert combos == 160
    # whole-line mode, the ["k1,k2"] legacy form of Keys, an optional group that does not take part, an empty group
    for cfg, values in (({"SourceKey": "content", "Regex": "(.*)", "Keys": ["all"]}, ["a b", "", "x\ny"]),
                        ({"SourceKey": "content", "Regex": "(.*)", "Keys": []}, ["whole"]),
                        ({"SourceKey": "content", "Regex": r"(\w+) (\w+)", "Keys": ["k1,k2"]}, ["a b", "ab"]),
                        ({"SourceKey": "content", "Regex": r"(\d+)(?: (\w+))?", "Keys": ["num", "word"]}, ["12", "12 ab", "x"]),
                        ({"SourceKey": "content", "Regex": r"(\w*)-(\w*)", "Keys": ["l", "r"], "KeepingSourceWhenParseSucceed": True}, ["-", "a-", "-b"])):
        if not cfg["Keys"]:
            with pytest.raises(ValueError):
                RefProcessor(cfg)
            with pytest.raises(ValueError):
                HostProcessor(double, cfg)
            continue
        _same_as_reference(double, cfg, {"events": [{"contents": {"content": v}, "timestamp": 1, "type": 1} for v in values]}, "shapes")
    # what Init refuses, it refuses on both sides
    for bad in ({"Regex": "a", "Keys": ["k"]}, {"SourceKey": "content", "Keys": ["k"]}, {"SourceKey": "content", "Regex": "(", "Keys": ["k"]},
                {"SourceKey": "content", "Regex": "a"}, {"SourceKey": "", "Regex": "a", "Keys": ["k"]}, {"SourceKey": 1, "Regex": "a", "Keys": ["k"]},
                {"SourceKey": "content", "Regex": "a", "Keys": "k"}, {"SourceKey": "content", "Regex": "a", "Keys": [1]}):
        with pytest.raises(ValueError):
            RefProcessor(bad)
        with pytest.raises(ValueError):
            HostProcessor(double, bad)


def test_generated_configs_beside_the_reference_s_processor(double):
    """Generated parser configs (the generator of tests/test_plugin_slot_reference.py: groups that may not take part, fewer / more Keys than
    groups, Keys colliding with the source key, the renamed source key and the raw-log keys, every option on and off) through this build
    of the product's host code -- the bulk stitch of the stand-in, the per-key stitch of the reference-shaped and reference-model builds --
    beside the reference's own processor: refused alike, the same events, counters and alarm texts.  (A 7 000-group run: no difference.)"""
    if not os.path.isdir(REF):
        pytest.skip("needs the reference tree (/root/reference)")
    rng = random.Random(4)
    regexes = [(r"(\w+)\t(\w+).*", 2), (r"(\w+) (\d{3}) (.*)", 3), (r"(\S+)", 1), ("(.*)", 1), (r"(a)|(b)", 2), (r"(\w+)\t?(\w*)", 2), (r"no groups", 0),
               (r"(?:x)(\d+)y", 1)]
    key_pool = ["k1", "k2", "k3", "content", "raw", "__raw_log__", "__raw__", "other"]
    lines = ["GET\t200 rest", "POST 404 ua", "nomatch", "", "a\tb", "x\ty z", "a", "b", "x12y", "no groups", "GET 200 curl", "ünï\tcödé"]
    groups = refused = 0
    for trial in range(200):
        rx, ngroups = rng.choice(regexes)
        nkeys = rng.choice([ngroups, ngroups, ngroups, max(0, ngroups - 1), ngroups + 1, 0])
        config = {"SourceKey": rng.choice(["content", "content", "other"]), "Regex": rx, "Keys": [rng.choice(key_pool) for _ in range(nkeys)]}
        for opt in ("KeepingSourceWhenParseFail", "KeepingSourceWhenParseSucceed", "CopingRawLog"):
            if rng.random() < 0.5:
                config[opt] = rng.random() < 0.7
        if rng.random() < 0.4:
            config["RenamedSourceKey"] = rng.choice(key_pool + [""])
        try:
            ref = RefProcessor(config)
        except ValueError:
            ref = None
        try:
            mine = HostProcessor(double, config)
        except ValueError:
            mine = None
        assert (mine is None) == (ref is None), config
        if ref is None:
            refused += 1
            continue
        double.hd_want_alarms(mine.h)
        for _ in range(3):
            events = []
            for j in range(rng.randint(0, 10)):
                kind = rng.random()
                contents = [[config["SourceKey"], rng.choice(lines)]]
                if kind > 0.8:
                    contents.append([rng.choice(key_pool), "pre-existing"])
                if kind > 0.95:
                    contents = [["elsewhere", "v"]]
                events.append({"contents": contents, "timestamp": 1 + j, "type": 1})
            g = {"events": events}
            if rng.random() < 0.3:
                g["metadata"] = {"log.file.offset": "__file_offset__"}
            want, got = ref.process(g), mine.process(g)
            assert got == want, (config, g)
            assert [m for _, m in mine.alarms()] == [m for _, m in ref.take_alarms()], (config, g)
            groups += 1
        c = mine.counters()
        assert {k: c[k] for k in ("discarded", "out_failed", "out_key_not_found", "out_successful")} == ref.counters(), config
    assert groups > 400 and refused > 30
