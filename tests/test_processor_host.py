"""Host logic of processor_parse_regex_gpu that needs no device: Init parity with the reference
(ProcessorParseRegexNative.cpp:29-106), whole-line mode (:68,147-148,170-174 -- no regex engine involved),
key-not-found / non-log events (:135-143), the JSON fixture round trip, and the loud failure without a GPU."""
import json
import os

import pytest

from loongcollector_amd import binding as B
from loongcollector_amd.processor import EventGroup, Processor, ProcessorInitError
from oracle.processor_oracle import ProcessorOracle


def vectors(golden_dir):
    with open(os.path.join(golden_dir, "reference_unittest_vectors.json")) as f:
        return {c["name"]: c for c in json.load(f)["cases"]}


def test_reference_init_cases(golden_dir):
    v = vectors(golden_dir)
    assert Processor(v["OnSuccessfulInit"]["config"]).keys == ["k1", "k2"]   # legacy ["k1,k2"] form
    assert Processor(v["TestInit"]["config"]).keys == ["content"]


@pytest.mark.parametrize("cfg,msg", [
    ({"Regex": "a", "Keys": ["k"]}, "mandatory param SourceKey is missing"),
    ({"SourceKey": "c", "Keys": ["k"]}, "mandatory param Regex is missing"),
    ({"SourceKey": "c", "Regex": "(", "Keys": ["k"]}, "mandatory string param Regex is not a valid regex"),
    ({"SourceKey": "c", "Regex": "a"}, "mandatory param Keys is missing"),
    ({"SourceKey": "c", "Regex": "a", "Keys": []}, "mandatory list param Keys is empty"),
    ({"SourceKey": "", "Regex": "a", "Keys": ["k"]}, "mandatory string param SourceKey is empty"),
    ({"SourceKey": 3, "Regex": "a", "Keys": ["k"]}, "param SourceKey is not of type string"),
    # (round 6: r"(a)\1" initialises -- back-references run on the device backtracking engine, tests/test_backref.py)
    ({"SourceKey": "c", "Regex": r"(?R)a", "Keys": ["k"]}, "cannot be executed by the GPU engines"),
])
def test_init_failures_mirror_the_reference(cfg, msg):
    with pytest.raises(ProcessorInitError, match=msg):
        Processor(cfg)
    if "GPU engines" not in msg:
        with pytest.raises(ValueError):
            ProcessorOracle(cfg)


def test_wrongly_typed_optional_params_only_warn():
    p = Processor({"SourceKey": "c", "Regex": "(.*)", "Keys": ["k"], "KeepingSourceWhenParseFail": "yes"})
    assert p.keys == ["k"]


def test_whole_line_mode_reference_case_runs_without_a_device(golden_dir):
    case = vectors(golden_dir)["TestProcessWholeLine"]
    p = Processor(case["config"])
    g = EventGroup({"events": case["events"]})
    p.process(g)
    assert [dict(c) for c in g.contents()] == case["expect_contents"]
    c = p.counters()
    for name, want in case["expect_counters"].items():
        assert c[name] == want, name
    assert c["out_successful_events_total"] == 2 and c["in_size_bytes"] > 0


def test_key_not_found_and_non_log_events_are_kept_and_counted():
    p = Processor({"SourceKey": "content", "Regex": "(.*)", "Keys": ["k"]})
    g = EventGroup({"events": [{"contents": {"other": "x"}, "timestamp": 1, "type": 1},
                               {"content": "raw bytes", "timestamp": 2, "type": 4},
                               {"contents": {"content": "hello"}, "timestamp": 3, "type": 1}]})
    p.process(g)
    assert g.contents() == [[("other", "x")], None, [("k", "hello")]]
    c = p.counters()
    assert (c["out_key_not_found_events_total"], c["out_failed_events_total"], c["out_successful_events_total"]) == (1, 1, 1)
    assert c["in_events_total"] == 3 and c["out_events_total"] == 3


def test_fixture_round_trip_and_content_order():
    fx = {"events": [{"contents": [["b", "2"], ["a", "1"]], "timestamp": 7, "timestampNanosecond": 9, "type": 1}],
          "metadata": {"log.file.path_resolved": "/var/log/x.log"}}
    g = EventGroup(fx)
    assert g.contents() == [[("b", "2"), ("a", "1")]]
    d = g.to_dict()
    assert d["metadata"]["log.file.path_resolved"] == "/var/log/x.log"
    assert d["events"][0]["timestampNanosecond"] == 9
    # object form is inserted in key order, like jsoncpp's getMemberNames() in the reference fixture loader
    assert EventGroup({"events": [{"contents": {"b": "2", "a": "1"}, "timestamp": 1, "type": 1}]}).contents() == [[("a", "1"), ("b", "2")]]


def test_regex_mode_fails_loudly_without_a_device_and_leaves_the_group_untouched():
    if B.device_count() > 0:
        pytest.skip("a HIP device is present")
    p = Processor({"SourceKey": "content", "Regex": r"(\w+)\t(\w+).*", "Keys": ["k1", "k2"]})
    g = EventGroup({"events": [{"contents": {"content": "a\tb"}, "timestamp": 1, "type": 1}]})
    with pytest.raises(B.GpuUnavailableError):
        p.process(g)
    assert g.contents() == [[("content", "a\tb")]]
    assert p.counters()["in_events_total"] == 0


def test_processor_interface_symbol_matches_cprocessor_h_layout():
    import ctypes

    class Iface(ctypes.Structure):
        _fields_ = [("version", ctypes.c_int), ("name", ctypes.c_char_p), ("language", ctypes.c_char_p),
                    ("init", ctypes.c_void_p), ("finalize", ctypes.c_void_p), ("process", ctypes.c_void_p)]

    lib = B.load()
    iface = Iface.in_dll(lib, "processor_interface")
    assert iface.version == 100                       # PROCESSOR_INTERFACE_VERSION, CProcessor.h:23
    assert iface.name == b"processor_parse_regex_gpu"
    assert iface.init and iface.finalize and iface.process
