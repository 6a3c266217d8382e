"""processor_filter_regex_native on the device.  CPU part: the filter oracle against the
reference's own unit-test vectors, Init precedence and errors through the C ABI, the non-UTF-8 blanking routine.
GPU part (-m gpu): the same vectors and random event groups through lc_filter_process against the oracle.

Reference: core/plugin/processor/ProcessorFilterNative.cpp, core/unittest/processor/ProcessorFilterNativeUnittest.cpp."""
import json
import os
import random

import pytest

from loongcollector_amd import binding as B
from loongcollector_amd import processor as P
from loongcollector_amd.processor import EventGroup, Filter, ProcessorInitError
from oracle.filter_oracle import FilterOracle, none_utf8


@pytest.fixture(scope="module")
def vectors(golden_dir):
    with open(os.path.join(golden_dir, "filter_vectors.json"), encoding="utf-8") as f:
        return json.load(f)


def _group(events):
    return EventGroup({"events": [{"contents": c, "timestamp": 12345678901, "timestampNanosecond": 0, "type": 1}
                                  for c in events]})


def test_oracle_reproduces_the_reference_unit_test_vectors(vectors):
    for c in vectors["cases"]:
        o = FilterOracle(c["config"])
        got = o.process([{k: v.encode("utf-8") for k, v in e.items()} for e in c["in"]])
        assert [{k: v.decode("utf-8") for k, v in e.items()} for e in got] == c["out"], c["cite"]


def test_init_modes_and_errors_through_the_c_abi(vectors):
    for c in vectors["init_ok"]:
        assert Filter(c["config"]).mode == c["mode"], c["cite"]
        assert FilterOracle(c["config"]).mode == c["mode"]
    for c in vectors["init_fail"]:
        with pytest.raises(ProcessorInitError):
            Filter(c["config"])
        with pytest.raises(ValueError):
            FilterOracle(c["config"])
    # ConditionExp wins over FilterKey/FilterRegex, which win over Include (ProcessorFilterNative.cpp:33-143)
    both = {"ConditionExp": {"key": "a", "exp": "x", "type": "regex"}, "FilterKey": ["a"], "FilterRegex": ["("]}
    assert Filter(both).mode == "expression"
    assert Filter({"FilterKey": ["a"], "FilterRegex": ["x"], "Include": {"b": "("}}).mode == "rule"


def test_non_utf8_blanking_rules():
    """noneUtf8 :297-379: stray continuation bytes, truncated sequences, overlong 2-byte forms, > U+10FFFF, 0xF8..0xFF"""
    cases = {
        b"plain ascii": b"plain ascii",
        "héllo 你好 \U0001F600".encode("utf-8"): "héllo 你好 \U0001F600".encode("utf-8"),
        b"a\x80b": b"a b",                      # stray continuation byte
        b"a\xc3": b"a ",                        # truncated 2-byte sequence
        b"\xc0\xafx": b" \x20x".replace(b"\x20x", b" x"),   # overlong: lead blanked, then the stray continuation byte
        b"\xe4\xbd": b"  ",                     # truncated 3-byte sequence: lead blanked, continuation is then stray
        b"\xe0\x80\x80": b"   ",                # 3-byte form below U+0800
        b"\xf4\x90\x80\x80": b"    ",           # above U+10FFFF
        b"\xf8\x88\x80\x80\x80": b"     ",      # 5-byte lead
        b"\xed\xa0\x80": b"\xed\xa0\x80",       # surrogates are NOT rejected by the reference routine
    }
    for raw, want in cases.items():
        bad, fixed = none_utf8(raw)
        assert fixed == want, raw
        assert bad == (raw != want)
        assert P.none_utf8(raw) == (bad, fixed)      # the routine compiled into the library
    rng = random.Random(5)
    alphabet = [0x41, 0x20, 0x7F, 0x80, 0xBF, 0xC0, 0xC2, 0xDF, 0xE0, 0xE4, 0xED, 0xEF, 0xF0, 0xF4, 0xF5, 0xF8, 0xFF, 0xA0, 0x90]
    for _ in range(4000):
        raw = bytes(rng.choice(alphabet) for _ in range(rng.randint(0, 9)))
        assert P.none_utf8(raw) == none_utf8(raw), raw


def test_no_cpu_path():
    if B.load().lc_device_count() > 0:
        pytest.skip("a HIP device is present")
    f = Filter({"Include": {"a": "x.*"}})
    g = _group([{"a": "xy"}])
    with pytest.raises(B.GpuUnavailableError):
        f.process(g)
    assert len(g.to_dict()["events"]) == 1       # untouched
    Filter({}).process(g)                        # bypass mode has no regex leaf and needs no device
    assert len(g.to_dict()["events"]) == 1


@pytest.mark.gpu
def test_reference_vectors_on_the_device(vectors):
    for c in vectors["cases"]:
        f = Filter(c["config"])
        g = _group(c["in"])
        f.process(g)
        d = g.to_dict()
        got = [e["contents"] for e in d["events"]] if d else []
        assert got == c["out"], c["cite"]
        assert f.counters() == {"in_events_total": len(c["in"]), "out_events_total": len(c["out"])}


@pytest.mark.gpu
@pytest.mark.parametrize("config", [
    {"Include": {"status": "2\\d\\d|30[14]", "method": "GET|HEAD"}},
    {"FilterKey": ["path", "ua"], "FilterRegex": ["/api/.*", ".*(?:bot|curl).*"]},
    {"ConditionExp": {"operator": "or", "operands": [
        {"operator": "not", "operands": [{"type": "regex", "key": "status", "exp": "[23]\\d\\d"}]},
        {"operator": "and", "operands": [{"type": "regex", "key": "path", "exp": "/admin(?:/.*)?"},
                                         {"type": "regex", "key": "ip", "exp": "10\\.\\d+\\.\\d+\\.\\d+"}]}]},
     "DiscardingNonUTF8": True},
])
def test_random_event_groups_against_the_oracle(config):
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
        e = {}
        for k, vals in fields.items():
            if rng.random() < 0.85:
                e[k] = rng.choice(vals)
        events.append(e)
    o = FilterOracle(config)
    want = o.process([{k: v.encode("utf-8") for k, v in e.items()} for e in events])
    f = Filter(config)
    g = _group(events)
    f.process(g)
    d = g.to_dict()
    got = [e["contents"] for e in d["events"]] if d else []
    assert 0 < len(want) < len(events)
    assert got == [{k: v.decode("utf-8") for k, v in e.items()} for e in want]


@pytest.mark.gpu
def test_groups_of_alternating_sizes_over_more_than_257_trips():
    """The zero-copy trip's completion word must not be reachable by status bytes: long and short groups in
    turn on ONE runner thread, the long group's verdicts at [64..68) spelling the short group's trip number where bytes allow, well past
    trips 256 / 257 (the first numbers two or three status bytes can spell).  tests/test_filter_host_double.py holds the CPU twin."""
    f = Filter({"Include": {"k": "yes.*"}})
    rng = random.Random(7)
    trip = 0
    for _ in range(300):
        nxt = trip + 2
        flags = [rng.random() < 0.5 for _ in range(200)]
        for b in range(4):
            flags[64 + b] = ((nxt >> (8 * b)) & 0xFF) == 1
        for flags_now in (flags, [rng.random() < 0.5 for _ in range(64)]):
            trip += 1
            g = _group([{"k": ("yes%d" if fl else "no%d") % i} for i, fl in enumerate(flags_now)])
            f.process(g)
            d = g.to_dict()
            got = [e["contents"]["k"] for e in d["events"]] if d else []
            assert got == ["yes%d" % i for i, fl in enumerate(flags_now) if fl], trip
