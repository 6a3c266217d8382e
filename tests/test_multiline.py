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
