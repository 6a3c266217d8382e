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
    """MultilineOptions.cpp:170-205, :250-262"""
    m = Multiline(StartPattern="Exception.*", ContinuePattern=r"\s+at\s.*", EndPattern=r"\s*\.\.\.\d+ more")
    assert m.patterns == {"start": True, "continue": False, "end": True} and m.is_multiline   # all three: continue dropped
    m = Multiline(ContinuePattern=r"\s+at\s.*")
    assert m.patterns == {"start": False, "continue": False, "end": False} and not m.is_multiline   # continue alone: ignored
    assert Multiline(StartPattern=".*").patterns["start"] is False                # nothing left after stripping '.*'
    assert Multiline(EndPattern="x$").patterns["end"] is True
    with pytest.raises(MultilineInitError):
        Multiline(StartPattern="(")
    o = MultilineOracle(StartPattern="a", ContinuePattern="b", EndPattern="c")
    assert o.cont is None and o.is_multiline


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
])
def test_random_buffers_against_the_oracle(config):
    rng = random.Random(41)
    pool = [b"2024-01-04 boom", b"  at com.example.A.b(A.java:1)", b"[ERROR] x", b"BEGIN tx", b"END7", b"END", b"}", b"{",
            b"stmt;", b"noise", b"", b"\tcontinued", b"2024-13-99 not checked"]
    o = MultilineOracle(**config)
    m = Multiline(**config)
    for _ in range(200):
        val = b"\n".join(rng.choice(pool) for _ in range(rng.randint(0, 12)))
        if rng.random() < 0.2:
            val += b"\n"
        assert m.split(val) == o.split(val), (config, val)
