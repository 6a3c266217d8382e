"""Go plugin processor_regex: the oracle against the reference's own test vectors and the
Init errors on the CPU; the device path against both with -m gpu.  Reference: plugins/processor/regex/regex.go, regex_test.go."""
import json
import os
import random

import pytest

from loongcollector_amd import binding as B
from loongcollector_amd.grok import GoRegex, GrokInitError
from oracle.go_regex_oracle import GoRegexOracle


@pytest.fixture(scope="module")
def vectors(golden_dir):
    with open(os.path.join(golden_dir, "go_regex_vectors.json"), encoding="utf-8") as f:
        return json.load(f)


def test_oracle_reproduces_the_reference_test_vectors(vectors):
    for c in vectors["cases"]:
        o = GoRegexOracle(**c["config"])
        got = o.process_log([(k, v.encode("utf-8")) for k, v in c["in"]])
        assert [[k, v.decode("utf-8")] for k, v in got] == c["out"], c["cite"]


def test_init_errors(vectors):
    for c in vectors["init_fail"]:
        with pytest.raises(GrokInitError):
            GoRegex(**c["config"])
        with pytest.raises(ValueError):
            GoRegexOracle(**c["config"])
    GoRegex(Regex="(a)", Keys=["k"])   # compiles without a device


def test_no_cpu_path():
    if B.load().lc_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(B.GpuUnavailableError):
        GoRegex(Regex="(a)", Keys=["k"]).process_logs([[("content", "a")]])


@pytest.mark.gpu
def test_reference_vectors_on_the_device(vectors):
    for c in vectors["cases"]:
        got = GoRegex(**c["config"]).process_logs([[tuple(kv) for kv in c["in"]]])
        assert [list(kv) for kv in got[0]] == c["out"], c["cite"]


@pytest.mark.gpu
@pytest.mark.parametrize("config", [
    {"Regex": r"(\w+)=(\d*)(?: (\w+))?", "Keys": ["k", "v", "next"]},
    {"Regex": r"^(\S+) (\S+)$", "Keys": ["a", "b"], "FullMatch": True, "KeepSourceIfParseError": False},
    {"Regex": r"\[(.*?)\] (.*)", "Keys": ["tag", "rest"], "SourceKey": "msg", "KeepSource": True},
])
def test_random_logs_against_the_oracle(config):
    rng = random.Random(23)
    words = ["a=1", "b=", "[x] tail\nmore", "[] y", "plain", "k=22 next", "two words", "", "x=9 y=8"]
    logs = []
    for _ in range(1500):
        log = [(rng.choice(["content", "msg", "other"]), rng.choice(words)) for _ in range(rng.randint(0, 3))]
        logs.append(log)
    o = GoRegexOracle(**config)
    want = [[(k, v.decode("utf-8")) for k, v in o.process_log([(k, v.encode("utf-8")) for k, v in log])] for log in logs]
    assert GoRegex(**config).process_logs(logs) == want
