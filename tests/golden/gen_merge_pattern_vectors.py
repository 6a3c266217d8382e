"""Writes tests/golden/multiline_merge_pattern_vectors.json: what the REFERENCE's own merge processor (ProcessorMergeMultilineLogNative.cpp
+ MultilineOptions.cpp, compiled from /root/reference into oracle/_ref/libref_processor.so) leaves of random line groups under configs on
which it reads the patterns differently from the splitter -- a trailing '$' stripped, ".*" alone no pattern, ContinuePattern dropped when
all three are given (MultilineOptions.cpp:170-200,250-266) -- and under a few on which it does not.  Needs the reference tree; the GPU box
has only the vectors.  Run from the repo root:  python tests/golden/gen_merge_pattern_vectors.py"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_multiline_host_double as T  # noqa: E402


def main():
    T._double()   # (binds the harness prototypes)
    rng = random.Random(5)
    pool = [ln.decode() for ln in T.POOL]
    cases = []
    for config in T.MERGE_CONFIGS:
        ref = T.RefPlugin("processor_merge_multiline_log_native", dict(config, MergeType="regex"))
        before = (0, 0)
        for trial in range(24):
            n = rng.choice([1, 2, 3, 5, 8, 13, 40])
            lines = [rng.randrange(len(pool)) for _ in range(n)]
            data = "\n".join(pool[t] for t in lines).encode()
            out = T._ref_lines(ref, data)
            now = T._ref_merge_counters(ref)
            cases.append({"config": config, "in": lines, "out": [[ts, dict(kv)["content"]] for ts, kv in out],
                          "counters": [now[0] - before[0], now[1] - before[1]]})
            before = now
    path = os.path.join(ROOT, "tests", "golden", "multiline_merge_pattern_vectors.json")
    with open(path, "w", encoding="utf-8") as f:
        json.dump({"what": "reference merge processor (regex mode) on line groups; in: indices into lines; out: [timestamp of the event "
                           "kept (1 + index of its first line), merged content]; counters: merged / unmatched events of the group",
                   "generator": "tests/golden/gen_merge_pattern_vectors.py", "lines": pool, "cases": cases}, f, indent=0)
    print(path, len(cases), "cases")


if __name__ == "__main__":
    main()
