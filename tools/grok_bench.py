#!/usr/bin/env python3
"""BASELINE.json configs[2] on one MI355X: processor_grok with the example_config pattern list (the Match entries of
tests/golden/grok_config3.json that the device engines can run), mixed 128..4096 B lines resident in HBM.

One "step" = one lc_grok_match_device() call over the whole batch (all patterns, all rounds).  A parity gate compares
the device result with the Grok oracle on a sample before anything is timed; the oracle's own speed on that sample is
reported beside the GPU number.  Prints one JSON line (same field names as bench.py).

    python tools/grok_bench.py --lines 65536 --steps 5 --warmup 1
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", type=int, default=1 << 16)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-sample-lines", type=int, default=2000)
    ap.add_argument("--patterns", type=int, default=0, help="use only the first N supported patterns (0 = all)")
    args = ap.parse_args()

    import torch

    from loongcollector_amd import binding
    from loongcollector_amd.grok import Grok, GrokInitError
    from loongcollector_amd.grok_corpus import grok_lines

    if not torch.cuda.is_available():
        raise SystemExit("grok_bench.py needs a HIP device: the Grok matcher has no CPU path")
    dev = torch.device("cuda", 0)
    with open(os.path.join(ROOT, "tests", "golden", "grok_config3.json"), encoding="utf-8") as f:
        cfg = json.load(f)
    supported, refused = [], []
    for m in cfg["match"]:
        try:
            Grok(Match=[m], CustomPatterns=cfg["custom_patterns"], AnchoredFirst=False)   # (a probe: no warm-up thread)
            supported.append(m)
        except GrokInitError as e:
            refused.append((m, str(e).split(": ", 1)[-1][:80]))
    if args.patterns:
        supported = supported[:args.patterns]
    g = Grok(Match=supported, CustomPatterns=cfg["custom_patterns"])
    t0 = time.perf_counter()
    g.wait_ready()   # the anchored searches are compiled behind Init; the timed steps should see the matcher at full speed
    warm_s = time.perf_counter() - t0
    engines = [g.engine(i) for i in range(g.n_match)]

    values = grok_lines(args.lines)
    n = len(values)
    length = np.array([len(v) for v in values], dtype=np.uint32)
    off = np.zeros(n, dtype=np.uint32)
    off[1:] = np.cumsum(length[:-1], dtype=np.uint64).astype(np.uint32)
    data = np.frombuffer(b"".join(values) + b"\0" * 16, dtype=np.uint8).copy()
    total_bytes = int(length.sum())
    d_data = torch.from_numpy(data).to(dev)
    d_off = torch.from_numpy(off.view(np.int32)).to(dev)
    d_len = torch.from_numpy(length.view(np.int32)).to(dev)
    row = g.row_ints
    d_pattern = torch.empty(n, dtype=torch.int32, device=dev)
    d_first = torch.empty((n, row), dtype=torch.int32, device=dev)
    d_extra = torch.empty((n + 1024, row + 2), dtype=torch.int32, device=dev)
    d_nextra = torch.zeros(1, dtype=torch.int32, device=dev)
    d_scratch = torch.empty(g.scratch_bytes(n), dtype=torch.uint8, device=dev)

    def step():
        g.match_device(d_data, d_off, d_len, n, d_pattern, d_first, d_extra, d_nextra, d_scratch)

    step()
    torch.cuda.synchronize()
    pattern = d_pattern.cpu().numpy()
    first = d_first.cpu().numpy()

    # ---- parity gate + CPU baseline on a sample (the oracle is the checker, never the measured path)
    from oracle.grok_oracle import GrokOracle
    o = GrokOracle(supported, custom_patterns=cfg["custom_patterns"])
    sample = min(args.cpu_sample_lines, n)
    t0 = time.perf_counter()
    want = [o.process_value(values[i]) for i in range(sample)]
    cpu_s = time.perf_counter() - t0
    for i, (res, fields) in enumerate(want):
        if (pattern[i] >= 0) != (res == 0):
            raise SystemExit("PARITY FAILURE: line %d pattern %d vs oracle result %d" % (i, pattern[i], res))
        if res == 0:
            cols = g.columns(int(pattern[i]))
            got = []
            seen = {}
            for c, key in enumerate(cols):   # first contributing match only (further matches are in d_extra)
                b, e = int(first[i][2 + 2 * c]), int(first[i][3 + 2 * c])
                if key is None:
                    continue
                if key not in seen:
                    seen[key] = len(got)
                    got.append([key, -1, -1])
                if b >= 0 and b >= got[seen[key]][1]:
                    got[seen[key]][1:] = [b, e]
            got = [(k, values[i][b:e]) for k, b, e in got if e > b]
            if got != fields[:len(got)] or (not got and fields):
                raise SystemExit("PARITY FAILURE: line %d fields differ from the oracle" % i)
    sample_bytes = int(length[:sample].sum())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    hist = np.bincount(pattern[pattern >= 0], minlength=g.n_match)
    out = {
        "metric": "Grok lines/s (50-pattern example_config list, mixed 128-4096B lines) per MI355X",
        "value": round(n * args.steps / elapsed, 1), "unit": "lines/s", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "MBps": round(total_bytes * args.steps / elapsed / 1e6, 2),
        "config": {"workload": "configs[2]: processor_grok, %d of the 50 example_config log-format patterns (ordered, first "
                               "match wins), %d lines of %d..%d B (mean %d) resident in HBM"
                               % (len(supported), n, int(length.min()), int(length.max()), int(length.mean())),
                   "patterns_supported": len(supported), "patterns_refused": refused,
                   "engines": {"tdfa": engines.count(binding.LC_ENGINE_TDFA), "nfa": engines.count(binding.LC_ENGINE_NFA)},
                   "matched_lines": int((pattern >= 0).sum()), "undecidable_lines": int((pattern == -2).sum()),
                   "undecidable_line_indices": [int(i) for i in np.nonzero(pattern == -2)[0][:32]],
                   "extra_match_rows": int(d_nextra.cpu()[0]),
                   "patterns_hit": int((hist > 0).sum()),
                   "warm_up_s": round(warm_s, 2)},
        # algorithmic HBM bytes of one step: every value read once + 4 B offset + 4 B length + the result row (pattern id +
        # first-match row) written once per line.  The NFA kernels are nowhere near it: they are bound by the dependent table
        # reads of a byte-step (one line per wavefront), not by HBM -- the fraction says how far.
        "roofline": {"bound": "hbm", "achieved": round((total_bytes + n * (8 + 4 + 4 * row)) * args.steps / elapsed / 1e9, 3),
                     "peak": 8000.0, "unit": "GB/s",
                     "frac": round((total_bytes + n * (8 + 4 + 4 * row)) * args.steps / elapsed / 1e9 / 8000.0, 6), "traffic": None,
                     "kernel": "nfa_match_kernel (40 of the 50 patterns) + tdfa kernels (10 patterns, prefix screens)",
                     "algorithmic_bytes_per_step": int(total_bytes + n * (8 + 4 + 4 * row))},
        "cpu_baseline": {"value": round(sample / cpu_s, 1), "unit": "lines/s", "cores": 1, "kind": "port",
                         "MBps": round(sample_bytes / cpu_s / 1e6, 3),
                         "sample": "first %d lines of the batch: oracle/grok_oracle.py over oracle/bt_regex.c "
                                   "(processGrok restated), 1 thread" % sample},
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
