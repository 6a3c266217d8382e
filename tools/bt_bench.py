#!/usr/bin/env python3
"""tools/bt_bench.py -- what the device backtracking engine (LC_ENGINE_BT, csrc/bt_vm.hpp) costs beside the tagged DFA.

Regex A of the headline on Apache-combined 512 B lines resident in HBM, once on the handle's own engine (tdfa_stream_kernel) and once with
LC_ENGINE_BT asked for; then a pattern only the backtracking engine can run (the same line shape with a back-reference: the client address
must come back in the referer).  Every leg is checked against the oracle on the first 4 096 lines before it is timed (the oracle is the
checker, never the measured path).  One JSON line per leg.
  python tools/bt_bench.py [--lines 65536,262144] [--steps 10]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loongcollector_amd import binding as B  # noqa: E402
from loongcollector_amd import corpus  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lines", default="65536,262144")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from oracle.oracle import OracleRegex
    dev = torch.device("cuda:0")
    # the back-reference leg: "<ip> ... "<referer holding the ip again>" ..." -- group 1 must come back inside group 9
    backref = rb'([\d\.]+) \S+ \S+ \[(\S+) \S+\] \"(\w+) ([^\\"]*)\" ([\d\.]+) (\d+) (\d+) (\d+|-) \"([^\\"]*)\" \"\1 ([^\\"]*)\"'
    for n in [int(x) for x in args.lines.split(",")]:
        data, off, length = corpus.apache_batch(n, "A", poison_every=0)
        # two lines of three carry their client address again at the head of the user agent (what the back-reference leg asks for)
        data = data.copy()
        rows = data.reshape(n, -1)
        for i in range(n):
            if i % 3 == 2:
                continue
            row = rows[i].tobytes()
            ip = row[:row.index(b" ")]
            ua = row.rindex(b'"', 0, row.rindex(b'"')) + 1
            if ua + len(ip) + 1 < row.rindex(b'"'):
                rows[i, ua:ua + len(ip) + 1] = np.frombuffer(ip + b" ", dtype=np.uint8)
        d_data = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
        d_off = torch.from_numpy(np.ascontiguousarray(off, dtype=np.uint32).view(np.int32)).to(dev)
        total = int(length.sum())
        for name, pattern, engine in (("regex A, tagged DFA (the handle's engine)", corpus.REGEX_A, B.LC_ENGINE_AUTO),
                                      ("regex A, LC_ENGINE_BT", corpus.REGEX_A, B.LC_ENGINE_BT),
                                      ("regex A with a back-reference (\\1 at the head of the user agent), LC_ENGINE_BT", backref, B.LC_ENGINE_AUTO)):
            rx = B.GpuRegex(pattern, engine=engine if engine == B.LC_ENGINE_BT else B.LC_ENGINE_AUTO)
            G = rx.groups
            d_caps = torch.empty((n, 2 * G), dtype=torch.int32, device=dev)
            d_status = torch.empty((n,), dtype=torch.uint8, device=dev)
            st = torch.cuda.current_stream().cuda_stream

            def step():
                rx.match_device(d_data, d_off, None, n, d_caps, d_status, ngroups=G, sep_bytes=1, stream=st)
            step()
            torch.cuda.synchronize()
            k = min(n, 4096)
            exp_caps, exp_status = OracleRegex(pattern).fullmatch_batch(data, off[:k], length[:k])
            if not (np.array_equal(d_status.cpu().numpy()[:k], exp_status) and np.array_equal(d_caps.cpu().numpy()[:k], exp_caps)):
                raise SystemExit("PARITY FAILURE: %s" % name)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                step()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.steps
            print(json.dumps({"leg": name, "engine": rx.info()["engine"], "lines": n, "bytes": total, "ms_per_batch": round(ms, 4),
                              "GBps": round(total / ms / 1e6, 2), "matched": int((d_status == 1).sum().item()),
                              "gave_up": int((d_status == 3).sum().item()), "parity": "first %d lines equal the oracle" % k}), flush=True)


if __name__ == "__main__":
    main()
