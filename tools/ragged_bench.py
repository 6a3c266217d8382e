#!/usr/bin/env python3
"""Kernel rate on the config-5 style ragged corpus (70 % nginx lines 128-2048 B, 30 % JSON lines that must fail)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loongcollector_amd import binding, corpus  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
data, off, length = corpus.mixed_batch(n)
reps = 8
data = np.tile(data, reps)
span = int(off[-1])
off = np.concatenate([off[:-1] + np.uint32(span * r) for r in range(reps)] + [np.array([span * reps], np.uint32)])
length = np.tile(length, reps)
n *= reps
dev = torch.device("cuda:0")
rx = binding.GpuRegex(corpus.REGEX_B)
G = rx.groups
d_data = torch.from_numpy(data).to(dev)
d_off = torch.from_numpy(off.view(np.int32)).to(dev)
d_caps = torch.empty((n, 2 * G), dtype=torch.int32, device=dev)
d_status = torch.empty((n,), dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream()
d_scratch = torch.empty((binding.sched_scratch_bytes(n) // 4 + 1,), dtype=torch.int32, device=dev)
for _ in range(3):
    rx.match_device_ragged(d_data, d_off, None, n, d_caps, d_status, d_scratch, sep_bytes=1, stream=s.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    rx.match_device_ragged(d_data, d_off, None, n, d_caps, d_status, d_scratch, sep_bytes=1, stream=s.cuda_stream)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print("ragged %-17s: %d lines, %.0f MB, %.3f ms -> %.1f GB/s parsed (device counting sort + match)"
      % ("lc_..._ragged", n, length.sum() / 1e6, dt * 1e3, length.sum() / dt / 1e9))
for order_name in ("file order", "sorted by length"):
    if order_name != "file order":
        perm = np.argsort(length, kind="stable")
        d_o = torch.from_numpy(off[:-1][perm].view(np.int32).copy()).to(dev)
        d_l = torch.from_numpy(length[perm].view(np.int32).copy()).to(dev)
    else:
        d_o, d_l = d_off, None
    for _ in range(3):
        rx.match_device(d_data, d_o, d_l, n, d_caps, d_status, sep_bytes=1, stream=s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        rx.match_device(d_data, d_o, d_l, n, d_caps, d_status, sep_bytes=1, stream=s.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("ragged %-17s: %d lines, %.0f MB, %.3f ms -> %.1f GB/s parsed, matched %.1f%%"
          % (order_name, n, length.sum() / 1e6, dt * 1e3, length.sum() / dt / 1e9, 100 * float(d_status.float().mean())))
