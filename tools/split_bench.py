#!/usr/bin/env python3
"""Line-split kernels (f1) on a raw read buffer resident in HBM: time per call and achieved HBM rate.
Algorithmic bytes: 2 reads of the buffer (count pass + scatter pass) + 4 B per line written."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loongcollector_amd import binding as B, corpus  # noqa: E402
from oracle.split_oracle import split_lines  # noqa: E402  (checker only)

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
data, off, length = corpus.mixed_batch(int((mib << 20) / 640))
nb = int(off[-1])
d = torch.from_numpy(data[:nb].copy()).to(dev)
d_off = torch.empty(len(length) + 8, dtype=torch.int32, device=dev)
d_n = torch.zeros(1, dtype=torch.int32, device=dev)
d_s = torch.empty(B.split_scratch_bytes(nb) // 4 + 1, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
B.split_lines_device(d, nb, d_off, d_n, d_s, stream=s)
torch.cuda.synchronize()
n = int(d_n.item())
assert n == len(length), (n, len(length))
assert np.array_equal(d_off[:n + 1].cpu().numpy().view(np.uint32), off), "split offsets differ from the corpus"
exp = split_lines(data[:1 << 20].tobytes())
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(20):
    B.split_lines_device(d, nb, d_off, d_n, d_s, stream=s)
ev1.record()
torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 20
algo = 2 * nb + 4 * (n + 1)
print("split: %d MiB, %d lines: %.3f ms per call (count + scan + scatter) -> %.0f GB/s of input, %.0f GB/s algorithmic HBM traffic "
      "(2 reads + offsets) = %.3f of 8 TB/s" % (nb >> 20, n, ms, nb / ms / 1e6, algo / ms / 1e6, algo / ms / 1e6 / 8000))
