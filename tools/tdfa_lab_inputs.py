#!/usr/bin/env python3
"""Inputs for tools/tdfa_lab.hip: the headline batch (BASELINE configs[1]) and the compact TDFA tables of regex A, in one
binary file.  usage: tdfa_lab_inputs.py out.bin [n_lines]"""
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from loongcollector_amd import binding as B, corpus  # noqa: E402

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
kind = sys.argv[3] if len(sys.argv) > 3 else "A"
# LAB_EMPTY_EVERY=k: every k-th line gets an EMPTY referrer field ("" -- the capture group matches the empty string, a transition
# that stamps two registers at once), same line length
data, off, length = corpus.apache_batch(n, kind, empty_every=int(os.environ.get("LAB_EMPTY_EVERY", "0")))
rx = B.GpuRegex(corpus.REGEX_A if kind == "A" else corpus.REGEX_B)
blob = rx.table(B.LC_TABLE_TDFA_WIDE_BLOB, np.uint32)
assert blob is not None, "no compact blob"
info = rx.info()
data = np.concatenate([data, np.zeros((-len(data)) % 16 + 64, np.uint8)])
with open(out, "wb") as f:
    f.write(struct.pack("<8I", 0x4C414254, n, len(data), blob.nbytes, rx.groups, (int(blob[3]) & 0xFFFF) - 1, int(blob[15]), 0))  # (packed registers, without the dummy)
    f.write(data.tobytes())
    f.write(off.astype(np.uint32).tobytes())
    f.write(blob.tobytes())
print("wrote %s: %d lines, %d data bytes, blob %d bytes, block %d" % (out, n, len(data), blob.nbytes, int(blob[15])))
