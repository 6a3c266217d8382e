#!/bin/bash
# round 3: the ONE-STAMP pair table (LC_TDFA_PAIR=2) on the headline batch -- tools/tdfa_lab.hip, bit-exact against the single-byte walk
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
LC_TDFA_PAIR=2 LC_TDFA_COMPACT=512 python tools/tdfa_lab_inputs.py /tmp/lab_pair1.bin > /dev/null || exit 1
timeout 120 scratch/tdfa_lab /tmp/lab_pair1.bin 20 2>&1 | cut -c1-170 | tee gpurun_out/r3_lab_pair1.txt
