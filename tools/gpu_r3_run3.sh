#!/bin/bash
# round 3, GPU call 3: byte-pair chunks with ONE stamp per pair (timing only) against the product, same lab process
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
LC_TDFA_PAIR=1 LC_TDFA_COMPACT=512 python tools/tdfa_lab_inputs.py /tmp/lab_pair.bin > /dev/null || exit 1
python tools/tdfa_lab_inputs.py /tmp/lab_in.bin > /dev/null || exit 1
echo "== pair tables (512 lanes)"; timeout 300 scratch/tdfa_lab /tmp/lab_pair.bin 20 | cut -c1-170
echo "== product tables"; LAB_ONLY=1 timeout 300 scratch/tdfa_lab /tmp/lab_in.bin 20 | cut -c1-170
