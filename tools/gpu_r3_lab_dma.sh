#!/bin/bash
# round 3: the LDS-DMA staging variants of tdfa_stream_kernel (tools/tdfa_lab.hip, LAB_DMA) on the headline batch
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python tools/tdfa_lab_inputs.py /tmp/lab_in.bin > /dev/null || exit 1
LC_TDFA_PAIR=1 LC_TDFA_COMPACT=512 python tools/tdfa_lab_inputs.py /tmp/lab_pair.bin > /dev/null || exit 1
echo "== product tables"; LAB_DMA=1 timeout 200 scratch/tdfa_lab /tmp/lab_in.bin 20 2>&1 | cut -c1-170 | tee gpurun_out/r3_lab_dma.txt
echo "== pair tables (512 lanes)"; LAB_DMA=1 timeout 200 scratch/tdfa_lab /tmp/lab_pair.bin 20 2>&1 | cut -c1-170 | tee -a gpurun_out/r3_lab_dma.txt
