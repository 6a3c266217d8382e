#!/bin/bash
# round 3: tools/tdfa_lab.hip LAB_DMA variants on the headline batch -> gpurun_out/$1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python tools/tdfa_lab_inputs.py /tmp/lab_in.bin > /dev/null || exit 1
LAB_DMA=1 timeout 200 scratch/tdfa_lab /tmp/lab_in.bin 20 2>&1 | cut -c1-170 | tee gpurun_out/${1:-r3_lab.txt}
