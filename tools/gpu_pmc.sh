#!/bin/bash
# tools/gpu_pmc.sh TAG [ENV=VAL ...]: kernel-trace stats + two SQ counter passes of bench.py on the GPU box -> gpurun_out/TAG_*
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof_stats -o r1 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/$TAG/stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAVE_CYCLES -d $R/gpurun_out/$TAG/prof_pmc1 -o r1 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d $R/gpurun_out/$TAG/prof_pmc2 -o r1 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/$TAG/pmc2.log 2>&1
cd $R && python tools/prof_summary.py gpurun_out/$TAG > gpurun_out/$TAG/summary.txt 2>&1
rm -rf gpurun_out/$TAG/prof_*
cat gpurun_out/$TAG/summary.txt | grep -v "rocclr\|at::native"
