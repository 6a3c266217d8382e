#!/bin/bash
# round 4, call 7: which entries are the long poles of a 16 Ki Grok step (per-entry trace + timeline)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4_grok_6
LC_GROK_TRACE=1 GPU_MAX_HW_QUEUES=16 timeout 300 python tools/grok_bench.py --lines 16384 --steps 1 --warmup 3 --no-sequential-check --cpu-sample-lines 50 2> gpurun_out/r4_grok_6/trace.err >/dev/null
grep "grok plan" gpurun_out/r4_grok_6/trace.err | tail -80 > gpurun_out/r4_grok_6/trace.txt
cd /tmp && export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=16 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4_grok_6/prof -o r1 -- python $R/tools/grok_bench.py --lines 16384 --steps 2 --warmup 3 --no-sequential-check --cpu-sample-lines 50 > $R/gpurun_out/r4_grok_6/prof.log 2>&1
cd $R
python tools/grok_timeline.py gpurun_out/r4_grok_6/prof 600 > gpurun_out/r4_grok_6/grok_timeline.txt 2>&1
rm -rf gpurun_out/r4_grok_6/prof
tail -3 gpurun_out/r4_grok_6/trace.txt
