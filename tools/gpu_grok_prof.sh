#!/bin/bash
# rocprofv3 kernel trace of one Grok batch size: tools/gpu_grok_prof.sh LINES
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-16384}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/grok_prof_$N -o r1 -- python $R/tools/grok_bench.py --lines $N --steps 2 --warmup 1 --no-sequential-check --cpu-sample-lines 50 > $R/gpurun_out/grok_prof_$N.log 2>&1
cd $R
DB=$(dirname $(find gpurun_out/grok_prof_$N -name "r1_results.db" | head -1))
python tools/grok_prof_summary.py $DB > gpurun_out/grok_prof_${N}_summary.txt 2>&1
python tools/grok_timeline.py $DB 400 > gpurun_out/grok_timeline_$N.txt 2>&1
rm -rf gpurun_out/grok_prof_$N
head -60 gpurun_out/grok_timeline_$N.txt
