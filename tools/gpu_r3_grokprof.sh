#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3gp
O=gpurun_out/r3gp
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/grok_prof -o r1 -- python $R/tools/grok_bench.py --lines 16384 --steps 3 --warmup 2 --no-sequential-check --cpu-sample-lines 100 > $R/$O/grok_prof.log 2>&1
cd $R && python tools/grok_prof_summary.py $O/grok_prof > $O/grok_rocprofv3.txt 2>&1; rm -rf $O/grok_prof; head -44 $O/grok_rocprofv3.txt | cut -c1-140
