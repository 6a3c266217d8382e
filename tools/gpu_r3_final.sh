#!/bin/bash
# round 3, final evidence: what the driver runs at round end (GPU tests, smoke, the default bench line) + the kernel traces
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3final
O=gpurun_out/r3final
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File \"/usr" | tail -6 | cut -c1-300 > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -1 | cut -c1-300
timeout 420 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-420 $O/bench_n1.json; tail -2 $O/bench_n1.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof_stats -o r1 -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-configs > $R/$O/stats.log 2>&1
cd $R && python tools/prof_summary.py $O > $O/tdfa_kernel_rocprofv3.txt 2>&1; rm -rf $O/prof_stats; head -8 $O/tdfa_kernel_rocprofv3.txt | cut -c1-140
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/grok_prof -o r1 -- python $R/tools/grok_bench.py --lines 16384 --steps 3 --warmup 2 --no-sequential-check --cpu-sample-lines 100 > $R/$O/grok_prof.log 2>&1
cd $R && python tools/grok_prof_summary.py $O/grok_prof > $O/grok_rocprofv3.txt 2>&1; rm -rf $O/grok_prof; head -12 $O/grok_rocprofv3.txt | cut -c1-140
