#!/bin/bash
# tools/gpu_round2_refresh.sh: the measured numbers DESIGN.md / README.md / profiles/round2_* quote, in one gpurun call:
# default bench line, its kernel trace and HBM counters (separate --pmc passes), the Grok bench at three batch sizes + its trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | cut -c1-200 > gpurun_out/r2f_pytest_gpu.log; tail -2 gpurun_out/r2f_pytest_gpu.log
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -2 | cut -c1-300
timeout 400 python bench.py > gpurun_out/r2f_bench.json 2> gpurun_out/r2f_bench.err; cut -c1-600 gpurun_out/r2f_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > $R/gpurun_out/prof_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o r1 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $R/gpurun_out/prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o r1 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $R/gpurun_out/prof_write.log 2>&1
cd $R && python tools/prof_summary.py gpurun_out > gpurun_out/r2f_prof_summary.txt 2>&1; head -40 gpurun_out/r2f_prof_summary.txt | cut -c1-160
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
for n in 16384 65536 262144 1048576; do
  timeout 400 python tools/grok_bench.py --lines $n --steps 3 --warmup 1 --cpu-sample-lines 300 > gpurun_out/r2f_grok_$n.json 2> gpurun_out/r2f_grok_$n.err
  echo "grok n=$n $(tail -1 gpurun_out/r2f_grok_$n.json | cut -c90-200)"
done
LC_GROK_TRACE=1 timeout 300 python tools/grok_bench.py --lines 262144 --steps 1 --warmup 1 --cpu-sample-lines 100 > /dev/null 2> gpurun_out/r2f_grok_trace.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/grok_prof -o r1 -- python $R/tools/grok_bench.py --lines 262144 --steps 1 --warmup 0 --cpu-sample-lines 100 > $R/gpurun_out/grok_prof.log 2>&1
cd $R && python tools/grok_prof_summary.py gpurun_out/grok_prof > gpurun_out/r2f_grok_prof_summary.txt 2>&1; head -16 gpurun_out/r2f_grok_prof_summary.txt | cut -c1-120
rm -rf gpurun_out/grok_prof
