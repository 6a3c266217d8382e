#!/bin/bash
# helper run on the GPU box by gpurun (tools/gpu_check.sh [test|notest] [prof]): parity tests + bench + optional rocprof passes (outputs under gpurun_out/)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
if [ "$1" != "notest" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
fi
timeout 300 python bench.py --steps 20 --warmup 3 ${BENCH_ARGS} > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -2 gpurun_out/bench.err
if [ "$2" == "prof" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o r1 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/prof_stats.log 2>&1
  timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_fetch -o r1 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/prof_fetch.log 2>&1
  timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_write -o r1 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/prof_write.log 2>&1
  i=0
  IFS='|' read -ra SETS <<< "${PMC_EXTRA}"
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout 400 rocprofv3 --pmc $set -d $R/gpurun_out/prof_pmc$i -o r1 -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/prof_pmc$i.log 2>&1
  done
  cd $R && python tools/prof_summary.py gpurun_out > gpurun_out/prof_summary.txt 2>&1; cat gpurun_out/prof_summary.txt
fi
