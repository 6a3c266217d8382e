#!/bin/bash
# tools/gpu_r6.sh WHAT...: round-6 measurement calls (one gpurun call each; outputs under gpurun_out/r6/)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6
for what in "$@"; do
case $what in
suite)
  timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | cut -c1-300 | tee gpurun_out/r6/pytest_gpu.txt ;;
inagent_trace)
  # where a 16-thread in-agent Grok run spends its time: the host stages of every merged batch
  LC_GROK_TRACE=1 timeout 600 python tools/grok_inagent_bench.py --threads 1,16 --groups 20 > gpurun_out/r6/inagent.json 2> gpurun_out/r6/inagent_trace.txt
  cat gpurun_out/r6/inagent.json | cut -c1-400
  grep "grok host batch" gpurun_out/r6/inagent_trace.txt | tail -30 | cut -c1-250 ;;
inagent)
  timeout 600 python tools/grok_inagent_bench.py --threads 1,4,16,32 --groups 40 2>/dev/null | cut -c1-330 | tee gpurun_out/r6/inagent_$(date +%H%M).json ;;
headline)
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
  for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'])"; done ;;
lazy)
  timeout 900 python -m pytest tests/test_lazy_tdfa.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-300
  for lz in 1 0; do LC_LAZY_TDFA=$lz GPU_MAX_HW_QUEUES=16 timeout 900 python tools/grok_bench.py --lines 1000,16384,65536 --steps 10 --warmup 2 --cpu-sample-lines 1500 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('lazy=$lz', d['config']['workload'][-60:-20], d['ms_per_step'], 'ms', d['config']['batch'], d['config'].get('lazy_automata'))"; done ;;
groktests)
  timeout 1500 python -m pytest tests/test_gpu_grok.py tests/test_go_regex.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-300 | tee gpurun_out/r6/pytest_grok.txt ;;
esac
done
