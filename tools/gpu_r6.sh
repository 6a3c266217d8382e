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
groktests)
  timeout 1500 python -m pytest tests/test_gpu_grok.py tests/test_go_regex.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-300 | tee gpurun_out/r6/pytest_grok.txt ;;
esac
done
