#!/bin/bash
# tools/gpu_round4_last3.sh: the Grok GPU tests and the 16 Ki step with the eighth additional anchored automaton (CISCOFW713172)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/last3
timeout 80 python -m pytest tests/test_gpu_grok.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-200 | tee gpurun_out/last3/pytest_grok.txt
timeout 70 python tools/grok_bench.py --lines 16384 --steps 10 --warmup 3 > gpurun_out/last3/grok.json 2> gpurun_out/last3/grok.err
cut -c1-330 gpurun_out/last3/grok.json; tail -2 gpurun_out/last3/grok.err | cut -c1-200
