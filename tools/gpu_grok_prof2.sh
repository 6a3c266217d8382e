#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_gpu_grok.py -x -q 2>&1 | tail -3
timeout 600 python tools/grok_inagent_bench.py --threads 1,4,16,32 2>&1 | tail -4 | tee gpurun_out/grok_inagent.json | cut -c1-200
echo "== 3 patterns only (a single log source)"
timeout 600 python tools/grok_inagent_bench.py --threads 1,4,16 --patterns 3 2>&1 | tail -3 | cut -c1-200
