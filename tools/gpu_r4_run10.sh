#!/bin/bash
# round 4, call 10: Grok with measured per-entry costs (calibration batches) and longest-first dealing
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_gpu_grok.py tests/test_multiline.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300
GPU_MAX_HW_QUEUES=16 bash tools/gpu_grok_profile.sh r4_grok_9 16384 2>&1 | grep -v "^void\|^grok_\|^nfa_\|^sched\|^__amd\|^tdfa\|^run_cap" | head -30
GPU_MAX_HW_QUEUES=16 timeout 300 python tools/grok_inagent_bench.py 2>/dev/null | tail -3 | cut -c1-600
