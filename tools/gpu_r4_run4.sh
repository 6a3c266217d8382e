#!/bin/bash
# round 4, call 4: Grok with remainder screens
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_gpu_grok.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-300
GPU_MAX_HW_QUEUES=16 bash tools/gpu_grok_profile.sh r4_grok_3 16384
