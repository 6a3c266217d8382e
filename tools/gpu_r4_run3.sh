#!/bin/bash
# round 4, call 3: Grok after atomic elision + NFA early exit / steady-run skip / anchored NFA searches / TDFA absorbing state
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests/test_gpu_grok.py tests/test_gpu_parity.py tests/test_multiline.py tests/test_go_regex.py tests/test_filter.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-300
GPU_MAX_HW_QUEUES=16 bash tools/gpu_grok_profile.sh r4_grok_2 16384
