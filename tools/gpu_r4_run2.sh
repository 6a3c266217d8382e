#!/bin/bash
# round 4, call 2: configs[2] on the BASELINE-shaped corpus (128..4096 B, 44 winning entries) before any Grok change; new multiline tests
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_grok.py tests/test_multiline.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-300
GPU_MAX_HW_QUEUES=16 bash tools/gpu_grok_profile.sh r4_grok_before 16384
