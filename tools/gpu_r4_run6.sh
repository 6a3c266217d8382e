#!/bin/bash
# round 4, call 6: Grok with the phased round 0 (one post launch, deferred second chance, one remainder launch), the wave TDFA kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_gpu_grok.py tests/test_gpu_parity.py tests/test_go_regex.py -m gpu -q -x 2>&1 | tail -12 | cut -c1-300
GPU_MAX_HW_QUEUES=16 bash tools/gpu_grok_profile.sh r4_grok_5 16384 2>&1 | grep -v "^void\|^grok_\|^nfa_\|^sched\|^__amd\|^tdfa\|^run_cap" | head -30
