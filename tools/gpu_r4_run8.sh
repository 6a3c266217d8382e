#!/bin/bash
# round 4, call 8: Grok with shadow levels + wave kernel for LDS-size automata
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_gpu_grok.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -8 | cut -c1-300
GPU_MAX_HW_QUEUES=16 bash tools/gpu_grok_profile.sh r4_grok_7 16384 2>&1 | grep -v "^void\|^grok_\|^nfa_\|^sched\|^__amd\|^tdfa\|^run_cap" | head -30
