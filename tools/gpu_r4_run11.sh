#!/bin/bash
# round 4, call 11: doomed-spawn (quasi-steady) skipping in the NFA kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_gpu_grok.py tests/test_gpu_parity.py tests/test_go_regex.py tests/test_filter.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300
GPU_MAX_HW_QUEUES=16 bash tools/gpu_grok_profile.sh r4_grok_10 16384 2>&1 | grep -v "^void\|^grok_\|^nfa_\|^sched\|^__amd\|^tdfa\|^run_cap" | head -30
echo "== LC_NFA_NO_QUASI=1"
LC_NFA_NO_QUASI=1 GPU_MAX_HW_QUEUES=16 timeout 300 python tools/grok_bench.py --lines 1000,16384 --steps 5 --warmup 4 --no-sequential-check --cpu-sample-lines 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][-60:], d['ms_per_step'])"
