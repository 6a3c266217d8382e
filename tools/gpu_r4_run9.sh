#!/bin/bash
# round 4, call 9: Grok -- one fork for the levels, NFA programs staged in LDS for small batches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1500 python -m pytest tests/test_gpu_grok.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300
GPU_MAX_HW_QUEUES=16 bash tools/gpu_grok_profile.sh r4_grok_8 16384 2>&1 | grep -v "^void\|^grok_\|^nfa_\|^sched\|^__amd\|^tdfa\|^run_cap" | head -30
for sb in 0; do
  echo "== LC_NFA_SMALL_BATCH=$sb"
  LC_NFA_SMALL_BATCH=$sb GPU_MAX_HW_QUEUES=16 timeout 300 python tools/grok_bench.py --lines 1000,16384 --steps 5 --warmup 4 --no-sequential-check --cpu-sample-lines 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][-60:], d['ms_per_step'])"
done
