#!/bin/bash
# round 4, call 13: run_capture_kernel with 16-byte loads; the doomed-spawn GPU test
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_grok.py tests/test_gpu_parity.py -m gpu -q -x -k "grok or doomed or run_capture or golden" 2>&1 | tail -4 | cut -c1-300
GPU_MAX_HW_QUEUES=16 timeout 300 python tools/grok_bench.py --lines 1000,16384,65536 --steps 5 --warmup 4 --no-sequential-check --cpu-sample-lines 300 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['workload'][-60:], d['ms_per_step'])"
