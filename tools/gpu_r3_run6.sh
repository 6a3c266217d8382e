#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python tools/grok_bench.py --lines 1000,4096,16384,65536 --steps 5 --warmup 4 --no-sequential-check --cpu-sample-lines 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['config']['workload'].split(', ')[2][:14], d['value'], 'lines/s', d['ms_per_step'], 'ms')
"
