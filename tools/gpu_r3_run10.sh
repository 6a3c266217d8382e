#!/bin/bash
# round 3, call 10: Grok worker streams vs hardware queues (GPU_MAX_HW_QUEUES defaults to 4: streams beyond that share a queue)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for cfg in "8 4" "8 8" "16 16" "16 24"; do
  set -- $cfg
  echo "== LC_GROK_STREAMS=$1 GPU_MAX_HW_QUEUES=$2"
  LC_GROK_STREAMS=$1 GPU_MAX_HW_QUEUES=$2 timeout 300 python tools/grok_bench.py --lines 1000,16384,65536 --steps 5 --warmup 4 --no-sequential-check --cpu-sample-lines 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['config']['workload'].split(', ')[2][:14], d['value'], 'lines/s', d['ms_per_step'], 'ms')
"
done
