#!/bin/bash
# round 3 experiments: hardware queues / worker streams for the speculative Grok path, and the 256 Ki timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
for q in 4 8; do
  for s in 4 8; do
    echo "== GPU_MAX_HW_QUEUES=$q LC_GROK_STREAMS=$s"
    GPU_MAX_HW_QUEUES=$q LC_GROK_STREAMS=$s timeout 600 python tools/grok_bench.py --lines 1000,16384,65536,262144 --steps 5 --warmup 2 --no-sequential-check --cpu-sample-lines 200 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['config']['workload'].split('patterns (')[1][:10], d['config']['workload'].split(', ')[2][:14], d['value'], 'lines/s', d['ms_per_step'], 'ms')
"
  done
done
bash tools/gpu_grok_prof.sh 262144 > /dev/null 2>&1
grep -v "0.00[0-9] \+[0-9]" gpurun_out/grok_timeline_262144.txt | head -120
