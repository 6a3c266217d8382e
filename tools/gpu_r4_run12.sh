#!/bin/bash
# round 4, call 12: worker streams x hardware queues with the round-4 Grok kernels (the round-3 answer was 16 x 16)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for q in 4 8 16; do for s in 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q LC_GROK_STREAMS=$s"
  GPU_MAX_HW_QUEUES=$q LC_GROK_STREAMS=$s timeout 300 python tools/grok_bench.py --lines 1000,16384,65536 --steps 5 --warmup 4 --no-sequential-check --cpu-sample-lines 50 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ', d['config']['lines_per_batch'] if 'lines_per_batch' in d['config'] else d['config']['workload'][-58:-40], d['ms_per_step'])"
done; done 2>&1 | tee gpurun_out/r4_grok_streams.txt
