#!/bin/bash
# round 3: the middle ground -- 8 hardware queues: Grok with 16 and 8 worker streams, and the multi-threaded host legs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for s in 16 8; do
echo "== GPU_MAX_HW_QUEUES=8 LC_GROK_STREAMS=$s"
GPU_MAX_HW_QUEUES=8 LC_GROK_STREAMS=$s timeout 300 python tools/grok_bench.py --lines 1000,16384 --steps 5 --warmup 4 --no-sequential-check --cpu-sample-lines 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['config']['workload'].split(', ')[2][:14], d['value'], 'lines/s', d['ms_per_step'], 'ms')
"
done
q=8
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --no-configs > gpurun_out/r3_hwq$q.json 2>gpurun_out/r3_hwq$q.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3_hwq$q.json").read()); e=d["end_to_end"]
print("queues=$q", "in_agent", e["in_agent_MBps"], "columnar", e["in_agent_columnar_MBps"], "pipeline", e["pipeline"]["fused_MBps"], "filter", e["filter"]["MBps"], "multiline", e["multiline"]["MBps"], "host", e["host_MBps"])
PY
