#!/bin/bash
# round 3: do 16 hardware queues cost the multi-threaded host paths anything?  (bench.py end_to_end legs, GPU_MAX_HW_QUEUES 4 vs 16)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for q in 4 16; do
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --no-configs > gpurun_out/r3_hwq$q.json 2>gpurun_out/r3_hwq$q.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_hwq$q.json").read()); e=d["end_to_end"]
    print("queues=$q", "in_agent", e["in_agent_MBps"], "columnar", e["in_agent_columnar_MBps"], "pipeline", e["pipeline"]["fused_MBps"], "filter", e["filter"]["MBps"], "multiline", e["multiline"]["MBps"], "host", e["host_MBps"])
except Exception as ex:
    print("failed", ex); print(open("gpurun_out/r3_hwq$q.err").read()[-1500:])
PY
done
