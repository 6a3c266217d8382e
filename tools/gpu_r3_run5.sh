#!/bin/bash
# round 3, GPU call 5: Grok with the global-table automata scheduled first; the new end_to_end legs; pending groups per pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_grok.py -x -q 2>&1 | tail -3
timeout 600 python tools/grok_bench.py --lines 1000,16384,65536 --steps 5 --warmup 4 --no-sequential-check --cpu-sample-lines 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['config']['workload'].split(', ')[2][:14], d['value'], 'lines/s', d['ms_per_step'], 'ms')
"
timeout 600 python bench.py --no-cpu-baseline --no-configs > gpurun_out/r3_bench_e2e.json 2>gpurun_out/r3_bench_e2e.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_bench_e2e.json").read())
    e=d["end_to_end"]
    print("in_agent", e["in_agent_MBps"], "columnar", e["in_agent_columnar_MBps"]); print("pipeline", e["pipeline"]["fused_MBps"], e["pipeline"]["three_steps_MBps"])
    print("multiline", e["multiline"]["MBps"], e["multiline"]["three_patterns_MBps"]); print("filter", e["filter"]["MBps"])
except Exception as ex:
    print("e2e failed", ex); print(open("gpurun_out/r3_bench_e2e.err").read()[-2500:])
PY
timeout 300 python bench.py --config 4 > gpurun_out/r3_bench_config4.json 2>gpurun_out/r3_bench_config4.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_bench_config4.json").read())
    print("config4", d["value"], d["roofline"]["frac"], d["pending_groups_per_pipeline"])
except Exception as ex:
    print("config4 failed", ex); print(open("gpurun_out/r3_bench_config4.err").read()[-2500:])
PY
