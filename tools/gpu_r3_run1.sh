#!/bin/bash
# round 3, GPU call 1: the device-first multiline / filter paths (parity + throughput), the row-stride A/B on the stream kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_multiline.py tests/test_filter.py tests/test_gpu_pipeline.py tests/test_gpu_processor.py -m gpu -q -x 2>&1 | grep -v "^  File \"/usr" | tail -25 | cut -c1-300 > gpurun_out/r3_pytest_ml.log; cat gpurun_out/r3_pytest_ml.log
for pad in 0 1; do
  LC_TDFA_ROW_PAD=$pad timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-configs > gpurun_out/r3_bench_pad$pad.json 2>gpurun_out/r3_bench_pad$pad.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_bench_pad$pad.json").read())
    print("row pad=$pad", "MB/s", d["value"], "ms/step", d["ms_per_step"], "kernel ms", d["roofline"]["avg_kernel_ms"], "frac", d["roofline"]["frac"], d["config"].get("lds_table_bytes"))
except Exception as e:
    print("row pad=$pad failed", e); print(open("gpurun_out/r3_bench_pad$pad.err").read()[-1500:])
PY
done
timeout 600 python bench.py --no-cpu-baseline --no-configs > gpurun_out/r3_bench_e2e.json 2>gpurun_out/r3_bench_e2e.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_bench_e2e.json").read())
    e=d["end_to_end"]
    print("in_agent", e["in_agent_MBps"]); print("pipeline", e["pipeline"]["fused_MBps"], e["pipeline"]["three_steps_MBps"])
    print("multiline", e["multiline"]["MBps"], e["multiline"]["three_patterns_MBps"]); print("filter", e["filter"]["MBps"])
except Exception as ex:
    print("e2e failed", ex); print(open("gpurun_out/r3_bench_e2e.err").read()[-2500:])
PY
