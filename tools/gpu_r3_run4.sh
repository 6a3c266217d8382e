#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for st in 0 400 800 1400 2000; do
  LC_TDFA_STAGGER=$st timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-configs > gpurun_out/r3_stag_$st.json 2>gpurun_out/r3_stag_$st.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_stag_$st.json").read())
    print("stagger ticks=$st", "ms/step", d["ms_per_step"], "kernel ms", d["roofline"]["avg_kernel_ms"], "frac", d["roofline"]["frac"])
except Exception as e:
    print("stagger=$st failed", e); print(open("gpurun_out/r3_stag_$st.err").read()[-1500:])
PY
done
