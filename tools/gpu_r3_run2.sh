#!/bin/bash
# round 3, GPU call 2: how the stream kernel scales with resident workgroups (latency- or pipe-bound?), multiline bench without Python in the loop
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for extra in 0 11264 24576 60000; do
  LC_TDFA_EXTRA_LDS=$extra timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-configs > gpurun_out/r3_occ_$extra.json 2>gpurun_out/r3_occ_$extra.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_occ_$extra.json").read())
    print("extra LDS=$extra", "ms/step", d["ms_per_step"], "kernel ms", d["roofline"]["avg_kernel_ms"], "frac", d["roofline"]["frac"], d["roofline"]["kernels_launched"])
except Exception as e:
    print("extra=$extra failed", e); print(open("gpurun_out/r3_occ_$extra.err").read()[-1500:])
PY
done
python - <<PY
import json, sys
sys.path.insert(0, ".")
import bench
print(json.dumps(bench.measure_multiline([1, 4, 16])))
PY
