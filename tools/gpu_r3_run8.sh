#!/bin/bash
# round 3, call 8: next-generation prefetch in the DMA kernel -- lab + bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
python tools/tdfa_lab_inputs.py /tmp/lab_in.bin > /dev/null || exit 1
echo "== product tables"; LAB_TRACE=1 LAB_DMA=1 timeout 200 scratch/tdfa_lab /tmp/lab_in.bin 20 2>&1 | cut -c1-170 | tee gpurun_out/r3_lab_prefetch.txt
for pf in 1 0; do
LC_TDFA_PREFETCH=$pf timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-configs > gpurun_out/r3_bench_pf$pf.json 2>gpurun_out/r3_bench_pf$pf.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3_bench_pf$pf.json").read())
    print("bench prefetch=$pf", "MB/s", d["value"], "ms/step", d["ms_per_step"], "kernel ms", d["roofline"]["avg_kernel_ms"], "frac", d["roofline"]["frac"], d["roofline"]["kernels_launched"])
except Exception as e:
    print("bench failed", e); print(open("gpurun_out/r3_bench_pf$pf.err").read()[-1500:])
PY
done
