#!/bin/bash
# tools/gpu_ab.sh: GPU parity suite, then bench.py once per LC_TDFA_COMPACT variant (0 = the default 32-bit kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^  File \"/usr" | tail -25 | cut -c1-220 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
for spec in ${VARIANTS:-0 256 512 1024}; do
  v=${spec%%p}; pair=0; [ "$v" != "$spec" ] && pair=1       # "512p" = compact 512 with the byte-pair table
  LC_TDFA_PAIR=$pair LC_TDFA_COMPACT=$v timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c$v.json 2>gpurun_out/bench.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_c$v.json").read())
print("compact=$spec", "MB/s", d["value"], "ms/step", d["ms_per_step"], "kernel ms", d["roofline"]["avg_kernel_ms"], "frac", d["roofline"]["frac"])
PY
done
