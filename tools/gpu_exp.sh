#!/bin/bash
# tools/gpu_exp.sh: bench.py against experimental builds of the library (loongcollector_amd/lib/exp/liblc_<name>.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
for spec in $EXPS; do
  name=${spec%%:*}; compact=${spec#*:}
  lib=$R/loongcollector_amd/lib/exp/liblc_$name.so
  [ "$name" == "base" ] && lib=$R/loongcollector_amd/lib/liblc_regex_gpu.so
  LC_REGEX_GPU_LIB=$lib LC_TDFA_COMPACT=$compact timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_exp.json 2>gpurun_out/bench.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_exp.json").read())
    print("$name compact=$compact", "MB/s", d["value"], "kernel ms", d["roofline"]["avg_kernel_ms"], "frac", d["roofline"]["frac"])
except Exception as e:
    print("$name compact=$compact FAILED", e, open("gpurun_out/bench.err").read()[-300:])
PY
done
