#!/bin/bash
# tools/gpu_round4_last2.sh: the bounded-window in-agent legs on the device -- bench.py's end_to_end block as the driver's line runs
# it (smaller headline batch, no CPU baseline, no other configs), the processors' GPU tests, the native bench (parts a, b, c).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/last2 scratch
timeout 150 python bench.py --lines 262144 --steps 3 --warmup 1 --no-cpu-baseline --no-configs > gpurun_out/last2/bench_e2e.json 2> gpurun_out/last2/bench_e2e.err
echo "bench rc $?"; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/last2/bench_e2e.json").read().strip().splitlines()[-1])
    e = d["end_to_end"]
    print({k: e[k] for k in e if k.startswith("in_agent") and not k.endswith("what")})
    print("pipeline", e.get("pipeline", {}).get("fused_MBps"), "filter", e.get("filter", {}).get("MBps"), "multiline", e.get("multiline", {}).get("MBps"))
except Exception as ex:
    print("no bench line:", ex)
PY
tail -3 gpurun_out/last2/bench_e2e.err | cut -c1-300
timeout 60 python -m pytest tests/test_gpu_processor.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-200 | tee gpurun_out/last2/pytest_processors.txt
g++ -O2 -std=c++17 -I include tools/inagent_bench.cpp -o scratch/inagent_bench -L loongcollector_amd/lib -llc_regex_gpu -lpthread &&
  LD_LIBRARY_PATH=loongcollector_amd/lib:/opt/rocm/lib timeout 40 scratch/inagent_bench 256000 1000 1 16 2>&1 | tee gpurun_out/last2/inagent.txt
