#!/bin/bash
# round 3, call 9: tdfa_l2_kernel with learned self loops -- parity (L2 tests + the Grok suite), Grok at four batch sizes, in-agent shape
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "l2 or L2 or global" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_grok.py -m gpu -x -q 2>&1 | tail -3
timeout 400 python tools/grok_bench.py --lines 1000,16384,65536,262144 --steps 5 --warmup 4 --cpu-sample-lines 200 > gpurun_out/r3_grok_bench2.json 2> gpurun_out/r3_grok_bench2.err
python - <<PY
import json
for l in open("gpurun_out/r3_grok_bench2.json"):
    d = json.loads(l); c = d["config"]
    print(c["workload"].split(", ")[2][:14], d["value"], "lines/s", d["ms_per_step"], "ms", c.get("parity"))
PY
tail -2 gpurun_out/r3_grok_bench2.err
timeout 300 python tools/grok_inagent_bench.py --threads 1,16 --group 1000 --groups 20 2>/dev/null | cut -c1-260
