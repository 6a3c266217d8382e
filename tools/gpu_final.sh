#!/bin/bash
# tools/gpu_final.sh: what the driver runs at round end (GPU tests, smoke, default bench) + the Grok bench and its kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  File \"/usr" | tail -8 | cut -c1-300 > gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 300 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; cut -c1-420 gpurun_out/bench.json
timeout 200 python tools/grok_bench.py --lines 16384 --steps 3 --warmup 1 > gpurun_out/grok_bench.json 2> gpurun_out/grok_bench.err
timeout 200 python tools/grok_bench.py --lines 65536 --steps 3 --warmup 1 > gpurun_out/grok_bench64k.json 2>> gpurun_out/grok_bench.err
python - <<PY
import json
for f in ("gpurun_out/grok_bench.json", "gpurun_out/grok_bench64k.json"):
    d = json.loads(open(f).read()); c = d["config"]
    print(f, d["value"], "lines/s", d["ms_per_step"], "ms", "undecidable", c["undecidable_lines"], "matched", c["matched_lines"], "hit", c["patterns_hit"])
PY
if [ "$1" == "prof" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/grok_prof -o r1 -- python $R/tools/grok_bench.py --lines 16384 --steps 1 --warmup 0 --cpu-sample-lines 200 > $R/gpurun_out/grok_prof.log 2>&1
  cd $R && python tools/grok_prof_summary.py gpurun_out/grok_prof > gpurun_out/grok_prof_summary.txt 2>&1; head -14 gpurun_out/grok_prof_summary.txt | cut -c1-120
  rm -rf gpurun_out/grok_prof
fi
