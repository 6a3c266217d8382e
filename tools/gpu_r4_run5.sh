#!/bin/bash
# round 4, call 5: Grok with the wide kernel's shortcuts; byte-pair tables for regex B (headline) and for the STANDARD tables (small batches)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4_pair
timeout 900 python -m pytest tests/test_gpu_grok.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
GPU_MAX_HW_QUEUES=16 bash tools/gpu_grok_profile.sh r4_grok_4 16384 2>&1 | grep -v "^void\|^grok_\|^nfa_\|^sched\|^__amd\|^tdfa\|^run_cap" | head -30
show() { python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1', 'MB/s', d['value'], 'kernel ms', r['avg_kernel_ms'], 'frac', r['frac'], r['kernels_launched'])"; }
for pct in 1 2; do
  LC_TDFA_PAIR_DOUBLE_PCT=$pct timeout 200 python bench.py --regex B --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | show "regex B DOUBLE_PCT=$pct"
done
for pf in default 2; do
  if [ $pf = default ]; then unset LC_TDFA_PAIR; else export LC_TDFA_PAIR=$pf; fi
  timeout 400 python bench.py --no-cpu-baseline --no-configs > gpurun_out/r4_pair/e2e_$pf.json 2> gpurun_out/r4_pair/e2e_$pf.err
  timeout 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/r4_pair/cfg4_$pf.json 2> gpurun_out/r4_pair/cfg4_$pf.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4_pair/e2e_$pf.json").read()); e=d["end_to_end"]
    print("LC_TDFA_PAIR=$pf: in_agent", e.get("in_agent_MBps"), "columnar", e.get("in_agent_columnar_MBps"), "pipeline", e.get("pipeline",{}).get("fused_MBps"), "host", e.get("host_path_MBps"), "headline kernel ms", d["roofline"]["avg_kernel_ms"])
except Exception as ex:
    print("e2e $pf failed", ex); print(open("gpurun_out/r4_pair/e2e_$pf.err").read()[-800:])
try:
    c=json.loads(open("gpurun_out/r4_pair/cfg4_$pf.json").read().strip().splitlines()[-1])
    print("   config 4:", c["value"], c["unit"], "frac", c["roofline"]["frac"], str(c.get("config",{}))[:300])
except Exception as ex:
    print("cfg4 $pf failed", ex); print(open("gpurun_out/r4_pair/cfg4_$pf.err").read()[-800:])
PY
done
