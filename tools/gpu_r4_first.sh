#!/bin/bash
# next round, first GPU call: what round 3 switched on or prepared without being able to time it in the driver's configuration.
#   1. the default bench line (one-stamp pair tables for regex A) against LC_TDFA_PAIR=0 (single-byte tables), kernel trace of the default
#   2. the full GPU suite
#   3. regex B on the pair table (forced), and the pair table for the STANDARD tables (LC_TDFA_PAIR=2 without LC_TDFA_COMPACT): parity + in-agent legs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r4first
O=gpurun_out/r4first
for pf in default 0; do
  if [ $pf = default ]; then unset LC_TDFA_PAIR; else export LC_TDFA_PAIR=$pf; fi
  timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-configs > $O/bench_pair_$pf.json 2> $O/bench_pair_$pf.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_pair_$pf.json").read())
    print("LC_TDFA_PAIR=$pf", "MB/s", d["value"], "ms/step", d["ms_per_step"], "kernel ms", d["roofline"]["avg_kernel_ms"], "frac", d["roofline"]["frac"], d["roofline"]["kernels_launched"])
except Exception as e:
    print("bench failed", e); print(open("$O/bench_pair_$pf.err").read()[-1500:])
PY
done
unset LC_TDFA_PAIR
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof_stats -o r1 -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-configs > $R/$O/stats.log 2>&1
cd $R && python tools/prof_summary.py $O > $O/tdfa_kernel_rocprofv3.txt 2>&1; rm -rf $O/prof_stats; head -6 $O/tdfa_kernel_rocprofv3.txt | cut -c1-140
for pct in 1 3; do echo "regex B, LC_TDFA_PAIR_DOUBLE_PCT=$pct"; LC_TDFA_PAIR_DOUBLE_PCT=$pct timeout 200 python bench.py --regex B --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | cut -c1-400; done
LC_TDFA_PAIR=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_processor.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300
LC_TDFA_PAIR=2 timeout 300 python bench.py --no-cpu-baseline --no-configs 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); e=d['end_to_end']
print('standard tables with pairs: in_agent', e['in_agent_MBps'], 'columnar', e['in_agent_columnar_MBps'], 'pipeline', e['pipeline']['fused_MBps'])"
