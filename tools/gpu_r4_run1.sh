#!/bin/bash
# round 4, call 1: the shipped headline kernel's evidence, the new forced one-stamp tests, the full GPU suite, pair table A/B runs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "one_stamp" 2>&1 | tail -15 | cut -c1-400
bash tools/gpu_evidence.sh 4
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 | cut -c1-300 | tee gpurun_out/r4_pytest_gpu.txt
for pf in 0; do
  LC_TDFA_PAIR=$pf timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('LC_TDFA_PAIR=$pf', 'MB/s', d['value'], 'kernel ms', r['avg_kernel_ms'], 'frac', r['frac'], r['kernels_launched'])"
done
for pct in 1 3; do
  LC_TDFA_PAIR_DOUBLE_PCT=$pct timeout 200 python bench.py --regex B --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('regex B, DOUBLE_PCT=$pct', 'MB/s', d['value'], 'kernel ms', r['avg_kernel_ms'], 'frac', r['frac'], r['kernels_launched'])"
done
