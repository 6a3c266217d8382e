#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_gpu_grok.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for st in 1 0; do
if [ $st = 0 ]; then export LC_TDFA_L2_NO_STAGE=1; else unset LC_TDFA_L2_NO_STAGE; fi
echo "== programs staged in LDS: $st"
timeout 600 python tools/grok_bench.py --lines 1000,16384,65536 --steps 5 --warmup 4 --no-sequential-check --cpu-sample-lines 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['config']['workload'].split(', ')[2][:14], d['value'], 'lines/s', d['ms_per_step'], 'ms')
"
done
