#!/bin/bash
# round 3: Grok on the GPU box -- parity tests of both paths, then configs[2] at several batch sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
if [ "$1" != "notest" ]; then
timeout 1500 python -m pytest tests/test_gpu_grok.py -x -q 2>&1 | tail -15 > gpurun_out/pytest_grok.log; tail -15 gpurun_out/pytest_grok.log
fi
LC_GROK_TRACE=1 timeout 900 python tools/grok_bench.py --lines ${GROK_LINES:-1000,16384,65536,262144,1048576} --steps 5 --warmup 4 > gpurun_out/grok_bench.json 2> gpurun_out/grok_bench.err
cat gpurun_out/grok_bench.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['config']['workload'].split(', ')[2][:14], d['value'], 'lines/s', d['ms_per_step'], 'ms', d['config']['batch'], d['config']['parity'])
"
grep "grok plan" gpurun_out/grok_bench.err | awk '{k=$4; last[k]=$0} END {for (k in last) print last[k]}'
tail -3 gpurun_out/grok_bench.err
