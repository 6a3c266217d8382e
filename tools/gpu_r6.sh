#!/bin/bash
# tools/gpu_r6.sh WHAT...: round-6 measurement calls (one gpurun call each; outputs under gpurun_out/r6/)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r6
for what in "$@"; do
case $what in
suite)
  timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | cut -c1-300 | tee gpurun_out/r6/pytest_gpu.txt ;;
inagent_trace)
  # where a 16-thread in-agent Grok run spends its time: the host stages of every merged batch
  LC_GROK_TRACE=1 timeout 600 python tools/grok_inagent_bench.py --threads 1,16 --groups 20 > gpurun_out/r6/inagent.json 2> gpurun_out/r6/inagent_trace.txt
  cat gpurun_out/r6/inagent.json | cut -c1-400
  grep "grok host batch" gpurun_out/r6/inagent_trace.txt | tail -30 | cut -c1-250 ;;
inagent)
  timeout 600 python tools/grok_inagent_bench.py --threads 1,4,16,32 --groups 40 2>/dev/null | cut -c1-330 | tee gpurun_out/r6/inagent_$(date +%H%M).json ;;
a8ab)
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_pair1_tables.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
  for rep in 1 2; do for a8 in 1 0; do LC_TDFA_CMAPA8=$a8 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cmapa8=$a8', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_launched'])"; done; done
  LC_TDFA_CMAPA8=1 timeout 300 python bench.py --regex B --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('regexB a8=1', d['ms_per_step'], d['roofline']['frac'])" ;;
headline)
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
  for i in 1 2; do timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'])"; done ;;
lazy)
  timeout 900 python -m pytest tests/test_lazy_tdfa.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-300
  for lz in 1 0; do LC_LAZY_TDFA=$lz GPU_MAX_HW_QUEUES=16 timeout 900 python tools/grok_bench.py --lines 1000,16384,65536 --steps 10 --warmup 2 --cpu-sample-lines 1500 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('lazy=$lz', d['config']['workload'][-60:-20], d['ms_per_step'], 'ms', d['config']['batch'], d['config'].get('lazy_automata'))"; done ;;
kprof)
  # per-kernel times of a 1000-value Grok step (rocprofv3 kernel trace)
  cd /tmp && export TMPDIR=/tmp
  GPU_MAX_HW_QUEUES=16 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6/kprof -o r1 -- python $R/tools/grok_bench.py --lines ${KPROF_LINES:-1000} --steps 3 --warmup 2 --no-sequential-check --cpu-sample-lines 50 > $R/gpurun_out/r6/kprof.log 2>&1
  cd $R; python tools/grok_prof_summary.py gpurun_out/r6/kprof 2>&1 | head -30 | cut -c1-150; rm -rf gpurun_out/r6/kprof ;;
screenab)
  for rep in 1 2; do for sw in 1 0; do LC_GROK_SCREEN_WAVE=$sw GPU_MAX_HW_QUEUES=16 timeout 900 python tools/grok_bench.py --lines 1000,16384,65536 --steps 10 --warmup 2 --cpu-sample-lines 300 --no-sequential-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('screen_wave=$sw', d['config']['workload'][-60:-40], d['ms_per_step'], 'ms')"; done; done ;;
flatab)
  for rep in 1 2; do for fz in 1 0; do LC_GROK_FLAT=$fz GPU_MAX_HW_QUEUES=16 timeout 900 python tools/grok_bench.py --lines 1000,16384,65536 --steps 10 --warmup 2 --cpu-sample-lines 300 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('flat=$fz', d['config']['workload'][-60:-40], d['ms_per_step'], 'ms', d['config']['parity']['both_paths_agree_on_every_line'])"; done; done ;;
scaledab)
  for rep in 1 2; do for fz in 1 0; do LC_GROK_SCREEN_SCALED=$fz GPU_MAX_HW_QUEUES=16 timeout 900 python tools/grok_bench.py --lines 1000,16384,65536 --steps 10 --warmup 2 --cpu-sample-lines 300 --no-sequential-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('screen_scaled=$fz', d['config']['workload'][-60:-40], d['ms_per_step'], 'ms')"; done; done ;;
stageab)
  for rep in 1 2; do for fz in 1 0; do LC_GROK_FUSED_STAGE=$fz GPU_MAX_HW_QUEUES=16 timeout 900 python tools/grok_bench.py --lines 1000,16384 --steps 10 --warmup 2 --cpu-sample-lines 300 --no-sequential-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('fused_stage=$fz', d['config']['workload'][-60:-40], d['ms_per_step'], 'ms')"; done; done ;;
fusedab)
  for rep in 1 2; do for fz in 1 0; do LC_GROK_FUSED_ROUND0=$fz GPU_MAX_HW_QUEUES=16 timeout 900 python tools/grok_bench.py --lines 1000,16384,65536 --steps 10 --warmup 2 --cpu-sample-lines 300 --no-sequential-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('fused=$fz', d['config']['workload'][-60:-40], d['ms_per_step'], 'ms')"; done; done
  LC_GROK_TRACE=1 GPU_MAX_HW_QUEUES=16 timeout 300 python tools/grok_bench.py --lines 16384 --steps 2 --warmup 1 --no-sequential-check --cpu-sample-lines 50 2>&1 >/dev/null | grep "grok plan: n\|in one launch\|grok plan 2a: entry" | tail -12 | cut -c1-230 ;;
agroktests)
  timeout 1500 python -m pytest tests/test_gpu_grok.py tests/test_go_regex.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-300 | tee gpurun_out/r6/pytest_grok.txt ;;
esac
done
