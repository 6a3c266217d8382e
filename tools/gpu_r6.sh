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
aheadab)
  # the remainder screens queued ahead of the host's read of round 0's counts, on / off (parity gates of grok_bench.py included)
  for rep in 1 2; do for ah in 1 0; do LC_GROK_REMAINDER_AHEAD=$ah GPU_MAX_HW_QUEUES=16 timeout 900 python tools/grok_bench.py --lines 1000,16384,65536 --steps 10 --warmup 8 --cpu-sample-lines 300 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ahead=$ah', d['config']['workload'][-60:-40], d['ms_per_step'], 'ms', d['config'].get('parity_ok', d['config'].get('parity')))"; done; done | tee gpurun_out/r6/aheadab.txt ;;
bigab)
  # the 70-130 KB screens staged into a CU's whole LDS (phase 1: LC_GROK_BIG_SCREENS; remainder screens: LC_GROK_BIG_REMAINDER), measured again
  for rep in 1 2; do for cfg in "0 0" "1 0" "0 1" "1 1"; do set -- $cfg; LC_GROK_BIG_SCREENS=$1 LC_GROK_BIG_REMAINDER=$2 GPU_MAX_HW_QUEUES=16 timeout 900 python tools/grok_bench.py --lines 1000,16384,65536 --steps 10 --warmup 8 --cpu-sample-lines 100 --no-sequential-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('big_screens=$1 big_remainder=$2', d['config']['workload'][-60:-40], d['ms_per_step'], 'ms')"; done; done | tee gpurun_out/r6/bigab.txt ;;
prefixab)
  # the prefix screen instead of a relaxed screen that is not staged into LDS (LC_GROK_SCREEN_PREFIX 0 / 1 / 2), with and without the big screens staged
  for rep in 1 2; do for cfg in "0 0" "2 0" "2 1" "1 0"; do set -- $cfg; LC_GROK_SCREEN_PREFIX=$1 LC_GROK_BIG_SCREENS=$2 LC_GROK_BIG_REMAINDER=$2 GPU_MAX_HW_QUEUES=16 timeout 900 python tools/grok_bench.py --lines 1000,16384,65536 --steps 10 --warmup 8 --cpu-sample-lines 300 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('screen_prefix=$1 big=$2', d['config']['workload'][-60:-40], d['ms_per_step'], 'ms', d['config']['batch']['pairs'], d['config']['parity']['both_paths_agree_on_every_line'])"; done; done | tee gpurun_out/r6/prefixab.txt ;;
relaxab)
  # relaxed screens relaxed further until their table can be staged into LDS (LC_GROK_RELAX_PREFER_BYTES), with and without the big screens staged
  for rep in 1 2; do for cfg in "0 0" "45056 0" "153600 1" "45056 1"; do set -- $cfg; LC_GROK_RELAX_PREFER_BYTES=$1 LC_GROK_BIG_SCREENS=$2 LC_GROK_BIG_REMAINDER=$2 GPU_MAX_HW_QUEUES=16 timeout 900 python tools/grok_bench.py --lines 1000,16384,65536 --steps 10 --warmup 8 --cpu-sample-lines 300 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('relax_prefer=$1 big=$2', d['config']['workload'][-60:-40], d['ms_per_step'], 'ms', d['config']['batch']['pairs'], d['config']['parity']['both_paths_agree_on_every_line'])"; done; done | tee gpurun_out/r6/relaxab.txt ;;
persistab)
  # the headline kernel with persistent wavefronts (default) and with one workgroup per block (LC_TDFA_PERSIST=0): parity first, then A/B
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pair1_tables.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
  for rep in 1 2; do for pz in 1 0; do LC_TDFA_PERSIST=$pz timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('persist=$pz', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_launched'], d['config'].get('parity'))"; done; done | tee gpurun_out/r6/persistab.txt
  for L in 2097152 524288 300000; do LC_TDFA_PERSIST=1 timeout 300 python bench.py --lines $L --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('persist=1 lines $L', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_launched'])"; done | tee -a gpurun_out/r6/persistab.txt
  LC_TDFA_PERSIST=1 timeout 300 python bench.py --regex B --steps 30 --warmup 5 --no-cpu-baseline --no-e2e --no-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('regexB persist=1', d['ms_per_step'], d['roofline']['frac'])" | tee -a gpurun_out/r6/persistab.txt ;;
groktests)
  timeout 1500 python -m pytest tests/test_gpu_grok.py tests/test_go_regex.py -m gpu -q -x 2>&1 | tail -5 | cut -c1-300 | tee gpurun_out/r6/pytest_grok.txt ;;
esac
done
