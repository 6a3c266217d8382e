#!/bin/bash
# tools/gpu_r5.sh STAGE: the GPU calls of round 5 (profiles/round5_* come from them; profiles/ROUND5.md indexes the files).
#   first    full GPU suite, configs[3] (packed multi-pipeline launch) bench line + kernel trace, the Grok step's profile
#   multi    configs[3] only
#   second   full GPU suite, configs[3], bench.py's end_to_end block (in-agent legs incl. the reference-shaped build)
#   grok     Grok GPU tests + the Grok step's profile (bench lines, phase trace, kernel trace, timeline)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/r5_$1; mkdir -p $O scratch
multi() {
  timeout 200 python bench.py --config 4 --steps 50 --warmup 5 > $O/multi.json 2> $O/multi.err; cut -c1-900 $O/multi.json; tail -2 $O/multi.err | cut -c1-300
  (cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/multi_prof -o r1 -- python $R/bench.py --config 4 --steps 50 --warmup 5 > $R/$O/multi_prof.log 2>&1)
  python tools/grok_prof_summary.py $O/multi_prof > $O/multi_kernel_rocprofv3.txt 2>&1; rm -rf $O/multi_prof
  head -12 $O/multi_kernel_rocprofv3.txt | cut -c1-200
}
case "$1" in
first)
  timeout 400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | cut -c1-300 | tee $O/pytest_gpu.txt
  multi
  GPU_MAX_HW_QUEUES=16 bash tools/gpu_grok_profile.sh r5_first_grok 16384 2>&1 | head -12 | cut -c1-250 ;;
multi) multi ;;
second)
  timeout 500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | cut -c1-300 | tee $O/pytest_gpu.txt
  multi
  timeout 400 python bench.py --lines 262144 --steps 3 --warmup 1 --no-cpu-baseline --no-configs > $O/bench_e2e.json 2> $O/bench_e2e.err
  echo "bench rc $?"; tail -c 6000 $O/bench_e2e.json | tr ',' '\n' | grep -A5 '"in_agent' | head -60; tail -3 $O/bench_e2e.err | cut -c1-300 ;;
grok)
  timeout 120 python -m pytest tests/test_gpu_grok.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-200 | tee $O/pytest_grok.txt
  GPU_MAX_HW_QUEUES=16 bash tools/gpu_grok_profile.sh r5_grok_$2 16384 2>&1 | head -14 | cut -c1-250 ;;
plan)
  # round 5's Grok plan: the knobs one by one (LC_GROK_WIDE_FIRST / BREADTH / EARLY_ROUNDS / REMAINDER_LITERAL / BIG_SCREENS), then the profile
  export LC_TABLE_CACHE_DIR=/tmp/lctab GPU_MAX_HW_QUEUES=16
  timeout 700 python -m pytest tests/test_gpu_grok.py tests/test_gpu_decide.py -m gpu -q -x 2>&1 | tail -4 | cut -c1-300 | tee $O/pytest_grok.txt
  timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "nfa or wide or overflow or golden or search or atomic or lookaround or resumed or doomed or run_capt or wave or small_automata or global_memory" 2>&1 | tail -4 | cut -c1-300 | tee $O/pytest_parity.txt
  ab() {
    name=$1; shift
    env "$@" timeout 200 python tools/grok_bench.py --lines 1000,16384 --steps 10 --warmup 8 --no-sequential-check --cpu-sample-lines 100 > $O/ab_$name.json 2> $O/ab_$name.err
    python - <<PY
import json
out = []
for l in open("$O/ab_$name.json"):
    if l.startswith("{"):
        d = json.loads(l); out.append("%s lines %.3f ms" % (d["config"]["workload"].split("), ")[1].split(" lines")[0], d["ms_per_step"]))
print("%-12s" % "$name", " | ".join(out))
PY
  }
  ab default
  ab no_wide LC_GROK_WIDE_FIRST=0
  ab no_breadth LC_GROK_BREADTH=0
  ab no_early LC_GROK_EARLY_ROUNDS=0
  ab no_remlit LC_GROK_REMAINDER_LITERAL=0
  ab no_bound LC_GROK_BOUND=0
  ab no_won LC_GROK_REMAINDER_WON=0
  ab slice512 LC_GROK_SLICE=512
  ab no_inchain LC_GROK_REMAINDER_INCHAIN=0
  ab no_poststream LC_GROK_POST_IN_STREAM=0
  ab no_lazy LC_GROK_LAZY_SYNC3=0
  ab all_off LC_GROK_WIDE_FIRST=0 LC_GROK_BREADTH=0 LC_GROK_EARLY_ROUNDS=0 LC_GROK_REMAINDER_LITERAL=0 LC_GROK_BOUND=0 LC_GROK_REMAINDER_WON=0 LC_GROK_SLICE=512 LC_GROK_REMAINDER_INCHAIN=0 LC_GROK_POST_IN_STREAM=0 LC_GROK_LAZY_SYNC3=0
  bash tools/gpu_grok_profile.sh r5_plan 16384 2>&1 | head -14 | cut -c1-250 ;;
big)
  # the 64 Ki-value batch: the phase trace, and the small-batch path forced onto it (LC_GROK_SMALL_BATCH)
  export LC_TABLE_CACHE_DIR=/tmp/lctab GPU_MAX_HW_QUEUES=16
  for SB in 32768 65536; do
    echo "## LC_GROK_SMALL_BATCH=$SB"
    LC_GROK_SMALL_BATCH=$SB timeout 300 python tools/grok_bench.py --lines 65536 --steps 10 --warmup 8 --no-sequential-check --cpu-sample-lines 100 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  %.3f ms/step  %s' % (d['ms_per_step'], d['config']['batch']))"
    LC_GROK_SMALL_BATCH=$SB LC_GROK_TRACE=1 timeout 300 python tools/grok_bench.py --lines 65536 --steps 1 --warmup 8 --no-sequential-check --cpu-sample-lines 50 2>&1 >/dev/null | grep "grok plan" | tail -60 | cut -c1-170 > $O/trace_$SB.txt
    grep "plan 2a" $O/trace_$SB.txt | sort -t'|' -k2 | awk -F'measured round 0 ' '{print $2" | "$1}' | sort -g -r | head -8 | cut -c1-150
    tail -1 $O/trace_$SB.txt
  done
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o r1 -- python $R/tools/grok_bench.py --lines 65536 --steps 3 --warmup 8 --no-sequential-check --cpu-sample-lines 50 > $R/$O/prof.log 2>&1
  cd $R; python tools/grok_timeline.py $O/prof 900 > $O/grok_timeline_64k.txt 2>&1; python tools/grok_prof_summary.py $O/prof > $O/grok_rocprofv3_64k.txt 2>&1; rm -rf $O/prof
  head -30 $O/grok_rocprofv3_64k.txt | cut -c1-140 ;;
big2)
  export LC_TABLE_CACHE_DIR=/tmp/lctab GPU_MAX_HW_QUEUES=16
  run() { echo "## $*"; env "$@" timeout 300 python tools/grok_bench.py --lines ${LINES:-65536} --steps 10 --warmup 8 --no-sequential-check --cpu-sample-lines 100 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  %.3f ms/step' % d['ms_per_step'])"; }
  run LC_GROK_SMALL_BATCH=65536
  run LC_GROK_SMALL_BATCH=65536 LC_GROK_SLICE=512
  run LC_GROK_SMALL_BATCH=65536 LC_GROK_SLICE=1024
  run LC_GROK_SMALL_BATCH=65536 LC_GROK_WIDE_FIRST=0
  run LC_GROK_SMALL_BATCH=65536 LC_TDFA_WAVE_MAX=32768
  LINES=131072 run LC_GROK_SMALL_BATCH=65536
  LINES=131072 run LC_GROK_SMALL_BATCH=131072
  LINES=262144 run LC_GROK_SMALL_BATCH=65536
  LINES=262144 run LC_GROK_SMALL_BATCH=262144
  cd /tmp && export TMPDIR=/tmp
  LC_GROK_SMALL_BATCH=65536 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o r1 -- python $R/tools/grok_bench.py --lines 65536 --steps 3 --warmup 8 --no-sequential-check --cpu-sample-lines 50 > $R/$O/prof.log 2>&1
  cd $R; python tools/grok_timeline.py $O/prof 900 > $O/grok_timeline_64k.txt 2>&1; python tools/grok_prof_summary.py $O/prof > $O/grok_rocprofv3_64k.txt 2>&1; rm -rf $O/prof
  head -16 $O/grok_rocprofv3_64k.txt | cut -c1-140 ;;
ab64)
  # the plan's knobs on a 64 Ki-value batch
  export LC_TABLE_CACHE_DIR=/tmp/lctab GPU_MAX_HW_QUEUES=16
  run() { printf "%-44s" "$*"; env "$@" timeout 300 python tools/grok_bench.py --lines 65536 --steps 10 --warmup 8 --no-sequential-check --cpu-sample-lines 100 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  %.3f ms/step' % d['ms_per_step'])"; }
  run X=default
  run LC_GROK_POST_IN_STREAM=0
  run LC_GROK_BREADTH=0
  run LC_GROK_REMAINDER_INCHAIN=0
  run LC_GROK_LAZY_SYNC3=0
  run LC_GROK_BOUND=0
  run LC_GROK_STREAMS=8
  run LC_TDFA_WAVE_MAX=16384
  run LC_NFA_GLOBAL_KB=160
  run X=default ;;
classlists)
  # the thread-list kernels on follow lists by byte class (device_tables.h NF_OFF_CSTART) against the whole follow lists
  export LC_TABLE_CACHE_DIR=/tmp/lctab GPU_MAX_HW_QUEUES=16
  timeout 700 python -m pytest tests/test_gpu_grok.py tests/test_gpu_decide.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
  timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_processor.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-300
  for V in "X=classlists" "LC_NFA_NO_CLASS_LISTS=1"; do
    echo "## $V"
    env $V LC_BENCH_ANCHORED=1 LC_BENCH_ENGINE=nfa LC_BENCH_REPS=5 timeout 200 python tools/grok_pattern_bench.py '%{CISCOFW313005}' '%{CISCOFW302013_302014_302015_302016}' '%{HAPROXYHTTP}' 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "engine\|with literal\|>= 2 KiB" | cut -c1-150
    env $V timeout 300 python tools/grok_bench.py --lines 1000,16384,65536 --steps 10 --warmup 8 --no-sequential-check --cpu-sample-lines 100 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  grok %s: %.3f ms/step' % (d['config']['workload'].split('), ')[1].split(' lines')[0], d['ms_per_step']))"
  done ;;
*) echo "usage: $0 first|multi|grok|plan|big|big2|ab64|classlists"; exit 2 ;;
esac
