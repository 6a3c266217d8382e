#!/bin/bash
# tools/gpu_late.sh STAGE: the three short GPU calls of late round 4 (after the host-side work of its second half), kept as one
# script.  profiles/round4_late_* come from them (profiles/ROUND4.md).
#   suite   the full GPU suite, the native in-agent bench, the Grok steps (1000 / 16 Ki values)
#   agent   bench.py's end_to_end block as the driver's line runs it (bounded-window in-agent legs), the processors' GPU tests,
#           the native in-agent bench (all groups alive / 16 groups alive, minor faults per group)
#   grok    the Grok GPU tests and the 16 Ki step
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; O=gpurun_out/late_$1; mkdir -p $O scratch
inagent() {
  g++ -O2 -std=c++17 -I include tools/inagent_bench.cpp -o scratch/inagent_bench -L loongcollector_amd/lib -llc_regex_gpu -lpthread &&
    LD_LIBRARY_PATH=loongcollector_amd/lib:/opt/rocm/lib timeout 60 scratch/inagent_bench 256000 1000 1 16 2>&1 | tee $O/inagent.txt
}
case "$1" in
suite)
  timeout 240 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | cut -c1-300 | tee $O/pytest_gpu.txt
  inagent
  timeout 150 python tools/grok_bench.py --lines 1000,16384 --steps 10 --warmup 3 > $O/grok.json 2> $O/grok.err
  cut -c1-330 $O/grok.json; tail -3 $O/grok.err | cut -c1-300 ;;
agent)
  timeout 150 python bench.py --lines 262144 --steps 3 --warmup 1 --no-cpu-baseline --no-configs > $O/bench_e2e.json 2> $O/bench_e2e.err
  echo "bench rc $?"; tail -c 3000 $O/bench_e2e.json | tr ',' '\n' | grep -A4 '"in_agent' | head -24; tail -3 $O/bench_e2e.err | cut -c1-300
  timeout 60 python -m pytest tests/test_gpu_processor.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-200 | tee $O/pytest_processors.txt
  inagent ;;
grok)
  timeout 80 python -m pytest tests/test_gpu_grok.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-200 | tee $O/pytest_grok.txt
  timeout 70 python tools/grok_bench.py --lines 16384 --steps 10 --warmup 3 > $O/grok.json 2> $O/grok.err
  cut -c1-330 $O/grok.json; tail -2 $O/grok.err | cut -c1-200 ;;
*) echo "usage: $0 suite|agent|grok"; exit 2 ;;
esac
