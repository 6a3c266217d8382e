#!/bin/bash
# tools/gpu_round_final.sh ROUND: what a round's profiles/ files come from, in one gpurun call --
#   the full GPU suite, the driver's bench line (all configs), the headline kernel's evidence (trace + PMC), the Grok step's kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
RND=$1
cd $R; mkdir -p gpurun_out/final
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 | cut -c1-300 | tee gpurun_out/final/pytest_gpu.txt
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -2 | cut -c1-300
timeout 900 python bench.py > gpurun_out/final/bench_n1.json 2> gpurun_out/final/bench_n1.err; cut -c1-400 gpurun_out/final/bench_n1.json; tail -2 gpurun_out/final/bench_n1.err | cut -c1-300
timeout 200 python bench.py --regex B --steps 30 --warmup 3 --no-cpu-baseline --no-e2e --no-configs > gpurun_out/final/bench_regexB.json 2>/dev/null; cut -c1-200 gpurun_out/final/bench_regexB.json
bash tools/gpu_evidence.sh $RND
GPU_MAX_HW_QUEUES=16 bash tools/gpu_grok_profile.sh final_grok 16384 2>&1 | head -12 | cut -c1-250
