#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3final
timeout 420 python bench.py > gpurun_out/r3final/bench_n1.json 2> gpurun_out/r3final/bench_n1.err; cut -c1-300 gpurun_out/r3final/bench_n1.json; tail -3 gpurun_out/r3final/bench_n1.err
