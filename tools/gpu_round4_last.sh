#!/bin/bash
# tools/gpu_round4_last.sh: the round's last GPU minutes, after the host-side work of its second half (processor stitch, event model,
# faster tagged-DFA construction, seven more anchored Grok automata): the full GPU suite, the in-agent shape natively, the Grok step.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/last scratch
timeout 240 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | cut -c1-300 | tee gpurun_out/last/pytest_gpu.txt
g++ -O2 -std=c++17 -I include tools/inagent_bench.cpp -o scratch/inagent_bench -L loongcollector_amd/lib -llc_regex_gpu -lpthread &&
  LD_LIBRARY_PATH=loongcollector_amd/lib:/opt/rocm/lib timeout 60 scratch/inagent_bench 256000 1000 1 16 2>&1 | tee gpurun_out/last/inagent.txt
timeout 150 python tools/grok_bench.py --lines 1000,16384 --steps 10 --warmup 3 > gpurun_out/last/grok.json 2> gpurun_out/last/grok.err
cut -c1-330 gpurun_out/last/grok.json; tail -3 gpurun_out/last/grok.err | cut -c1-300
