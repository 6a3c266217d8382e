#!/bin/bash
# round 3, evidence call: the driver's default bench line, the kernel trace of the same command, Grok at four batch sizes + its trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r3ev
O=gpurun_out/r3ev
timeout 420 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-600 $O/bench_n1.json; tail -2 $O/bench_n1.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof_stats -o r1 -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-configs > $R/$O/stats.log 2>&1
cd $R && python tools/prof_summary.py $O > $O/tdfa_kernel_rocprofv3.txt 2>&1; rm -rf $O/prof_stats; head -12 $O/tdfa_kernel_rocprofv3.txt | cut -c1-140
timeout 400 python tools/grok_bench.py --lines 1000,16384,65536,262144 --steps 5 --warmup 4 --cpu-sample-lines 200 > $O/grok_bench.json 2> $O/grok_bench.err
python - <<PY
import json
for l in open("$O/grok_bench.json"):
    d = json.loads(l); c = d["config"]
    print(c["workload"].split(", ")[2][:14], d["value"], "lines/s", d["ms_per_step"], "ms", c.get("parity"), d.get("roofline", {}).get("frac"))
PY
tail -2 $O/grok_bench.err
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/grok_prof -o r1 -- python $R/tools/grok_bench.py --lines 16384 --steps 3 --warmup 2 --no-sequential-check --cpu-sample-lines 100 > $R/$O/grok_prof.log 2>&1
cd $R && python tools/grok_prof_summary.py $O/grok_prof > $O/grok_rocprofv3.txt 2>&1; rm -rf $O/grok_prof; head -16 $O/grok_rocprofv3.txt | cut -c1-140
