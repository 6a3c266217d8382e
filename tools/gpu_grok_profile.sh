#!/bin/bash
# tools/gpu_grok_profile.sh TAG [LINES]: configs[2] on the GPU box -- bench lines at 1000 / 16 Ki / 64 Ki values, the phase trace of a step
# (LC_GROK_TRACE), and the rocprofv3 kernel trace + timeline of a LINES-value step (default 16384)  -> gpurun_out/TAG/
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; LINES=${2:-16384}
O=$R/gpurun_out/$TAG
mkdir -p $O; cd $R
timeout 600 python tools/grok_bench.py --lines 1000,16384,65536 --steps 5 --warmup 8 --no-sequential-check --cpu-sample-lines 200 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
for l in open("$O/bench.json"):
    if not l.startswith("{"): continue
    d = json.loads(l)
    print("grok %s  %.3f ms/step  %.0f lines/s  frac %s  hit %s  batch %s" % (d["config"]["workload"].split("patterns")[1][:60], d["ms_per_step"], d["value"], d["roofline"]["frac"], d["config"]["patterns_hit"], d["config"]["batch"]))
PY
tail -3 $O/bench.err | cut -c1-300
LC_GROK_TRACE=1 timeout 300 python tools/grok_bench.py --lines $LINES --steps 1 --warmup 8 --no-sequential-check --cpu-sample-lines 50 2>&1 >/dev/null | grep "grok plan" > $O/trace_all.txt
# (the last batch's lines: from the last "2a" block on)
python - <<PY > $O/trace.txt
lines = open("$O/trace_all.txt").read().splitlines()
ends = [i for i, l in enumerate(lines) if l.startswith("grok plan: n ")]
start = ends[-2] + 1 if len(ends) >= 2 else 0
print("\n".join(lines[start:]))
PY
rm -f $O/trace_all.txt; tail -2 $O/trace.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/grok_prof -o r1 -- python $R/tools/grok_bench.py --lines $LINES --steps 3 --warmup 8 --no-sequential-check --cpu-sample-lines 50 > $O/grok_prof.log 2>&1
cd $R
python tools/grok_prof_summary.py $O/grok_prof > $O/grok_rocprofv3.txt 2>&1
python tools/grok_timeline.py $O/grok_prof 600 > $O/grok_timeline.txt 2>&1
rm -rf $O/grok_prof
head -48 $O/grok_rocprofv3.txt | cut -c1-140
head -3 $O/grok_timeline.txt
