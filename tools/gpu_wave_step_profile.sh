#!/bin/bash
# tools/gpu_wave_step_profile.sh TAG [PATTERN]: tdfa_wave_kernel on ONE Grok entry (default %{CISCOFW105003}: a search with a GREEDYDATA in
# front of a literal -- every space of the free text behind starts an attempt) over the values of the configs[2] corpus that carry its
# literal: time per launch with the tables in L2 and in LDS, then SQ counter passes of the kernel  -> gpurun_out/TAG.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; PAT=${2:-%{CISCOFW105003\}}
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export LC_BENCH_PREFER_WAVE=1 LC_BENCH_REPS=5
{
echo "## time per launch, tables in LDS (LC_TDFA_WAVE_LDS_TRANS=1)"
LC_TDFA_WAVE_LDS_TRANS=1 timeout 200 python tools/grok_pattern_bench.py "$PAT" 2>&1 | grep -v Warning | tail -7
echo "## time per launch, tables in L2 (default)"
timeout 200 python tools/grok_pattern_bench.py "$PAT" 2>&1 | grep -v Warning | tail -7
} > $O.txt
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
P2="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  LC_BENCH_REPS=1 timeout 300 rocprofv3 --pmc $P -d $O/p$i -o r --output-format csv -- python $R/tools/grok_pattern_bench.py "$PAT" > $O/run$i.log 2>&1
done
cd $R && python - <<PY >> $O.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "tdfa_wave" not in k: continue
        acc[k[:40] + " grid=" + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("## SQ counters per dispatch (tables in L2: the default)")
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-26s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf $O
cat $O.txt | cut -c1-200
