#!/bin/bash
# tools/gpu_nfa_step_profile.sh TAG [PATTERN]: the thread-list engine's byte step on ONE Grok entry (default %{HAPROXYHTTP}, anchored
# search, the values of the configs[2] corpus that carry its literal) -- time per launch with the program read from L2 and with the
# program staged in LDS (LC_NFA_SMALL_BATCH), then SQ counter passes of the kernel  -> gpurun_out/TAG.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; PAT=${2:-%{HAPROXYHTTP\}}
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export LC_BENCH_ANCHORED=1 LC_BENCH_ENGINE=nfa LC_BENCH_REPS=5
{
echo "## time per launch, program in L2 (default)"
timeout 200 python tools/grok_pattern_bench.py "$PAT" 2>&1 | grep -v Warning | tail -6
echo "## time per launch, program staged in LDS (LC_NFA_STAGE_SMALL=1)"
LC_NFA_STAGE_SMALL=1 timeout 200 python tools/grok_pattern_bench.py "$PAT" 2>&1 | grep -v Warning | tail -6
} > $O.txt
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
P2="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
i=0
for MODE in l2 lds; do
for P in "$P1" "$P2"; do
  i=$((i+1))
  if [ $MODE = lds ]; then export LC_NFA_STAGE_SMALL=1; else unset LC_NFA_STAGE_SMALL; fi
  LC_BENCH_REPS=1 timeout 300 rocprofv3 --pmc $P -d $O/p$i -o r --output-format csv -- python $R/tools/grok_pattern_bench.py "$PAT" > $O/run$i.log 2>&1
  echo $MODE > $O/p$i/mode.txt 2>/dev/null
done
done
cd $R && python - <<PY >> $O.txt
import csv, glob, collections, os
for mode in ("l2", "lds"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob("$O/p*")):
        if open(os.path.join(d, "mode.txt")).read().strip() != mode: continue
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "")
                if "nfa_match_kernel" not in k and "nfa_wide" not in k: continue
                acc[k[:64] + " grid=" + row.get("Grid_Size", "?")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("## SQ counters per dispatch, program in %s" % mode)
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print("   %-26s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf $O
cat $O.txt | cut -c1-200
