#!/bin/bash
# tools/lab_pmc.sh TAG INPUTS: SQ counter passes of scratch/tdfa_lab on the GPU box -> gpurun_out/TAG.txt (per kernel averages)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; IN=$2
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
P1="SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAVE_CYCLES"
P2="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
P3="SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAIT_INST_VMEM"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  LAB_ONLY=1 timeout 300 rocprofv3 --pmc $P -d $R/gpurun_out/$TAG/p$i -o r --output-format csv -- $R/scratch/tdfa_lab $IN 3 > $R/gpurun_out/$TAG/run$i.log 2>&1
done
cd $R && python - <<PY > gpurun_out/$TAG.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/$TAG/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "tdfa" not in k: continue
        acc[k[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-26s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf gpurun_out/$TAG
cat gpurun_out/$TAG.txt
