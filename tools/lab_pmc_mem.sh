#!/bin/bash
# tools/lab_pmc_mem.sh TAG INPUTS [LABBIN]: memory-side counter passes (one counter group per pass, as the guide prescribes)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; IN=$2; BIN=${3:-$R/scratch/tdfa_lab}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  LAB_ONLY=1 timeout 300 rocprofv3 --pmc $P -d $R/gpurun_out/$TAG/p$i -o r --output-format csv -- $BIN $IN 3 > $R/gpurun_out/$TAG/run$i.log 2>&1
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
        print("   %-30s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
tail -3 gpurun_out/$TAG/run1.log >> gpurun_out/$TAG.txt
rm -rf gpurun_out/$TAG
cat gpurun_out/$TAG.txt
