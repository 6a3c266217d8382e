#!/bin/bash
# tools/gpu_evidence.sh ROUND [quick]: the evidence files of the headline kernel AS SHIPPED (defaults, no LC_* knob), in one gpurun call
#   gpurun_out/evidence/tdfa_kernel_rocprofv3.txt   rocprofv3 --kernel-trace --stats of the bench command  -> profiles/roundN_tdfa_kernel_rocprofv3.txt
#   gpurun_out/evidence/traffic.json                FETCH_SIZE / WRITE_SIZE passes (separate)               -> profiles/roundN_traffic.json
#   gpurun_out/evidence/sq_counters.txt             two SQ passes (LDS, VALU, waits)                         -> profiles/roundN_tdfa_sq_counters.txt
#   gpurun_out/evidence/bench_line.json             the bench line of the same command (HIP-event kernel time: must agree with the trace)
R=${GRAFT_REPO_ROOT:-/root/repo}
RND=$1
O=$R/gpurun_out/evidence
mkdir -p $O
BENCH="python $R/bench.py --no-cpu-baseline --no-e2e --no-configs"
cd $R
timeout 200 $BENCH --steps 40 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err || tail -5 $O/bench_line.err
python - <<PY
import json
d = json.loads(open("$O/bench_line.json").read())
r = d["roofline"]
print("bench line: MB/s", d["value"], "ms/step", d["ms_per_step"], "kernel ms", r["avg_kernel_ms"], "frac", r["frac"], r["kernels_launched"])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o r1 -- $BENCH --steps 40 --warmup 5 > $O/stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/prof_fetch -o r1 -- $BENCH --steps 5 --warmup 1 > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/prof_write -o r1 -- $BENCH --steps 5 --warmup 1 > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAVE_CYCLES -d $O/prof_pmc1 -o r1 -- $BENCH --steps 5 --warmup 1 > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d $O/prof_pmc2 -o r1 -- $BENCH --steps 5 --warmup 1 > $O/pmc2.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/prof_pmc3 -o r1 -- $BENCH --steps 5 --warmup 1 > $O/pmc3.log 2>&1
cd $R
python tools/make_traffic_json.py $O $O/traffic.json --round $RND
python tools/prof_summary.py $O > $O/summary_all.txt 2>&1
grep -v "rocclr\|at::native\|vectorized_elementwise\|Cijk" $O/summary_all.txt | sed -n '1,/== PMC pass/p' | grep -v "== PMC pass" > $O/tdfa_kernel_rocprofv3.txt
sed -n '/== PMC pass/,$p' $O/summary_all.txt > $O/sq_counters.txt
rm -rf $O/prof_*
head -8 $O/tdfa_kernel_rocprofv3.txt | cut -c1-150
cat $O/sq_counters.txt | cut -c1-150
# batch-size sweep (round 5): the same kernel on batches whose payload fits the Infinity Cache (256 MB) and on ones that do not -- if a
# line costs the same either way, what the kernel fetches twice through the fabric (the second 64-byte half of each 128-byte L2 line)
# is served in front of HBM and is not what bounds it  -> gpurun_out/evidence/size_sweep.txt
{
echo "# lines  payload_MiB  ms/step  ns/line  roofline.frac   (python bench.py --lines N --steps 20 --warmup 3, HBM-resident batch re-read every step)"
for N in 65536 131072 262144 524288 1048576 2097152; do
  timeout 300 $BENCH --lines $N --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); n = $N
print('%8d  %8.0f  %.4f  %.2f  %.4f' % (n, n * 513 / 2**20, d['ms_per_step'], d['ms_per_step'] * 1e6 / n, d['roofline']['frac']))"
done
} > $O/size_sweep.txt
cat $O/size_sweep.txt
# occupancy probe (round 6): the same launch with ONE resident workgroup per CU instead of two (LC_TDFA_EXTRA_LDS pads the workgroup's LDS) --
# how the time scales with the waves in flight says whether the kernel waits for a pipe or for latency  -> gpurun_out/evidence/occupancy_probe.txt
{
echo "# LC_TDFA_EXTRA_LDS  workgroups/CU  ms/step  roofline.frac   (python bench.py --steps 20 --warmup 3)"
for X in 0 8192; do
  LC_TDFA_EXTRA_LDS=$X timeout 300 $BENCH --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%8d  %d  %.4f  %.4f' % ($X, 1 if $X else 2, d['ms_per_step'], d['roofline']['frac']))"
done
} > $O/occupancy_probe.txt
cat $O/occupancy_probe.txt
