#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs: per-kernel time stats and PMC averages.  Usage: tools_prof_summary.py DIR"""
import collections
import os
import sqlite3
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
p = os.path.join(d, "prof_stats", "r1_results.db")
if os.path.exists(p):
    cur = sqlite3.connect(p).cursor()
    print("== kernel-trace stats (rocprofv3 --kernel-trace --stats): name, calls, total_us, avg_us, pct")
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-60s %6d %12.1f %10.2f %6.2f" % (name[:60], calls, total / 1e3 if total > 1e6 else total, avg / 1e3 if avg > 1e5 else avg, pct))
    try:
        rows = list(cur.execute("select name, vgpr_count, sgpr_count, lds_size, workgroup_x, grid_x from kernels group by name"))
        for r in rows:
            print("   resources:", r[0][:50], "vgpr", r[1], "sgpr", r[2], "lds", r[3], "wg", r[4], "grid", r[5])
    except Exception as e:  # noqa
        print("   (no resource columns)", e)
for sub in sorted(os.listdir(d)):
    p = os.path.join(d, sub, "r1_results.db")
    if not sub.startswith("prof_") or sub == "prof_stats" or not os.path.exists(p):
        continue
    cur = sqlite3.connect(p).cursor()
    agg = collections.defaultdict(list)
    for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        agg[(k[:48], c)].append(v)
    print("== PMC pass %s (per-dispatch averages)" % sub)
    for (k, c), v in sorted(agg.items()):
        if "match_kernel" in k or "stream_kernel" in k or "split" in k:
            print("%-50s %-22s n=%3d avg=%16.2f" % (k, c, len(v), sum(v) / len(v)))
