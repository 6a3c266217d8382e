#!/usr/bin/env python3
"""Summary of a rocprofv3 --kernel-trace --stats run of tools/grok_bench.py (rocpd sqlite): time per kernel and the slowest
dispatches.  Usage: grok_prof_summary.py DIR   (DIR holds r1_results.db)"""
import os
import sqlite3
import sys

d = sys.argv[1]
cur = sqlite3.connect(os.path.join(d, "r1_results.db")).cursor()
print("## per kernel (top_kernels), times in ms")
print("%-58s %6s %10s %9s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "%"))
for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-58s %6d %10.3f %9.3f %6.2f" % (name[:58], calls, total / 1e3, avg / 1e3, pct))  # the view is in microseconds
try:
    rows = list(cur.execute('select name, grid_x, ("end" - start), lds_size from kernels order by ("end" - start) desc limit 12'))
    print("\n## slowest dispatches (values handed to the automaton = grid/256*4 for the NFA kernel, grid for the TDFA kernel)")
    print("%-46s %8s %9s %8s" % ("kernel", "grid/64", "ms", "lds_B"))
    for name, grid, dur, lds in rows:
        print("%-46s %8d %9.3f %8d" % (name[:46], grid // 64, dur / 1e6, lds))
except Exception as e:  # noqa: BLE001 -- the column set differs between rocprofv3 versions
    print("(no per-dispatch columns: %s)" % e)
