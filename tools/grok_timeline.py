#!/usr/bin/env python3
"""Timeline of the LAST Grok batch in a rocprofv3 --kernel-trace run of tools/grok_bench.py (rocpd sqlite): every dispatch
between the last grok_literal_index_kernel and the end, with its start relative to that kernel, duration, queue and grid.
Usage: grok_timeline.py DIR [max_rows]   (DIR holds r1_results.db)"""
import os
import sqlite3
import sys

d = sys.argv[1]
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
cur = sqlite3.connect(os.path.join(d, "r1_results.db")).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = 'select name, start, "end", grid_x, %s from kernels order by start' % (qcol or "0")
rows = list(cur.execute(sel))
starts = [i for i, r in enumerate(rows) if r[0].startswith(("grok_literal_index_kernel", "grok_literal_chunk_kernel", "grok_literal_lds_kernel", "grok_mask_fill_kernel"))]
if not starts:
    raise SystemExit("no Grok batch in the trace")
first = starts[-1]
t0 = rows[first][1]
print("# last batch: %d dispatches, %.3f ms from the literal pass to the end of the last kernel" % (
    len(rows) - first, (max(r[2] for r in rows[first:]) - t0) / 1e6))
print("%9s %9s %6s %8s  %s" % ("start_ms", "dur_ms", "queue", "grid/64", "kernel"))
busy = {}
for name, s, e, grid, q in rows[first:first + limit]:
    print("%9.3f %9.3f %6s %8d  %s" % ((s - t0) / 1e6, (e - s) / 1e6, q, grid // 64, name[:70]))
for name, s, e, grid, q in rows[first:]:
    k = name.split("(")[0][:60]
    b = busy.setdefault(k, [0, 0.0])
    b[0] += 1
    b[1] += (e - s) / 1e6
print("\n# per kernel in this batch")
for k, (c, t) in sorted(busy.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %5d %9.3f ms" % (k, c, t))
