#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of the headline kernel -> profiles/roundN_traffic.json (what bench.py quotes as roofline.traffic).

    tools/make_traffic_json.py DIR OUT.json [--round N]

DIR holds prof_fetch/ and prof_write/ (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, SEPARATE passes of
`bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --no-configs`, as MI355X_MICROARCH.md prescribes) and bench_line.json (the JSON
line of the same command, for the kernel's name and the workload).  Correction (same guide, HBM section): on gfx950 FETCH_SIZE
tallies the 128-byte requests of 16 B/lane coalesced reads at 64 B -> doubled; WRITE_SIZE is uncalibrated there -> quoted raw, and
beside it the lower bound (raw fetch + write)."""
import collections
import json
import os
import sqlite3
import sys


def per_dispatch(dbdir, counter):
    p = os.path.join(dbdir, "r1_results.db")
    cur = sqlite3.connect(p).cursor()
    acc = collections.defaultdict(list)
    for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
        if c == counter:
            acc[k].append(v)
    return acc


def main():
    d, out = sys.argv[1], sys.argv[2]
    rnd = sys.argv[sys.argv.index("--round") + 1] if "--round" in sys.argv else "?"
    line = json.loads(open(os.path.join(d, "bench_line.json")).read().strip().splitlines()[-1])
    fetch = per_dispatch(os.path.join(d, "prof_fetch"), "FETCH_SIZE")
    write = per_dispatch(os.path.join(d, "prof_write"), "WRITE_SIZE")
    # the dominant kernel = the tdfa_stream_kernel instantiation with the largest fetch; the mop-up launch is listed beside it
    rows = []
    for k in fetch:
        if "tdfa" not in k and "nfa" not in k:
            continue
        f = fetch[k]
        w = write.get(k, [0.0])
        rows.append({"kernel": k[:160], "dispatches": len(f), "fetch_size_kb_raw": round(sum(f) / len(f), 2),
                     "write_size_kb_raw": round(sum(w) / len(w), 2)})
    rows.sort(key=lambda r: -r["fetch_size_kb_raw"])
    main_row = rows[0]
    fetch_kb = sum(r["fetch_size_kb_raw"] for r in rows)
    write_kb = sum(r["write_size_kb_raw"] for r in rows)
    cfg = line["config"]
    algo = line["roofline"]["algorithmic_bytes_per_launch"]
    hbm = int(fetch_kb * 1024 * 2 + write_kb * 1024)
    j = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `bench.py --steps 5 --warmup 1 --no-cpu-baseline "
                   "--no-e2e --no-configs` (tools/gpu_evidence.sh, round %s)" % rnd,
         "lines": cfg["lines_per_batch"], "regex": "A" if "regex A" in cfg["workload"] else "B", "line_bytes": 512, "engine": cfg["engine"],
         "kernel": main_row["kernel"], "kernels_launched": line["roofline"]["kernels_launched"], "per_kernel": rows,
         "fetch_size_kb_raw": round(fetch_kb, 2), "write_size_kb_raw": round(write_kb, 2),
         "correction": "FETCH_SIZE x2 (gfx950: the counter tallies 128-B requests at 64 B for 16 B/lane coalesced reads, "
                       "MI355X_MICROARCH.md HBM section); WRITE_SIZE uncorrected (uncalibrated there)",
         "hbm_bytes_per_launch": hbm, "lower_bound_bytes_per_launch": int((fetch_kb + write_kb) * 1024),
         "algorithmic_bytes_per_launch": algo, "ratio_to_algorithmic": round(hbm / algo, 3),
         "compact_tables": cfg.get("compact_tables")}
    with open(out, "w") as f:
        json.dump(j, f, indent=1)
    print(json.dumps({k: j[k] for k in ("kernel", "fetch_size_kb_raw", "write_size_kb_raw", "hbm_bytes_per_launch", "ratio_to_algorithmic")}))


if __name__ == "__main__":
    main()
