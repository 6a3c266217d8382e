#!/bin/bash
# rebuild with each flag set; report kernel ms and FETCH_SIZE (KB) per dispatch
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for f in "$@"; do
  LC_EXTRA_CXXFLAGS="$f" python -m loongcollector_amd.build --force > /dev/null 2>&1
  r=$(python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_kernel_ms'])")
  rm -rf /tmp/pf; (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1)
  fs=$(python - <<'PY'
import sqlite3,glob
p=glob.glob('/tmp/pf/**/r1_results.db', recursive=True)
cur=sqlite3.connect(p[0]).cursor()
v=[r[0] for r in cur.execute("select value from counters_collection where kernel_name like '%match_kernel%'")]
print(round(sum(v)/len(v)))
PY
)
  echo "flags=[$f] kernel_ms=$r FETCH_KB=$fs"
done
