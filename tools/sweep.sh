#!/bin/bash
# usage: tools/sweep.sh "<flags1>" "<flags2>" ...   -- rebuilds with each flag set and prints kernel ms
cd ${GRAFT_REPO_ROOT:-/root/repo}
for f in "$@"; do
  LC_EXTRA_CXXFLAGS="$f" python -m loongcollector_amd.build --force > /dev/null 2>&1
  r=$(python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_kernel_ms'], d['value'])")
  echo "flags=[$f] kernel_ms,value = $r"
done
