#!/bin/bash
# tools/gpu_screen_split.sh [LINES] [screen|rem]: diagnosis -- every screen of configs[2] (LC_GROK_SCREEN_SPLIT), or every entry's remainder
# screens (LC_GROK_REM_SPLIT), in a launch of its own, timed by a kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
LINES=${1:-16384}; WHAT=${2:-screen}
O=$R/gpurun_out/${WHAT}_split; mkdir -p $O; cd $R
LC_GROK_TRACE=1 timeout 300 python tools/grok_bench.py --lines $LINES --steps 1 --warmup 8 --no-sequential-check --cpu-sample-lines 50 2> $O/trace_all.txt >/dev/null
grep "grok screen of entry" $O/trace_all.txt | sort -u > $O/screens.txt
grep "grok plan 2d" $O/trace_all.txt | tail -46 > $O/in_play.txt; rm -f $O/trace_all.txt
cd /tmp && export TMPDIR=/tmp
if [ $WHAT = rem ]; then export LC_GROK_REM_SPLIT=1; else export LC_GROK_SCREEN_SPLIT=1; fi
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r1 -- python $R/tools/grok_bench.py --lines $LINES --steps 3 --warmup 8 --no-sequential-check --cpu-sample-lines 50 > $O/prof.log 2>&1
cd $R
python tools/grok_timeline.py $O/prof 600 > $O/timeline.txt 2>&1
rm -rf $O/prof
cut -c1-150 $O/screens.txt
if [ $WHAT = rem ]; then cut -c1-120 $O/in_play.txt; grep -n "grok_remainder" $O/timeline.txt | head -100 | cut -c1-90; else grep -n "grok_screen_all_kernel" $O/timeline.txt | head -60 | cut -c1-90; fi
