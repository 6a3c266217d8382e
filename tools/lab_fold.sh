#!/bin/bash
# tools/lab_fold.sh: what folding multi-stamp register programs into set registers buys (regex_handle.cpp planTdfaFold).
# The headline batch with an empty referrer field on every k-th line, tables packed with and without the fold.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for k in 0 10 2; do
  for fold in 1 0; do
    if [ $fold = 0 ]; then export LC_TDFA_NO_FOLD=1; else unset LC_TDFA_NO_FOLD; fi
    LAB_EMPTY_EVERY=$k python tools/tdfa_lab_inputs.py /tmp/lab_in.bin > /dev/null || exit 1
    echo "== empty referrer every $k lines, fold=$fold"
    LAB_ONLY=1 timeout 300 scratch/tdfa_lab /tmp/lab_in.bin 20 | grep -v "bare chain\|no output"
  done
done
