#!/bin/bash
# multilevel preconditioner: iteration counts of a batched 16-shift solve, two-level vs child levels
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c21
mkdir -p $O
export BATCH_ONLY=1
for N in 106 236; do
  for lv in 2 3 4; do
    echo "== N=$N RICADI_LEVELS=$lv"
    RICADI_LEVELS=$lv timeout -k 10 600 python tools/batch_probe.py $N 16 2>&1 | tail -2
  done
done 2>&1 | tee $O/probe.log
exit 0
