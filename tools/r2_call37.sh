#!/bin/bash
# multi-shift tile SpMM at cfg2 with the groups split over more workgroups
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c37
mkdir -p $O
echo "== per-group kernel"; timeout -k 10 300 python tools/spmm_batch_pmc.py 58 16 200 2>&1 | grep "us per launch"
for ys in 1 2 4 8 16; do
  echo "== multi-shift kernel, ysplit $ys"
  RICADI_MS_SPMM=2 RICADI_MS_YSPLIT=$ys timeout -k 10 300 python tools/spmm_batch_pmc.py 58 16 200 2>&1 | grep "us per launch"
done
for ys in 1 2 4; do
  echo "== cfg5 multi-shift kernel, ysplit $ys"
  RICADI_MS_YSPLIT=$ys timeout -k 10 300 python tools/spmm_batch_pmc.py 236 16 50 2>&1 | grep "us per launch"
done
exit 0
