#!/bin/bash
# multilevel preconditioner: level-2 aggregate sizes (velocity, pressure) against iteration counts
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c22
mkdir -p $O
export BATCH_ONLY=1 RICADI_LEVELS=3
for cfg in "236 2 2" "236 2 3" "236 8 6" "236 8 12" "236 4 3" "106 2 2" "106 2 3" "106 1 1"; do
  set -- $cfg
  echo "== N=$1 L2 aggregates $2 / $3"
  RICADI_L2_AV=$2 RICADI_L2_AP=$3 RICADI_VERBOSE=1 timeout -k 10 300 python tools/batch_probe.py $1 16 2>&1 | grep -v amdgpu.ids | tail -2
done 2>&1 | tee $O/probe.log
exit 0
