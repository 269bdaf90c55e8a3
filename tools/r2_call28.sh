#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c28
mkdir -p $O
for N in 236 58; do
  for v in "5 0" "7 1" "7 2" "7 3"; do
    set -- $v
    echo "== N=$N RICADI_ARNOLDI16=$1 RICADI_UD16=$2"
    RICADI_ARNOLDI16=$1 RICADI_UD16=$2 timeout -k 10 300 python tools/arnoldi_probe.py $N 16 30 2>&1 | grep -v amdgpu.ids | cut -c1-75
  done
done 2>&1 | tee $O/probe.log
exit 0
