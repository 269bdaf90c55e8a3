#!/bin/bash
# robustness probes with the final defaults
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c41
mkdir -p $O
timeout -k 10 600 python tools/robustness_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/robust.log
timeout -k 10 300 python tools/check_precond.py 30 0.05 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/check_precond.log
exit 0
