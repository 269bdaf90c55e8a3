#!/bin/bash
# base aggregates grown in steps of 1.5 until the gentle child level fits: cfg5 / cfg4 / cfg3 cycles, probe at N=236
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c25
mkdir -p $O
BATCH_ONLY=1 timeout -k 10 300 python tools/batch_probe.py 236 16 2>&1 | grep -v amdgpu.ids | tail -1
BATCH_ONLY=1 timeout -k 10 300 python tools/batch_probe.py 150 16 2>&1 | grep -v amdgpu.ids | tail -1
BATCH_ONLY=1 RICADI_LEVELS=2 timeout -k 10 300 python tools/batch_probe.py 150 16 2>&1 | grep -v amdgpu.ids | tail -1
for w in cfg5 cfg4; do
  st=1; wu=0; [ $w = cfg4 ] && wu=1
  timeout -k 10 900 python bench.py --workload $w --steps $st --warmup $wu > $O/$w.json 2> $O/$w.err; cut -c1-160 $O/$w.json; echo
done
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
exit 0
