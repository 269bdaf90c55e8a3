#!/bin/bash
# gentle third level: operator variants at n = 1e5, cfg4 cycle, GPU tests with the level enabled
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c23
mkdir -p $O
for lv in 2 3; do
  echo "== RICADI_LEVELS=$lv"
  RICADI_LEVELS=$lv timeout -k 10 600 python tools/ml_probe.py 106 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $O/probe.log
for lv in 2 3; do
  RICADI_LEVELS=$lv timeout -k 10 900 python bench.py --workload cfg4 --steps 1 --warmup 1 > $O/cfg4_l$lv.json 2> $O/cfg4_l$lv.err
  echo "cfg4 LEVELS=$lv: $(cut -c1-120 $O/cfg4_l$lv.json)"
done
RICADI_LEVELS=3 timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests_l3.log 2>&1
echo "pytest rc=$?" >> $O/gputests_l3.log
tail -3 $O/gputests_l3.log
exit 0
