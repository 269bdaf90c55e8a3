#!/bin/bash
# register-resident diagonal-block inverse of the block Gauss-Jordan
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c39
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
timeout -k 10 300 python tools/ml_probe.py 58 2>&1 | grep -v amdgpu.ids | cut -c1-120
timeout -k 10 300 python tools/ml_probe.py 106 2>&1 | grep -v amdgpu.ids | cut -c1-120
for i in 1 2; do
timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/bench$i.json 2> $O/bench$i.err; cut -c1-200 $O/bench$i.json | cut -c75-200
done
timeout -k 10 900 python bench.py --workload cfg3 --steps 2 --warmup 1 > $O/cfg3.json 2> $O/cfg3.err; cut -c1-130 $O/cfg3.json
exit 0
