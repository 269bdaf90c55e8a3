#!/bin/bash
# coarse apply with 4 chunks per pass and unconditional grouped loads; reduce_partials with 8 loads in flight
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c44
mkdir -p $O
timeout -k 10 300 python tools/kernel_classes.py 58 16 100 2>&1 | grep -E "coarse|dots|block_v|restrict"
timeout -k 10 300 python tools/kernel_classes.py 236 16 30 2>&1 | grep -E "coarse|dots|block_v|restrict"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
for i in 1 2 3; do
timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/bench$i.json 2> $O/bench$i.err; cut -c75-200 $O/bench$i.json
done
timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/cfg5.json 2> $O/cfg5.err; cut -c1-130 $O/cfg5.json
timeout -k 10 900 python bench.py --workload cfg4 --steps 1 --warmup 1 > $O/cfg4.json 2> $O/cfg4.err; cut -c1-130 $O/cfg4.json
exit 0
