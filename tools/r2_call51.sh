#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c51
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
for i in 1 2 3; do
timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/bench$i.json 2> $O/bench$i.err; cut -c75-200 $O/bench$i.json
done
timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/cfg5.json 2> $O/cfg5.err; cut -c1-130 $O/cfg5.json
exit 0
