#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c53
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
timeout -k 10 600 python bench.py --workload cfg3 --steps 2 --warmup 1 > $O/cfg3.json 2> $O/cfg3.err; echo "cfg3 $(cut -c1-110 $O/cfg3.json)"
timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err; cut -c75-200 $O/bench.json
exit 0
