#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c2
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -4 $O/gputests.log
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
tail -3 $O/bench.err
cut -c1-600 $O/bench.json
