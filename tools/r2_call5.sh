#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c5
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -4 $O/gputests.log
RICADI_TIMING=1 timeout -k 10 600 python bench.py --no-large-roofline --no-cpu-baseline --no-extras --steps 2 > $O/bench_t.json 2> $O/bench_t.err
grep "ricadi timing" $O/bench_t.err | tail -4
timeout -k 10 600 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
cut -c1-220 $O/bench.json
RICADI_SYNC_RECOMPRESS=1 timeout -k 10 600 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/bench_sync.json 2> $O/bench_sync.err
cut -c1-220 $O/bench_sync.json
