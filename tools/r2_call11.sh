#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c11
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
( time timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
echo "bench rc=$?"; tail -3 $O/bench.time
cut -c1-260 $O/bench.json
timeout -k 10 600 python bench.py --workload cfg3 --steps 2 --warmup 1 > $O/cfg3.json 2> $O/cfg3.err; cut -c1-200 $O/cfg3.json; echo
timeout -k 10 600 python bench.py --workload cfg4 --steps 1 --warmup 1 > $O/cfg4.json 2> $O/cfg4.err; cut -c1-200 $O/cfg4.json; echo
timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 0 > $O/cfg5.json 2> $O/cfg5.err; cut -c1-200 $O/cfg5.json; echo
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
