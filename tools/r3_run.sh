#!/bin/bash
# round-3 GPU call: tests, smoke, bench.  Usage: tools/r3_run.sh TAG [pytest -k expr]
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/$1
mkdir -p $O
if [ -n "$2" ]; then KX=(-k "$2"); else KX=(); fi
timeout -k 10 1000 python -m pytest tests -m gpu -x -q "${KX[@]}" > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -15 $O/gputests.log
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
( time timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
echo "bench rc=$?"; tail -3 $O/bench.time; cut -c1-300 $O/bench.json; echo; tail -5 $O/bench.err
exit 0
