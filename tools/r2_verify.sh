#!/bin/bash
# end-of-round verification of the committed tree: GPU suite, smoke, default bench line
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2verify
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-230 $O/bench.json
exit 0
