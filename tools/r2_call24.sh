#!/bin/bash
# three-level default: GPU suite, headline, cfg3-5 cycles
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c24
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/bench.json 2> $O/bench.err
cut -c1-200 $O/bench.json; echo
for w in cfg3 cfg4 cfg5; do
  st=2; wu=1; [ $w = cfg5 ] && st=1 && wu=0; [ $w = cfg4 ] && st=1
  timeout -k 10 900 python bench.py --workload $w --steps $st --warmup $wu > $O/$w.json 2> $O/$w.err; cut -c1-160 $O/$w.json; echo
done
exit 0
