#!/bin/bash
# rectangle sweep with all indices / gathers / tile loads issued in groups
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c45
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
for i in 1 2 3; do
timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/bench$i.json 2> $O/bench$i.err; cut -c75-200 $O/bench$i.json
done
timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/cfg5.json 2> $O/cfg5.err; cut -c1-130 $O/cfg5.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
f=$(find $GRAFT_REPO_ROOT/$O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $GRAFT_REPO_ROOT/$O/kernel_stats.csv; rm -rf $GRAFT_REPO_ROOT/$O/prof
python - <<'PY'
import csv, os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r2c45/kernel_stats.csv')))
for r in rows[:12]:
    print("%-60s %6s %8.1f us %5.1f%%"%(r['Name'].replace('void ','').split('(')[0][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
exit 0
