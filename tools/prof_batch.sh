#!/bin/bash
# rocprofv3 kernel stats of the batched solve alone (cfg2; default 16 shifts).  Run on the GPU box:
#   bash tools/prof_batch.sh [tag] [groups]
tag=${1:-pb}
ng=${2:-16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export BATCH_ONLY=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -o $tag -- python tools/batch_probe.py 58 $ng > gpurun_out/$tag.log 2>&1 || exit 1
cp $(find gpurun_out/$tag -name "*kernel_stats.csv") gpurun_out/${tag}_stats.csv
rm -rf gpurun_out/$tag
grep "G=" gpurun_out/$tag.log
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/${tag}_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)/1e6
print("total kernel ms %.1f" % tot)
for r in rows[:22]:
    print("%-44s %7d %8.1f us %8.1f ms %5.1f%%"%(r['Name'].split('(')[0].replace('void ','').replace('ricadi::','')[:44],int(r['Calls']),float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6,float(r['Percentage'])))
PY
