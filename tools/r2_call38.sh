#!/bin/bash
# kernel-time profile of the cfg4 cycle (m = 66 panels)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2c38
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg4 -o cfg4 -- python $R/bench.py --workload cfg4 --steps 1 --warmup 0 > $O/cfg4.log 2>&1
f=$(find $O/cfg4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/cfg4_kernel_stats.csv; rm -rf $O/cfg4
head -25 $O/cfg4_kernel_stats.csv | cut -c1-150
exit 0
