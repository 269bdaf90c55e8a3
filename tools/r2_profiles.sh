#!/bin/bash
# round-2 profiles: rocprofv3 kernel stats of the bench run, of the K1 launches (cfg2 / cfg5) and of
# the kernel classes; PMC passes (FETCH_SIZE, WRITE_SIZE separately) of the K1 launches.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2prof
mkdir -p $O
cd /tmp
stats() { # tag, program args...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o $tag -- python "$@" > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats.csv
  rm -rf $O/$tag
}
pmc() { # tag, counter, program args...
  tag=$1; ctr=$2; shift; shift
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${tag}_$ctr -o p -- python "$@" > $O/${tag}_$ctr.log 2>&1
  f=$(find $O/${tag}_$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_avg.py "$f" spmm_blocked > $O/${tag}_$ctr.txt
  rm -rf $O/${tag}_$ctr
}
cd $R
stats bench bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1
echo "bench stats done"
stats spmm58 tools/spmm_batch_pmc.py 58 16 200
stats spmm236 tools/spmm_batch_pmc.py 236 16 50
stats classes58 tools/kernel_classes.py 58 16 100
echo "stats done"
for ctr in FETCH_SIZE WRITE_SIZE; do
  pmc spmm58 $ctr tools/spmm_batch_pmc.py 58 16 30
  pmc spmm236 $ctr tools/spmm_batch_pmc.py 236 16 10
done
ls $O
cat $O/spmm58_FETCH_SIZE.txt $O/spmm58_WRITE_SIZE.txt $O/spmm236_FETCH_SIZE.txt $O/spmm236_WRITE_SIZE.txt
grep "us per launch" $O/spmm58.log $O/spmm236.log $O/classes58.log
