#!/bin/bash
# PMC passes (FETCH_SIZE, WRITE_SIZE separately) of the batched K1 launches as the hot path issues them now
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2pmc
mkdir -p $O
cd /tmp
pmc() { # tag, counter, program args...
  tag=$1; ctr=$2; shift; shift
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${tag}_$ctr -o p -- python "$@" > $O/${tag}_$ctr.log 2>&1
  f=$(find $O/${tag}_$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_avg.py "$f" spmm_blocked > $O/${tag}_$ctr.txt
  rm -rf $O/${tag}_$ctr
}
cd $R
for ctr in FETCH_SIZE WRITE_SIZE; do
  pmc spmm58 $ctr tools/spmm_batch_pmc.py 58 16 30
  pmc spmm236 $ctr tools/spmm_batch_pmc.py 236 16 10
done
cat $O/spmm58_FETCH_SIZE.txt $O/spmm58_WRITE_SIZE.txt $O/spmm236_FETCH_SIZE.txt $O/spmm236_WRITE_SIZE.txt
exit 0
