#!/bin/bash
# round-3 record: GPU tests, smoke, full bench line, 2-rank rehearsal line, cfg3-5 cycles, rocprofv3 summaries,
# PMC passes (FETCH_SIZE / WRITE_SIZE separately) of the K1 launches.  Usage on the box: tools/r3_final.sh [TAG] [dre]
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3final}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
( time timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
echo "bench rc=$?"; tail -3 $O/bench.time; cut -c1-240 $O/bench.json; echo
timeout -k 10 600 python bench.py --gpus 2 --rehearse-one-gpu --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
echo "2-rank rehearsal rc=$?"; cut -c1-200 $O/bench_2ranks_one_gpu.json; echo
for w in cfg3 cfg4 cfg5; do
  st=2; wu=1; [ $w = cfg5 ] && st=1; [ $w = cfg4 ] && st=1
  timeout -k 10 900 python bench.py --workload $w --steps $st --warmup $wu > $O/$w.json 2> $O/$w.err; cut -c1-160 $O/$w.json; echo
done
if [ "$2" = dre ]; then
  timeout -k 10 900 python bench.py --workload cfg4-dre --steps 1 --warmup 0 > $O/cfg4_dre.json 2> $O/cfg4_dre.err; cut -c1-200 $O/cfg4_dre.json; echo
fi
cd /tmp
stats() { tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o $tag -- python "$@" > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats.csv; rm -rf $O/$tag; }
pmc() { tag=$1; ctr=$2; shift; shift
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${tag}_$ctr -o p -- python "$@" > $O/${tag}_$ctr.log 2>&1
  f=$(find $O/${tag}_$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_avg.py "$f" spmm_blocked > $O/${tag}_$ctr.txt
  rm -rf $O/${tag}_$ctr; }
stats bench $R/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1
cd $R
stats classes58 tools/kernel_classes.py 58 16 100
stats spmm58 tools/spmm_batch_pmc.py 58 16 200
stats spmm236 tools/spmm_batch_pmc.py 236 16 50
grep "us per launch" $O/spmm58.log $O/spmm236.log $O/classes58.log | cut -c1-200
for ctr in FETCH_SIZE WRITE_SIZE; do
  pmc spmm58 $ctr tools/spmm_batch_pmc.py 58 16 30
  pmc spmm236 $ctr tools/spmm_batch_pmc.py 236 16 10
done
cat $O/spmm58_FETCH_SIZE.txt $O/spmm58_WRITE_SIZE.txt $O/spmm236_FETCH_SIZE.txt $O/spmm236_WRITE_SIZE.txt
exit 0
