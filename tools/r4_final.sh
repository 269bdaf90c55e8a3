#!/bin/bash
# round-4 record, in two gpurun calls (each < 20 min).  Usage on the box:  tools/r4_final.sh TAG a|b
#  a: GPU tests, smoke, the default bench line, 2-rank rehearsal line, rocprofv3 kernel statistics of the bench, of the
#     launch classes and of the K1 launches at cfg2 / cfg5 sizes, PMC passes (FETCH_SIZE / WRITE_SIZE separately) of K1
#  b: cfg3 (steady Newton step to K) with its extras, the cfg3-cycle / cfg4 / cfg5 passes through the boundary, cfg4-dre,
#     rocprofv3 kernel statistics of a two-time-step cfg4-dre run, 4- and 5-rank rehearsals on the one GPU
#  c: ONE call for a re-record when few GPU-minutes are left: a, then b without the full cfg4-dre sweep (200 s)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r4final}
mkdir -p $O
stats() { tag=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o $tag -- python "$@" > $O/$tag.log 2>&1 )
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats.csv; rm -rf $O/$tag; }
pmc() { tag=$1; ctr=$2; shift; shift
  ( cd /tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${tag}_$ctr -o p -- python "$@" > $O/${tag}_$ctr.log 2>&1 )
  f=$(find $O/${tag}_$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/pmc_avg.py "$f" spmm_blocked > $O/${tag}_$ctr.txt
  rm -rf $O/${tag}_$ctr; }
MODE=${2:-a}
if [ $MODE = a ] || [ $MODE = p ] || [ $MODE = c ]; then
  if [ $MODE = a ] || [ $MODE = c ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
  rc=$?; echo "pytest rc=$rc" >> $O/gputests.log; tail -3 $O/gputests.log
  [ $rc = 124 ] || [ $rc = 137 ] && { echo "GPU tests killed at their limit: no further GPU step in this call"; exit 1; }
  python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
  ( time timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
  echo "bench rc=$?"; tail -3 $O/bench.time; cut -c1-240 $O/bench.json; echo
  timeout -k 10 600 python bench.py --gpus 2 --rehearse-one-gpu --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks.err
  echo "2-rank rehearsal rc=$?"; cut -c1-200 $O/bench_2ranks_one_gpu.json; echo
  stats bench $R/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1
  fi
  stats classes58 $R/tools/kernel_classes.py 58 16 100
  stats spmm58 $R/tools/spmm_batch_pmc.py 58 16 200
  stats spmm236 $R/tools/spmm_batch_pmc.py 236 16 50
  grep "us per launch" $O/spmm58.log $O/spmm236.log $O/classes58.log | cut -c1-200
  for ctr in FETCH_SIZE WRITE_SIZE; do
    pmc spmm58 $ctr $R/tools/spmm_batch_pmc.py 58 16 30
    pmc spmm236 $ctr $R/tools/spmm_batch_pmc.py 236 16 10
  done
  cat $O/spmm58_FETCH_SIZE.txt $O/spmm58_WRITE_SIZE.txt $O/spmm236_FETCH_SIZE.txt $O/spmm236_WRITE_SIZE.txt
fi
if [ $MODE = b ] || [ $MODE = c ]; then
  timeout -k 10 900 python bench.py --workload cfg3 --steps 3 --warmup 1 --no-large-roofline > $O/cfg3.json 2> $O/cfg3.err; echo "cfg3 rc=$?"; cut -c1-200 $O/cfg3.json; echo
  for w in cfg3-cycle cfg4 cfg5; do
    st=2; wu=1; [ $w = cfg5 ] && st=1; [ $w = cfg4 ] && st=1
    timeout -k 10 900 python bench.py --workload $w --steps $st --warmup $wu > $O/$w.json 2> $O/$w.err; cut -c1-160 $O/$w.json; echo
  done
  if [ $MODE = b ]; then
  timeout -k 10 900 python bench.py --workload cfg4-dre --steps 1 --warmup 0 > $O/cfg4_dre.json 2> $O/cfg4_dre.err; cut -c1-200 $O/cfg4_dre.json; echo
  fi
  [ $MODE = c ] && for n in 4 5; do
    timeout -k 10 300 python bench.py --gpus $n --rehearse-one-gpu --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_${n}ranks_one_gpu.json 2> $O/bench_${n}ranks.err
    echo "$n-rank rehearsal rc=$?"; cut -c1-200 $O/bench_${n}ranks_one_gpu.json; echo
  done
  stats dre2 $R/bench.py --workload cfg4-dre --nts 2 --steps 1 --warmup 0 --no-cpu-baseline
  [ $MODE = b ] && for n in 4 5; do
    timeout -k 10 600 python bench.py --gpus $n --rehearse-one-gpu --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_${n}ranks_one_gpu.json 2> $O/bench_${n}ranks.err
    echo "$n-rank rehearsal rc=$?"; cut -c1-200 $O/bench_${n}ranks_one_gpu.json; echo
  done
fi
exit 0
