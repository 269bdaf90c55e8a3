#!/bin/bash
# final round-2 record: GPU tests, smoke, full bench line, cfg3-5 cycles, rocprofv3 summaries
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2final9
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -1
( time timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
echo "bench rc=$?"; tail -3 $O/bench.time; cut -c1-240 $O/bench.json; echo
for w in cfg3 cfg4 cfg5; do
  st=2; wu=1; [ $w = cfg5 ] && st=1; [ $w = cfg4 ] && st=1
  timeout -k 10 900 python bench.py --workload $w --steps $st --warmup $wu > $O/$w.json 2> $O/$w.err; cut -c1-160 $O/$w.json; echo
done
cd /tmp
stats() { tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o $tag -- python "$@" > $O/$tag.log 2>&1
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats.csv; rm -rf $O/$tag; }
stats bench $R/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1
cd $R
stats classes58 tools/kernel_classes.py 58 16 100
stats spmm58 tools/spmm_batch_pmc.py 58 16 200
stats spmm236 tools/spmm_batch_pmc.py 236 16 50
grep "us per launch" $O/spmm58.log $O/spmm236.log | cut -c1-200
# attribution at cfg5 (warm): one feature off at a time
for t in "RICADI_FGMRES=0" "RICADI_ARNOLDI16=0" "RICADI_LEVELS=2" "RICADI_X32=0" "RICADI_H16=0" "RICADI_COARSE_GJ=0"; do
  env $t timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/cfg5_off.json 2> $O/cfg5_off.err
  echo "cfg5 with $t: $(cut -c1-110 $O/cfg5_off.json)"
done
exit 0
