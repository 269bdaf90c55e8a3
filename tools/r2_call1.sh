#!/bin/bash
# round 2, GPU call 1: new tests (minus the cfg2 fixture test) + where does the C++ sweep path lose time
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c1
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not cfg2_newton" -s > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -5 $O/gputests.log
for cc in 0 512 100000; do
  RICADI_CC=$cc timeout -k 10 300 python bench.py --steps 3 --warmup 1 --sequential --cpp-sweeps --sweep-width 16 --no-cpu-baseline --no-large-roofline > $O/cpp_cc$cc.json 2> $O/cpp_cc$cc.err
  echo "cc=$cc: $(cut -c1-200 $O/cpp_cc$cc.json)"
done
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-large-roofline > $O/py.json 2> $O/py.err
echo "py: $(cut -c1-200 $O/py.json)"
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_cpp -o cpp -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --sequential --cpp-sweeps --sweep-width 16 --no-cpu-baseline --no-large-roofline > $GRAFT_REPO_ROOT/$O/prof_cpp.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_cpp -name "*kernel_stats*" | head -3
f=$(find $O/prof_cpp -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f" > $O/prof_cpp_kernel_stats_head.csv
# drop the big trace files (64 MiB merge limit)
find $O/prof_cpp -name "*kernel_trace*" -delete
