#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c49
mkdir -p $O
timeout -k 10 300 python tools/kernel_classes.py 58 16 100 2>&1 | grep -E "block_v"
for i in 1 2; do
timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/bench$i.json 2> $O/bench$i.err; cut -c75-200 $O/bench$i.json
done
RICADI_BA_PLAIN=1 timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/bench_p.json 2> $O/bench_p.err; echo "plain: $(cut -c75-200 $O/bench_p.json)"
for v in 0 1; do
RICADI_BA_PLAIN=$v timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/cfg5_$v.json 2> $O/cfg5_$v.err; echo "cfg5 BA_PLAIN=$v $(cut -c1-110 $O/cfg5_$v.json)"
done
RICADI_BA_PLAIN=0 timeout -k 10 900 python bench.py --workload cfg3 --steps 2 --warmup 1 > $O/cfg3.json 2> $O/cfg3.err; echo "cfg3 $(cut -c1-110 $O/cfg3.json)"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
exit 0
