#!/bin/bash
# plain block-Jacobi sweeps through the rectangle kernel (grouped loads) vs block_apply_kernel
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c48
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/$tag.json 2> $O/$tag.err
  python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("%-28s value %7.2f ms %7.1f solves/step %d its/solve %5.1f Kerr %.1e"%(sys.argv[2], d['value'], d['ms_per_step'], d['config']['shift_solves_per_step'], d['config']['gmres_iters_per_shift_solve'], d['config']['K_rel_diff_vs_oracle']))
PY
}
run plain RICADI_BA_PLAIN=1
run rect A=1
run plainb RICADI_BA_PLAIN=1
run rectb A=1
for v in 1 0; do
RICADI_BA_PLAIN=$v timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/cfg5_$v.json 2> $O/cfg5_$v.err; echo "cfg5 BA_PLAIN=$v $(cut -c1-110 $O/cfg5_$v.json)"
RICADI_BA_PLAIN=$v timeout -k 10 900 python bench.py --workload cfg4 --steps 1 --warmup 1 > $O/cfg4_$v.json 2> $O/cfg4_$v.err; echo "cfg4 BA_PLAIN=$v $(cut -c1-110 $O/cfg4_$v.json)"
done
exit 0
