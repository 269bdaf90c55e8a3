#!/bin/bash
# A/B: flexible GMRES (Z_j kept in FP32, no preconditioner application at the cycle end) vs right preconditioning
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c20
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
run right RICADI_FGMRES=0
run flex A=1
run right2 RICADI_FGMRES=0
run flex2 A=1
for w in cfg3 cfg5; do
  st=2; wu=1; [ $w = cfg5 ] && st=1 && wu=0
  for v in 0 1; do
    RICADI_FGMRES=$v timeout -k 10 900 python bench.py --workload $w --steps $st --warmup $wu > $O/${w}_f$v.json 2> $O/${w}_f$v.err
    echo "$w FGMRES=$v: $(cut -c1-120 $O/${w}_f$v.json)"
  done
done
exit 0
