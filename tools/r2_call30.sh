#!/bin/bash
# third level whenever the base coarse problem exceeds coarse_max / 2; parity-ordered tile rows A/B
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c30
mkdir -p $O
for v in 0 1; do
  echo "== RICADI_SB_PARITY=$v"
  RICADI_SB_PARITY=$v timeout -k 10 300 python tools/spmm_batch_pmc.py 58 16 200 2>&1 | grep "us per launch"
  RICADI_SB_PARITY=$v timeout -k 10 300 python tools/spmm_batch_pmc.py 236 16 50 2>&1 | grep "us per launch"
done
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/$tag.json 2> $O/$tag.err
  python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("%-28s value %7.2f ms %7.1f solves/step %d its/solve %5.1f Kerr %.1e"%(sys.argv[2], d['value'], d['ms_per_step'], d['config']['shift_solves_per_step'], d['config']['gmres_iters_per_shift_solve'], d['config']['K_rel_diff_vs_oracle']))
PY
}
run par0 RICADI_SB_PARITY=0
run par1 RICADI_SB_PARITY=1
run par0b RICADI_SB_PARITY=0
run par1b RICADI_SB_PARITY=1
for w in cfg3 cfg4 cfg5; do
  st=2; wu=1; [ $w = cfg5 ] && st=1 && wu=0; [ $w = cfg4 ] && st=1
  timeout -k 10 900 python bench.py --workload $w --steps $st --warmup $wu > $O/$w.json 2> $O/$w.err; cut -c1-130 $O/$w.json; echo
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
exit 0
