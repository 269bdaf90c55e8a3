#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c31
mkdir -p $O
for N in 58 236; do
  for ti in 1 2; do
    echo "== N=$N RICADI_COARSE_TI=$ti"
    RICADI_COARSE_TI=$ti timeout -k 10 300 python tools/kernel_classes.py $N 16 50 2>&1 | grep -E "coarse|block_v|restrict"
  done
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
run ti1 RICADI_COARSE_TI=1
run ti2 RICADI_COARSE_TI=2
run ti1b RICADI_COARSE_TI=1
run ti2b RICADI_COARSE_TI=2
RICADI_TIMING=1 timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/timing.json 2> $O/timing.err
grep -i "timing\|setup\|ms" $O/timing.err | tail -8
exit 0
