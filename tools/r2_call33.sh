#!/bin/bash
# coarse inverses by block Gauss-Jordan (batched GEMMs) vs rocSOLVER getrf + getri
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c33
mkdir -p $O
for v in 0 1; do
  echo "== RICADI_COARSE_GJ=$v"
  RICADI_COARSE_GJ=$v timeout -k 10 300 python tools/ml_probe.py 58 2>&1 | grep -v amdgpu.ids | cut -c1-200
  RICADI_COARSE_GJ=$v timeout -k 10 300 python tools/ml_probe.py 106 2>&1 | grep -v amdgpu.ids | cut -c1-200
done 2>&1 | tee $O/probe.log
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
run gj0 RICADI_COARSE_GJ=0
run gj1 RICADI_COARSE_GJ=1
run gj0b RICADI_COARSE_GJ=0
run gj1b RICADI_COARSE_GJ=1
for w in cfg3 cfg4; do
  for v in 0 1; do
    st=2; wu=1; [ $w = cfg4 ] && st=1
    RICADI_COARSE_GJ=$v timeout -k 10 900 python bench.py --workload $w --steps $st --warmup $wu > $O/${w}_gj$v.json 2> $O/${w}_gj$v.err
    echo "$w GJ=$v: $(cut -c1-120 $O/${w}_gj$v.json)"
  done
done
exit 0
