#!/bin/bash
# dot kernels with 256-row workgroups and the cross-workgroup sum fused in (RICADI_DOTS16X = 0 / 1 / 2)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c47
mkdir -p $O
for v in 0 1 2; do
  echo "== RICADI_DOTS16X=$v"
  RICADI_DOTS16X=$v timeout -k 10 300 python tools/arnoldi_probe.py 58 16 50 2>&1 | grep -v amdgpu.ids | cut -c1-82
done
for v in 0 1; do
  echo "== RICADI_DOTS16X=$v"
  RICADI_DOTS16X=$v timeout -k 10 300 python tools/arnoldi_probe.py 236 16 20 2>&1 | grep -v amdgpu.ids | cut -c1-82
done
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
run d0 RICADI_DOTS16X=0
run d2 RICADI_DOTS16X=2
run d1 RICADI_DOTS16X=1
run d0b RICADI_DOTS16X=0
run d2b RICADI_DOTS16X=2
for v in 0 2; do
RICADI_DOTS16X=$v timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/cfg5_$v.json 2> $O/cfg5_$v.err; echo "cfg5 DOTS16X=$v $(cut -c1-110 $O/cfg5_$v.json)"
done
exit 0
