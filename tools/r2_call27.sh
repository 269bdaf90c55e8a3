#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c27
mkdir -p $O
for N in 236 58; do
  for v in 0 7; do
    echo "== N=$N RICADI_ARNOLDI16=$v"
    RICADI_ARNOLDI16=$v timeout -k 10 300 python tools/arnoldi_probe.py $N 16 30 2>&1 | grep -v amdgpu.ids
  done
done 2>&1 | tee $O/probe.log
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/$tag.json 2> $O/$tag.err
  python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("%-28s value %7.2f ms %7.1f solves/step %d its/solve %5.1f Kerr %.1e"%(sys.argv[2], d['value'], d['ms_per_step'], d['config']['shift_solves_per_step'], d['config']['gmres_iters_per_shift_solve'], d['config']['K_rel_diff_vs_oracle']))
PY
}
run mask0 RICADI_ARNOLDI16=0
run mask5 RICADI_ARNOLDI16=5
run mask7 RICADI_ARNOLDI16=7
run mask0b RICADI_ARNOLDI16=0
run mask5b RICADI_ARNOLDI16=5
run mask7b RICADI_ARNOLDI16=7
for v in 0 5 7; do
  RICADI_ARNOLDI16=$v timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 0 > $O/cfg5_a$v.json 2> $O/cfg5_a$v.err
  echo "cfg5 ARNOLDI16=$v: $(cut -c1-120 $O/cfg5_a$v.json)"
done
exit 0
