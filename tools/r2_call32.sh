#!/bin/bash
# gain() from the device-resident factor; first restart-cycle length after the Arnoldi rewrite
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c32
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
run cyc10 RICADI_CYC0=10
run cyc8 RICADI_CYC0=8
run cyc12 RICADI_CYC0=12
run cyc14 RICADI_CYC0=14
run cyc18 RICADI_CYC0=18
run cyc10b RICADI_CYC0=10
run cyc12r40 RICADI_CYC0=12 RICADI_OPTS=gmres_restart=40
for c in 10 14 20; do
  RICADI_CYC0=$c timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 0 > $O/cfg5_c$c.json 2> $O/cfg5_c$c.err
  echo "cfg5 CYC0=$c: $(cut -c1-120 $O/cfg5_c$c.json)"
done
exit 0
