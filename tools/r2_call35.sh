#!/bin/bash
# operator on the FP32-stored Z_j (RICADI_X32) vs on the FP64 z, second round (raw loads first, conversion at the LDS store)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c35
mkdir -p $O
for v in 0 1; do
  echo "== RICADI_X32=$v"
  RICADI_X32=$v timeout -k 10 300 python tools/spmm_batch_pmc.py 58 16 200 2>&1 | grep "us per launch"
  RICADI_X32=$v timeout -k 10 300 python tools/spmm_batch_pmc.py 236 16 50 2>&1 | grep "us per launch"
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
run x0 RICADI_X32=0
run x1 RICADI_X32=1
run x0b RICADI_X32=0
run x1b RICADI_X32=1
for rep in a b; do
  for v in 0 1; do
    RICADI_X32=$v timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 0 > $O/cfg5_x$v$rep.json 2> $O/cfg5_x$v$rep.err
    echo "cfg5 X32=$v ($rep): $(cut -c1-120 $O/cfg5_x$v$rep.json)"
  done
done
for v in 0 1; do
  RICADI_X32=$v timeout -k 10 900 python bench.py --workload cfg3 --steps 2 --warmup 1 > $O/cfg3_x$v.json 2> $O/cfg3_x$v.err
  echo "cfg3 X32=$v: $(cut -c1-120 $O/cfg3_x$v.json)"
done
exit 0
