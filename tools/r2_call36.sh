#!/bin/bash
# preconditioner input from the FP16 basis vector (RICADI_H16), FP64 copy of the current vector not written
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c36
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
run h0 RICADI_H16=0
run h1 RICADI_H16=1
run h0b RICADI_H16=0
run h1b RICADI_H16=1
for rep in a b; do
  for v in 0 1; do
    RICADI_H16=$v timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/cfg5_h$v$rep.json 2> $O/cfg5_h$v$rep.err
    echo "cfg5 H16=$v ($rep): $(cut -c1-120 $O/cfg5_h$v$rep.json)"
  done
done
for v in 0 1; do
  RICADI_H16=$v timeout -k 10 900 python bench.py --workload cfg3 --steps 2 --warmup 1 > $O/cfg3_h$v.json 2> $O/cfg3_h$v.err
  echo "cfg3 H16=$v: $(cut -c1-120 $O/cfg3_h$v.json)"
done
exit 0
