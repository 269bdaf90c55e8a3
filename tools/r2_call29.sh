#!/bin/bash
# Arnoldi kernels with 16-byte loads (padded LDS): GPU suite, A/B of the headline, cfg3/cfg5; cfg3 with a smaller dense coarse
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c29
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
run mask0 RICADI_ARNOLDI16=0
run mask7 RICADI_ARNOLDI16=7
run mask0b RICADI_ARNOLDI16=0
run mask7b RICADI_ARNOLDI16=7
run cm1024 RICADI_OPTS=coarse_max=1024
for v in 0 7; do
  RICADI_ARNOLDI16=$v timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 0 > $O/cfg5_a$v.json 2> $O/cfg5_a$v.err
  echo "cfg5 ARNOLDI16=$v: $(cut -c1-120 $O/cfg5_a$v.json)"
  RICADI_ARNOLDI16=$v timeout -k 10 900 python bench.py --workload cfg3 --steps 2 --warmup 1 > $O/cfg3_a$v.json 2> $O/cfg3_a$v.err
  echo "cfg3 ARNOLDI16=$v: $(cut -c1-120 $O/cfg3_a$v.json)"
done
for cm in 2048 1800; do
  RICADI_OPTS=coarse_max=$cm timeout -k 10 900 python bench.py --workload cfg3 --steps 2 --warmup 1 > $O/cfg3_cm$cm.json 2> $O/cfg3_cm$cm.err
  echo "cfg3 coarse_max=$cm: $(cut -c1-120 $O/cfg3_cm$cm.json)"
done
RICADI_OPTS=coarse_max=2048 timeout -k 10 900 python bench.py --workload cfg4 --steps 1 --warmup 1 > $O/cfg4_cm2048.json 2> $O/cfg4_cm2048.err
echo "cfg4 coarse_max=2048: $(cut -c1-120 $O/cfg4_cm2048.json)"
exit 0
