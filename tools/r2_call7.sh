#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c7
mkdir -p $O
run() { # tag, env...
  tag=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/$tag.json 2> $O/$tag.err
  python - "$O/$tag.json" "$tag" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("%-28s value %7.2f ms %7.1f its/solve %5.1f Kerr %.1e"%(sys.argv[2], d['value'], d['ms_per_step'], d['config']['gmres_iters_per_shift_solve'], d['config']['K_rel_diff_vs_oracle']))
PY
}
run base A=1
run nofuse_jt RICADI_NOFUSE_JT=1
run tail2 RICADI_TAIL_GROUPS=2
run tail4 RICADI_TAIL_GROUPS=4
run tail6 RICADI_TAIL_GROUPS=6
run tail4_r50 RICADI_TAIL_GROUPS=4 RICADI_OPTS=gmres_restart=50
run tail8_r50 RICADI_TAIL_GROUPS=8 RICADI_OPTS=gmres_restart=50
run cyc16 RICADI_CYC0=16
run cyc10 RICADI_CYC0=10
