#!/bin/bash
# Same-call A/B of bench variants on the GPU box (boxes of the pool differ by ~5 %, so variants are
# only comparable inside ONE gpurun call).  Usage, from the repo root on the box:
#   tools/ab.sh OUTTAG "tag1:ENV1=a ENV2=b" "tag2:" ...      [BENCH_ARGS="--no-large-roofline ..."]
# Every variant runs `python bench.py $BENCH_ARGS` with its environment and prints one summary row;
# the JSON lines land in gpurun_out/OUTTAG/<tag>.json.  (Replaces the 51 one-off r2_call*.sh of round 2.)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
O=gpurun_out/$1; shift
mkdir -p "$O"
ARGS=${BENCH_ARGS:---no-large-roofline --no-cpu-baseline --no-extras}
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  env $envs timeout -k 10 600 python bench.py $ARGS > "$O/$tag.json" 2> "$O/$tag.err" || { echo "$tag: FAILED (see $O/$tag.err)"; tail -3 "$O/$tag.err"; continue; }
  python - "$O/$tag.json" "$tag" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
c = d["config"]
print("%-24s value %8.2f  ms/step %8.1f  solves/step %s  its/solve %s  K err %s" % (
    sys.argv[2], d["value"], d["ms_per_step"], c.get("shift_solves_per_step"),
    c.get("gmres_iters_per_shift_solve"), c.get("K_rel_diff_vs_oracle")))
PY
done
