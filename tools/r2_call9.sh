#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c9
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -3 $O/gputests.log
cat > /tmp/tk.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, numpy as np
from optconpy_amd import _lib, problems as pb
torch.cuda.set_device(0)
pr = pb.ricc_problem(58, 0.05)
ctx = _lib.Context(0)
ctx.set_operator((-pr.A-pr.Nc).T.tocsr(), pr.M.T.tocsr(), pr.J)
ms = pb.logshifts(1.0, 3e3, 16)
for G in (16, 8, 2):
    out = {}
    for k in ("coarse", "block_v", "spmm", "spmm_sy", "dots", "update_dots", "update", "restrict", "precond"):
        ctx.time_kernel_dev(k, ms[:G], [1.0]*G, 16, nvec=7, reps=10)
        out[k] = round(1e3*min(ctx.time_kernel_dev(k, ms[:G], [1.0]*G, 16, nvec=7, reps=100) for _ in range(3)), 1)
    print("G=%d" % G, out, flush=True)
PY
python /tmp/tk.py 2>/dev/null | tee $O/tk32.log
RICADI_COARSE16=1 python /tmp/tk.py 2>/dev/null | tee $O/tk16.log
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
run coarse16 RICADI_COARSE16=1
run base_again A=1
