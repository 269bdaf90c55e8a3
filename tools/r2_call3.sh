#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c3
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -4 $O/gputests.log
timeout -k 10 900 python bench.py --no-large-roofline --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c3/bench.json'))
print({k:d[k] for k in ('value','ms_per_step','value_python_sweep_driver','value_fp64_storage')})
print(d['config']['K_rel_diff_vs_oracle'], d['roofline_tsqr_mfma'])
PY
