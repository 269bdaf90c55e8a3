#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c6
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1
echo "pytest rc=$?" >> $O/gputests.log
tail -4 $O/gputests.log
timeout -k 10 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
tail -2 $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2c6/bench.json'))
print({k:d.get(k) for k in ('value','ms_per_step','value_python_sweep_driver','value_fp64_storage')})
print(d['config']['K_rel_diff_vs_oracle'], d['config']['gmres_iters_per_shift_solve'])
for key in ('roofline','roofline_cfg5'):
    r=d[key]; print(key, {k:r.get(k) for k in ('us_per_launch','frac','frac_batched_form','single_panel_us_per_launch','single_panel_frac')})
for key in ('roofline_kernels','roofline_kernels_cfg5'):
    print(key, {k:(v['us_per_launch'],v['frac']) if isinstance(v,dict) else v for k,v in d[key].items()})
PY
