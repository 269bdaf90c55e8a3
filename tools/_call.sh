cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3x7; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q > $O/t.log 2>&1; tail -5 $O/t.log
timeout -k 10 600 python bench.py --no-large-roofline --no-cpu-baseline > $O/b.json 2> $O/b.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r3x7/b.json"))
print({k:d[k] for k in d if k.startswith(("value","ms_per","pcie")) or k=="metric"})
print(d["config"]["K_rel_diff_vs_oracle"], d["roofline_tsqr_mfma"])
PY
tail -3 $O/b.err
