cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3x10; mkdir -p $O
RICADI_OPTS="coarse_max=8192" timeout -k 10 900 python bench.py --workload cfg5 --steps 1 --warmup 1 > $O/cfg5_c8192.json 2> $O/cfg5_c8192.err; cut -c1-160 $O/cfg5_c8192.json; echo; tail -3 $O/cfg5_c8192.err
RICADI_OPTS="coarse_max=8192" timeout -k 10 900 python bench.py --workload cfg4 --steps 1 --warmup 1 > $O/cfg4_c8192.json 2> $O/cfg4_c8192.err; cut -c1-160 $O/cfg4_c8192.json; echo
python - <<'PY'
import json
for w in ['cfg5_c8192','cfg4_c8192']:
    try:
        d=json.load(open('gpurun_out/r3x10/%s.json'%w)); c=d['config']
        print(w, d['value'], d['ms_per_step'], {k:c[k] for k in c if 'iter' in k or 'level' in k or 'coarse' in k or 'nonconv' in k})
    except Exception as e: print(w, e)
PY
