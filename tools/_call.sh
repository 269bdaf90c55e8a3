cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 120 python tools/qr_probe.py 26450 456 2>&1 | grep -v amdgpu.ids
timeout -k 10 120 python tools/recompress_probe.py 26450 768 2>&1 | tail -1
python - <<'PY'
import sys; sys.path.insert(0,'.')
import bench, torch
from optconpy_amd import _lib
ctx=_lib.Context(0); ctx.set_dims(26450)
print(bench.gram_mfma(ctx, 26450, 512)); print(bench.gram_mfma(ctx, 26450, 128))
PY
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "tsqr or compress or gain or update" 2>&1 | tail -2
tools/ab.sh r3x18 "pref:" "pref2:"
