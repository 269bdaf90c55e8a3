cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r3x1
timeout -k 10 300 python -m pytest tests/test_gpu_round3.py -x -q -k recompress > gpurun_out/r3x1/t.log 2>&1; tail -5 gpurun_out/r3x1/t.log
timeout -k 10 120 python tools/recompress_probe.py 26450 768 2>&1 | tail -2
RICADI_RECOMPRESS_EIG=1 timeout -k 10 120 python tools/recompress_probe.py 26450 768 2>&1 | tail -2
timeout -k 10 120 python tools/recompress_probe.py 26450 1200 2>&1 | tail -2
RICADI_RECOMPRESS_EIG=1 timeout -k 10 120 python tools/recompress_probe.py 26450 1200 2>&1 | tail -2
tools/ab.sh r3x1 "pchol:" "eig:RICADI_RECOMPRESS_EIG=1"
