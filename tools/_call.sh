cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3x7; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/t.log 2>&1; tail -3 $O/t.log
