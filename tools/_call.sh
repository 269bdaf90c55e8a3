cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3x13; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -6 $O/gputests.log
