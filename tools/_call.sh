cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3x5; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -3 $O/gputests.log
tools/ab.sh r3x5 "fused:" "unfused:RICADI_SWEEP_UNFUSED=1" "fused2:"
RICADI_TIMING=1 timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep "timing" | tail -2
