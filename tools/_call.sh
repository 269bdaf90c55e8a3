cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r3x4; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -4 $O/gputests.log
RICADI_TIMING=1 timeout -k 10 300 python bench.py --no-large-roofline --no-cpu-baseline --no-extras > $O/b.json 2> $O/b.err; cut -c1-300 $O/b.json; grep -i "timing\|setup\|sweep" $O/b.err | tail -12
