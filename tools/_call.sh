cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python tools/robustness_probe.py 2>&1 | grep "^N="
timeout 600 python tools/ml_probe.py 58 2>&1 | tail -4
tools/ab.sh r3x16 "sa05:"
