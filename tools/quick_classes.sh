#!/bin/bash
# stage timings at cfg2 / cfg5 sizes + the default bench line of the tree as it is.  Usage on the box: tools/quick_classes.sh TAG [ENV=..]
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/${1:-quick}; shift
mkdir -p $O
env "$@" timeout -k 10 300 python tools/kernel_classes.py 58 16 100 > $O/classes58.log 2>&1 || { tail -5 $O/classes58.log; exit 1; }
echo "== N=58"; grep "us per launch" $O/classes58.log | grep -v " 0.[0-9][0-9] us"
tools/ab.sh $(basename $O)/ab "tree:$*"
env "$@" timeout -k 10 500 python tools/kernel_classes.py 236 16 20 > $O/classes236.log 2>&1 || { tail -5 $O/classes236.log; exit 1; }
echo "== N=236"; grep "us per launch" $O/classes236.log | grep -v " 0.[0-9][0-9] us"
