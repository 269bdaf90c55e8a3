#!/bin/bash
# same-call A/B of ONE of the library's 0/1 environment switches: stage timings of the lockstep
# iteration at cfg2 / cfg5 sizes (tools/kernel_classes.py) and the default bench line (tools/ab.sh), switch on and off.
# Usage on the box: tools/switch_ab.sh TAG SWITCH   (e.g. RICADI_MID32, RICADI_W32, RICADI_BLOCKS16)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/${1:-mid32}
SW=${2:-RICADI_MID32}
mkdir -p $O
for v in 1 0; do
  env $SW=$v timeout -k 10 300 python tools/kernel_classes.py 58 16 100 > $O/classes58_$v.log 2>&1 || { tail -5 $O/classes58_$v.log; exit 1; }
  echo "== N=58 $SW=$v"; grep "pc_\|precond\|spmm" $O/classes58_$v.log | grep -v " 0.[0-9][0-9] us"
done
tools/ab.sh ${1:-mid32}/ab "on:$SW=1" "off:$SW=0"
for v in 1 0; do
  env $SW=$v timeout -k 10 500 python tools/kernel_classes.py 236 16 20 > $O/classes236_$v.log 2>&1 || { tail -5 $O/classes236_$v.log; exit 1; }
  echo "== N=236 $SW=$v"; grep "pc_\|precond\|spmm" $O/classes236_$v.log | grep -v " 0.[0-9][0-9] us"
done
