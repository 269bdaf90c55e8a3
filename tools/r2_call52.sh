#!/bin/bash
# cfg3: multi-shift SpMM (and with it the FP32 operator input) forced vs chosen by size
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c52
mkdir -p $O
for rep in a b; do
for v in 1 2; do
RICADI_MS_SPMM=$v timeout -k 10 600 python bench.py --workload cfg3 --steps 2 --warmup 1 > $O/cfg3_$v$rep.json 2> $O/cfg3_$v$rep.err; echo "cfg3 MS_SPMM=$v $(cut -c1-110 $O/cfg3_$v$rep.json)"
done
done
exit 0
