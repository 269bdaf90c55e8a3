#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c4
mkdir -p $O
RICADI_TIMING=1 timeout -k 10 600 python bench.py --no-large-roofline --no-cpu-baseline --no-extras --steps 2 > $O/bench_t.json 2> $O/bench_t.err
grep "ricadi timing" $O/bench_t.err | tail -8
