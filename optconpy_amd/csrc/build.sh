#!/bin/bash
# Build libricadi_hip.so for MI355X (gfx950).  hipcc cross-compiles without a GPU.
# One object per source file (kernel families + solver + host logic), compiled in parallel, then one link.
set -euo pipefail
cd "$(dirname "$0")"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function"
SRCS="ricadi_spmm.hip ricadi_arnoldi.hip ricadi_precond.hip ricadi_dense.hip ricadi_solver.hip ricadi_host.cpp"
mkdir -p build
pids=()
for s in $SRCS; do
  hipcc $FLAGS -c "$s" -o "build/${s%.*}.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
objs=""
for s in $SRCS; do objs="$objs build/${s%.*}.o"; done
hipcc --offload-arch=gfx950 -fPIC -shared $objs -o ../libricadi_hip.so -lrocsolver -lrocblas -lrccl
