#!/bin/bash
# Build libricadi_hip.so for MI355X (gfx950).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Wall -Wno-unused-function \
  ricadi_kernels.hip ricadi_solver.hip ricadi_host.cpp \
  -o ../libricadi_hip.so -lrocsolver -lrocblas -lrccl
