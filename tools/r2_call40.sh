#!/bin/bash
# rehearsal of bench.py's N > 1 code path: 2 ranks on the one GPU, gloo collectives (not a measurement)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r2c40
mkdir -p $O
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 1 --rehearse-one-gpu > $O/n2.json 2> $O/n2.err
echo "rc=$?"; cut -c1-400 $O/n2.json; tail -3 $O/n2.err
exit 0
