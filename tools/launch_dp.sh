#!/bin/bash
# One rank per GPU over RCCL/xGMI on ONE node (what `python bench.py --gpus N` re-executes itself as):
#   tools/launch_dp.sh N [bench.py arguments ...]      e.g.  tools/launch_dp.sh 8 --steps 200 --config 4
set -euo pipefail
N=${1:?usage: launch_dp.sh N [bench args]}; shift
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=$(python3 -c 'import socket; s=socket.socket(); s.bind(("127.0.0.1",0)); print(s.getsockname()[1])')
cd "$(dirname "$0")/.."
exec python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 \
  --master-port "$PORT" bench.py --gpus "$N" "$@"
