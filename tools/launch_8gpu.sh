#!/usr/bin/env bash
# One node, one process per MI355X, RCCL over xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).
#   tools/launch_8gpu.sh [N=8] [extra bench.py args]      e.g.  tools/launch_8gpu.sh 8 --mode train --train-batch 16
# bench.py reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment torchrun sets, brackets the timed region with
# barrier + synchronize, takes the MAX over ranks and prints ONE JSON line on rank 0.  Inference runs independent replicas
# (images are independent: no data-path collective); --mode train reports the DDP training step (bucketed gradient all-reduce
# overlapped with backward, one-message BN buffer broadcast, 16-byte loss all-reduce).
set -euo pipefail
N="${1:-8}"; shift || true
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0            # the host driver only supports dmabuf IPC
export NCCL_MIN_NCHANNELS="${NCCL_MIN_NCHANNELS:-16}"   # ring all-reduce over point-to-point xGMI is per-link bound: use the links
exec python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29500}" \
     bench.py --gpus "$N" --steps "${STEPS:-50}" --warmup "${WARMUP:-10}" "$@"
