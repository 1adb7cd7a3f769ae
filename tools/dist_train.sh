#!/usr/bin/env bash
# counterpart of the reference's tools/dist_train.sh: one process per GPU over RCCL
CONFIG=$1
GPUS=$2
PORT=${PORT:-29500}
export HSA_ENABLE_IPC_MODE_LEGACY=0
PYTHONPATH="$(dirname $0)/..":$PYTHONPATH \
python -m torch.distributed.run --nnodes=1 --nproc-per-node=$GPUS --master-addr 127.0.0.1 --master-port=$PORT \
    $(dirname "$0")/train.py $CONFIG --launcher pytorch ${@:3}
