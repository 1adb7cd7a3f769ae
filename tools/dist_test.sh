#!/usr/bin/env bash
# counterpart of the reference's tools/dist_test.sh
CONFIG=$1
CHECKPOINT=$2
GPUS=$3
PORT=${PORT:-29500}
export HSA_ENABLE_IPC_MODE_LEGACY=0
PYTHONPATH="$(dirname $0)/..":$PYTHONPATH \
python -m torch.distributed.run --nnodes=1 --nproc-per-node=$GPUS --master-addr 127.0.0.1 --master-port=$PORT \
    $(dirname "$0")/test.py $CONFIG $CHECKPOINT --launcher pytorch ${@:4}
