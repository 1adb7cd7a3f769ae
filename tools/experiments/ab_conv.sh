#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of two builds of the library (box-to-box spread is ~5-10 %):
#   gpurun -- 'bash tools/experiments/ab_conv.sh gpurun_tmp/lib_old.so "python tools/conv_bench.py" 150-190'
OLD=$1; CMD=$2; COLS=${3:-1-200}
LIB=boosting-r-cnn_amd/lib/libbrcnn_hip.so
cp $LIB /tmp/new.so
for i in 1 2; do
  cp /tmp/new.so $LIB; $CMD 2>&1 | tail -14 | cut -c1-30,$COLS > /tmp/new$i.txt
  cp $OLD $LIB; $CMD 2>&1 | tail -14 | cut -c$COLS > /tmp/old$i.txt
done
cp /tmp/new.so $LIB
echo "new | old | new | old"; paste /tmp/new1.txt /tmp/old1.txt /tmp/new2.txt /tmp/old2.txt
