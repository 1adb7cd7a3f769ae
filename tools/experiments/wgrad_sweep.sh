#!/bin/bash
# Run ON THE GPU BOX: sweep the M-slicing of the wgrad kernels (BRCNN_WG_SLOTS / BRCNN_WG_MINROWS):
#   gpurun -- 'DT="bf16 2" SLOTS="512 1024" ROWS="512 1024" bash tools/experiments/wgrad_sweep.sh'
for t in $SLOTS; do for r in $ROWS; do
  echo -n "slots $t minrows $r: "
  BRCNN_WG_SLOTS=$t BRCNN_WG_MINROWS=$r python tools/wgrad_bench.py $DT 2>&1 | grep TF | awk '{printf "%s ", $(NF-1)}'
  echo
done; done
