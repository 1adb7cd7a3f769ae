cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 8 12 15 3; do echo "dbg=$d"; BRCNN_STEM_DBG=$d python tools/experiments/r06/stem_bench.py 2>&1 | grep -o "torch.*fused.*" | sed 's/stem.*maxpool [0-9.]* us//'; done
