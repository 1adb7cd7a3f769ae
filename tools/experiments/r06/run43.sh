cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_bf16_gpu.py tests/test_f16_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "bottleneck_tail" 2>&1 | tail -8
timeout 300 python tools/experiments/r06/tail_bench.py 2>&1 | tail -4
