cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_bf16_gpu.py tests/test_f16_gpu.py tests/test_golden_gpu.py -x -q -m gpu -k "small_kernels or groupnorm or gn or tower or golden or multi_level" 2>&1 | tail -2
python tools/experiments/gn_bench.py 2>/dev/null | tail -8
