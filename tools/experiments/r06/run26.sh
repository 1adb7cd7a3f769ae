cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_bf16_gpu.py tests/test_f16_gpu.py tests/test_ops_gpu.py tests/test_golden_gpu.py -x -q -m gpu -k "stem or golden or backbone or resnet" 2>&1 | tail -3
python tools/experiments/r06/stem_bench.py 2>&1 | tail -4
