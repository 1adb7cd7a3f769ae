cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_f16_gpu.py -x -q -m gpu 2>&1 | tail -3
python tools/train_layers.py 2>/dev/null | grep "dgrad_bn_backward_ex" | head -12
