cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "early_rpn_backward_gives" 2>&1 | grep "AssertionError: (\|passed\|failed" | cut -c1-200; done
