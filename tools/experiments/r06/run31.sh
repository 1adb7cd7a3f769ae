cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -x -q -m gpu -k "groupnorm or gn or small_kernels or sgd or optim or tower or golden or train_step" 2>&1 | tail -3
python tools/experiments/gn_bench.py 2>/dev/null | tail -8
