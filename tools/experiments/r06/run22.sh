cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "graphed_trunk" 2>&1 | tail -1; done
timeout 1800 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06/pytest_gpu_g.txt 2>&1; echo "rc=$?" >> gpurun_out/r06/pytest_gpu_g.txt
tail -3 gpurun_out/r06/pytest_gpu_g.txt | cut -c1-200
