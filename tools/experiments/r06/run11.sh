cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06/pytest_gpu_c.txt 2>&1; echo "rc=$?" >> gpurun_out/r06/pytest_gpu_c.txt
tail -5 gpurun_out/r06/pytest_gpu_c.txt
