cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "device_path_equals_reference_chain" 2>&1 | grep -v "^E  .*tensor\|^  *\[" | tail -25
