cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "reference_proposals_golden or configs1_size" 2>&1 | grep -v "^E  .*tensor(\|^  *\[" | tail -30
