cd $GRAFT_REPO_ROOT
python tools/experiments/r06/roi_gather_tail.py 2>/dev/null
