cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
python tools/experiments/r06/aten_train.py 2>/dev/null > gpurun_out/r06/aten_train.txt
head -100 gpurun_out/r06/aten_train.txt
