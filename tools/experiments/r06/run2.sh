cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
for rep in 1 2; do
python tools/experiments/r06/no_wgrad.py train
python tools/experiments/r06/no_wgrad.py freeze
BRCNN_WGRAD_STREAM=0 python tools/experiments/r06/no_wgrad.py train
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/no_wgrad.log
