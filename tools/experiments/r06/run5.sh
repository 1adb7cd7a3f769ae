cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/ -x -q -m gpu --deselect "tests/test_train_gpu.py::test_early_rpn_backward_gives_the_same_step" > gpurun_out/r06/pytest_gpu_b.txt 2>&1; echo "rc=$?" >> gpurun_out/r06/pytest_gpu_b.txt
tail -12 gpurun_out/r06/pytest_gpu_b.txt
timeout 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "early_rpn_backward_gives" 2>&1 | tail -5
b() { env $1 python bench.py --mode train --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
print('$1', 'train ms_per_step', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median',0),3), 'slack', d.get('host_slack_at_sync_ms'))"; }
for i in 1 2 3; do
b "X=1"
b "BRCNN_FUSE_FAN_IN=0 BRCNN_FC_PACK=0 BRCNN_ROI_ADDEND=0"
done 2>&1 | tee gpurun_out/r06/ab_glue.log
