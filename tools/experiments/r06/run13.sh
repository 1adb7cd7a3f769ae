cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06/pytest_gpu_d.txt 2>&1; echo "rc=$?" >> gpurun_out/r06/pytest_gpu_d.txt
tail -5 gpurun_out/r06/pytest_gpu_d.txt
b() { env $1 python bench.py --mode train --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
s=d['stages_ms']
print('$1', 'train ms_per_step', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median',0),3), 'slack', round(d.get('host_slack_at_sync_ms'),2), 'rpn_bwd', s['rpn_loss_and_backward'], 'roi', s['roi_align'], 'fc', s['fc_head'], 'bwd', s['backward'])"; }
for i in 1 2 3; do b "X=1"; done 2>&1 | tee gpurun_out/r06/bench_e.log
