cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_bf16_gpu.py tests/test_train_gpu.py tests/test_stress_gpu.py -x -q -m gpu 2>&1 | tail -6
b() { env $1 python bench.py --mode train --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
print('$1', 'train ms_per_step', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median',0),3), 'slack', round(d.get('host_slack_at_sync_ms'),2), 'bwd', d['stages_ms']['backward'])"; }
for i in 1 2 3; do
b "X=1"
b "BRCNN_BN_REDUCE_DEFER=0"
done 2>&1 | tee gpurun_out/r06/ab_bndefer.log
