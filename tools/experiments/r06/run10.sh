cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
b() { env $1 python bench.py --mode train --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
s=d['stages_ms']
print('$1', 'train ms_per_step', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median',0),3), 'slack', round(d.get('host_slack_at_sync_ms'),2), 'rpn_bwd', s['rpn_loss_and_backward'], 'roi', s['roi_align'], 'fc', s['fc_head'], 'bwd', s['backward'])"; }
for i in 1 2 3; do
b "BRCNN_HOLD_RPN_WGRAD=1"
b "BRCNN_HOLD_RPN_WGRAD=2"
b "BRCNN_HOLD_RPN_WGRAD=0"
done 2>&1 | tee gpurun_out/r06/ab_hold.log
