# A/B of the deferred slab reductions (csrc/wgrad_defer.hip) on one box: bf16 train step of bench.py, interleaved
for i in 1 2 3; do
for v in "0 16" "1 4" "1 16" "1 64"; do
set -- $v
echo "DEFER=$1 ITEMS=$2"
BRCNN_WGRAD_DEFER=$1 BRCNN_WGRAD_DEFER_ITEMS=$2 python bench.py --mode train --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d.get('train',d)
        print('  ms', round(d['ms_per_step'],3), 'median', t.get('step_ms_median'), 'host_enqueue', t.get('host_enqueue_ms_median'), 'slack', t.get('host_slack_at_sync_ms'), 'loss', t.get('loss_last_timed', t.get('loss')))
"
done; done
