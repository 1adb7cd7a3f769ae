for i in 1 2 3; do
for v in 0 1; do
echo "PACK_STREAM=$v"
BRCNN_PACK_STREAM=$v python bench.py --mode train --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d.get('step_stats',{}).get('step_ms_median'), d.get('loss_last_timed'), {k:v for k,v in d.items() if 'slack' in k})
"
done; done
