# HIP_FORCE_DEV_KERNARG (kernel arguments in device memory instead of host-coherent memory: a launch's argument fetch
# does not cross the host link): bf16 train step (~770 launches) and the fp32 inference pass, interleaved
for i in 1 2 3; do
for v in 0 1; do
echo "HIP_FORCE_DEV_KERNARG=$v"
HIP_FORCE_DEV_KERNARG=$v python bench.py --mode train --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d.get('train',d)
        print('  train ms', round(d['ms_per_step'],3), 'median', round(t.get('step_ms_median'),3), 'slack', round(t.get('host_slack_at_sync_ms'),2), 'loss', t.get('loss_last_timed', t.get('loss')))
"
done; done
for v in 0 1; do
echo "HIP_FORCE_DEV_KERNARG=$v"
HIP_FORCE_DEV_KERNARG=$v python bench.py --mode inference --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)
        print('  inference ms', round(d['ms_per_step'],3), 'bs1', d.get('latency_bs1_ms'))
"
done
