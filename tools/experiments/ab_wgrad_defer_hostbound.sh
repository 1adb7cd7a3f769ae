# the same A/B with every launch made dearer for the host (rocprofv3 --kernel-trace: the step becomes launch-bound,
# profiles/r05_notes.md): what the deferral is for
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do
for v in 0 1; do
echo "under rocprofv3 --kernel-trace: DEFER=$v"
BRCNN_WGRAD_DEFER=$v rocprofv3 --kernel-trace -d /tmp/prof_ab_$v -- python3 $R/bench.py --mode train --steps 30 --warmup 8 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d.get('train',d)
        print('  ms', round(d['ms_per_step'],3), 'median', t.get('step_ms_median'), 'slack', t.get('host_slack_at_sync_ms'), 'loss', t.get('loss_last_timed', t.get('loss')))
"
done; done
