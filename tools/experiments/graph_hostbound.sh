# does the graph-replayed trunk win where the host cannot keep up?  bf16 train leg under rocprofv3 --kernel-trace (every
# launch dearer for the host, zero slack at the sampler's synchronisation), eager against BRCNN_BENCH_GRAPH_TRUNK=1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do
for v in 0 1; do
echo "under rocprofv3 --kernel-trace: graph trunk=$v"
BRCNN_BENCH_GRAPH_TRUNK=$v timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_g_$v -- python3 $R/bench.py --mode train --steps 30 --warmup 8 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); t=d.get('train',d)
        print('  ms', round(d['ms_per_step'],3), 'median', round(t.get('step_ms_median'),3), 'slack', round(t.get('host_slack_at_sync_ms'),3), 'graph', t.get('graph_trunk'))
"
done; done
