cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "roi or early_rpn or train_step" 2>&1 | tail -4
b() { env $1 python bench.py --mode train --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
print('$1', 'train ms_per_step', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median',0),3), 'slack', round(d.get('host_slack_at_sync_ms'),2), 'stages', d.get('stages_ms'))"; }
for i in 1 2; do
b "X=1"
b "BRCNN_FUSE_FAN_IN=0 BRCNN_FC_PACK=0 BRCNN_ROI_ADDEND=0"
done 2>&1 | tee gpurun_out/r06/ab_glue2.log
python bench.py --steps 20 --warmup 5 2>gpurun_out/r06/bench_mid.err | tail -1 > gpurun_out/r06/bench_mid.json; python -c "
import json; d=json.load(open('gpurun_out/r06/bench_mid.json')); print({k:d[k] for k in ('value','ms_per_step','roofline')}); print(d['train']['value'], d['train']['ms_per_step'], d['train']['roofline'])"
