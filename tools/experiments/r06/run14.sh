cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06/pytest_gpu_e.txt 2>&1; echo "rc=$?" >> gpurun_out/r06/pytest_gpu_e.txt
tail -12 gpurun_out/r06/pytest_gpu_e.txt | cut -c1-300
for i in 1 2; do
for e in "X=1" "BRCNN_PYRAMID_BUFFER=0"; do
env $e python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l); t=d['train']
print('$e', 'inference img/s', round(d['value'],2), 'ms', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median',0),3), '| train ms', round(t['ms_per_step'],3), 'median', round(t.get('step_ms_median',0),3))"
done; done 2>&1 | tee gpurun_out/r06/ab_pyrbuf.log
