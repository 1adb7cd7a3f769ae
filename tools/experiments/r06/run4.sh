cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06/pytest_gpu_a.txt 2>&1; echo "rc=$?" >> gpurun_out/r06/pytest_gpu_a.txt
tail -15 gpurun_out/r06/pytest_gpu_a.txt
python tools/experiments/r06/aten_train.py 2>/dev/null > gpurun_out/r06/aten_train2.txt
head -60 gpurun_out/r06/aten_train2.txt
for i in 1 2; do python bench.py --mode train --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
print('train ms_per_step', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median',0),3), 'loss', d.get('loss_last_timed_step'))"; done
