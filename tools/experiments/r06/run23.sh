cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 1800 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06/pytest_gpu_h.txt 2>&1; echo "rc=$?" >> gpurun_out/r06/pytest_gpu_h.txt
grep -v "tensor(\|^E   *\[\|^  *\[" gpurun_out/r06/pytest_gpu_h.txt | tail -25 | cut -c1-220
for i in 1 2; do python bench.py --mode train --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
l=[x for x in sys.stdin if x.startswith('{')][-1]; d=json.loads(l)
s=d['stages_ms']
print('train ms_per_step', round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median',0),3), 'fc', s['fc_head'], 'roi', s['roi_align'], 'bwd', s['backward'])"; done
