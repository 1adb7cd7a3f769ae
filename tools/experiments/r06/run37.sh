cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "topk or proposal or rpn or golden or get_bboxes or fullsize or configs1" 2>&1 | tail -3
python tools/op_bench.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
for k, v in d.items():
    if 'topk' in k or 'nms' in k or 'sort' in k: print(k, round(v['us'], 1))
"
python bench.py --mode inference --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inf', d['value'], d['ms_per_step'], d['stages_ms'])
"
python bench.py --mode inference --dtype bf16 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inf bf16', d['value'], d['ms_per_step'], d['stages_ms'])
"
