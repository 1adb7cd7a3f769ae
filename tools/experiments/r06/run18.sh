cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_train_gpu.py tests/test_bf16_gpu.py -x -q -m gpu -k "roi or early_rpn or train_step or device_path" 2>&1 | grep -v "tensor(\|^E   *\[\|^  *\[" | tail -25
python tools/experiments/r06/roi_gather_tail.py 2>/dev/null
python tools/op_bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items():
    if 'roialign' in k: print(k, round(v['us'],1))"
