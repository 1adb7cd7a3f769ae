cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_bf16_gpu.py tests/test_f16_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "roi or gather or extract or early_rpn" 2>&1 | tail -3
python tools/op_bench.py 2>/dev/null > gpurun_out/ob.json; python -c "
import json
d=json.load(open('gpurun_out/ob.json'))
for k,v in d.items():
    if 'roialign' in k: print(k, round(v['us'],1))
"
