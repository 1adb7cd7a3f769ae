cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "roi" 2>&1 | tail -2
python tools/op_bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items():
    if 'roialign' in k: print(k, round(v['us'],1))"
