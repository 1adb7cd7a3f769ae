cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_f16_gpu.py tests/test_train_gpu.py tests/test_stress_gpu.py -x -q -m gpu -k "wgrad or weight_grad or train_step or reproduc or backward" 2>&1 | tail -3
python tools/wgrad_pp_bench.py 75 2>/dev/null | tail -4
for i in 1 2; do python bench.py --mode train --train-dtype bf16 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d.get('train', d)
print('train ms', t['ms_per_step'], t['step_ms_median'])
"; done
