cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15
python bench.py --mode inference --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inf', d['value'], d['ms_per_step'], d['roofline']['frac'])
"
python bench.py --mode train --train-dtype bf16 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d.get('train', d)
print('train ms', t['ms_per_step'], t['step_ms_median'])
"
