cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
python bench.py --mode inference --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inf', d['value'], d['ms_per_step'], d['roofline']['frac'], d['stages_ms'])
"
BRCNN_FUSE_BLOCK_TAIL=0 python bench.py --mode inference --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inf (two launches)', d['value'], d['ms_per_step'], d['roofline']['frac'])
"
python bench.py --mode inference --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inf', d['value'], d['ms_per_step'], d['roofline']['frac'])
"
