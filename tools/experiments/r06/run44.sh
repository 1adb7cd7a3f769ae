cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
for v in 0 1 0 1; do
BRCNN_FUSE_BLOCK_TAIL=$v python bench.py --mode train --train-dtype bf16 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d.get('train', d)
print('fuse=$v train ms', round(t['ms_per_step'],3), round(t['step_ms_median'],3))
"
done
for v in 0 1; do
BRCNN_FUSE_BLOCK_TAIL=$v python bench.py --mode inference --dtype bf16 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$v inf bf16', round(d['value'],1), round(d['ms_per_step'],3))
"
done
