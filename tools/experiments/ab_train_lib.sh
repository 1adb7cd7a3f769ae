# same-box A/B of the bf16 train step under two builds of the library (see ab_lib.sh): usage: bash tools/experiments/ab_train_lib.sh <dir of the other build> [rounds]
P=$PWD/boosting-r-cnn_amd/$1/libbrcnn_hip.so; R=${2:-3}
one() { python bench.py --no-cpu-baseline --mode train --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); t = d.get('train') or d
        print('$1 train %.3f ms' % t['ms_per_step'])
"; }
for i in $(seq $R); do BRCNN_LIB_PATH=$P one other; one new; done
