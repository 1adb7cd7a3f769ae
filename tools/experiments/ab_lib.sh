# same-box A/B of two builds of libbrcnn_hip.so: the in-tree one against boosting-r-cnn_amd/lib_prev/libbrcnn_hip.so (built from
# an older commit in a git worktree and copied there; *.so is git-ignored but travels with gpurun).  Alternates the two,
# prints the headline (fp32 inference), the bf16 inference and the bf16 train step of each run.
# usage (on the GPU box): bash tools/experiments/ab_lib.sh [rounds]
R=${1:-2}
one() {
  python bench.py --no-cpu-baseline $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); t = d.get('train')
        print('$1', d['dtype'], 'inference %.2f img/s %.3f ms' % (d['value'], d['ms_per_step']), ('train %.3f ms' % t['ms_per_step']) if t else '')
"
}
for i in $(seq $R); do
  BRCNN_LIB_PATH=$PWD/boosting-r-cnn_amd/lib_prev/libbrcnn_hip.so BRCNN_FUSE_BN3_BWD=0 one prev ""
  one new ""
  BRCNN_LIB_PATH=$PWD/boosting-r-cnn_amd/lib_prev/libbrcnn_hip.so BRCNN_FUSE_BN3_BWD=0 one prev "--dtype bf16 --mode inference"
  one new "--dtype bf16 --mode inference"
done
