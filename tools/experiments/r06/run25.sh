cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for i in 1 2; do
for v in base new; do
  if [ $v = base ]; then export BRCNN_LIB_PATH=$GRAFT_REPO_ROOT/boosting-r-cnn_amd/lib/ab/libbrcnn_base.so; else unset BRCNN_LIB_PATH; fi
  echo "== $v $i"
  python tools/op_bench.py > gpurun_out/r06/ob_$v$i.json 2> gpurun_out/r06/ob_$v$i.err || tail -5 gpurun_out/r06/ob_$v$i.err
  python -c "
import sys, json
d = json.load(open('gpurun_out/r06/ob_$v$i.json'))
for k, v in d.items():
    if 'roialign' in k: print(k, round(v['us'], 1))
"
done; done
