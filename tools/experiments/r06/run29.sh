cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r06/pytest_gpu_i.txt
cat gpurun_out/r06/pytest_gpu_i.txt
python bench.py --no-cpu-baseline > gpurun_out/r06/bench_i.json 2> gpurun_out/r06/bench_i.err
python - <<'P'
import json
d = json.loads(open('gpurun_out/r06/bench_i.json').read().strip().splitlines()[-1])
print('inference img/s', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline'].get('frac_of_bound'))
t = d.get('train', {})
print('train ms', t.get('ms_per_step'), t.get('step_ms_median'), 'img/s', t.get('value'))
for k in ('inference_bf16', 'bf16'):
    if k in d: print(k, json.dumps(d[k])[:300])
P
