"""torch.profiler view of one inference step: which host call sites launch the small kernels"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
from torch.profiler import profile, ProfilerActivity
m, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_utdac.py', torch.device('cuda', 0))
m.set_compute_dtype(os.environ.get('BRCNN_DTYPE', 'f32'))
img, metas = bench.synthetic_batch(8, 'cuda')
for _ in range(3):
    with torch.no_grad():
        out = m.simple_test_device(img, metas, rescale=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    with torch.no_grad():
        out = m.simple_test_device(img, metas, rescale=True)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(group_by_stack_n=8), key=lambda r: -r.self_device_time_total)
for r in rows[:45]:
    st = [f.split('/')[-1] for f in r.stack if 'boosting' in f or 'bench' in f][:3]
    print(f'{r.key[:48]:48s} n={r.count:4d} cuda={r.self_device_time_total:9.1f}us  ' + ' <- '.join(st))
