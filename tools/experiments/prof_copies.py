import sys, os, torch, collections
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    with torch.no_grad():
        out = m.simple_test_device(img, metas, rescale=True)
    torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::clone', 'aten::to', 'aten::_to_copy', 'aten::contiguous', 'aten::fill_', 'aten::zero_'):
        c[(e.name, str(e.input_shapes)[:80])] += 1
for k, v in c.most_common(40):
    print(v, k)
