import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
import bench
from brcnn import profiling
m, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_utdac.py', torch.device('cuda', 0))
m = m.eval().freeze_for_inference()
m.set_compute_dtype(os.environ.get('BRCNN_DTYPE', 'f32'))
img, metas = bench.synthetic_batch(8, 'cuda')
for _ in range(2):
    with torch.no_grad(): m.simple_test_device(img, metas, rescale=True)
tab = profiling.per_layer_table(m, img, metas)
tot = sum(t[1] for t in tab)
agg = {}
for (shape, ms, tf, gbs) in tab:
    k = shape
    a = agg.setdefault(k, [0, 0.0, tf, gbs])
    a[0] += 1; a[1] += ms
print(f'total conv ms {tot:.2f}')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K = k
    print(f'M={M:8d} N={N:5d} K={K:6d} x{a[0]:2d}  {a[1]:7.3f} ms  {a[2]:6.1f} TF/s  {a[3]:7.0f} GB/s')
# whole-step breakdown with events
import time
def timeit(fn, n=3):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1000, r
with torch.no_grad():
    t_feat, feats = timeit(lambda: m.extract_feat_nhwc(img))
    t_rpn, outs = timeit(lambda: m.rpn_head.forward_nhwc(list(feats)))
    t_prop, (dets, num) = timeit(lambda: m.rpn_head.get_bboxes_padded(*outs, metas))
    t_roi, _ = timeit(lambda: m.roi_head.simple_test_padded(feats, dets, num, metas, rescale=True))
print(f'backbone+neck {t_feat:.2f} ms, rpn tower {t_rpn:.2f}, proposals {t_prop:.2f}, roi head+nms {t_roi:.2f}')
