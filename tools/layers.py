import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd())
import bench
from brcnn import profiling
m, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_utdac.py', torch.device('cuda', 0))
m = m.eval().freeze_for_inference()
m.set_compute_dtype(os.environ.get('BRCNN_DTYPE', 'f32'))
if os.environ.get('BRCNN_TILE_HOOK'):       # e.g. -9 / -10: split-K of few-tile launches by heuristic / forced
    from brcnn import lib as _lib
    for v in os.environ['BRCNN_TILE_HOOK'].split(','):
        assert _lib.load().brcnn_conv_set_tile_bf16(int(v)) == 0
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
peak = profiling.FP32_MFMA_PEAK_TFLOPS if os.environ.get('BRCNN_DTYPE', 'f32') == 'f32' else profiling.BF16_MFMA_PEAK_TFLOPS
floor_all = 0.0
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K = k
    # floor of the launch: MFMA peak of the dtype or 6.3 TB/s over its algorithmic bytes (x, weights, residual, output once)
    per = a[1] / a[0]
    t_mfma = 2.0 * M * N * K / (peak * 1e12) * 1e3
    t_hbm = a[3] * 1e9 * per * 1e-3 / (profiling.HBM_STREAM_TBS * 1e12) * 1e3
    floor_all += max(t_mfma, t_hbm) * a[0]
    print(f'M={M:8d} N={N:5d} K={K:6d} x{a[0]:2d}  {a[1]:7.3f} ms  {a[2]:6.1f} TF/s  {a[3]:7.0f} GB/s  bound={"hbm " if t_hbm > t_mfma else "mfma"} '
          f'{max(t_mfma, t_hbm) / per:5.2f} of it')
print(f'sum of the per-launch floors {floor_all:.2f} ms = {floor_all / tot:.3f} of the measured conv time')
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
