"""stem (7x7/s2 + BN + ReLU) and max-pool launches alone, batch 8 x 3 x 800 x 1344, every dtype"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import brcnn  # noqa
from brcnn import ops
DEV = 'cuda:0'
def timed(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
img = torch.randn(8, 3, 800, 1344, device=DEV)
wt = torch.randn(64, 3, 7, 7, device=DEV) / 12
sc, sh = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)
for dt in (torch.float32, torch.bfloat16, torch.float16):
    wp = ops.pack_stem_weight(wt, dt)
    y = ops.stem7x7s2_nchw(img, wp, sc, sh, True)
    print(str(dt), 'stem %.1f us' % timed(lambda: ops.stem7x7s2_nchw(img, wp, sc, sh, True)),
          'maxpool %.1f us' % timed(lambda: ops.maxpool3x3s2_nhwc(y)), end=' ')
    wq = ops.pack_stem_pool_weight(wt, dt)
    print('fused %.1f us' % timed(lambda: ops.stem7x7s2_pool_nchw(img, wq, sc, sh)))
