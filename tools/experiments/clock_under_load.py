"""Effective shader clock while the conv kernels run (no profiler attached): a one-lane probe kernel
on a second stream counts shader cycles against the constant 100 MHz counter."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import brcnn
from brcnn import ops, lib
L = lib.load()
def probe(work, ms=20):
    out = torch.zeros(2, dtype=torch.int64, device='cuda')
    side = torch.cuda.Stream()
    work(); torch.cuda.synchronize()
    L.brcnn_clock_probe(out.data_ptr(), int(ms * 1e5), side.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); n = 0
    while not side.query():
        work(); n += 1
    e1.record(); torch.cuda.synchronize()
    o = out.tolist()
    return o[1] / o[0] * 0.1, n, e0.elapsed_time(e1)
idle = probe(lambda: None)
print('idle-ish (probe only)         %.3f GHz' % idle[0])
for dt in ('f32', 'bf16'):
    N, H, W, Ci, Co, k = 8, 100, 168, 256, 256, 3
    x = torch.randn(N, H, W, Ci, device='cuda'); w = torch.randn(Co, k, k, Ci, device='cuda') * 0.05
    if dt == 'bf16': x, w = x.bfloat16(), w.bfloat16()
    sc = torch.rand(Co, device='cuda') + 0.5; sh = torch.randn(Co, device='cuda')
    ghz, n, ms = probe(lambda: ops.conv2d_nhwc(x, w, sc, sh, None, True, 1, 1), ms=60)
    fl = 2.0 * N * H * W * Co * k * k * Ci
    tf = fl * n / ms / 1e9
    peak = (157.3 if dt == 'f32' else 2500.0) * ghz / 2.4
    print('%s 3x3 256->256 M=134400: %.3f GHz under load, %d launches in %.1f ms = %.1f TF/s = %.1f %% of the peak at that clock (%.0f TF/s)'
          % (dt, ghz, n, ms, tf, 100 * tf / peak, peak))
