"""stage-1 Bottleneck tail (3x3 64->64 + 1x1 64->256 + identity) as two launches and as one, fp32, batch 8 x 200 x 336"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import brcnn  # noqa
from brcnn import ops
DEV = 'cuda:0'
def timed(f, n=20):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
s2, b2 = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV)
s3, b3 = torch.rand(256, device=DEV) + 0.5, torch.randn(256, device=DEV)
for dt in (torch.float32, torch.bfloat16, torch.float16):
    x = torch.randn(8, 200, 336, 64, device=DEV).to(dt); idn = torch.randn(8, 200, 336, 256, device=DEV).to(dt)
    w2 = (torch.randn(64, 3, 3, 64, device=DEV) / 24).to(dt); w3 = (torch.randn(256, 1, 1, 64, device=DEV) / 8).to(dt)
    two = lambda: ops.conv2d_nhwc(ops.conv2d_nhwc(x, w2, s2, b2, None, True, 1, 1), w3, s3, b3, idn, True, 1, 0)
    one = lambda: ops.bottleneck_tail_nhwc(x, w2, s2, b2, w3, s3, b3, idn)
    print(str(dt), 'equal', torch.equal(two(), one()), 'two launches %.1f us   one launch %.1f us' % (timed(two), timed(one)))
