"""RPN head conv (3x3, 256 -> 54 outputs, M = 179 200) and other narrow-N layers on the fp32 tile shapes"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import brcnn  # noqa
from brcnn import ops, lib
L = lib.load()
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, N, H, W, Ci, Co, k in [('rpn head 3x3 256->54', 8, 140, 160, 256, 54, 3), ('s1 3x3 64->64', 8, 200, 336, 64, 64, 3),
                                 ('s1 1x1 256->64', 8, 200, 336, 256, 64, 1)]:
    x = torch.randn(N, H, W, Ci, device='cuda'); w = torch.randn(Co, k, k, Ci, device='cuda') * 0.05
    sh = torch.randn(Co, device='cuda')
    fl = 2.0 * N * H * W * Co * k * k * Ci
    out = []
    for wm, nt in ((0, 0), (1, 1), (2, 1), (2, 2)):
        L.brcnn_conv_set_tile(wm, nt)
        ms = bench(lambda: ops.conv2d_nhwc(x, w, None, sh, None, False, 1, k // 2))
        out.append(f'wm{wm}nt{nt}: {ms * 1000:7.1f} us {fl / ms / 1e9:6.1f} TF')
    L.brcnn_conv_set_tile(0, 0)
    print(f'{name:24s} ' + ' | '.join(out))
