"""the fp32 layers of the inference step (batch 8 x 800 x 1344) under the workgroup tiles of conv_igemm.hip
(brcnn_conv_set_tile): 64x64 (default), 128x64, 128x128, and the eight-phase kernel forced"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import brcnn  # noqa
from brcnn import ops, lib
L = lib.load()


def timed(fn, n=9):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


layers = [(200, 336, 64, 256, 1, True), (200, 336, 64, 64, 1, False), (200, 336, 256, 64, 1, False), (200, 336, 64, 64, 3, False),
          (100, 168, 128, 512, 1, True), (100, 168, 512, 128, 1, False), (100, 168, 128, 128, 3, False), (100, 168, 256, 256, 1, False),
          (50, 84, 256, 1024, 1, True), (50, 84, 1024, 256, 1, False), (50, 84, 256, 256, 3, False), (50, 84, 512, 256, 1, False),
          (25, 42, 512, 2048, 1, True), (25, 42, 2048, 512, 1, False), (25, 42, 512, 512, 3, False), (100, 168, 256, 256, 3, False)]
tiles = [('64x64', (1, 1)), ('128x64', (2, 1)), ('128x128', (2, 2)), ('pp', (-2, 2))]
for H, W, K, N, k, res in layers:
    M = 8 * H * W
    x = torch.randn(8, H, W, K, device='cuda')
    w = torch.randn(N, k, k, K, device='cuda') * 0.05
    sc = torch.rand(N, device='cuda') + 0.5; sh = torch.randn(N, device='cuda')
    r = torch.randn(8, H, W, N, device='cuda') if res else None
    fl = 2.0 * M * N * K * k * k
    L.brcnn_conv_set_tile(0, 0); L.brcnn_conv_set_tile(-2, 1)
    ref = ops.conv2d_nhwc(x, w, scale=sc, shift=sh, residual=r, relu=True, pad=k // 2)
    ms = timed(lambda: ops.conv2d_nhwc(x, w, scale=sc, shift=sh, residual=r, relu=True, pad=k // 2))
    line = f'M={M:6d} K={K * k * k:5d} N={N:4d} res={int(res)} | heuristic {ms * 1e3:7.1f} us {fl / ms / 1e9:6.1f} TF |'
    for name, (a, b) in tiles:
        if a == -2:
            L.brcnn_conv_set_tile(0, 0); L.brcnn_conv_set_tile(-2, 2)
        else:
            L.brcnn_conv_set_tile(-2, 0); L.brcnn_conv_set_tile(a, b)
        try:
            y = ops.conv2d_nhwc(x, w, scale=sc, shift=sh, residual=r, relu=True, pad=k // 2)
        except Exception:
            line += f' {name}: n/a'
            continue
        same = torch.equal(y, ref)
        ms = timed(lambda: ops.conv2d_nhwc(x, w, scale=sc, shift=sh, residual=r, relu=True, pad=k // 2))
        line += f' {name}:{ms * 1e3:7.1f}{"" if same else "!"}'
    L.brcnn_conv_set_tile(0, 0); L.brcnn_conv_set_tile(-2, 1)
    print(line, flush=True)
