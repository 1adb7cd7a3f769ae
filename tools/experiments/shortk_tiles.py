"""the 1x1 layers of the bf16 conv stack (batch 8 x 800 x 1344) under every workgroup tile of conv_igemm_bf16.hip
(brcnn_conv_set_tile_bf16): is the heuristic's choice the best one, and how far is the best from the streaming rate"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import brcnn  # noqa
from brcnn import ops, lib
L = lib.load()


def timed(fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] != 'f16') else torch.float16
layers = [(200, 336, 64, 256, True), (200, 336, 64, 64, False), (200, 336, 256, 64, False), (100, 168, 128, 512, True), (100, 168, 512, 128, False),
          (50, 84, 256, 1024, True), (50, 84, 1024, 256, False), (25, 42, 512, 2048, True), (25, 42, 2048, 512, False),
          (100, 168, 256, 256, False), (50, 84, 512, 256, False)]
tiles = [0, 11, 21, 22, 81, 82, 164, 42, 381, 382, 322, 342, 8844]
for H, W, K, N, res in layers:
    M = 8 * H * W
    x = torch.randn(8, H, W, K, device='cuda').to(dt)
    w = (torch.randn(N, 1, 1, K, device='cuda') * 0.05).to(dt)
    sc = torch.rand(N, device='cuda') + 0.5; sh = torch.randn(N, device='cuda')
    r = torch.randn(8, H, W, N, device='cuda').to(dt) if res else None
    by = (M * K + M * N * (2 if res else 1) + N * K) * 2
    line = f'M={M:6d} K={K:4d} N={N:4d} res={int(res)} floor {by / 5.8e6:6.1f} us |'
    ref = None
    for t in tiles:
        L.brcnn_conv_set_tile_bf16(t)
        try:
            y = ops.conv2d_nhwc(x, w, scale=sc, shift=sh, residual=r, relu=True)
        except Exception:
            line += f' {t}: n/a'
            continue
        if ref is None: ref = y
        same = torch.equal(y, ref)
        ms = timed(lambda: ops.conv2d_nhwc(x, w, scale=sc, shift=sh, residual=r, relu=True))
        line += f' {t}:{ms * 1e3:6.1f}{"" if same else "!"}'
    L.brcnn_conv_set_tile_bf16(0)
    print(line, flush=True)
