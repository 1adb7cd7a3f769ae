"""bf16 layers with 64 / 128 output channels and the mid-size 3x3 layers under every tile of conv_igemm_bf16.hip"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import brcnn  # noqa
from brcnn import ops, lib
L = lib.load()


def timed(fn, n=11):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


layers = [(200, 336, 64, 64, 3), (200, 336, 256, 64, 1), (100, 168, 128, 128, 3), (100, 168, 512, 128, 1), (100, 168, 256, 128, 1),
          (50, 84, 256, 256, 3), (50, 84, 1024, 256, 1), (25, 42, 512, 512, 3), (25, 42, 2048, 512, 1), (50, 84, 256, 1024, 1)]
tiles = [0, 11, 21, 22, 81, 82, 164, 42, 381, 382, 322, 342, 8844]
for H, W, K, N, k in layers:
    M = 8 * H * W
    x = torch.randn(8, H, W, K, device='cuda').bfloat16()
    w = (torch.randn(N, k, k, K, device='cuda') * 0.05).bfloat16()
    sc = torch.rand(N, device='cuda') + 0.5; sh = torch.randn(N, device='cuda')
    fl = 2.0 * M * N * K * k * k
    line = f'M={M:6d} K={K * k * k:5d} N={N:4d} |'
    best = None
    for t in tiles:
        L.brcnn_conv_set_tile_bf16(t)
        try:
            ops.conv2d_nhwc(x, w, scale=sc, shift=sh, relu=True, pad=k // 2)
        except Exception:
            continue
        ms = timed(lambda: ops.conv2d_nhwc(x, w, scale=sc, shift=sh, relu=True, pad=k // 2))
        line += f' {t}:{ms * 1e3:6.1f}'
        if t == 0: base = ms
        if best is None or ms < best[1]: best = (t, ms)
    L.brcnn_conv_set_tile_bf16(0)
    print(line + f' | heuristic {fl / base / 1e9:5.0f} TF/s, best tile {best[0]} {100 * (base / best[1] - 1):+.0f} %', flush=True)
