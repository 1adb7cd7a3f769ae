"""GroupNorm(+ReLU) forward / backward on the RPN tower's concatenated maps (batch 8, five levels, 256 channels, bf16):
time per call and the streaming rate of the bytes the passes have to move"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import brcnn  # noqa
from brcnn import ops
LV = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
rows = 8 * sum(h * w for h, w in LV)
C, G = 256, 32


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(rows, C, device='cuda').to(dt); dy = torch.randn(rows, C, device='cuda').to(dt)
    g = torch.rand(C, device='cuda') + 0.5; b = torch.randn(C, device='cuda') * 0.1
    esz = x.element_size()
    y, st = ops.groupnorm_nhwc_multi(x, g, b, G, 8, LV, relu=True, return_stats=True)
    ms = timed(lambda: ops.groupnorm_nhwc_multi(x, g, b, G, 8, LV, relu=True, return_stats=True))
    print(f'{dt} forward  (stats: read x; apply: read x, write y = 3 passes): {ms * 1e3:7.1f} us {3 * rows * C * esz / ms / 1e9:5.2f} TB/s')
    ms = timed(lambda: ops.groupnorm_nhwc_multi_backward(dy, x, st, g, b, G, 8, LV, True))
    print(f'{dt} backward (reduce: read x, dy; apply: read x, dy, write dx = 5 passes): {ms * 1e3:7.1f} us {5 * rows * C * esz / ms / 1e9:5.2f} TB/s')
