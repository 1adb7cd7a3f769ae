"""GroupNorm(+ReLU) forward / backward over the five-level RPN tower tensor (batch 8, 256 channels, 32 groups):
time per call (HIP events on the launch stream) and algorithmic GB/s (fwd: 2 passes read + 1 write; bwd: 2 x 2 reads
+ 1 write)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import brcnn  # noqa
from brcnn import ops

dev = 'cuda:0'
sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
B, C, G = 8, 256, 32
rows = sum(B * h * w for h, w in sizes)
for dtype in (torch.bfloat16, torch.float32):
    x = torch.randn(rows, C, device=dev).to(dtype)
    dy = torch.randn(rows, C, device=dev).to(dtype)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    y, stats = ops.groupnorm_nhwc_multi(x, gamma, beta, G, B, sizes, 1e-5, True, return_stats=True)
    fwd = lambda: ops.groupnorm_nhwc_multi(x, gamma, beta, G, B, sizes, 1e-5, True, return_stats=True)
    bwd = lambda: ops.groupnorm_nhwc_multi_backward(dy, x, stats, gamma, beta, G, B, sizes, True)
    for name, fn, streams in (('fwd', fwd, 3), ('bwd', bwd, 5)):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1000
        print(f'{str(dtype):16s} {name} {us:8.1f} us  {streams * x.numel() * x.element_size() / us / 1e3:7.1f} GB/s')
