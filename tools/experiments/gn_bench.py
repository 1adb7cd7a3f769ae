"""GroupNorm(+ReLU) forward / backward of the RPN tower tensor (batch 8, five pyramid levels of 800 x 1344, 256 channels,
32 groups) in isolation: HIP-event time per call and the streaming rate it corresponds to (forward: x read twice + y
written; backward: x, dy read twice + dx written).  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import brcnn  # noqa
from brcnn import ops

dev = torch.device('cuda:0')
sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
B, C, G = 8, 256, 32
rows = sum(B * h * w for h, w in sizes)
for dt in (torch.bfloat16, torch.float16, torch.float32):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(rows, C, generator=g).to(dev, dt)
    dy = torch.randn(rows, C, generator=g).to(dev, dt)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    es = x.element_size()

    def fwd():
        return ops.groupnorm_nhwc_multi(x, gamma, beta, G, B, sizes, 1e-5, True, return_stats=True)

    y, stats = fwd()

    def bwd():
        return ops.groupnorm_nhwc_multi_backward(dy, x, stats, gamma, beta, G, B, sizes, True)

    for name, fn, nbytes in (('forward', fwd, 3 * rows * C * es), ('backward', bwd, 5 * rows * C * es)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 40
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f'{str(dt):16s} {name:9s} {us:8.1f} us   {nbytes / us / 1e6:6.2f} TB/s of {nbytes / 1e6:.0f} MB')
