"""GroupNorm(32)+ReLU of the RPN tower (all 5 levels, batch 8): forward and backward timings"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import brcnn
from brcnn import ops
sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
B, C, G = 8, 256, 32
rows = sum(B * h * w for h, w in sizes)
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(rows, C, device='cuda').to(dt)
    dy = torch.randn(rows, C, device='cuda').to(dt)
    g = torch.rand(C, device='cuda') + 0.5
    b = torch.randn(C, device='cuda') * 0.1
    y, st = ops.groupnorm_nhwc_multi(x, g, b, G, B, sizes, 1e-5, True, return_stats=True)
    f = bench(lambda: ops.groupnorm_nhwc_multi(x, g, b, G, B, sizes, 1e-5, True))
    bw = bench(lambda: ops.groupnorm_nhwc_multi_backward(dy, x, st, g, b, G, B, sizes, True))
    nb = rows * C * x.element_size()
    print(f'{dt}: fwd {f:.1f} us ({3 * nb / f / 1e6:.2f} TB/s of 3 streams), bwd {bw:.1f} us ({5 * nb / bw / 1e6:.2f} TB/s of 5 streams)')
