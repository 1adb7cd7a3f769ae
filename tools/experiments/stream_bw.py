"""what a plain streaming kernel reaches on this box (bf16 add: 2 reads + 1 write) next to the short-K residual 1x1 layers
of the bf16 conv stack (x + residual read, out written): the gap is what a better-streaming short-K kernel could win"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import brcnn  # noqa
from brcnn import ops


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
    esz = 2 if dt == torch.bfloat16 else 4
    for M, N, K, hw in [(537600, 256, 64, (200, 336)), (134400, 512, 128, (100, 168)), (33600, 1024, 256, (50, 84)),
                        (537600, 64, 256, (200, 336))]:
        a = torch.randn(M, N, device='cuda').to(dt); b = torch.randn(M, N, device='cuda').to(dt); c = torch.empty_like(a)
        ms = timed(lambda: torch.add(a, b, out=c))
        line = f'{str(dt)[6:]:9s} M={M} N={N} K={K}: add {ms * 1e3:6.1f} us {3 * M * N * esz / ms / 1e9:6.2f} TB/s'
        ms = timed(lambda: c.copy_(a))
        line += f' | copy {ms * 1e3:6.1f} us {2 * M * N * esz / ms / 1e9:6.2f} TB/s'
        x = torch.randn(8, hw[0], hw[1], K, device='cuda').to(dt)
        w = (torch.randn(N, 1, 1, K, device="cuda") * 0.05).to(dt)
        sc = torch.rand(N, device='cuda') + 0.5; sh = torch.randn(N, device='cuda')
        res = b.view(8, hw[0], hw[1], N)
        for label, r in (('conv+res', res), ('conv', None)):
            ms = timed(lambda: ops.conv2d_nhwc(x, w, scale=sc, shift=sh, residual=r, relu=True))
            by = (M * K + M * N * (2 if r is not None else 1)) * esz
            line += f' | {label} {ms * 1e3:6.1f} us {by / ms / 1e9:6.2f} TB/s {2.0 * M * N * K / ms / 1e9:6.1f} TF'
        print(line, flush=True)
