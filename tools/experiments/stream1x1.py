"""persistent short-K 1x1 kernel (conv1x1_stream_bf16.hip, hook brcnn_conv_set_tile_bf16(-17)) against the one-tile-per-workgroup
kernels (-15): bit equality on a set of shapes (ragged M, residual / ReLU, bf16 / fp16), then timings on the layers of the
batch-8 inference step"""
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


def run(x, w, sc, sh, r, relu, mode):
    assert L.brcnn_conv_set_tile_bf16(mode) == 0
    try:
        return ops.conv2d_nhwc(x, w, scale=sc, shift=sh, residual=r, relu=relu)
    finally:
        L.brcnn_conv_set_tile_bf16(-16)


bad = 0
g = torch.Generator().manual_seed(3)
for dt in (torch.bfloat16, torch.float16):
    for (n, h, w_, k, co, res, relu) in [(2, 64, 64, 64, 256, True, True), (1, 65, 67, 64, 128, False, True), (3, 37, 41, 128, 512, True, False),
                                         (8, 100, 168, 128, 512, True, True), (2, 50, 84, 64, 384, False, False), (1, 64, 65, 128, 128, True, True)]:
        x = torch.randn(n, h, w_, k, generator=g).to(dt).cuda()
        wt = (torch.randn(co, 1, 1, k, generator=g) * 0.05).to(dt).cuda()
        sc = (torch.rand(co, generator=g) + 0.5).cuda(); sh = torch.randn(co, generator=g).cuda()
        r = torch.randn(n, h, w_, co, generator=g).to(dt).cuda() if res else None
        a = run(x, wt, sc, sh, r, relu, -15)
        b = run(x, wt, sc, sh, r, relu, -17)
        torch.cuda.synchronize()
        same = torch.equal(a, b)
        bad += not same
        print(dt, (n, h, w_, k, co, res, relu), 'equal' if same else f'DIFFERENT max {float((a.float() - b.float()).abs().max()):.4g} at {int((a != b).sum())} of {a.numel()}', flush=True)
print('mismatching cases:', bad)
for H, W, K, N, res in [(200, 336, 64, 256, True), (200, 336, 64, 256, False), (100, 168, 128, 512, True), (100, 168, 128, 512, False), (200, 336, 64, 128, False)]:
    M = 8 * H * W
    x = torch.randn(8, H, W, K, device='cuda').bfloat16()
    wt = (torch.randn(N, 1, 1, K, device='cuda') * 0.05).bfloat16()
    sc = torch.rand(N, device='cuda') + 0.5; sh = torch.randn(N, device='cuda')
    r = torch.randn(8, H, W, N, device='cuda').bfloat16() if res else None
    by = (M * K + M * N * (2 if res else 1)) * 2
    line = f'M={M} K={K} N={N} res={int(res)} floor {by / 5.8e6:6.1f} us |'
    for mode, name in ((-15, 'tile per workgroup'), (-17, 'persistent')):
        ms = timed(lambda: run(x, wt, sc, sh, r, True, mode))
        line += f' {name}: {ms * 1e3:6.1f} us {by / ms / 1e9:5.2f} TB/s |'
    print(line, flush=True)
