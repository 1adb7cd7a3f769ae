"""split-K of few-tile launches (eight-phase kernels): time and result against the unsplit launch, fp32 and bf16"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import brcnn
from brcnn import ops, lib
L = lib.load()


def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


shapes = [('s4 3x3 512->512', 8, 25, 42, 512, 512, 3), ('p4 3x3 256->256', 8, 25, 42, 256, 256, 3),
          ('s4 1x1 2048->512', 8, 25, 42, 2048, 512, 1), ('s4 1x1 512->2048', 8, 25, 42, 512, 2048, 1),
          ('s3 3x3 256->256', 8, 50, 84, 256, 256, 3), ('s4 1x1 1024->2048', 8, 25, 42, 1024, 2048, 1)]
for dt in ('f32', 'bf16'):
    for name, N, H, W, Ci, Co, k in shapes:
        torch.manual_seed(0)
        x = torch.randn(N, H, W, Ci, device='cuda'); w = torch.randn(Co, k, k, Ci, device='cuda') * 0.03
        sc = torch.rand(Co, device='cuda') + 0.5; sh = torch.randn(Co, device='cuda')
        if dt == 'bf16':
            x, w = x.bfloat16(), w.bfloat16()
        fl = 2.0 * N * H * W * Co * Ci * k * k
        call = lambda: ops.conv2d_nhwc(x, w, sc, sh, None, True, 1, k // 2)
        res = {}
        line = f'{dt} {name:20s}'
        for tag, force, par in (('auto', 1, -8), ('pp unsplit', 128, -8), ('pp split-K', 128, -9), ('pp256 split-K', 256, -9)):
            if dt == 'f32':
                assert L.brcnn_conv_set_tile(-2, force) == 0
            else:
                assert L.brcnn_conv_set_tile_bf16(8844 if force != 1 else 0) == 0
                if force == 128: continue
            assert L.brcnn_conv_set_tile_bf16(par) == 0
            y = call(); y2 = call()
            res[tag] = y
            us = bench(call)
            err = (y.float() - res['auto'].float()).abs().max().item() / res['auto'].float().abs().max().item()
            line += f' | {tag}: {us:7.1f} us {fl / us / 1e6:6.1f} TF err {err:.1e} rep {torch.equal(y, y2)}'
        print(line)
        L.brcnn_conv_set_tile(-2, 1); L.brcnn_conv_set_tile_bf16(0); L.brcnn_conv_set_tile_bf16(-9)
