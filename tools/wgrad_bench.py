"""weight-gradient kernel per layer shape and tile: time, TFLOP/s and the result against the 128x128 tile.
usage: python tools/wgrad_bench.py [bf16|f32] [tile ...]   (bf16 tiles: 1 = 64x64, 2 = 128x128, 4 = 256x256)"""
import sys, os, torch, ctypes
sys.path.insert(0, os.getcwd())
import brcnn
from brcnn import ops, lib
L = lib.load()
BF = len(sys.argv) > 1 and sys.argv[1] == 'bf16'
# bf16 tile ids; suffix 'a': reduce the M slices with fp32 atomics instead of slabs + second stage
tiles = sys.argv[2:] or ['0']
LV = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
shapes = [('tower 3x3 256->256 x5', 8, LV, 256, 256, 3, 1, 1), ('rpn_l0 3x3 256->256', 8, LV[:1], 256, 256, 3, 1, 1),
          ('s3 3x3 256->256', 8, LV[1:2], 256, 256, 3, 1, 1), ('s2 1x1 128->512', 8, LV[:1], 128, 512, 1, 1, 0),
          ('s3 1x1 1024->256', 8, LV[1:2], 1024, 256, 1, 1, 0), ('s3 1x1 256->1024', 8, LV[1:2], 256, 1024, 1, 1, 0),
          ('s4 3x3 512->512', 8, LV[2:3], 512, 512, 3, 1, 1), ('s4 1x1 2048->512', 8, LV[2:3], 2048, 512, 1, 1, 0),
          ('fc 12544->1024', 4096, [(1, 1)], 12544, 1024, 1, 1, 0), ('fc 1024->1024', 4096, [(1, 1)], 1024, 1024, 1, 1, 0),
          ('s2 3x3 128->128', 8, LV[:1], 128, 128, 3, 1, 1), ('pafpn 3x3 s2 256->256', 8, LV[:1], 256, 256, 3, 2, 1)]


def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for name, N, lv, Ci, Co, k, st, pd in shapes:
    outs = [ops.conv_out_size(H, W, k, k, st, pd) for H, W in lv]
    M = sum(N * ho * wo for ho, wo in outs)
    torch.manual_seed(0)
    x = torch.randn(sum(N * H * W for H, W in lv), Ci, device='cuda'); dy = torch.randn(M, Co, device='cuda')
    if BF: x, dy = x.bfloat16(), dy.bfloat16()
    hs = (ctypes.c_int * len(lv))(*[h for h, _ in lv]); ws = (ctypes.c_int * len(lv))(*[w for _, w in lv])
    fl = 2.0 * M * Co * k * k * Ci
    ref = None
    line = f'{name:24s} M={M:7d}'
    for ts in tiles:
        # tile id, optional 'a' (atomics) and 'pNN' (NN percent of a generation of workgroups per launch)
        base, _, pct = ts.partition('p')
        t = int(base.rstrip('a'))
        if BF:
            L.brcnn_conv_set_tile_wgrad_bf16(t)
            L.brcnn_conv_set_tile_wgrad_bf16(10 if base.endswith('a') else 11)
            L.brcnn_conv_set_tile_wgrad_bf16(2000 + int(pct or 75))
        dw = torch.zeros(Co, k, k, Ci, device='cuda')
        call = lambda: L.brcnn_conv2d_wgrad_nhwc_multi(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, len(lv), hs, ws, Ci, Co, k, k,
                                                       st, pd, 1 if BF else 0, None)
        assert call() == 0
        torch.cuda.synchronize()
        got = dw.clone()
        if ref is None: ref = got
        err = float((got - ref).abs().max() / ref.abs().max())
        ms = bench(call)
        line += f' | t{ts}: {ms * 1000:7.1f} us {fl / ms / 1e9:6.1f} TF err {err:.1e}'
    print(line)
