"""the weight-gradient shapes of the bf16 train step (batch 8 x 800 x 1344, COCO recipe): two-buffer kernel
(conv_wgrad_bf16.hip, heuristic tile) against the eight-phase 256 x 256 kernel (conv_wgrad_pp_bf16.hip), each with its
slab reduction; time per launch (HIP events over 10 launches), TFLOP/s, and the distance between the two results.
usage: python tools/wgrad_pp_bench.py [pct ...]   (percent of the CUs per eight-phase launch; default 75 100)"""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
import brcnn  # noqa
from brcnn import lib, ops
L = lib.load()
pcts = [int(v) for v in sys.argv[1:]] or [75, 100]
LV = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
shapes = [('tower 3x3 256->256 x5 levels', 8, LV, 256, 256, 3, 1, 1, 4), ('neck 3x3 256->256 50x84', 8, LV[1:2], 256, 256, 3, 1, 1, 9),
          ('neck 3x3 256->256 100x168', 8, LV[:1], 256, 256, 3, 1, 1, 1), ('s3 1x1 1024->256', 8, LV[1:2], 1024, 256, 1, 1, 0, 6),
          ('s3 1x1 256->1024', 8, LV[1:2], 256, 1024, 1, 1, 0, 6), ('s4 3x3 512->512', 8, LV[2:3], 512, 512, 3, 1, 1, 3),
          ('s4 1x1 512->2048', 8, LV[2:3], 512, 2048, 1, 1, 0, 3), ('s4 1x1 2048->512', 8, LV[2:3], 2048, 512, 1, 1, 0, 2),
          ('lateral 1x1 512->256 100x168', 8, LV[:1], 512, 256, 1, 1, 0, 2), ('s3 down 1x1 512->1024 s2', 8, LV[:1], 512, 1024, 1, 2, 0, 1),
          ('s3 1x1 512->256 100x168', 8, LV[:1], 512, 256, 1, 1, 0, 1), ('lateral 1x1 1024->256', 8, LV[1:2], 1024, 256, 1, 1, 0, 1),
          ('neck 3x3 256->256 25x42', 8, LV[2:3], 256, 256, 3, 1, 1, 3), ('pafpn 3x3 s2 256->256', 8, LV[:1], 256, 256, 3, 2, 1, 1),
          ('fc 12544->1024', 4096, [(1, 1)], 12544, 1024, 1, 1, 0, 1), ('fc 1024->1024', 4096, [(1, 1)], 1024, 1024, 1, 1, 0, 1),
          ('s2 3x3 128->128 (not taken)', 8, LV[:1], 128, 128, 3, 1, 1, 0),
          ('diagnostic: plain 1x1 2304->256, M of the tower', 8, [(140, 160)], 2304, 256, 1, 1, 0, 0),
          ('diagnostic: 3x3 256->256 one map 140x160', 8, [(140, 160)], 256, 256, 3, 1, 1, 0)]


def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


tot = {}
for name, N, lv, Ci, Co, k, st, pd, times in shapes:
    outs = [ops.conv_out_size(H, W, k, k, st, pd) for H, W in lv]
    M = sum(N * ho * wo for ho, wo in outs)
    torch.manual_seed(0)
    x = torch.randn(sum(N * H * W for H, W in lv), Ci, device='cuda').bfloat16()
    dy = torch.randn(M, Co, device='cuda').bfloat16()
    hs = (ctypes.c_int * len(lv))(*[h for h, _ in lv]); ws = (ctypes.c_int * len(lv))(*[w for _, w in lv])
    fl = 2.0 * M * Co * k * k * Ci
    h = lib.stream_handle()
    line = f'{name:30s} M={M:7d} x{times}'
    ref = None
    for label, hooks in [('two-buffer', [20])] + [(f'pp {p}%{"" if f else " sep"}', [22, 4000 + p, 30 + f]) for p in pcts for f in (1, 0)]:
        for hk in hooks:
            assert L.brcnn_conv_set_tile_wgrad_bf16(hk) == 0
        dw = torch.zeros(Co, k, k, Ci, device='cuda')
        call = lambda: L.brcnn_conv2d_wgrad_nhwc_multi(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), N, len(lv), hs, ws, Ci, Co, k, k,
                                                       st, pd, 1, h)
        assert call() == 0
        torch.cuda.synchronize()
        got = dw.clone()
        if ref is None: ref = got
        err = float((got - ref).abs().max() / ref.abs().max())
        ms = bench(call)
        tot[label] = tot.get(label, 0.0) + ms * times
        line += f' | {label}: {ms * 1000:7.1f} us {fl / ms / 1e9:6.1f} TF d {err:.0e}'
    print(line, flush=True)
L.brcnn_conv_set_tile_wgrad_bf16(21)
L.brcnn_conv_set_tile_wgrad_bf16(4075)
print('per step (launch counts of the train step):', {k: round(v, 3) for k, v in tot.items()}, 'ms')
