"""RoIAlign forward time against the ORDER of the RoIs (same RoIs, permuted): does locality pay now that the kernel is
bound by fabric re-fetches?  python tools/experiments/roi_order.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import brcnn  # noqa
from brcnn import ops
from tests import util
DEV = 'cuda'
B = 8
strides = [8, 16, 32, 64, 128]
sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
g = torch.Generator().manual_seed(0)
feats = [torch.randn(B, h, w, 256, generator=g).to(DEV) for h, w in sizes]


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def keys(rois):
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lv = torch.floor(torch.log2(scale / 56 + 1e-6)).clamp(0, 4).long()
    st = torch.tensor(strides, dtype=torch.float32)[lv]
    yc = ((rois[:, 2] + rois[:, 4]) * 0.5 / st).long()           # centre row / column on the RoI's own level
    xc = ((rois[:, 1] + rois[:, 3]) * 0.5 / st).long()
    return rois[:, 0].long(), lv, yc, xc


def order(rois, kind):
    img, lv, yc, xc = keys(rois)
    if kind == 'as delivered (image-major, random inside)':
        return torch.arange(len(rois))
    if kind == 'image, level, y':
        k = ((img * 8 + lv) * 256 + yc) * 256 + xc
    elif kind == 'image, level, 12-row band, x':
        k = ((img * 8 + lv) * 32 + yc // 12) * 256 + xc
    elif kind == 'image, level, 6-row band, x':
        k = ((img * 8 + lv) * 64 + yc // 6) * 256 + xc
    elif kind == 'level, image, 12-row band, x':
        k = ((lv * 64 + img) * 32 + yc // 12) * 256 + xc
    elif kind == 'image, level, 16x16 tile (row-major tiles)':
        k = (((img * 8 + lv) * 32 + yc // 16) * 32 + xc // 16) * 1024 + (yc % 16) * 16 + (xc % 16)
    return torch.argsort(k, stable=True)


for per_img in (256, 512, 2000):
    K = per_img * B
    rois = util.rand_rois(K, B, 1333., 800., seed=per_img, min_size=16., max_size=800.)
    rois = rois[torch.argsort(rois[:, 0], stable=True)].contiguous()
    ref = None
    for kind in ['as delivered (image-major, random inside)', 'image, level, y', 'image, level, 12-row band, x',
                 'image, level, 6-row band, x', 'level, image, 12-row band, x', 'image, level, 16x16 tile (row-major tiles)']:
        perm = order(rois, kind)
        rg = rois[perm].contiguous().to(DEV)
        out = ops.roi_extract(feats, rg, 7, strides, 56, 0)
        out = out[0] if isinstance(out, tuple) else out
        if ref is None:
            ref = out
        else:
            inv = torch.empty_like(perm); inv[perm] = torch.arange(len(perm))
            assert torch.equal(out[inv.to(DEV)], ref)
        t = timed(lambda: ops.roi_extract(feats, rg, 7, strides, 56, 0))
        print(f'{per_img:5d} x {B}  {kind:48s} {t:8.1f} us', flush=True)
