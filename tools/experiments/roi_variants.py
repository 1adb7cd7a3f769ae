"""RoIAlign forward: rows per wave x visiting order, at the op-bench sizes (python tools/experiments/roi_variants.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import brcnn  # noqa
from brcnn import ops, lib
from tests import util
L = lib.load()
DEV = 'cuda'
B = 8
strides = [8, 16, 32, 64, 128]
sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
g = torch.Generator().manual_seed(0)
feats = [torch.randn(B, h, w, 256, generator=g).to(DEV) for h, w in sizes]


def timed(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for per_img in (256, 512, 1000, 2000):
    K = per_img * B
    rois = util.rand_rois(K, B, 1333., 800., seed=per_img, min_size=16., max_size=800.)
    rois = rois[torch.argsort(rois[:, 0], stable=True)].contiguous().to(DEV)
    L.brcnn_roi_align_set_exact(11); L.brcnn_roi_align_set_exact(20)
    ref, lref = ops.roi_extract(feats, rois, 7, strides, 56, 0)
    for rpw, name_r in ((11, 'row/wave'), (17, 'RoI/wave')):
        for od, name_o in ((20, 'as given'), (22, 'band order')):
            L.brcnn_roi_align_set_exact(rpw); L.brcnn_roi_align_set_exact(od)
            out, lv = ops.roi_extract(feats, rois, 7, strides, 56, 0)
            assert torch.equal(out, ref) and torch.equal(lv, lref), (per_img, name_r, name_o)
            t = timed(lambda: ops.roi_extract(feats, rois, 7, strides, 56, 0))
            print(f'{per_img:5d} x {B}  {name_r:9s} {name_o:10s} {t:8.1f} us', flush=True)
L.brcnn_roi_align_set_exact(10); L.brcnn_roi_align_set_exact(21)
