"""RoIAlign forward: the prepared-record form (round 5) against the footprint form it replaces, at the op-bench sizes, fp32
and bf16 pyramids; equality of the two and the capped fraction of 8 TB/s (pyramid counted once + output).
python tools/experiments/roi_prepared.py"""
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
feats32 = [torch.randn(B, h, w, 256, generator=g).to(DEV) for h, w in sizes]


def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for dt in (torch.float32, torch.bfloat16):
    feats = [f.to(dt) for f in feats32]
    es = feats[0].element_size()
    pyramid = sum(f[:, :, :, :].numel() for f in feats[:4]) * es
    for per_img in (256, 512, 1000, 2000):
        K = per_img * B
        rois = util.rand_rois(K, B, 1333., 800., seed=per_img, min_size=16., max_size=800.)
        rois = rois[torch.argsort(rois[:, 0], stable=True)].contiguous().to(DEV)
        capped = pyramid + K * 49 * 256 * es
        L.brcnn_roi_align_set_exact(30)
        ref, lref = ops.roi_extract(feats, rois, 7, strides, 56, 0)
        t_old = timed(lambda: ops.roi_extract(feats, rois, 7, strides, 56, 0))
        L.brcnn_roi_align_set_exact(31)
        out, lv = ops.roi_extract(feats, rois, 7, strides, 56, 0)
        t_new = timed(lambda: ops.roi_extract(feats, rois, 7, strides, 56, 0))
        same = torch.equal(out, ref)
        worst = (out.float() - ref.float()).abs().max().item()
        frac_rows = (out != ref).flatten(1).any(1).float().mean().item()
        print(f'{str(dt)[6:]:9s} {per_img:5d} x {B}: footprint {t_old:7.1f} us ({capped / t_old / 8e6:.3f} of 8 TB/s capped)   prepared '
              f'{t_new:7.1f} us ({capped / t_new / 8e6:.3f})   levels equal {torch.equal(lv, lref)}  bits equal {same} '
              f'(max |d| {worst:.2e}, RoIs that differ {frac_rows:.4f})', flush=True)
L.brcnn_roi_align_set_exact(30)
