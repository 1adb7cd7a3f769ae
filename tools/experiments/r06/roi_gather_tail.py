"""RoIAlign gradient gather against the RoI size mix: is the launch the serial chain of the coarse levels' tiles?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import brcnn  # noqa
from brcnn.autograd import roi_extract_autograd
from tests import util
DEV = 'cuda'
B = 8
strides = [8, 16, 32, 64, 128]
sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
g = torch.Generator().manual_seed(0)
feats = [torch.randn(B, h, w, 256, generator=g).to(DEV) for h, w in sizes]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ts[n // 2] * 1e3


for per_img in (512, 2000):
    for lo, hi in ((16., 800.), (16., 110.), (16., 220.), (230., 440.), (450., 800.)):
        K = per_img * B
        rois = util.rand_rois(K, B, 1333., 800., seed=per_img, min_size=lo, max_size=hi)
        rois = rois[torch.argsort(rois[:, 0], stable=True)].contiguous().to(DEV)
        fr = [f.clone().requires_grad_() for f in feats]
        go = torch.randn(K, 7, 7, 256, device=DEV)
        out = roi_extract_autograd(fr, rois, 7, strides, 56, 0)

        def bwd():
            for f in fr:
                f.grad = None
            out.backward(go, retain_graph=True)
        t_fb = timed(bwd)
        print(f'{per_img} x 8, sizes {lo:.0f}-{hi:.0f} px: backward (record + gather + 5 zero-size ops) {t_fb:8.1f} us', flush=True)
