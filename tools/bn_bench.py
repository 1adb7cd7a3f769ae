"""microbenchmark of the bn_act kernels (affine form vs eval-BN form): GB/s of algorithmic traffic"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import brcnn  # noqa
from brcnn.autograd import bn_act_autograd, bn_eval_act_autograd

dev = 'cuda:0'
for dtype in (torch.bfloat16, torch.float32):
    for rows, c in [(8 * 200 * 336, 256), (8 * 100 * 168, 512), (8 * 50 * 84, 1024), (8 * 25 * 42, 2048), (8 * 100 * 168, 128)]:
        z = torch.randn(rows, c, device=dev).to(dtype)
        res = torch.randn(rows, c, device=dev).to(dtype)
        bn = torch.nn.BatchNorm2d(c).to(dev).eval()
        scale, shift = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
        for name, fn in (('affine', lambda: bn_act_autograd(z, scale, shift, res, True)),
                         ('bn_eval', lambda: bn_eval_act_autograd(z, bn, res, True))):
            with torch.no_grad():
                for _ in range(3):
                    fn()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20):
                    fn()
                e.record()
                torch.cuda.synchronize()
            us = s.elapsed_time(e) / 20 * 1000
            nbytes = 3 * z.numel() * z.element_size()
            print(f'{str(dtype):16s} rows {rows:7d} C {c:5d} {name:8s} {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s')
