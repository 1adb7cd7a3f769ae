"""microbenchmark of the eval-BN + ReLU kernels (`brcnn_bn_eval_act_forward/backward`) on the ResNet-50
stage 2-4 shapes of the train step: algorithmic GB/s per launch (HIP events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import brcnn  # noqa
from brcnn import lib as L
from brcnn.ops import _ptr, _dt

dev = 'cuda:0'
lib = L.load()
NOZ = os.environ.get('BN_READ_OUT') == '1'
tot = {}
for dtype in (torch.bfloat16, torch.float32):
    for rows, c, cnt_plain, cnt_res in [(8 * 100 * 168, 128, 8, 0), (8 * 100 * 168, 512, 1, 4), (8 * 50 * 84, 256, 12, 0),
                                        (8 * 50 * 84, 1024, 1, 6), (8 * 25 * 42, 512, 6, 0), (8 * 25 * 42, 2048, 1, 3)]:
        z = torch.randn(rows, c, device=dev).to(dtype)
        res = torch.randn(rows, c, device=dev).to(dtype)
        dout = torch.randn(rows, c, device=dev).to(dtype)
        out, dz, dres = torch.empty_like(z), torch.empty_like(z), torch.empty_like(z)
        g, b, m = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev), torch.randn(c, device=dev)
        v = torch.rand(c, device=dev) + 0.5
        dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
        nb = lib.brcnn_bn_act_backward_workspace_bytes(rows, c, _dt(z))
        ws = torch.empty(max(nb, 4), dtype=torch.uint8, device=dev)
        for has_res, cnt in ((False, cnt_plain), (True, cnt_res)):
            if cnt == 0:
                continue
            fwd = lambda: lib.brcnn_bn_eval_act_forward(_ptr(z), _ptr(g), _ptr(b), _ptr(m), _ptr(v), 1e-5, _ptr(res) if has_res else None,
                                                        _ptr(out), rows, c, 1, _dt(z), None)
            bwd = lambda: lib.brcnn_bn_eval_act_backward(_ptr(dout), _ptr(out) if has_res or NOZ else None, _ptr(z), _ptr(g), _ptr(b), _ptr(m), _ptr(v), 1e-5, _ptr(dz),
                                                         _ptr(dres) if has_res else None, _ptr(dg), _ptr(db), _ptr(ws), nb, rows, c, 1,
                                                         _dt(z), None)
            for name, fn, streams in (('fwd', fwd, 3 if has_res else 2), ('bwd', bwd, 5 if has_res else (4 if NOZ else 3))):
                for _ in range(3):
                    assert fn() == 0
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(20):
                    fn()
                e.record()
                torch.cuda.synchronize()
                us = s.elapsed_time(e) / 20 * 1000
                nbytes = streams * z.numel() * z.element_size()
                tot[(str(dtype), name)] = tot.get((str(dtype), name), 0.0) + us * cnt
                print(f'{str(dtype):16s} rows {rows:7d} C {c:5d} res {int(has_res)} {name} {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s  x{cnt}')
for k, v in tot.items():
    print(k, f'{v / 1000:.3f} ms per step (R50 stages 2-4)')
