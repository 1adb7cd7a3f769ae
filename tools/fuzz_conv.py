"""Randomised shapes through the conv kernels (forward, autograd) against float64 torch: run once on the GPU box
after kernel changes (python tools/fuzz_conv.py [cases] [seed])."""
import sys, os, random
sys.path.insert(0, os.getcwd())
import numpy as np, torch, torch.nn.functional as F
import brcnn
from brcnn import ops
from brcnn.autograd import conv2d_nhwc_autograd


def run(cases=150, seed=0, verbose=True):
    rng = random.Random(seed)
    g = torch.Generator().manual_seed(1)
    bad = 0
    for it in range(cases):
        dt = rng.choice([torch.float32, torch.float32, torch.bfloat16])
        cmul = 64 if dt == torch.bfloat16 else 32
        cin = cmul * rng.randint(1, 6)
        cout = rng.choice([4, 16, 20, 21, 54, 64, 96, 128, 160, 256, 320, 512]) if dt == torch.float32 else 8 * rng.randint(1, 48)
        k = rng.choice([1, 1, 3, 3, 3])
        stride = rng.choice([1, 1, 1, 2])
        pad = k // 2 if rng.random() < 0.8 else 0
        n = rng.randint(1, 3)
        h, w = rng.randint(k, 41), rng.randint(k, 47)
        res, relu, sc = rng.random() < 0.4, rng.random() < 0.5, rng.random() < 0.6
        x = torch.randn(n, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, k, k, generator=g) / np.sqrt(cin * k * k)
        scale = torch.rand(cout, generator=g) + 0.5 if sc else None
        shift = torch.randn(cout, generator=g)
        if dt == torch.bfloat16:
            x, wt = x.bfloat16().float(), wt.bfloat16().float()
        ho, wo = ops.conv_out_size(h, w, k, k, stride, pad)
        r = torch.randn(n, cout, ho, wo, generator=g) if res else None
        if r is not None and dt == torch.bfloat16:
            r = r.bfloat16().float()
            if cout % 8: r = None
        ref = F.conv2d(x.double(), wt.double(), None, stride, pad)
        if scale is not None: ref = ref * scale.double().view(1, -1, 1, 1)
        ref = ref + shift.double().view(1, -1, 1, 1)
        if r is not None: ref = ref + r.double()
        if relu: ref = ref.relu()
        d = lambda t: None if t is None else t.cuda()
        xg = x.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
        wg = wt.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
        rg = None if r is None else r.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
        y = ops.conv2d_nhwc(xg, wg, d(scale), d(shift), rg, relu, stride, pad).float().permute(0, 3, 1, 2).cpu().double()
        tol = 3e-5 if dt == torch.float32 else 1e-2
        err = (y - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
        ok = err < tol
        # autograd (fp32 / bf16), no residual
        xa = x.permute(0, 2, 3, 1).contiguous().to(dt).cuda().requires_grad_()
        wa = wt.cuda().requires_grad_()
        ya = conv2d_nhwc_autograd(xa, wa, None, stride, pad)
        go = torch.randn(n, cout, ho, wo, generator=g)
        if dt == torch.bfloat16: go = go.bfloat16().float()
        ya.backward(go.permute(0, 2, 3, 1).contiguous().to(dt).cuda())
        xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
        F.conv2d(xr, wr, None, stride, pad).backward(go.double())
        tg = 1e-4 if dt == torch.float32 else 2e-2
        e_dx = (xa.grad.float().permute(0, 3, 1, 2).cpu().double() - xr.grad).abs().max().item() / max(xr.grad.abs().max().item(), 1e-6)
        e_dw = (wa.grad.cpu().double() - wr.grad).abs().max().item() / max(wr.grad.abs().max().item(), 1e-6)
        ok = ok and e_dx < tg and e_dw < tg
        if not ok:
            bad += 1
            print('FAIL', dt, (n, cin, h, w, cout, k, stride, pad), 'res', res, 'relu', relu, 'err', err, e_dx, e_dw)
    return bad


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    failed = run(n, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print('cases', n, 'failed', failed)
