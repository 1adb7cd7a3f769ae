"""Localise the intermittent BatchNorm-gradient mismatch of the ResLayer backward pass (tests/test_stress_gpu.py).
python tools/experiments/race_hunt.py [reps]: runs the pass under several configurations and reports, per configuration,
how many repetitions differed from the first one, in which gradients and elements."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import brcnn  # noqa
from brcnn import autograd as A, blocks, lib as _lib
from brcnn.backbones import Bottleneck, ResLayer

DEV = 'cuda:0'
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
L = _lib.load()


class Load:
    def __init__(self):
        self.stream = torch.cuda.Stream(DEV)
        g = torch.Generator().manual_seed(1)
        self.a = [torch.randn(n, n, generator=g).to(DEV, torch.bfloat16) for n in (4096, 1536, 6144)]
        self.c = [torch.empty_like(a) for a in self.a]

    def push(self, i):
        with torch.cuda.stream(self.stream):
            for j in range(3):
                k = (i + j) % 3
                torch.mm(self.a[k], self.a[k], out=self.c[k])


def build(dtype, channels_last):
    torch.manual_seed(47)
    layer = ResLayer(Bottleneck, 512, 128, 4, 1).to(DEV)
    for m in layer.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    layer.eval()
    if channels_last:
        blocks.conv_weights_channels_last(layer)
    x = torch.randn(8, 50, 84, 512, device=DEV).to(dtype)
    return layer, x


def one_pass(layer, x, fused, dtype):
    saved = (A.FUSE_BN_BACKWARD_INTO_DGRAD, A.FUSE_RESIDUAL_BN_BACKWARD)
    A.FUSE_BN_BACKWARD_INTO_DGRAD = A.FUSE_RESIDUAL_BN_BACKWARD = fused
    try:
        layer.zero_grad()
        xd = x.clone().requires_grad_()
        out = layer.forward_nhwc(xd)
        go = torch.randn(out.shape, device=DEV, generator=torch.Generator(DEV).manual_seed(3)).to(dtype)
        out.backward(go)
        A.join_side_streams()
        return out.detach(), xd.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()}
    finally:
        A.FUSE_BN_BACKWARD_INTO_DGRAD, A.FUSE_RESIDUAL_BN_BACKWARD = saved


def run(name, dtype=torch.float16, side=True, load=True, channels_last=True, fused_modes=(True, False)):
    A.WGRAD_SIDE_STREAM = side
    layer, x = build(dtype, channels_last)
    ld = Load() if load else None
    first, bad = {}, {}
    detail = []
    try:
        for rep in range(REPS):
            fused = fused_modes[rep % len(fused_modes)]
            if ld:
                ld.push(rep)
            cur = one_pass(layer, x, fused, dtype)
            if fused not in first:
                first[fused] = cur
                continue
            ref = first[fused]
            if not torch.equal(cur[0], ref[0]):
                bad[('out', fused)] = bad.get(('out', fused), 0) + 1
            if not torch.equal(cur[1], ref[1]):
                bad[('dx', fused)] = bad.get(('dx', fused), 0) + 1
            for k, g in cur[2].items():
                if not torch.equal(g, ref[2][k]):
                    bad[(k, fused)] = bad.get((k, fused), 0) + 1
                    d = (g - ref[2][k]).flatten()
                    nz = d.nonzero().flatten()
                    if len(detail) < 12:
                        detail.append((rep, fused, k, tuple(g.shape), int(nz.numel()), nz[:8].tolist(), float(d.abs().max())))
    finally:
        A.WGRAD_SIDE_STREAM = True
    torch.cuda.synchronize()
    print(f'== {name}: {REPS} reps, mismatching (tensor, fused) -> count: {bad if bad else "none"}')
    for d in detail:
        print('   ', d)
    sys.stdout.flush()


if __name__ == '__main__':
    run('fp16, side stream, load, channels-last weights')
    run('fp16, side stream, NO load', load=False)
    run('fp16, NO side stream, load', side=False)
    run('fp16, side stream, load, contiguous weights (3x3 wgrad on the main stream)', channels_last=False)
    run('fp16, side stream, load, separate launches only', fused_modes=(False,))
    run('fp16, side stream, load, fused only', fused_modes=(True,))
    run('bf16, side stream, load', dtype=torch.bfloat16)
