"""Seeded synthetic weights: there is no network for checkpoints, so benches, smoke runs and the golden
fixtures all draw the parameters of a module from one CPU generator (the golden generator applies the
same function to the imported reference model, so both sides hold identical weights)."""
import torch


def seeded_state_dict(module, seed=0):
    """Deterministic synthetic weights (no checkpoint download is possible): every tensor of
    `module.state_dict()` is drawn, in sorted key order, from a CPU generator.  The golden
    generator applies the same function to the reference model, so both sides hold
    identical weights without shipping 184 MB of parameters."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    out = {}
    for k in sorted(sd.keys()):
        v = sd[k]
        if k.endswith('num_batches_tracked'):
            out[k] = torch.zeros_like(v)
            continue
        if k.endswith('running_mean'):
            out[k] = torch.randn(v.shape, generator=g) * 0.1
        elif k.endswith('running_var'):
            out[k] = torch.rand(v.shape, generator=g) + 0.5
        elif k.endswith('.scale'):
            out[k] = torch.tensor(1.0 + 0.1 * float(torch.randn((), generator=g)))
        elif v.dim() >= 2:       # conv / linear weights: He-style fan-in scaling
            fan_in = v[0].numel()
            std = (2.0 / fan_in) ** 0.5
            if 'rpn_cls' in k or 'rpn_iou' in k or 'rpn_reg' in k:
                std *= 0.5
            if 'fc_reg' in k:
                std *= 0.1
            out[k] = torch.randn(v.shape, generator=g) * std
        elif 'bn3.weight' in k:  # damp the residual branches so 16 blocks stay bounded
            out[k] = torch.rand(v.shape, generator=g) * 0.3 + 0.2
        elif k.endswith('.weight'):   # norm scales
            out[k] = torch.rand(v.shape, generator=g) * 0.5 + 0.75
        elif 'rpn_cls.bias' in k:
            out[k] = torch.randn(v.shape, generator=g) * 0.5 - 1.0
        else:                     # biases / norm shifts
            out[k] = torch.randn(v.shape, generator=g) * 0.05
        out[k] = out[k].to(v.dtype)
    return out
