"""`FusedSGD`: torch.optim.SGD (momentum, weight decay; dampening 0, no nesterov) + mmcv's OptimizerHook
gradient clipping + the fp16 recipes' static loss scaling as ONE pass over the parameters on the HIP kernels
`brcnn_sgd_step` / `brcnn_pack_conv_weights_batch` (csrc/optim.hip).

The reference: `optimizer = dict(type='SGD', lr=.., momentum=0.9, weight_decay=0.0001)`,
`optimizer_config = dict(grad_clip=dict(max_norm=35, norm_type=2))` (configs/_base_/schedules/schedule_1x.py:2-3,
configs/boosting_rcnn/boosting_rcnn_r50_pafpn_1x_utdac.py:130) -> torch's multi-tensor kernels, ~40 launches per
step.  Here: squared norm (fixed-order two stages), clip factor and the skip-on-inf decision on the device, one
update pass, and -- for the conv weights registered with `register_conv_weights` -- the forward / data-gradient
operands of the NEXT step written in the compute dtype in the same call, so that the per-layer weight packing
launches of the next forward disappear.  Same `param_groups` / `state_dict()` layout as torch.optim.SGD
(`momentum_buffer`), so checkpoints are interchangeable with the reference's.
"""
import ctypes

import torch

from . import lib as _L
from .ops import _dt, _ptr, _stream


PACK_LINEAR = __import__('os').environ.get('BRCNN_FC_PACK', '1') != '0'       # (A/B switch: register_conv_weights)


def _dense_layout(t):
    """'contiguous' / 'channels_last' when `t` covers its storage densely in that order, else None"""
    if t.is_contiguous():
        return 'contiguous'
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return 'channels_last'
    return None


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, **kwargs):
        if dampening != 0 or nesterov:
            raise NotImplementedError('FusedSGD: dampening / nesterov are not used by the recipes')
        if kwargs.get('maximize', False):
            raise NotImplementedError('FusedSGD: maximize')
        defaults = dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=False)
        super().__init__(params, defaults)
        self._conv = []              # [(param, fwd_buf, dgrad_buf)] of registered conv / linear weights, per compute dtype
        self._fcperm = []            # [(param, fwd_buf, dgrad_buf, C, ph * pw)]: the first FC (column-permuted operands)
        self._conv_dtype = None
        self.ctl = None              # device [grad norm, applied factor, skipped]
        for g in self.param_groups:
            for p in g['params']:
                if not p.is_cuda:
                    raise _L.BrcnnHipError('FusedSGD runs on the HIP device only (move the model first)')

    # ---- conv weight operands of the next step -------------------------------------------------
    def register_conv_weights(self, module, dtype):
        """keep the packed forward / data-gradient operands of every trainable, ungrouped nn.Conv2d weight -- and of the
        nn.Linear weights of the box head -- of `module` current in `dtype` (the compute dtype): written by `step()`,
        consumed by `autograd.ConvNHWCFunction` through `weight._brcnn_pack`.  Weights stored channels-last
        (`blocks.conv_weights_channels_last`) are read in that layout.  A Linear tagged `_brcnn_fc_perm = (C, ph, pw)`
        (the first FC behind the RoI extractor: its columns are (C, ph, pw), the NHWC features multiply (ph, pw, C))
        gets its operands in the permuted K order (`brcnn_pack_fc_weight_permuted`, `weight._brcnn_pack_perm`)."""
        mult = 32 if dtype == torch.float32 else 64
        self._conv, self._fcperm, self._conv_dtype = [], [], dtype
        mine = {id(p) for g in self.param_groups for p in g['params']}
        for m in module.modules():
            if isinstance(m, torch.nn.Conv2d) and m.groups == 1 and id(m.weight) in mine and m.weight.requires_grad \
                    and m.weight.shape[0] % mult == 0 and _dense_layout(m.weight) is not None:
                w = m.weight
                co, ci, kh, kw = w.shape
                self._conv.append((w, torch.empty((co, kh, kw, ci), dtype=dtype, device=w.device),
                                   torch.empty((ci, kh, kw, co), dtype=dtype, device=w.device)))
            elif PACK_LINEAR and isinstance(m, torch.nn.Linear) and (getattr(m, '_brcnn_pack_linear', False) or
                                                       getattr(m, '_brcnn_fc_perm', None) is not None) and \
                    id(m.weight) in mine and m.weight.requires_grad and m.weight.shape[0] % mult == 0 and m.weight.shape[1] % mult == 0 and m.weight.is_contiguous() and \
                    m.weight.dtype == torch.float32:
                w = m.weight
                co, k = w.shape
                perm = getattr(m, '_brcnn_fc_perm', None)
                f = torch.empty((co, 1, 1, k), dtype=dtype, device=w.device)
                d = torch.empty((k, 1, 1, co), dtype=dtype, device=w.device)
                if perm is not None and perm[0] % 64 == 0 and perm[0] * perm[1] * perm[2] == k and perm[1] * perm[2] <= 255:
                    self._fcperm.append((w, f, d, int(perm[0]), int(perm[1] * perm[2])))
                elif perm is None:
                    self._conv.append((w, f, d))
        self._pack(None)
        return len(self._conv) + len(self._fcperm)

    def _pack(self, ctl):
        lib = _L.load()
        for w, f, d, c, pn in self._fcperm:
            st = lib.brcnn_pack_fc_weight_permuted(_ptr(w), _ptr(f), _ptr(d), w.shape[0], c, pn, _dt(f), _ptr(ctl), _stream())
            _L.check(st, 'brcnn_pack_fc_weight_permuted')
            w._brcnn_pack_perm = (w._version, self._conv_dtype, f, d)
        if not self._conv:
            return
        n = len(self._conv)
        ws = (ctypes.c_void_p * n)(*[w.data_ptr() for w, _, _ in self._conv])
        fw = (ctypes.c_void_p * n)(*[f.data_ptr() for _, f, _ in self._conv])
        dg = (ctypes.c_void_p * n)(*[d.data_ptr() for _, _, d in self._conv])
        dims = (ctypes.c_int * (4 * n))(*[int(v) for w, _, _ in self._conv for v in (tuple(w.shape) + (1, 1))[:4]])
        cl = (ctypes.c_int * n)(*[int(w.dim() == 4 and _dense_layout(w) == 'channels_last') for w, _, _ in self._conv])
        st = lib.brcnn_pack_conv_weights_batch(ws, fw, dg, dims, cl, n, _dt(self._conv[0][1]), _ptr(ctl), _stream())
        _L.check(st, 'brcnn_pack_conv_weights_batch')
        for w, f, d in self._conv:
            w._brcnn_pack = (w._version, self._conv_dtype, f, d)

    # ---- the step ------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, max_norm=None, loss_scale=1.0, skip_nonfinite=None):
        """one optimizer step.  `max_norm`: clip the global L2 gradient norm first (None / <= 0: no clipping);
        `loss_scale`: the gradients carry this factor (fp16 static loss scaling) -- they are unscaled inside the
        update, and a non-finite norm skips the whole step (what GradScaler.step does).  `skip_nonfinite`: that
        skip rule; default: only with a loss scale (`loss_scale != 1`) -- an fp32 / bf16 run lets the inf / NaN through
        to the weights as clip_grad_norm_ + torch.optim.SGD would, so that a divergence is visible.  Returns the
        device tensor [grad norm, applied factor, skipped]."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ps, gs, bs, lrs, wds, has = [], [], [], [], [], []
        momentum = None
        for group in self.param_groups:
            if momentum is None:
                momentum = group['momentum']
            elif momentum != group['momentum']:
                raise NotImplementedError('FusedSGD: one momentum for all parameter groups')
            for p in group['params']:
                if p.grad is None:
                    continue
                if _dense_layout(p) is None or p.dtype != torch.float32:
                    raise _L.BrcnnHipError('FusedSGD: fp32 parameters in a dense layout (contiguous or channels-last) expected')
                # the update is element-wise by memory offset: gradient and momentum buffer in the parameter's layout
                g = p.grad
                if g.dtype != torch.float32 or g.stride() != p.stride():
                    g = torch.empty_like(p).copy_(g)
                state = self.state[p]
                buf = state.get('momentum_buffer')
                if buf is not None and (buf.stride() != p.stride() or buf.dtype != torch.float32):
                    buf = state['momentum_buffer'] = torch.empty_like(p).copy_(buf)     # e.g. a loaded checkpoint
                # a new buffer starts at zero with has = 1: momentum * 0 + d == d is torch's first-step value, and a step
                # the device skips (non-finite norm) leaves a defined buffer behind instead of uninitialised memory
                if buf is None and group['momentum'] != 0:
                    buf = state['momentum_buffer'] = torch.zeros_like(p)
                has.append(0 if buf is None else 1)
                ps.append(p)
                gs.append(g)
                bs.append(buf)
                lrs.append(group['lr'])
                wds.append(group['weight_decay'])
        if not ps:
            return loss
        n = len(ps)
        dev = ps[0].device
        # backstop: the gradients written on the weight-gradient side stream are complete before they are read, also
        # when a backward pass raised before its end-of-pass join ran
        from . import autograd as _A
        _A.join_side_streams(dev)
        lib = _L.load()
        numel = (ctypes.c_int64 * n)(*[p.numel() for p in ps])
        nb = lib.brcnn_sgd_workspace_bytes(n, numel)
        ws = torch.empty((nb + 3) // 4, dtype=torch.float32, device=dev)
        self.ctl = torch.empty(3, dtype=torch.float32, device=dev)
        pp = (ctypes.c_void_p * n)(*[p.data_ptr() for p in ps])
        gp = (ctypes.c_void_p * n)(*[g.data_ptr() for g in gs])
        bp = (ctypes.c_void_p * n)(*[0 if b is None else b.data_ptr() for b in bs])
        st = lib.brcnn_sgd_step(pp, gp, bp, numel, (ctypes.c_float * n)(*lrs), (ctypes.c_float * n)(*wds),
                                (ctypes.c_int * n)(*has), n, float(momentum or 0.0),
                                float(max_norm) if max_norm else 0.0, 1.0 / float(loss_scale),
                                int((loss_scale != 1.0) if skip_nonfinite is None else bool(skip_nonfinite)), _ptr(ws), nb, _ptr(self.ctl),
                                _stream())
        _L.check(st, 'brcnn_sgd_step')
        for p in ps:        # the kernels wrote through raw pointers: tell autograd / the packed-operand caches
            torch.autograd.graph.increment_version(p)
        self._pack(self.ctl)
        from .autograd import grad_arena
        grad_arena.new_step()       # the next backward's weight-gradient storage: one zero fill
        return self.ctl if loss is None else loss

    @property
    def last_grad_norm(self):
        """device scalar: the global gradient norm of the last step (before clipping, after unscaling)"""
        return None if self.ctl is None else self.ctl[0]
