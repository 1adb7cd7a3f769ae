"""TEST INFRASTRUCTURE ONLY -- ctypes front end of the C oracle (oracle/brcnn_oracle.c).

Mirrors the `mmcv.ops` Python API on CPU tensors so that (a) the parity tests can
compare the HIP path against it and (b) the golden-fixture generator can plug it
under the imported reference (`tests/golden/make_golden.py`) where mmcv-full is
absent.  The Python-level logic of `nms` / `batched_nms` / `soft_nms` restates
mmcv 1.4.0 `mmcv/ops/nms.py` (un-vendored dependency; call sites in the
reference: atss_rpn_head.py:756, bbox_nms.py:86, base_roi_extractor.py:54-60,
focal_loss.py:86).

Nothing in the product package imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, 'libbrcnn_oracle.so')
    src = os.path.join(_HERE, 'brcnn_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        f32p, i64p = ctypes.c_void_p, ctypes.c_void_p
        L.orc_roi_align_forward.argtypes = [f32p, f32p, f32p, f32p, f32p] + [ctypes.c_int] * 6 + \
            [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_roi_align_backward.argtypes = [f32p, f32p, f32p] + [ctypes.c_int] * 6 + \
            [ctypes.c_float, ctypes.c_int, ctypes.c_int]
        L.orc_roi_align_forward_range.argtypes = [f32p, f32p, f32p] + [ctypes.c_int] * 7 + \
            [ctypes.c_float, ctypes.c_int, ctypes.c_int]
        L.orc_nms.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_float, ctypes.c_int, i64p]
        L.orc_nms.restype = ctypes.c_int64
        L.orc_argsort_desc.argtypes = [f32p, ctypes.c_int64, i64p]
        L.orc_softnms.argtypes = [f32p, f32p, ctypes.c_int64, f32p, i64p, ctypes.c_float,
                                  ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int]
        L.orc_softnms.restype = ctypes.c_int64
        for fn in (L.orc_sigmoid_focal_loss_forward, L.orc_sigmoid_focal_loss_backward):
            fn.argtypes = [f32p, i64p, f32p, f32p, ctypes.c_int64, ctypes.c_int64,
                           ctypes.c_float, ctypes.c_float]
        L.orc_preprocess_u8.argtypes = [f32p, ctypes.c_int, ctypes.c_int, f32p] + [ctypes.c_int] * 5 + \
            [f32p, f32p, ctypes.c_int]
        _LIB = L
    return _LIB


def _f32(t):
    return t.detach().to('cpu', torch.float32).contiguous()


def _pair(x):
    return (x, x) if isinstance(x, int) else tuple(x)


# --------------------------------------------------------------------------- RoIAlign
def roi_align_forward(input, rois, output_size, spatial_scale=1.0, sampling_ratio=0,
                      pool_mode='avg', aligned=True):
    ph, pw = _pair(output_size)
    x, r = _f32(input), _f32(rois)
    assert r.dim() == 2 and r.size(1) == 5
    n, c, h, w = x.shape
    k = r.size(0)
    out = torch.zeros(k, c, ph, pw)
    mode = {'max': 0, 'avg': 1}[pool_mode]
    ay = torch.zeros_like(out) if mode == 0 else None
    ax = torch.zeros_like(out) if mode == 0 else None
    lib().orc_roi_align_forward(x.data_ptr(), r.data_ptr(), out.data_ptr(),
                                ay.data_ptr() if ay is not None else None,
                                ax.data_ptr() if ax is not None else None,
                                c, h, w, k, ph, pw, float(spatial_scale), int(sampling_ratio),
                                mode, int(bool(aligned)))
    return out


def roi_align_backward(grad_output, rois, input_shape, output_size, spatial_scale=1.0,
                       sampling_ratio=0, aligned=True):
    ph, pw = _pair(output_size)
    g, r = _f32(grad_output), _f32(rois)
    n, c, h, w = input_shape
    gi = torch.zeros(n, c, h, w)
    lib().orc_roi_align_backward(g.data_ptr(), r.data_ptr(), gi.data_ptr(), c, h, w, r.size(0),
                                 ph, pw, float(spatial_scale), int(sampling_ratio),
                                 int(bool(aligned)))
    return gi


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, rois, output_size, spatial_scale, sampling_ratio, pool_mode, aligned):
        ctx.save_for_backward(rois)
        ctx.cfg = (tuple(input.shape), output_size, spatial_scale, sampling_ratio, aligned)
        assert pool_mode == 'avg' or not input.requires_grad
        return roi_align_forward(input, rois, output_size, spatial_scale, sampling_ratio,
                                 pool_mode, aligned)

    @staticmethod
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        shape, output_size, spatial_scale, sampling_ratio, aligned = ctx.cfg
        gi = roi_align_backward(grad_output, rois, shape, output_size, spatial_scale,
                                sampling_ratio, aligned)
        return gi, None, None, None, None, None, None


def roi_align(input, rois, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg',
              aligned=True):
    return _RoIAlignFn.apply(input, rois, output_size, spatial_scale, sampling_ratio, pool_mode,
                             aligned)


class RoIAlign(torch.nn.Module):
    """mmcv.ops.RoIAlign signature (roi_layer=dict(type='RoIAlign', output_size=7,
    sampling_ratio=0) at base_roi_extractor.py:54-60)."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg',
                 aligned=True, use_torchvision=False):
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)
        self.pool_mode = pool_mode
        self.aligned = aligned

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio,
                         self.pool_mode, self.aligned)


# --------------------------------------------------------------------------- NMS family
def argsort_desc(scores):
    s = _f32(scores)
    order = torch.empty(s.numel(), dtype=torch.int64)
    lib().orc_argsort_desc(s.data_ptr(), s.numel(), order.data_ptr())
    return order


def nms(boxes, scores, iou_threshold, offset=0, score_threshold=0, max_num=-1):
    assert boxes.size(1) == 4 and boxes.size(0) == scores.size(0) and offset in (0, 1)
    b, s = _f32(boxes), _f32(scores)
    valid_inds = None
    if score_threshold > 0:
        valid = s > score_threshold
        valid_inds = torch.nonzero(valid, as_tuple=False).squeeze(1)
        b, s = b[valid].contiguous(), s[valid].contiguous()
    keep = torch.empty(max(b.size(0), 1), dtype=torch.int64)
    k = lib().orc_nms(b.data_ptr(), s.data_ptr(), b.size(0), float(iou_threshold), int(offset),
                      keep.data_ptr())
    inds = keep[:k]
    if max_num > 0:
        inds = inds[:max_num]
    if valid_inds is not None:
        inds = valid_inds[inds]
    inds = inds.to(boxes.device)
    dets = torch.cat((boxes[inds], scores[inds].reshape(-1, 1)), dim=1)
    return dets, inds


def soft_nms(boxes, scores, iou_threshold=0.3, sigma=0.5, min_score=1e-3, method='linear',
             offset=0):
    assert boxes.size(1) == 4 and boxes.size(0) == scores.size(0) and offset in (0, 1)
    method_dict = {'naive': 0, 'linear': 1, 'gaussian': 2}
    b, s = _f32(boxes), _f32(scores)
    n = b.size(0)
    dets = torch.zeros(max(n, 1), 5)
    inds = torch.zeros(max(n, 1), dtype=torch.int64)
    k = lib().orc_softnms(b.data_ptr(), s.data_ptr(), n, dets.data_ptr(), inds.data_ptr(),
                          float(iou_threshold), float(sigma), float(min_score),
                          method_dict[method], int(offset))
    return dets[:k].to(boxes.device), inds[:k].to(boxes.device)


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv 1.4.0 batched_nms: coordinate-offset trick below split_thr, per-id loop above."""
    nms_cfg_ = dict(nms_cfg)
    class_agnostic = nms_cfg_.pop('class_agnostic', class_agnostic)
    if class_agnostic:
        boxes_for_nms = boxes
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        boxes_for_nms = boxes + offsets[:, None]
    nms_type = nms_cfg_.pop('type', 'nms')
    nms_op = {'nms': nms, 'soft_nms': soft_nms}[nms_type]
    split_thr = nms_cfg_.pop('split_thr', 10000)
    if boxes_for_nms.shape[0] < split_thr:
        dets, keep = nms_op(boxes_for_nms, scores, **nms_cfg_)
        boxes = boxes[keep]
        scores = dets[:, 4]
    else:
        max_num = nms_cfg_.pop('max_num', -1)
        total_mask = scores.new_zeros(scores.size(), dtype=torch.bool)
        scores_after_nms = scores.new_zeros(scores.size())
        for id in torch.unique(idxs):
            mask = (idxs == id).nonzero(as_tuple=False).view(-1)
            dets, keep = nms_op(boxes_for_nms[mask], scores[mask], **nms_cfg_)
            total_mask[mask[keep]] = True
            scores_after_nms[mask[keep]] = dets[:, -1]
        keep = total_mask.nonzero(as_tuple=False).view(-1)
        # tie rule shared with the product: descending score, ascending index
        order = argsort_desc(scores_after_nms[keep])
        scores = scores_after_nms[keep][order]
        keep = keep[order]
        boxes = boxes[keep]
        if max_num > 0:
            keep, boxes, scores = keep[:max_num], boxes[:max_num], scores[:max_num]
    return torch.cat([boxes, scores[:, None]], -1), keep


# --------------------------------------------------------------------------- focal loss
def sigmoid_focal_loss_forward(input, target, gamma=2.0, alpha=0.25, weight=None):
    x = _f32(input)
    t = target.detach().to('cpu', torch.int64).contiguous()
    w = _f32(weight) if weight is not None else None
    out = torch.empty_like(x)
    lib().orc_sigmoid_focal_loss_forward(x.data_ptr(), t.data_ptr(),
                                         w.data_ptr() if w is not None else None,
                                         out.data_ptr(), x.size(0), x.size(1), float(gamma),
                                         float(alpha))
    return out


def sigmoid_focal_loss_backward(input, target, gamma=2.0, alpha=0.25, weight=None):
    x = _f32(input)
    t = target.detach().to('cpu', torch.int64).contiguous()
    w = _f32(weight) if weight is not None else None
    out = torch.empty_like(x)
    lib().orc_sigmoid_focal_loss_backward(x.data_ptr(), t.data_ptr(),
                                          w.data_ptr() if w is not None else None,
                                          out.data_ptr(), x.size(0), x.size(1), float(gamma),
                                          float(alpha))
    return out


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, target, gamma, alpha, weight, reduction):
        ctx.save_for_backward(input, target, weight)
        ctx.cfg = (gamma, alpha, reduction)
        out = sigmoid_focal_loss_forward(input, target, gamma, alpha, weight)
        if reduction == 'mean':
            return out.sum() / input.size(0)
        if reduction == 'sum':
            return out.sum()
        return out

    @staticmethod
    def backward(ctx, grad_output):
        input, target, weight = ctx.saved_tensors
        gamma, alpha, reduction = ctx.cfg
        g = sigmoid_focal_loss_backward(input, target, gamma, alpha, weight)
        g = g * grad_output
        if reduction == 'mean':
            g = g / input.size(0)
        return g, None, None, None, None, None


def sigmoid_focal_loss(input, target, gamma=2.0, alpha=0.25, weight=None, reduction='mean'):
    return _FocalFn.apply(input, target, gamma, alpha, weight, reduction)


# --------------------------------------------------------------------------- float64 brute force
def roi_align_f64(input, rois, output_size, spatial_scale, sampling_ratio=0, aligned=True):
    """Independent numpy float64 brute force of avg RoIAlign (no pre-calc tables), used to
    pin the C restatement."""
    x = input.detach().cpu().numpy().astype(np.float64)
    r = rois.detach().cpu().numpy().astype(np.float64)
    ph_n, pw_n = _pair(output_size)
    n, c, h, w = x.shape
    out = np.zeros((r.shape[0], c, ph_n, pw_n))
    off = 0.5 if aligned else 0.0
    for k in range(r.shape[0]):
        b = int(r[k, 0])
        sw, sh = r[k, 1] * spatial_scale - off, r[k, 2] * spatial_scale - off
        ew, eh = r[k, 3] * spatial_scale - off, r[k, 4] * spatial_scale - off
        rw, rh = ew - sw, eh - sh
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / ph_n, rw / pw_n
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rh / ph_n))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rw / pw_n))
        cnt = max(gh * gw, 1)
        for ph in range(ph_n):
            for pw in range(pw_n):
                acc = np.zeros(c)
                for iy in range(gh):
                    y = sh + ph * bh + (iy + 0.5) * bh / gh
                    for ix in range(gw):
                        xx = sw + pw * bw + (ix + 0.5) * bw / gw
                        yy = y
                        if yy < -1.0 or yy > h or xx < -1.0 or xx > w:
                            continue
                        yy, xx = max(yy, 0.0), max(xx, 0.0)
                        yl, xl = int(yy), int(xx)
                        if yl >= h - 1:
                            yh = yl = h - 1
                            yy = float(yl)
                        else:
                            yh = yl + 1
                        if xl >= w - 1:
                            xh = xl = w - 1
                            xx = float(xl)
                        else:
                            xh = xl + 1
                        ly, lx = yy - yl, xx - xl
                        hy, hx = 1 - ly, 1 - lx
                        acc += (hy * hx * x[b, :, yl, xl] + hy * lx * x[b, :, yl, xh] +
                                ly * hx * x[b, :, yh, xl] + ly * lx * x[b, :, yh, xh])
                out[k, :, ph, pw] = acc / cnt
    return out


_FLIP = {None: 0, 'horizontal': 1, 'vertical': 2, 'diagonal': 3}


def preprocess_u8(img_u8, new_w, new_h, pad_h, pad_w, flip_direction, mean, std, to_rgb=True):
    """Resize -> RandomFlip -> Normalize -> Pad of one uint8 BGR HxWx3 image: (3,pad_h,pad_w) fp32"""
    import numpy as np
    src = torch.as_tensor(np.ascontiguousarray(img_u8))
    assert src.dtype == torch.uint8 and src.dim() == 3 and src.shape[2] == 3
    out = torch.empty((3, pad_h, pad_w), dtype=torch.float32)
    m = torch.tensor([float(v) for v in mean], dtype=torch.float32)
    sd = torch.tensor([float(v) for v in std], dtype=torch.float32)
    st = lib().orc_preprocess_u8(src.data_ptr(), src.shape[0], src.shape[1], out.data_ptr(), int(new_h),
                                 int(new_w), int(pad_h), int(pad_w), _FLIP[flip_direction], m.data_ptr(),
                                 sd.data_ptr(), int(bool(to_rgb)))
    assert st == 0, st
    return out
