"""TEST INFRASTRUCTURE ONLY -- the CPU restatement of the whole inference pipeline, used as
(a) the checker for end-to-end plumbing tests on CPU and (b) bench.py's `cpu_baseline` leg.

It runs the SAME host-side module graph as the product but with every `brcnn.ops` entry
point swapped (inside the `patched()` context only) for a CPU restatement: PyTorch-CPU
convolutions / linear layers (what the reference itself executes on CPU) and the C oracle
for RoIAlign / NMS.  The product never imports this module and never selects it by itself.
"""
import contextlib
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import orc


def _conv2d_nhwc(x, w, scale=None, shift=None, residual=None, relu=False, stride=1, pad=0, out_f32=False):
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, stride, pad)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1)
    if residual is not None:
        y = y + residual
    if relu:
        y = y.relu()
    return y.contiguous()


def _split_rows(x_cat, batch, sizes):
    out, r0 = [], 0
    for h, w in sizes:
        n = batch * h * w
        out.append(x_cat[r0:r0 + n].view(batch, h, w, x_cat.shape[1]))
        r0 += n
    return out


def _conv2d_nhwc_multi(x_cat, w, batch, sizes, scale=None, shift=None, residual=None, relu=False,
                       stride=1, pad=0, out_f32=False):
    assert residual is None
    ys = [_conv2d_nhwc(x, w, scale, shift, None, relu, stride, pad) for x in _split_rows(x_cat, batch, sizes)]
    return torch.cat([y.reshape(-1, y.shape[3]) for y in ys], 0), [tuple(y.shape[1:3]) for y in ys]


def _groupnorm_multi(x_cat, gamma, beta, groups, batch, sizes, eps=1e-5, relu=False):
    ys = [_groupnorm(x, gamma, beta, groups, eps, relu) for x in _split_rows(x_cat, batch, sizes)]
    return torch.cat([y.reshape(-1, y.shape[3]) for y in ys], 0)


def _linear_nhwc(x, w, bias=None, relu=False, out_f32=False):
    y = F.linear(x, w, bias)
    return y.relu() if relu else y


def _pack_stem_weight(w, dtype=torch.float32):
    return w.detach().float()          # the CPU restatement convolves the NCHW image directly


def _stem7x7s2_nchw(img, w_packed, scale=None, shift=None, relu=True):
    y = F.conv2d(img, w_packed, None, 2, 3)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1)
    if relu:
        y = y.relu()
    return y.permute(0, 2, 3, 1).contiguous()


def _maxpool(x):
    return F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).contiguous()


def _stem7x7s2_pool_nchw(img, w_packed, scale=None, shift=None):
    return _maxpool(_stem7x7s2_nchw(img, w_packed, scale, shift, True))


def _groupnorm(x, gamma, beta, groups, eps=1e-5, relu=False):
    y = F.group_norm(x.permute(0, 3, 1, 2), groups, gamma, beta, eps)
    if relu:
        y = y.relu()
    return y.permute(0, 2, 3, 1).contiguous()


def _upsample_add_(dst, src):
    up = F.interpolate(src.permute(0, 3, 1, 2), size=dst.shape[1:3], mode='nearest')
    dst += up.permute(0, 2, 3, 1)
    return dst


def _nchw_to_nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _nhwc_to_nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def _roi_extract(feats_nhwc, rois, output_size, featmap_strides, finest_scale=56, sampling_ratio=0):
    """single_level_roi_extractor.py:36-115 with the C oracle's RoIAlign per level"""
    k, c = rois.size(0), feats_nhwc[0].shape[3]
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    out = torch.zeros(k, c, ph, pw)
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lvls = torch.floor(torch.log2(scale / finest_scale + 1e-6)).clamp(min=0, max=len(feats_nhwc) - 1).long()
    for i, f in enumerate(feats_nhwc):
        inds = (lvls == i).nonzero(as_tuple=False).squeeze(1)
        if inds.numel():
            out[inds] = orc.roi_align_forward(f.permute(0, 3, 1, 2).contiguous(), rois[inds], (ph, pw),
                                              1.0 / featmap_strides[i], sampling_ratio, 'avg', True)
    return out.permute(0, 2, 3, 1).contiguous(), lvls.to(torch.int32)


def _nms_ranges(boxes, scores, ranges, max_segment_len, iou_threshold, offset=0, max_keep=-1):
    n = boxes.size(0)
    keep = torch.zeros(max(n, 1), dtype=torch.int64)
    num = torch.zeros(ranges.size(0), dtype=torch.int32)
    for s in range(ranges.size(0)):
        b, e = int(ranges[s, 0]), int(ranges[s, 1])
        if e > b:
            _, k = orc.nms(boxes[b:e], scores[b:e], iou_threshold, offset, 0, max_keep)
            keep[b:b + k.numel()] = k + b
            num[s] = k.numel()
    return keep, num


def _soft_nms_ranges(boxes, scores, ranges, iou_threshold=0.3, sigma=0.5, min_score=1e-3, method=1, offset=0):
    n = boxes.size(0)
    dets = torch.zeros((max(n, 1), 5))
    inds = torch.zeros((max(n, 1),), dtype=torch.int64)
    num = torch.zeros(ranges.size(0), dtype=torch.int32)
    name = {0: 'naive', 1: 'linear', 2: 'gaussian'}[int(method)]
    for s_ in range(ranges.size(0)):
        b, e = int(ranges[s_, 0]), int(ranges[s_, 1])
        if e > b:
            d, k = orc.soft_nms(boxes[b:e], scores[b:e], iou_threshold, sigma, min_score, name, offset)
            dets[b:b + k.numel()] = d
            inds[b:b + k.numel()] = k + b
            num[s_] = k.numel()
    return dets, inds, num


def _rpn_score(cls, iou):
    return (cls.sigmoid() * iou.sigmoid()).sqrt()


def _rpn_topk(scores, k):
    """atss_rpn_head.py:727-737: stable descending sort of the level, first nms_pre"""
    out = []
    for sc in scores:
        n = sc.shape[1]
        if n > k:
            ranked, inds = sc.sort(dim=1, descending=True, stable=True)
            out.append((ranked[:, :k].contiguous(), inds[:, :k].contiguous()))
        else:
            out.append((sc, torch.arange(n).expand(sc.shape[0], n).contiguous()))
    return out


def _rpn_decode(topk_inds, bbox_pred, base_anchors, feat_hw, stride, means, stds, max_shape,
                min_size, wh_ratio_clip=16 / 1000, pred_scale=1.0):
    from brcnn.core import delta2bbox
    b, k = topk_inds.shape
    h, w = feat_hw
    a = base_anchors.size(0)
    sw, sh = (stride, stride) if isinstance(stride, int) else stride
    cell = topk_inds // a
    ai = topk_inds % a
    sx = ((cell % w) * sw).to(base_anchors)
    sy = ((cell // w) * sh).to(base_anchors)
    anchors = base_anchors[ai] + torch.stack([sx, sy, sx, sy], -1)
    d = torch.gather(bbox_pred.reshape(b, -1, 4), 1, topk_inds[..., None].expand(b, k, 4))
    if pred_scale != 1.0:
        d = d * pred_scale
    props = delta2bbox(anchors.view(-1, 4), d.view(-1, 4), means, stds, max_shape, wh_ratio_clip).view(b, k, 4)
    valid = ((props[..., 2] - props[..., 0]) > min_size) & ((props[..., 3] - props[..., 1]) > min_size)
    return props, valid.to(torch.uint8)


# ---- differentiable CPU restatements of brcnn.autograd (training path of the oracle pipeline) ---
def _conv2d_nhwc_autograd(x, weight, bias, stride, pad, with_skip=False, out_f32=False):
    y = F.conv2d(x.permute(0, 3, 1, 2), weight, bias, stride, pad)
    y = y.permute(0, 2, 3, 1).contiguous()
    y = y.float() if out_f32 else y
    return (y, x) if with_skip else y


def _linear_autograd(x, weight, bias, relu=False, out_f32=False):
    y = F.linear(x, weight, bias)
    y = y.relu() if relu else y
    return y.float() if out_f32 else y


def _roi_extract_autograd(feats_nhwc, rois, output_size, strides, finest_scale=56, sampling_ratio=0):
    k, c = rois.size(0), feats_nhwc[0].shape[3]
    ph, pw = (output_size, output_size) if isinstance(output_size, int) else output_size
    rois = rois.float()
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lvls = torch.floor(torch.log2(scale / finest_scale + 1e-6)).clamp(min=0, max=len(feats_nhwc) - 1).long()
    out = feats_nhwc[0].new_zeros(k, c, ph, pw)
    for i, f in enumerate(feats_nhwc):
        inds = (lvls == i).nonzero(as_tuple=False).squeeze(1)
        if inds.numel():
            r = orc.roi_align(f.permute(0, 3, 1, 2).contiguous(), rois[inds], (ph, pw), 1.0 / strides[i],
                              sampling_ratio, 'avg', True)
            out = out.index_copy(0, inds, r)
        else:
            out = out + f.sum() * 0.          # keeps every level in the graph, as the reference does
    return out.permute(0, 2, 3, 1).contiguous()


def _bn_act_autograd(z, scale, shift, res=None, relu=True):
    y = z * scale.to(z.dtype) + shift.to(z.dtype)
    if res is not None:
        y = y + res
    return y.relu() if relu else y


def _groupnorm_nhwc_autograd(x, gamma, beta, groups, eps, relu):
    z = F.group_norm(x.permute(0, 3, 1, 2).float(), groups, gamma, beta, eps).permute(0, 2, 3, 1).to(x.dtype)
    return (z.relu() if relu else z).contiguous()


_PATCH_AUTOGRAD = dict(groupnorm_nhwc_autograd=_groupnorm_nhwc_autograd, bn_act_autograd=_bn_act_autograd, conv2d_nhwc_autograd=_conv2d_nhwc_autograd, linear_autograd=_linear_autograd,
                       roi_extract_autograd=_roi_extract_autograd)

_PATCH = dict(pack_stem_weight=_pack_stem_weight, stem7x7s2_nchw=_stem7x7s2_nchw,
              pack_stem_pool_weight=_pack_stem_weight, stem7x7s2_pool_nchw=_stem7x7s2_pool_nchw,
              conv2d_nhwc=_conv2d_nhwc, conv2d_nhwc_multi=_conv2d_nhwc_multi,
              groupnorm_nhwc_multi=_groupnorm_multi, linear_nhwc=_linear_nhwc, maxpool3x3s2_nhwc=_maxpool,
              groupnorm_nhwc=_groupnorm, upsample_nearest_add_nhwc_=_upsample_add_,
              nchw_to_nhwc=_nchw_to_nhwc, nhwc_to_nchw=_nhwc_to_nchw, roi_extract=_roi_extract,
              nms_ranges=_nms_ranges, soft_nms_ranges=_soft_nms_ranges, rpn_score=_rpn_score, rpn_decode=_rpn_decode, rpn_topk=_rpn_topk,
              nms=orc.nms, soft_nms=orc.soft_nms, batched_nms=orc.batched_nms,
              roi_align=orc.roi_align, RoIAlign=orc.RoIAlign, sigmoid_focal_loss=orc.sigmoid_focal_loss)


def available_cpus():
    """host cores this process may actually use: min(affinity mask, cgroup cpu quota)"""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except Exception:
        pass
    return n


@contextlib.contextmanager
def patched():
    """swap brcnn.ops entry points for the CPU restatements (tests / cpu_baseline only)"""
    import brcnn  # noqa: F401
    from brcnn import ops
    from brcnn import autograd as ag
    saved = {k: getattr(ops, k) for k in _PATCH}
    saved_ag = {k: getattr(ag, k) for k in _PATCH_AUTOGRAD}
    for k, v in _PATCH.items():
        setattr(ops, k, v)
    for k, v in _PATCH_AUTOGRAD.items():
        setattr(ag, k, v)
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
        for k, v in saved_ag.items():
            setattr(ag, k, v)


def timed_baseline(cfg, seed=0, batch=1, budget_s=20.0, threads=None):
    """`cpu_baseline` object of the bench line: the oracle pipeline on the host cores, on a
    bounded sample (a few 1333x800 images) of the same workload."""
    import os
    from brcnn import build_detector
    from brcnn.synth import seeded_state_dict
    threads = threads or available_cpus()
    torch.set_num_threads(threads)
    with patched():
        m = build_detector(cfg.model)
        m.load_state_dict(seeded_state_dict(m, seed=seed))
        m.eval()
        g = torch.Generator().manual_seed(seed)
        img = torch.randn(batch, 3, 800, 1344, generator=g)
        metas = [dict(img_shape=(800, 1333, 3), pad_shape=(800, 1344, 3), ori_shape=(800, 1333, 3),
                      scale_factor=np.array([1., 1., 1., 1.], dtype=np.float32), flip=False)
                 for _ in range(batch)]
        with torch.no_grad():
            m.simple_test(img, metas, rescale=True)          # warm-up
            n, t0 = 0, time.perf_counter()
            while True:
                m.simple_test(img, metas, rescale=True)
                n += batch
                if time.perf_counter() - t0 > budget_s or n >= 16:
                    break
            dt = time.perf_counter() - t0
            # per-stage split of one more pass: host clock between the detector's stage marks
            from brcnn import profiling
            stages = profiling.stage_breakdown(lambda: m.simple_test(img, metas, rescale=True), cuda=False, iters=1)
    return {'value': n / dt, 'unit': 'images/sec', 'cores': threads, 'kind': 'port', 'stages_ms': stages,
            'sample': f'{n} synthetic 1333x800 images (batch {batch}), same model/config/weights, '
                      f'PyTorch-CPU conv/linear + C oracle RoIAlign/NMS, {dt:.1f} s'}
