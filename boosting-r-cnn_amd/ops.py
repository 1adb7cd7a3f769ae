"""Operator seam: the `mmcv.ops` Python API the reference calls, served by libbrcnn_hip.so.

Same names, argument meaning and error behaviour as mmcv 1.4.0's `RoIAlign / roi_align`,
`nms`, `batched_nms`, `soft_nms`, `sigmoid_focal_loss` (reference call sites:
roi_extractors/base_roi_extractor.py:54-60, single_level_roi_extractor.py:103,
atss_rpn_head.py:756, core/post_processing/bbox_nms.py:86, losses/focal_loss.py:86), plus
the conv-stack / RPN entry points that have no mmcv counterpart.

Every function takes tensors that live on a HIP device and enqueues on torch's current
stream.  There is no CPU path: CPU tensors raise.
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import lib as _L

LAYOUT_NCHW, LAYOUT_NHWC = 0, 1
DT_F32, DT_BF16, DT_BF16_OUT_F32, DT_F16, DT_F16_OUT_F32 = 0, 1, 2, 3, 4


def _dt(t):
    """C-ABI dtype code of an activation tensor (fp32, or bf16 for the bf16 MFMA path)"""
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float16:
        return DT_F16
    raise _L.BrcnnHipError(f'unsupported activation dtype {t.dtype} (fp32, bf16 or fp16)')


def _out_f32(dt):
    """the code that makes a 16-bit conv write an fp32 result"""
    return {DT_BF16: DT_BF16_OUT_F32, DT_F16: DT_F16_OUT_F32}[dt]


# ----------------------------------------------------------------------------- helpers
def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _L.BrcnnHipError(
                'brcnn.ops run on the HIP device only (got a CPU tensor); there is no CPU fallback')


def _stream():
    return _L.raw_stream_handle()


def _conv_stream():
    """for the entry points that launch a convolution: registers the stream's conv workspace on first sight"""
    return _L.stream_handle()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _pair(x):
    if isinstance(x, int):
        return (x, x)
    assert len(x) == 2
    return (int(x[0]), int(x[1]))


def _is_nhwc(t):
    """logical (N,C,H,W) tensor whose memory is (N,H,W,C) and not also plain-contiguous"""
    return t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and \
        not t.is_contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------- RoIAlign
class RoIAlignFunction(Function):
    @staticmethod
    def forward(ctx, input, rois, output_size, spatial_scale=1.0, sampling_ratio=0,
                pool_mode='avg', aligned=True):
        _require_gpu(input, rois)
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale = float(spatial_scale)
        ctx.sampling_ratio = int(sampling_ratio)
        assert pool_mode in ('max', 'avg')
        ctx.pool_mode = 0 if pool_mode == 'max' else 1
        ctx.aligned = bool(aligned)
        ctx.input_shape = input.size()
        assert rois.size(1) == 5, 'RoI must be (idx, x1, y1, x2, y2)!'
        assert input.dtype == torch.float32, 'roi_align: fp32 features expected'
        rois = rois.contiguous().float()
        n, c, h, w = input.shape
        k = rois.size(0)
        ph, pw = ctx.output_size
        nhwc = _is_nhwc(input) and ctx.pool_mode == 1 and c % 4 == 0
        ctx.nhwc = nhwc
        if nhwc:
            output = torch.zeros((k, c, ph, pw), dtype=input.dtype, device=input.device).contiguous(
                memory_format=torch.channels_last)
            x = input
        else:
            output = input.new_zeros((k, c, ph, pw))
            x = input.contiguous()
        argmax_y = argmax_x = None
        if ctx.pool_mode == 0:
            argmax_y = input.new_zeros(output.shape)
            argmax_x = input.new_zeros(output.shape)
        st = _L.load().brcnn_roi_align_forward(
            _ptr(x), _ptr(rois), _ptr(output), _ptr(argmax_y), _ptr(argmax_x), n, c, h, w, k, ph,
            pw, ctx.spatial_scale, ctx.sampling_ratio, ctx.pool_mode, int(ctx.aligned),
            LAYOUT_NHWC if nhwc else LAYOUT_NCHW, _stream())
        _L.check(st, 'brcnn_roi_align_forward')
        ctx.save_for_backward(rois)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        (rois,) = ctx.saved_tensors
        if ctx.pool_mode != 1:
            raise NotImplementedError('roi_align backward: avg pooling only on this path')
        n, c, h, w = ctx.input_shape
        ph, pw = ctx.output_size
        if ctx.nhwc:
            g = grad_output.contiguous(memory_format=torch.channels_last)
            grad_input = torch.zeros((n, c, h, w), dtype=g.dtype, device=g.device).contiguous(
                memory_format=torch.channels_last)
        else:
            g = grad_output.contiguous()
            grad_input = g.new_zeros((n, c, h, w))
        st = _L.load().brcnn_roi_align_backward(
            _ptr(g), _ptr(rois), _ptr(grad_input), n, c, h, w, rois.size(0), ph, pw,
            ctx.spatial_scale, ctx.sampling_ratio, int(ctx.aligned),
            LAYOUT_NHWC if ctx.nhwc else LAYOUT_NCHW, _stream())
        _L.check(st, 'brcnn_roi_align_backward')
        return grad_input, None, None, None, None, None, None


roi_align = RoIAlignFunction.apply


class RoIAlign(nn.Module):
    """mmcv.ops.RoIAlign: RoIAlign(output_size, spatial_scale=1.0, sampling_ratio=0,
    pool_mode='avg', aligned=True, use_torchvision=False); forward(input (N,C,H,W),
    rois (K,5)) -> (K,C,ph,pw).  A channels_last input selects the NHWC kernel and yields a
    channels_last output (same logical shape, same values)."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg',
                 aligned=True, use_torchvision=False):
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)
        self.pool_mode = pool_mode
        self.aligned = aligned
        self.use_torchvision = use_torchvision

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio,
                         self.pool_mode, self.aligned)

    def __repr__(self):
        return (f'{self.__class__.__name__}(output_size={self.output_size}, '
                f'spatial_scale={self.spatial_scale}, sampling_ratio={self.sampling_ratio}, '
                f'pool_mode={self.pool_mode}, aligned={self.aligned}, '
                f'use_torchvision={self.use_torchvision})')


_ROI_ORDER_MIN = []


def _roi_order_min_rois():
    if not _ROI_ORDER_MIN:
        _ROI_ORDER_MIN.append(int(_L.load().brcnn_roi_extract_order_min_rois()))
    return _ROI_ORDER_MIN[0]


def roi_extract(feats_nhwc, rois, output_size, featmap_strides, finest_scale=56, sampling_ratio=0):
    """Fused SingleRoIExtractor.forward (single_level_roi_extractor.py:57-115): level mapping
    + RoIAlign of every RoI on its own level in one launch.  `feats_nhwc`: list of (N,H,W,C)
    contiguous fp32 tensors.  Returns ((K,ph,pw,C) features, (K,) int32 levels)."""
    _require_gpu(rois, *feats_nhwc)
    import ctypes
    L = len(feats_nhwc)
    n, _, _, c = feats_nhwc[0].shape
    ph, pw = _pair(output_size)
    rois = rois.contiguous().float()
    k = rois.size(0)
    dt = _dt(feats_nhwc[0])
    out = torch.empty((k, ph, pw, c), dtype=feats_nhwc[0].dtype, device=rois.device)
    levels = torch.empty((k,), dtype=torch.int32, device=rois.device)
    for f in feats_nhwc:
        assert f.is_contiguous() and f.dtype == feats_nhwc[0].dtype and f.shape[0] == n and f.shape[3] == c
    ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats_nhwc])
    hs = (ctypes.c_int * L)(*[f.shape[1] for f in feats_nhwc])
    ws = (ctypes.c_int * L)(*[f.shape[2] for f in feats_nhwc])
    sc = (ctypes.c_float * L)(*[1.0 / s for s in featmap_strides])
    lib = _L.load()
    # caller-owned scratch: the visiting order (n int32) from the RoI count at which the library orders them, and the
    # per-RoI records of the prepared form (level mapping / geometry / axis weights once per RoI instead of per bin row)
    order = torch.empty((k,), dtype=torch.int32, device=rois.device) if k >= _roi_order_min_rois() else None
    nb = int(lib.brcnn_roi_extract_prep_workspace_bytes(k))          # 0 while the prepared form is off (the default)
    prep = torch.empty(nb, dtype=torch.uint8, device=rois.device) if nb else None
    st = lib.brcnn_roi_extract_forward_prepared(ptrs, hs, ws, sc, L, _ptr(rois), _ptr(out), _ptr(levels), n, c, k, ph, pw,
                                                int(sampling_ratio), float(finest_scale), dt, _ptr(order), _ptr(prep), nb,
                                                _stream())
    _L.check(st, 'brcnn_roi_extract_forward_prepared')
    return out, levels


# ----------------------------------------------------------------------------- NMS
def nms_segments(boxes, scores, seg_offsets, max_segment_len, iou_threshold, offset=0,
                 max_keep=-1):
    """back-to-back segments given as (S+1) offsets; see nms_ranges"""
    seg_offsets = seg_offsets.contiguous().to(torch.int32)
    ranges = torch.stack([seg_offsets[:-1], seg_offsets[1:]], 1)
    return nms_ranges(boxes, scores, ranges, max_segment_len, iou_threshold, offset, max_keep)


def nms_ranges(boxes, scores, ranges, max_segment_len, iou_threshold, offset=0, max_keep=-1):
    """Segmented greedy NMS, no host sync.  `ranges` (S,2) int32 [begin, end) per segment.
    Returns (keep (n,) int64, num_keep (S,) int32): segment s's survivors (global indices,
    score order) sit at keep[begin_s:][:num_keep[s]]."""
    _require_gpu(boxes, scores, ranges)
    boxes = boxes.contiguous().float()
    scores = scores.contiguous().float()
    ranges = ranges.to(torch.int32)
    seg_begin, seg_end = ranges[:, 0].contiguous(), ranges[:, 1].contiguous()
    n = boxes.size(0)
    S = seg_begin.numel()
    keep = torch.empty((max(n, 1),), dtype=torch.int64, device=boxes.device)
    num_keep = torch.zeros((S,), dtype=torch.int32, device=boxes.device)
    if n == 0:
        return keep[:0], num_keep
    lib = _L.load()
    max_segment_len = max(1, min(int(max_segment_len), n))
    wsb = lib.brcnn_nms_workspace_bytes(n, S, max_segment_len)
    ws = _ws(wsb, boxes.device)
    st = lib.brcnn_nms(_ptr(boxes), _ptr(scores), _ptr(seg_begin), _ptr(seg_end), S, n,
                       max_segment_len,
                       float(iou_threshold), int(offset), int(max_keep), _ptr(keep),
                       _ptr(num_keep), _ptr(ws), wsb, _stream())
    _L.check(st, 'brcnn_nms')
    return keep, num_keep


def nms_prepare(boxes, scores, ids, valid, level_sizes=None):
    """(B,T,4) f32, (B,T) f32, (B,T) int64, (B,T) bool -> compacted c_boxes, c_scores, c_ids, the offset
    boxes for the segmented NMS and the (B,2) int32 ranges, in one launch (brcnn_nms_prepare).
    `level_sizes` (host ints, sum == T): the slot's column ranges carry one id each (mmcv batched_nms
    above split_thr) -> ranges (B*L, 2), one segment per (image, id) (brcnn_nms_prepare_levels)."""
    _require_gpu(boxes, scores, ids, valid)
    B, T = scores.shape
    boxes, scores = boxes.contiguous().float(), scores.contiguous().float()
    ids = ids.contiguous().long()
    valid = valid.contiguous().to(torch.uint8) if valid.dtype != torch.bool else valid.contiguous().view(torch.uint8)
    c_boxes, nms_boxes = torch.empty_like(boxes), torch.empty_like(boxes)
    c_scores, c_ids = torch.empty_like(scores), torch.empty_like(ids)
    if level_sizes is not None:
        import ctypes
        L = len(level_sizes)
        ranges = torch.empty((B * L, 2), dtype=torch.int32, device=boxes.device)
        st = _L.load().brcnn_nms_prepare_levels(_ptr(boxes), _ptr(scores), _ptr(ids), _ptr(valid), _ptr(c_boxes),
                                                _ptr(c_scores), _ptr(c_ids), _ptr(nms_boxes), _ptr(ranges), B, T, L,
                                                (ctypes.c_int * L)(*[int(v) for v in level_sizes]), _stream())
        _L.check(st, 'brcnn_nms_prepare_levels')
        return c_boxes, c_scores, c_ids, nms_boxes, ranges
    ranges = torch.empty((B, 2), dtype=torch.int32, device=boxes.device)
    st = _L.load().brcnn_nms_prepare(_ptr(boxes), _ptr(scores), _ptr(ids), _ptr(valid), _ptr(c_boxes), _ptr(c_scores),
                                     _ptr(c_ids), _ptr(nms_boxes), _ptr(ranges), B, T, _stream())
    _L.check(st, 'brcnn_nms_prepare')
    return c_boxes, c_scores, c_ids, nms_boxes, ranges


def nms_collect_sorted(keep, num, ranges, c_boxes, c_scores, c_ids, K, L, new_scores5=None):
    """survivors of the per-(image, id) NMS re-sorted per image by score, first K (mmcv batched_nms above
    split_thr): dets (B,K,5) zero padded, ids_kept (B,K) (-1 padded), n_kept (B,) int32"""
    _require_gpu(keep, num, ranges, c_boxes, c_scores, c_ids, new_scores5)
    B, T = c_scores.shape
    dets = torch.empty((B, K, 5), dtype=torch.float32, device=c_boxes.device)
    ids_kept = torch.empty((B, K), dtype=torch.int64, device=c_boxes.device)
    n_kept = torch.empty((B,), dtype=torch.int32, device=c_boxes.device)
    st = _L.load().brcnn_nms_collect_sorted(_ptr(keep), _ptr(num), _ptr(ranges.contiguous()), _ptr(c_boxes),
                                            _ptr(c_scores), _ptr(c_ids), _ptr(new_scores5), _ptr(dets), _ptr(ids_kept),
                                            _ptr(n_kept), B, T, int(L), int(K), _stream())
    _L.check(st, 'brcnn_nms_collect_sorted')
    return dets, ids_kept, n_kept


def nms_collect(keep, num, c_boxes, c_scores, c_ids, K):
    """first min(num[b], K) survivors of every slot -> dets (B,K,5), ids_kept (B,K) (zeros / -1 padded)"""
    _require_gpu(keep, num, c_boxes, c_scores, c_ids)
    B, T = c_scores.shape
    dets = torch.empty((B, K, 5), dtype=torch.float32, device=c_boxes.device)
    ids_kept = torch.empty((B, K), dtype=torch.int64, device=c_boxes.device)
    st = _L.load().brcnn_nms_collect(_ptr(keep), _ptr(num), _ptr(c_boxes), _ptr(c_scores), _ptr(c_ids), _ptr(dets),
                                     _ptr(ids_kept), B, T, int(K), _stream())
    _L.check(st, 'brcnn_nms_collect')
    return dets, ids_kept


def nms(boxes, scores, iou_threshold, offset=0, score_threshold=0, max_num=-1):
    """mmcv.ops.nms: returns (dets (k,5), inds (k,) int64), score-descending."""
    assert isinstance(boxes, torch.Tensor) and isinstance(scores, torch.Tensor)
    assert boxes.size(1) == 4
    assert boxes.size(0) == scores.size(0)
    assert offset in (0, 1)
    _require_gpu(boxes, scores)
    b, s = boxes, scores
    valid_inds = None
    if score_threshold > 0:
        valid_mask = scores > score_threshold
        b, s = boxes[valid_mask], scores[valid_mask]
        valid_inds = torch.nonzero(valid_mask, as_tuple=False).squeeze(dim=1)
    n = b.size(0)
    if n == 0:
        inds = torch.empty((0,), dtype=torch.int64, device=boxes.device)
    else:
        seg = torch.tensor([0, n], dtype=torch.int32, device=boxes.device)
        keep, num_keep = nms_segments(b, s, seg, n, iou_threshold, offset, max_num)
        inds = keep[:int(num_keep.item())]
    if max_num > 0:
        inds = inds[:max_num]
    if valid_inds is not None:
        inds = valid_inds[inds]
    dets = torch.cat((boxes[inds], scores[inds].reshape(-1, 1)), dim=1)
    return dets, inds


def soft_nms_segments(boxes, scores, seg_offsets, iou_threshold=0.3, sigma=0.5, min_score=1e-3,
                      method=1, offset=0):
    """Segmented soft-NMS, no host sync: (dets (n,5), inds (n,) int64, num_keep (S,) int32)."""
    _require_gpu(boxes, scores, seg_offsets)
    boxes = boxes.contiguous().float()
    scores = scores.contiguous().float()
    seg_offsets = seg_offsets.contiguous().to(torch.int32)
    n, S = boxes.size(0), seg_offsets.numel() - 1
    dets = torch.zeros((max(n, 1), 5), dtype=torch.float32, device=boxes.device)
    inds = torch.zeros((max(n, 1),), dtype=torch.int64, device=boxes.device)
    num_keep = torch.zeros((S,), dtype=torch.int32, device=boxes.device)
    if n == 0:
        return dets[:0], inds[:0], num_keep
    lib = _L.load()
    wsb = lib.brcnn_softnms_workspace_bytes(n, S)
    ws = _ws(wsb, boxes.device)
    sb, se = seg_offsets[:-1].contiguous(), seg_offsets[1:].contiguous()
    st = lib.brcnn_softnms(_ptr(boxes), _ptr(scores), _ptr(sb), _ptr(se), S, n,
                           float(iou_threshold), float(sigma), float(min_score), int(method),
                           int(offset), _ptr(dets), _ptr(inds), _ptr(num_keep), _ptr(ws), wsb,
                           _stream())
    _L.check(st, 'brcnn_softnms')
    return dets, inds, num_keep


def soft_nms_ranges(boxes, scores, ranges, iou_threshold=0.3, sigma=0.5, min_score=1e-3, method=1, offset=0):
    """soft-NMS over arbitrary [begin, end) segments of a flat box list (`ranges` (S,2) int32), no
    host sync; segment s's picks sit at dets / inds [begin_s : begin_s + num_keep[s]] in pick order"""
    _require_gpu(boxes, scores, ranges)
    boxes = boxes.contiguous().float()
    scores = scores.contiguous().float()
    ranges = ranges.to(torch.int32)
    n, S = boxes.size(0), ranges.size(0)
    dets = torch.zeros((max(n, 1), 5), dtype=torch.float32, device=boxes.device)
    inds = torch.zeros((max(n, 1),), dtype=torch.int64, device=boxes.device)
    num_keep = torch.zeros((S,), dtype=torch.int32, device=boxes.device)
    if n == 0:
        return dets[:0], inds[:0], num_keep
    lib = _L.load()
    wsb = lib.brcnn_softnms_workspace_bytes(n, S)
    ws = _ws(wsb, boxes.device)
    sb, se = ranges[:, 0].contiguous(), ranges[:, 1].contiguous()
    st = lib.brcnn_softnms(_ptr(boxes), _ptr(scores), _ptr(sb), _ptr(se), S, n, float(iou_threshold), float(sigma),
                           float(min_score), int(method), int(offset), _ptr(dets), _ptr(inds), _ptr(num_keep),
                           _ptr(ws), wsb, _stream())
    _L.check(st, 'brcnn_softnms')
    return dets, inds, num_keep


def soft_nms(boxes, scores, iou_threshold=0.3, sigma=0.5, min_score=1e-3, method='linear',
             offset=0):
    """mmcv.ops.soft_nms: returns (dets (k,5) with decayed scores, inds (k,)) in pick order.
    (mmcv copies to the host and runs softnms_cpu; here the chain runs on the device.)"""
    assert isinstance(boxes, torch.Tensor) and isinstance(scores, torch.Tensor)
    assert boxes.size(1) == 4
    assert boxes.size(0) == scores.size(0)
    assert offset in (0, 1)
    method_dict = {'naive': 0, 'linear': 1, 'gaussian': 2}
    assert method in method_dict.keys()
    _require_gpu(boxes, scores)
    n = boxes.size(0)
    if n == 0:
        return boxes.new_zeros((0, 5)), torch.empty((0,), dtype=torch.int64, device=boxes.device)
    seg = torch.tensor([0, n], dtype=torch.int32, device=boxes.device)
    dets, inds, num_keep = soft_nms_segments(boxes, scores, seg, iou_threshold, sigma, min_score,
                                             method_dict[method], offset)
    k = int(num_keep.item())
    return dets[:k], inds[:k]


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.batched_nms (Python logic restated from mmcv 1.4.0): coordinate-offset trick
    below `split_thr` (default 10000) boxes, per-id NMS + re-sort above.  The per-id loop of
    the reference becomes ONE segmented launch (ids -> segments)."""
    nms_cfg_ = dict(nms_cfg)
    class_agnostic = nms_cfg_.pop('class_agnostic', class_agnostic)
    if class_agnostic:
        boxes_for_nms = boxes
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        boxes_for_nms = boxes + offsets[:, None]
    nms_type = nms_cfg_.pop('type', 'nms')
    if nms_type not in ('nms', 'soft_nms'):
        raise KeyError(f'unsupported nms type {nms_type}')
    nms_op = nms if nms_type == 'nms' else soft_nms
    split_thr = nms_cfg_.pop('split_thr', 10000)
    if boxes_for_nms.shape[0] < split_thr:
        dets, keep = nms_op(boxes_for_nms, scores, **nms_cfg_)
        boxes = boxes[keep]
        scores = dets[:, 4]
    else:
        max_num = nms_cfg_.pop('max_num', -1)
        # group by id: stable sort of ids keeps the original order inside a group
        order = torch.sort(idxs, stable=True)[1]
        ids_sorted = idxs[order]
        uniq, counts = torch.unique_consecutive(ids_sorted, return_counts=True)
        seg = torch.zeros(uniq.numel() + 1, dtype=torch.int32, device=boxes.device)
        seg[1:] = torch.cumsum(counts, 0).to(torch.int32)
        b_sorted, s_sorted = boxes_for_nms[order], scores[order]
        max_len = int(counts.max().item())
        total_mask = scores.new_zeros(scores.size(), dtype=torch.bool)
        scores_after_nms = scores.new_zeros(scores.size())
        if nms_type == 'nms':
            keep_buf, num_keep = nms_segments(b_sorted, s_sorted, seg, max_len,
                                              nms_cfg_.get('iou_threshold'),
                                              nms_cfg_.get('offset', 0), -1)
            pos = torch.arange(b_sorted.size(0), device=boxes.device)
            seg_id = torch.searchsorted(seg[1:].long(), pos, right=True)      # segment of each sorted row, no host sync
            valid = (pos - seg[:-1].long()[seg_id]) < num_keep.long()[seg_id]
            kept_sorted = keep_buf[valid]
            kept = order[kept_sorted]
            total_mask[kept] = True
            scores_after_nms[kept] = scores[kept]
        else:
            method = {'naive': 0, 'linear': 1, 'gaussian': 2}[nms_cfg_.get('method', 'linear')]
            dets, inds, num_keep = soft_nms_segments(
                b_sorted, s_sorted, seg, nms_cfg_.get('iou_threshold', 0.3),
                nms_cfg_.get('sigma', 0.5), nms_cfg_.get('min_score', 1e-3), method,
                nms_cfg_.get('offset', 0))
            pos = torch.arange(b_sorted.size(0), device=boxes.device)
            seg_id = torch.searchsorted(seg[1:].long(), pos, right=True)      # segment of each sorted row, no host sync
            valid = (pos - seg[:-1].long()[seg_id]) < num_keep.long()[seg_id]
            kept = order[inds[valid]]
            total_mask[kept] = True
            scores_after_nms[kept] = dets[valid, 4]
        keep = total_mask.nonzero(as_tuple=False).view(-1)
        scores, inds = scores_after_nms[keep].sort(descending=True, stable=True)
        keep = keep[inds]
        boxes = boxes[keep]
        if max_num > 0:
            keep = keep[:max_num]
            boxes = boxes[:max_num]
            scores = scores[:max_num]
    return torch.cat([boxes, scores[:, None]], -1), keep


# ----------------------------------------------------------------------------- focal loss
class SigmoidFocalLossFunction(Function):
    @staticmethod
    def forward(ctx, input, target, gamma=2.0, alpha=0.25, weight=None, reduction='mean'):
        _require_gpu(input, target, weight)
        assert isinstance(target, (torch.LongTensor, torch.cuda.LongTensor))
        assert input.dim() == 2
        assert target.dim() == 1
        assert input.size(0) == target.size(0)
        if weight is None:
            weight = input.new_empty(0)
        else:
            assert weight.dim() == 1
            assert input.size(1) == weight.size(0)
        ctx.reduction_dict = {'none': 0, 'mean': 1, 'sum': 2}
        assert reduction in ctx.reduction_dict.keys()
        ctx.gamma = float(gamma)
        ctx.alpha = float(alpha)
        ctx.reduction = ctx.reduction_dict[reduction]
        input = input.contiguous().float()
        target = target.contiguous()
        output = input.new_zeros(input.size())
        w = weight.contiguous().float() if weight.numel() > 0 else None
        st = _L.load().brcnn_sigmoid_focal_loss_forward(
            _ptr(input), _ptr(target), _ptr(w), _ptr(output), input.size(0), input.size(1),
            ctx.gamma, ctx.alpha, _stream())
        _L.check(st, 'brcnn_sigmoid_focal_loss_forward')
        if ctx.reduction == ctx.reduction_dict['mean']:
            output = output.sum() / input.size(0)
        elif ctx.reduction == ctx.reduction_dict['sum']:
            output = output.sum()
        ctx.save_for_backward(input, target, weight)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, target, weight = ctx.saved_tensors
        grad_input = input.new_zeros(input.size())
        w = weight.contiguous().float() if weight.numel() > 0 else None
        st = _L.load().brcnn_sigmoid_focal_loss_backward(
            _ptr(input), _ptr(target), _ptr(w), _ptr(grad_input), input.size(0), input.size(1),
            ctx.gamma, ctx.alpha, _stream())
        _L.check(st, 'brcnn_sigmoid_focal_loss_backward')
        grad_input *= grad_output
        if ctx.reduction == ctx.reduction_dict['mean']:
            grad_input /= input.size(0)
        return grad_input, None, None, None, None, None


sigmoid_focal_loss = SigmoidFocalLossFunction.apply


# ----------------------------------------------------------------------------- conv stack
def conv_out_size(h, w, kh, kw, stride, pad):
    return (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1


# ---- conv outputs written straight into a caller's buffer -------------------------------------------------------------
# The RPN runs every layer once over all pyramid levels, concatenated row-wise; the neck produces the levels one conv
# at a time.  `output_into(view)` makes the NEXT conv output of that size land in `view` (a slice of one (rows, C)
# buffer the neck allocated), so the levels lie back to back and `autograd.cat_rows` finds them already concatenated:
# no cat launch (183 MB read + written per fp32 batch of 8).
_NEXT_OUT = []


class output_into:
    def __init__(self, view):
        self.view = view

    def __enter__(self):
        _NEXT_OUT.append(self.view)
        return self

    def __exit__(self, *exc):
        if _NEXT_OUT and _NEXT_OUT[-1] is self.view:        # nobody took it
            _NEXT_OUT.pop()
        return False


def _alloc_out(shape, dtype, device):
    if _NEXT_OUT:
        t = _NEXT_OUT[-1]
        n = 1
        for v in shape:
            n *= int(v)
        if t is not None and t.numel() == n and t.dtype == dtype and t.device == device and t.is_contiguous():
            _NEXT_OUT.pop()
            return t.view(shape)
    return torch.empty(shape, dtype=dtype, device=device)


def conv2d_nhwc(x, w, scale=None, shift=None, residual=None, relu=False, stride=1, pad=0,
                out_f32=False):
    """y = act(conv(x, w) * scale + shift + residual); x (N,H,W,Cin), w (Cout,KH,KW,Cin),
    y (N,Ho,Wo,Cout), contiguous on the device.  fp32 tensors run the fp32 MFMA kernel; bf16
    x / w / residual (scale, shift stay fp32) run the bf16 kernel with fp32 accumulation and a
    bf16 result, or an fp32 result with `out_f32`."""
    _require_gpu(x, w, scale, shift, residual)
    assert x.dim() == 4 and w.dim() == 4 and x.is_contiguous() and w.is_contiguous()
    assert x.dtype == w.dtype, f'conv2d_nhwc: x {x.dtype} vs w {w.dtype}'
    dt = _dt(x)
    n, h, wd, cin = x.shape
    cout, kh, kw, cin2 = w.shape
    assert cin == cin2, f'conv2d_nhwc: Cin mismatch {cin} vs {cin2}'
    ho, wo = conv_out_size(h, wd, kh, kw, stride, pad)
    if dt != DT_F32 and out_f32:
        dt = _out_f32(dt)
    y = _alloc_out((n, ho, wo, cout), x.dtype if dt in (DT_BF16, DT_F16) else torch.float32, x.device)
    if residual is not None:
        assert residual.shape == y.shape and residual.is_contiguous() and residual.dtype == x.dtype
    st = _L.load().brcnn_conv2d_nhwc(_ptr(x), _ptr(w), _ptr(scale), _ptr(shift), _ptr(residual),
                                     _ptr(y), n, h, wd, cin, cout, kh, kw, int(stride), int(pad),
                                     int(bool(relu)), dt, _conv_stream())
    _L.check(st, 'brcnn_conv2d_nhwc')
    return y


def pack_grouped_weight(weight, groups):
    """(Cout, Cin/g, KH, KW) grouped-conv parameter -> (Cout, KH, KW, window) block-diagonal tiles
    for `conv2d_nhwc_grouped` (window = input channels seen by one 64-channel output tile)"""
    cout, cg_in, kh, kw = weight.shape
    cg_out = cout // groups
    assert cout % 64 == 0 and 64 % cg_out == 0, 'grouped conv: Cout % 64 == 0 and 64 % (Cout/groups) == 0'
    window = (64 // cg_out) * cg_in
    assert window % 32 == 0
    w = weight.detach().float().permute(0, 2, 3, 1)                       # (Cout, KH, KW, cg_in)
    out = torch.zeros((cout, kh, kw, window), dtype=torch.float32, device=weight.device)
    co = torch.arange(cout, device=weight.device)
    start = ((co // cg_out) * cg_in) - (co // 64) * window               # window position of each filter
    idx = (start[:, None] + torch.arange(cg_in, device=weight.device)[None, :])   # (Cout, cg_in)
    out.scatter_(3, idx[:, None, None, :].expand(cout, kh, kw, cg_in), w)
    return out.contiguous(), window


def conv2d_nhwc_grouped(x, w_tiles, window, scale=None, shift=None, residual=None, relu=False, stride=1, pad=0):
    """grouped conv on the MFMA kernel; `w_tiles`, `window` from pack_grouped_weight"""
    _require_gpu(x, w_tiles, scale, shift, residual)
    assert x.dim() == 4 and x.is_contiguous() and w_tiles.is_contiguous() and w_tiles.dtype == x.dtype
    n, h, wd, cin = x.shape
    cout, kh, kw, win = w_tiles.shape
    assert win == window
    dt = _dt(x)
    if dt != DT_F32 and window % 64:
        raise _L.BrcnnHipError(f'bf16 grouped conv needs a window that is a multiple of 64 channels (got {window})')
    ho, wo = conv_out_size(h, wd, kh, kw, stride, pad)
    y = torch.empty((n, ho, wo, cout), dtype=x.dtype, device=x.device)
    if residual is not None:
        assert residual.shape == y.shape and residual.is_contiguous()
    st = _L.load().brcnn_conv2d_nhwc_grouped(_ptr(x), _ptr(w_tiles), _ptr(scale), _ptr(shift), _ptr(residual),
                                             _ptr(y), n, h, wd, cin, cout, kh, kw, int(stride), int(pad),
                                             int(window), int(bool(relu)), dt, _conv_stream())
    _L.check(st, 'brcnn_conv2d_nhwc_grouped')
    return y


def conv2d_nhwc_multi(x_cat, w, batch, sizes, scale=None, shift=None, residual=None, relu=False,
                      stride=1, pad=0, out_f32=False):
    """The same conv over several back-to-back segments that share the weights (pyramid
    levels).  x_cat (sum_l batch*H_l*W_l, Cin) rows; `sizes` = [(H_l, W_l)].  Returns
    (y_cat (sum_l batch*Ho_l*Wo_l, Cout), [(Ho_l, Wo_l)])."""
    import ctypes
    _require_gpu(x_cat, w, scale, shift, residual)
    assert x_cat.dim() == 2 and x_cat.is_contiguous() and w.is_contiguous()
    cout, kh, kw, cin = w.shape
    assert x_cat.shape[1] == cin and x_cat.shape[0] == sum(batch * h * ww for h, ww in sizes)
    out_sizes = [conv_out_size(h, ww, kh, kw, stride, pad) for h, ww in sizes]
    rows = sum(batch * h * ww for h, ww in out_sizes)
    dt = _dt(x_cat)
    assert w.dtype == x_cat.dtype
    if dt != DT_F32 and out_f32:
        dt = _out_f32(dt)
    y = _alloc_out((rows, cout), x_cat.dtype if dt in (DT_BF16, DT_F16) else torch.float32, x_cat.device)
    if residual is not None:
        assert residual.shape == y.shape and residual.is_contiguous()
    L = len(sizes)
    hs = (ctypes.c_int * L)(*[h for h, _ in sizes])
    ws = (ctypes.c_int * L)(*[ww for _, ww in sizes])
    st = _L.load().brcnn_conv2d_nhwc_multi(_ptr(x_cat), _ptr(w), _ptr(scale), _ptr(shift),
                                           _ptr(residual), _ptr(y), batch, L, hs, ws, cin, cout, kh,
                                           kw, int(stride), int(pad), int(bool(relu)), dt,
                                           _conv_stream())
    _L.check(st, 'brcnn_conv2d_nhwc_multi')
    return y, out_sizes


def groupnorm_nhwc_multi(x_cat, gamma, beta, groups, batch, sizes, eps=1e-5, relu=False, return_stats=False):
    """GroupNorm(+ReLU) with statistics per (segment, image, group) over a concatenation of
    NHWC segments (rows, C).  `return_stats`: also the (mean, rstd) buffer the backward needs."""
    import ctypes
    _require_gpu(x_cat, gamma, beta)
    L = len(sizes)
    c = x_cat.shape[1]
    y = torch.empty_like(x_cat)
    ws = torch.empty((batch * L * groups * 2,), dtype=torch.float64, device=x_cat.device)
    hw = (ctypes.c_int * L)(*[h * w for h, w in sizes])
    st = _L.load().brcnn_groupnorm_nhwc_multi(_ptr(x_cat), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(ws),
                                              batch, L, hw, c, int(groups), float(eps),
                                              int(bool(relu)), _dt(x_cat), _stream())
    _L.check(st, 'brcnn_groupnorm_nhwc_multi')
    return (y, ws) if return_stats else y


def groupnorm_nhwc_multi_backward(dy, x_cat, stats, gamma, beta, groups, batch, sizes, relu):
    """(dx, dgamma, dbeta) of groupnorm_nhwc_multi; `stats` from its return_stats"""
    import ctypes
    _require_gpu(dy, x_cat, stats, gamma, beta)
    assert dy.shape == x_cat.shape and dy.dtype == x_cat.dtype and dy.is_contiguous() and x_cat.is_contiguous()
    L = len(sizes)
    c = x_cat.shape[1]
    hw = (ctypes.c_int * L)(*[h * w for h, w in sizes])
    lib = _L.load()
    nbytes = lib.brcnn_groupnorm_nhwc_multi_backward_workspace_bytes(batch, L, hw, c, int(groups))
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=x_cat.device)
    dx = torch.empty_like(x_cat)
    dgamma = torch.empty(c, dtype=torch.float32, device=x_cat.device)
    dbeta = torch.empty(c, dtype=torch.float32, device=x_cat.device)
    st = lib.brcnn_groupnorm_nhwc_multi_backward(_ptr(dy), _ptr(x_cat), _ptr(stats), _ptr(gamma), _ptr(beta),
                                                 _ptr(dx), _ptr(dgamma), _ptr(dbeta), _ptr(ws), nbytes, batch, L, hw,
                                                 c, int(groups), int(bool(relu)), _dt(x_cat), _stream())
    _L.check(st, 'brcnn_groupnorm_nhwc_multi_backward')
    return dx, dgamma, dbeta


def pack_stem_weight(w, dtype=torch.float32):
    """(Cout,3,7,7) parameter -> the stem kernel's K rows of 4-element pixels (128 bytes each):
    fp32 (Cout,7,1,32) = [co, kh, 0, kw*4 + c]; 16-bit (Cout,4,1,64) = [co, t, 0, r*32 + kw*4 + c] for filter row
    kh = 2t + r (two 8-pixel windows per K row: `brcnn_stem7x7s2_nchw`); zeros elsewhere"""
    cout = w.shape[0]
    p = torch.zeros((cout, 8, 8, 4), dtype=torch.float32, device=w.device)
    p[:, :7, :7, :3] = w.detach().float().permute(0, 2, 3, 1)
    if dtype == torch.float32:
        return p[:, :7].reshape(cout, 7, 1, 32).contiguous()
    return p.reshape(cout, 4, 1, 64).to(dtype).contiguous()


def stem7x7s2_nchw(img, w_packed, scale=None, shift=None, relu=True):
    """ResNet stem on the NCHW image: conv 7x7/s2/p3 (+scale/shift, ReLU) -> (N,Ho,Wo,Cout) NHWC"""
    _require_gpu(img, w_packed, scale, shift)
    n, c, h, w = img.shape
    assert c == 3 and img.dtype == torch.float32
    img = img.contiguous()
    cout = w_packed.shape[0]
    ho, wo = conv_out_size(h, w, 7, 7, 2, 3)
    y = torch.empty((n, ho, wo, cout), dtype=w_packed.dtype, device=img.device)
    lib = _L.load()
    ws = _ws(lib.brcnn_stem_workspace_bytes(n, h, w), img.device)
    st = lib.brcnn_stem7x7s2_nchw(_ptr(img), _ptr(w_packed), _ptr(scale), _ptr(shift), _ptr(y), _ptr(ws),
                                  n, h, w, cout, int(bool(relu)), _dt(y), _stream())
    _L.check(st, 'brcnn_stem7x7s2_nchw')
    return y


def pack_stem_pool_weight(w, dtype=torch.float32):
    """(64,3,7,7) parameter -> the LDS image of `brcnn_stem7x7s2_pool_nchw`'s weights.  K index of filter row kh,
    tap kw, channel c: fp32 k = kh*22 + kw*3 + c (one zero per row), stored [k // 2][co // 32][k % 2][co % 32];
    16-bit k = kh*32 + kw*4 + c, stored [co][232] (zeros elsewhere)"""
    cout = w.shape[0]
    assert tuple(w.shape) == (64, 3, 7, 7)
    wf = w.detach().float()
    if dtype == torch.float32:
        p = torch.zeros((7, 22, cout), dtype=torch.float32, device=w.device)
        p[:, :21] = wf.permute(2, 3, 1, 0).reshape(7, 21, cout)
        return p.reshape(77, 2, 2, 32).permute(0, 2, 1, 3).contiguous()
    p = torch.zeros((cout, 7, 8, 4), dtype=torch.float32, device=w.device)
    p[:, :, :7, :3] = wf.permute(0, 2, 3, 1)
    q = torch.zeros((cout, 232), dtype=torch.float32, device=w.device)
    q[:, :224] = p.reshape(cout, 224)
    return q.to(dtype).contiguous()


def stem7x7s2_pool_nchw(img, w_packed, scale=None, shift=None):
    """frozen ResNet stem in one launch: conv 7x7/s2/p3 (+scale/shift) + ReLU + max-pool 3x3/s2/p1 of the NCHW image
    -> (N,Hp,Wp,64) NHWC in w_packed's dtype (resnet.py:631-636)"""
    _require_gpu(img, w_packed, scale, shift)
    n, c, h, w = img.shape
    assert c == 3 and img.dtype == torch.float32
    img = img.contiguous()
    ho, wo = conv_out_size(h, w, 7, 7, 2, 3)
    hp, wp = conv_out_size(ho, wo, 3, 3, 2, 1)
    y = torch.empty((n, hp, wp, 64), dtype=w_packed.dtype, device=img.device)
    st = _L.load().brcnn_stem7x7s2_pool_nchw(_ptr(img), _ptr(w_packed), _ptr(scale), _ptr(shift), _ptr(y), n, h, w, 64,
                                             _dt(y), _stream())
    _L.check(st, 'brcnn_stem7x7s2_pool_nchw')
    return y


def bottleneck_tail_supported(x, w2, w3, identity):
    """shapes `bottleneck_tail_nhwc` covers: one dtype throughout (fp32 / bf16 / fp16), 64 -> 64 (3x3, packed (64,3,3,64))
    -> 256 channels, rows a multiple of 64 (fp32) / 128 (16-bit)"""
    dt = x.dtype
    if not (x.is_cuda and dt in (torch.float32, torch.bfloat16, torch.float16) and x.dim() == 4 and x.shape[3] == 64):
        return False
    rows = x.shape[0] * x.shape[1] * x.shape[2]
    return tuple(w2.shape) == (64, 3, 3, 64) and tuple(w3.shape) == (256, 1, 1, 64) and w2.dtype == dt and w3.dtype == dt and \
        identity.dtype == dt and tuple(identity.shape) == tuple(x.shape[:3]) + (256,) and \
        rows % (64 if dt == torch.float32 else 128) == 0


def bottleneck_tail_nhwc(x, w2, scale2, shift2, w3, scale3, shift3, identity):
    """relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(x))))) + identity) of a frozen stage-1 Bottleneck (resnet.py:263-302) in one
    launch; equal to the two `conv2d_nhwc` launches it replaces.  Weights packed (Cout,KH,KW,Cin), BN folded."""
    _require_gpu(x, w2, scale2, shift2, w3, scale3, shift3, identity)
    assert bottleneck_tail_supported(x, w2, w3, identity)
    x, identity = x.contiguous(), identity.contiguous()
    n, h, w, _ = x.shape
    y = torch.empty((n, h, w, 256), dtype=x.dtype, device=x.device)
    lib = _L.load()
    if x.dtype == torch.float32:
        st = lib.brcnn_bottleneck_tail_f32(_ptr(x), _ptr(w2.contiguous()), _ptr(scale2), _ptr(shift2), _ptr(w3.contiguous()),
                                           _ptr(scale3), _ptr(shift3), _ptr(identity), _ptr(y), n, h, w, _stream())
    else:
        st = lib.brcnn_bottleneck_tail_16(_ptr(x), _ptr(w2.contiguous()), _ptr(scale2), _ptr(shift2), _ptr(w3.contiguous()),
                                          _ptr(scale3), _ptr(shift3), _ptr(identity), _ptr(y), n, h, w, _dt(x), _stream())
    _L.check(st, 'brcnn_bottleneck_tail')
    return y


def linear_nhwc(x, w, bias=None, relu=False, out_f32=False):
    """x (M,K) @ w (N,K)^T + bias: the 1x1 case of the implicit GEMM with H=W=1."""
    m, k = x.shape
    y = conv2d_nhwc(x.reshape(m, 1, 1, k), w.reshape(w.shape[0], 1, 1, k), None, bias, None, relu,
                    out_f32=out_f32)
    return y.reshape(m, w.shape[0])


class _MaxPool3x3s2(Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        n, h, w, c = x.shape
        ho, wo = conv_out_size(h, w, 3, 3, 2, 1)
        y = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
        st = _L.load().brcnn_maxpool3x3s2_nhwc(_ptr(x), _ptr(y), n, h, w, c, _dt(x), _stream())
        _L.check(st, 'brcnn_maxpool3x3s2_nhwc')
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        n, h, w, c = x.shape
        dy = dy.to(x.dtype).contiguous()
        dx = torch.empty_like(x)
        st = _L.load().brcnn_maxpool3x3s2_nhwc_backward(_ptr(x), _ptr(y), _ptr(dy), _ptr(dx), n, h, w, c, _dt(x),
                                                        _stream())
        _L.check(st, 'brcnn_maxpool3x3s2_nhwc_backward')
        return dx


def maxpool3x3s2_nhwc(x):
    """ResNet stem max-pool (resnet.py:611) on an NHWC map; differentiable (a trainable stem,
    frozen_stages < 0, gets its gradient through `brcnn_maxpool3x3s2_nhwc_backward`)"""
    _require_gpu(x)
    if torch.is_grad_enabled() and x.requires_grad:
        return _MaxPool3x3s2.apply(x)
    n, h, w, c = x.shape
    ho, wo = conv_out_size(h, w, 3, 3, 2, 1)
    y = torch.empty((n, ho, wo, c), dtype=x.dtype, device=x.device)
    st = _L.load().brcnn_maxpool3x3s2_nhwc(_ptr(x), _ptr(y), n, h, w, c, _dt(x), _stream())
    _L.check(st, 'brcnn_maxpool3x3s2_nhwc')
    return y


def groupnorm_nhwc(x, gamma, beta, groups, eps=1e-5, relu=False):
    _require_gpu(x, gamma, beta)
    n, h, w, c = x.shape
    y = torch.empty_like(x)
    ws = torch.empty((n * groups * 2,), dtype=torch.float64, device=x.device)
    st = _L.load().brcnn_groupnorm_nhwc(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(ws), n,
                                        h * w, c, int(groups), float(eps), int(bool(relu)), _dt(x),
                                        _stream())
    _L.check(st, 'brcnn_groupnorm_nhwc')
    return y


def upsample_nearest_add_nhwc_(dst, src):
    """dst += nearest_upsample(src, size=dst.shape[1:3]) in place (FPN top-down step)."""
    _require_gpu(dst, src)
    n, hd, wd, c = dst.shape
    _, hs, ws_, _ = src.shape
    st = _L.load().brcnn_upsample_nearest_add_nhwc(_ptr(dst), _ptr(src), n, hd, wd, hs, ws_, c,
                                                   _dt(dst), _stream())
    _L.check(st, 'brcnn_upsample_nearest_add_nhwc')
    return dst


class _UpsampleAdd(Function):
    """dst + nearest_upsample(src, size=dst) on NHWC maps with its gradients (FPN top-down step in training)"""

    @staticmethod
    def forward(ctx, dst, src):
        _require_gpu(dst, src)
        dst, src = dst.contiguous(), src.contiguous()
        n, hd, wd, c = dst.shape
        _, hs, ws_, _ = src.shape
        out = torch.empty_like(dst)
        st = _L.load().brcnn_upsample_nearest_add_nhwc_out(_ptr(dst), _ptr(src), _ptr(out), n, hd, wd, hs, ws_, c,
                                                            _dt(dst), _stream())
        _L.check(st, 'brcnn_upsample_nearest_add_nhwc_out')
        ctx.shapes = (n, hd, wd, hs, ws_, c)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        n, hd, wd, hs, ws_, c = ctx.shapes
        dout = dout.contiguous()
        dsrc = None
        if ctx.needs_input_grad[1]:
            dsrc = torch.empty((n, hs, ws_, c), dtype=dout.dtype, device=dout.device)
            st = _L.load().brcnn_upsample_nearest_add_nhwc_backward(_ptr(dout), _ptr(dsrc), n, hd, wd, hs, ws_, c,
                                                                     _dt(dout), _stream())
            _L.check(st, 'brcnn_upsample_nearest_add_nhwc_backward')
        return (dout if ctx.needs_input_grad[0] else None), dsrc


def upsample_nearest_add_nhwc(dst, src):
    """differentiable dst + nearest_upsample(src, size=dst.shape[1:3]) (necks/fpn.py:178-181)"""
    assert dst.dtype == src.dtype and dst.shape[3] == src.shape[3] and dst.shape[3] % 4 == 0
    return _UpsampleAdd.apply(dst, src)


def colsum(x):
    """column sums of a (rows, C) fp32 / bf16 tensor as fp32 (C,): bias gradients, deterministic"""
    _require_gpu(x)
    assert x.dim() == 2 and x.is_contiguous()
    rows, c = x.shape
    c4 = c // 4
    if c % 4 or (c4 < 256 and c4 & (c4 - 1)):
        return x.float().sum(0)
    lib = _L.load()
    nb = lib.brcnn_colsum_workspace_bytes(rows, c)
    ws = torch.empty((nb + 3) // 4, dtype=torch.float32, device=x.device)
    out = torch.empty((c,), dtype=torch.float32, device=x.device)
    st = lib.brcnn_colsum(_ptr(x), _ptr(out), _ptr(ws), nb, rows, c, _dt(x), _stream())
    _L.check(st, 'brcnn_colsum')
    return out


def nchw_to_nhwc(x):
    _require_gpu(x)
    n, c, h, w = x.shape
    x = x.contiguous().float()
    y = torch.empty((n, h, w, c), dtype=torch.float32, device=x.device)
    st = _L.load().brcnn_nchw_to_nhwc(_ptr(x), _ptr(y), n, c, h * w, DT_F32, _stream())
    _L.check(st, 'brcnn_nchw_to_nhwc')
    return y


def nhwc_to_nchw(x):
    _require_gpu(x)
    n, h, w, c = x.shape
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    st = _L.load().brcnn_nhwc_to_nchw(_ptr(x.contiguous()), _ptr(y), n, c, h * w, DT_F32, _stream())
    _L.check(st, 'brcnn_nhwc_to_nchw')
    return y


# ----------------------------------------------------------------------------- RPN stage
def _last_dim_strided(t):
    """(rows, A) view info of a tensor whose last dim is dense and whose other dims collapse to
    one row stride (a channel slice of an NHWC tensor): returns (rows, A, row_stride)"""
    a = t.shape[-1]
    assert t.stride(-1) == 1
    rows = t.numel() // a
    rs = t.stride(-2) if t.dim() > 1 else a
    for d in range(t.dim() - 2, 0, -1):          # collapsible: stride[d-1] == stride[d]*shape[d]
        assert t.stride(d - 1) == t.stride(d) * t.shape[d], 'rpn ops need row-collapsible views'
    return rows, a, rs


def rpn_score(cls, iou):
    """sqrt(sigmoid(cls) * sigmoid(iou)) element-wise (atss_rpn_head.py:712-725).  cls / iou
    may be channel slices of a wider NHWC tensor (read in place); the result is dense."""
    _require_gpu(cls, iou)
    rows, a, cs = _last_dim_strided(cls)
    rows2, a2, is_ = _last_dim_strided(iou)
    assert (rows, a) == (rows2, a2)
    out = torch.empty(cls.shape, dtype=torch.float32, device=cls.device)
    st = _L.load().brcnn_rpn_score(_ptr(cls), _ptr(iou), _ptr(out), rows, a, cs, is_, _stream())
    _L.check(st, 'brcnn_rpn_score')
    return out


def rpn_topk(scores, k):
    """Per level top-k of the proposal scores (atss_rpn_head.py:727-737: sort descending, keep
    nms_pre) for all levels and images in one launch.  `scores`: list of (B, n_l) dense fp32;
    returns per level (score (B, min(k,n_l)), index (B, min(k,n_l)) int64) in (score
    descending, index ascending) order; a level with n_l <= k passes through unsorted."""
    import ctypes
    _require_gpu(*scores)
    L = len(scores)
    B = scores[0].shape[0]
    ns = [int(s.shape[1]) for s in scores]
    for s in scores:
        assert s.dim() == 2 and s.shape[0] == B and s.is_contiguous() and s.dtype == torch.float32
    if not 0 < k <= 4096:
        raise _L.BrcnnHipError(f'rpn_topk: k={k} outside (0, 4096]')
    dev = scores[0].device
    out_s = [torch.empty((B, min(k, n)), dtype=torch.float32, device=dev) for n in ns]
    out_i = [torch.empty((B, min(k, n)), dtype=torch.int64, device=dev) for n in ns]
    pa = lambda ts: (ctypes.c_void_p * L)(*[t.data_ptr() for t in ts])  # noqa: E731
    lib = _L.load()
    n_arr = (ctypes.c_int * L)(*ns)
    wsb = lib.brcnn_rpn_topk_workspace_bytes(n_arr, L, B, int(k))
    ws = _ws(wsb, dev)
    st = lib.brcnn_rpn_topk(pa(scores), n_arr, L, B, int(k), pa(out_s), pa(out_i), _ptr(ws), wsb, _stream())
    _L.check(st, 'brcnn_rpn_topk')
    return list(zip(out_s, out_i))


def rpn_decode(topk_inds, bbox_pred, base_anchors, feat_hw, stride, means, stds, max_shape,
               min_size, wh_ratio_clip=16 / 1000, pred_scale=1.0):
    """topk_inds (B,k) int64 flat anchor indices of one level; bbox_pred (B,H,W,4A) NHWC, possibly
    a channel slice of a wider tensor; deltas are multiplied by `pred_scale` first.
    Returns (proposals (B,k,4), valid (B,k) uint8)."""
    import ctypes
    _require_gpu(topk_inds, bbox_pred, base_anchors)
    b, k = topk_inds.shape
    h, w = feat_hw
    a = base_anchors.size(0)
    _, a4, pstride = _last_dim_strided(bbox_pred)
    assert a4 == 4 * a
    props = torch.empty((b, k, 4), dtype=torch.float32, device=bbox_pred.device)
    valid = torch.empty((b, k), dtype=torch.uint8, device=bbox_pred.device)
    m4 = (ctypes.c_float * 4)(*[float(v) for v in means])
    s4 = (ctypes.c_float * 4)(*[float(v) for v in stds])
    sw, sh = (stride, stride) if isinstance(stride, int) else stride
    mh, mw = (float(max_shape[0]), float(max_shape[1])) if max_shape is not None else (0.0, 0.0)
    st = _L.load().brcnn_rpn_decode(_ptr(topk_inds.contiguous()), _ptr(bbox_pred), int(pstride),
                                    float(pred_scale),
                                    _ptr(base_anchors.contiguous().float()), b, k, h, w, a,
                                    int(sw), int(sh), m4, s4, float(wh_ratio_clip), mh, mw,
                                    float(min_size), _ptr(props), _ptr(valid), _stream())
    _L.check(st, 'brcnn_rpn_decode')
    return props, valid


def rpn_decode_levels(topk_inds, bbox_preds, base_anchors, feat_hws, strides, means, stds, max_shape, min_size,
                      wh_ratio_clip=16 / 1000, pred_scales=None, max_shapes=None):
    """rpn_decode for all levels in one launch.  Per-level lists: topk_inds (B,k_l) int64, bbox_preds
    (B,H_l,W_l,4A) NHWC (possibly channel slices), base_anchors (A,4), feat_hws, strides.
    Returns proposals (B,T,4), valid (B,T) bool, ids (B,T) int64 (level index), T = sum k_l."""
    import ctypes
    L = len(topk_inds)
    _require_gpu(*topk_inds, *bbox_preds, *base_anchors)
    b = topk_inds[0].shape[0]
    a = base_anchors[0].size(0)
    inds = [t.contiguous() for t in topk_inds]
    bases = [t.contiguous().float() for t in base_anchors]
    pstr = []
    for p_ in bbox_preds:
        _, a4, ps = _last_dim_strided(p_)
        assert a4 == 4 * a
        pstr.append(int(ps))
    counts = [int(t.shape[1]) for t in inds]
    T = sum(counts)
    dev = bbox_preds[0].device
    props = torch.empty((b, T, 4), dtype=torch.float32, device=dev)
    valid = torch.empty((b, T), dtype=torch.bool, device=dev)
    ids = torch.empty((b, T), dtype=torch.int64, device=dev)
    ptrs = lambda ts: (ctypes.c_void_p * L)(*[t.data_ptr() for t in ts])     # noqa: E731
    ints = lambda vs: (ctypes.c_int * L)(*[int(v) for v in vs])               # noqa: E731
    sw = [s_ if isinstance(s_, int) else s_[0] for s_ in strides]
    sh = [s_ if isinstance(s_, int) else s_[1] for s_ in strides]
    m4 = (ctypes.c_float * 4)(*[float(v) for v in means])
    s4 = (ctypes.c_float * 4)(*[float(v) for v in stds])
    mh, mw = (float(max_shape[0]), float(max_shape[1])) if max_shape is not None else (0.0, 0.0)
    if isinstance(pred_scales, torch.Tensor):
        # the live Scale parameters, read on the device (training: no host read-back)
        _require_gpu(pred_scales)
        assert pred_scales.numel() == L and pred_scales.dtype == torch.float32 and pred_scales.is_contiguous()
        if max_shapes is not None:      # (B, 2) [h, w] per-image clip border on the device
            _require_gpu(max_shapes)
            assert max_shapes.shape == (b, 2) and max_shapes.dtype == torch.float32 and max_shapes.is_contiguous()
        st = _L.load().brcnn_rpn_decode_levels_dscale(
            ptrs(inds), ptrs(bbox_preds), ints(pstr), _ptr(pred_scales), _ptr(max_shapes), ptrs(bases), b, L, ints(counts),
            ints([h for h, _ in feat_hws]), ints([w for _, w in feat_hws]), a, ints(sw), ints(sh), m4, s4,
            float(wh_ratio_clip), mh, mw, float(min_size), _ptr(props), _ptr(valid), _ptr(ids), _stream())
        _L.check(st, 'brcnn_rpn_decode_levels_dscale')
        return props, valid, ids
    scales = (ctypes.c_float * L)(*[float(v) for v in (pred_scales or [1.0] * L)])
    st = _L.load().brcnn_rpn_decode_levels(ptrs(inds), ptrs(bbox_preds), ints(pstr), scales, ptrs(bases), b, L,
                                           ints(counts), ints([h for h, _ in feat_hws]), ints([w for _, w in feat_hws]),
                                           a, ints(sw), ints(sh), m4, s4, float(wh_ratio_clip), mh, mw, float(min_size),
                                           _ptr(props), _ptr(valid), _ptr(ids), _stream())
    _L.check(st, 'brcnn_rpn_decode_levels')
    return props, valid, ids


def rcnn_decode(probs, bbox_pred, dets, num, max_shape, scale_factor, num_classes, score_thr, means, stds,
                wh_ratio_clip=16 / 1000):
    """Fused second-stage candidates (brcnn_rcnn_decode): probs (B*K, C+1), bbox_pred (B*K, 4C), dets (B,K,5)
    padded proposals with priors, num (B,) int32, max_shape (B,2), scale_factor (B,4) or None.
    Returns boxes (B,K*C,4), scores (B,K*C), labels (B,K*C) int64, valid (B,K*C) bool."""
    import ctypes
    _require_gpu(probs, bbox_pred, dets, num, max_shape, scale_factor)
    B, K, _ = dets.shape
    C = int(num_classes)
    assert probs.shape == (B * K, C + 1) and bbox_pred.shape == (B * K, 4 * C)
    probs, bbox_pred, dets = probs.contiguous().float(), bbox_pred.contiguous().float(), dets.contiguous().float()
    dev = dets.device
    boxes = torch.empty((B, K * C, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((B, K * C), dtype=torch.float32, device=dev)
    labels = torch.empty((B, K * C), dtype=torch.int64, device=dev)
    valid = torch.empty((B, K * C), dtype=torch.bool, device=dev)
    m4 = (ctypes.c_float * 4)(*[float(v) for v in means])
    s4 = (ctypes.c_float * 4)(*[float(v) for v in stds])
    sf = scale_factor.contiguous().float() if scale_factor is not None else None
    st = _L.load().brcnn_rcnn_decode(_ptr(probs), _ptr(bbox_pred), _ptr(dets), _ptr(num.to(torch.int32).contiguous()),
                                     _ptr(max_shape.contiguous().float()), _ptr(sf), B, K, C, float(score_thr), m4, s4,
                                     float(wh_ratio_clip), _ptr(boxes), _ptr(scores), _ptr(labels), _ptr(valid), _stream())
    _L.check(st, 'brcnn_rcnn_decode')
    return boxes, scores, labels, valid


# --------------------------------------------------------------------------- input front door
_FLIP_CODE = {None: 0, 'horizontal': 1, 'vertical': 2, 'diagonal': 3}


def preprocess_u8(src_u8, out, new_w, new_h, flip_direction, mean, std, to_rgb=True):
    """Resize(new_w,new_h) -> flip -> Normalize -> Pad of one decoded uint8 BGR image
    (transforms.py Resize/RandomFlip/Normalize/Pad) in one pass.  `src_u8` (H,W,3) uint8 on the
    device, `out` (3,PH,PW) fp32 on the device (PH >= new_h, PW >= new_w; the rest is zeroed)."""
    import ctypes
    _require_gpu(src_u8, out)
    assert src_u8.dtype == torch.uint8 and src_u8.dim() == 3 and src_u8.shape[2] == 3 and src_u8.is_contiguous()
    assert out.dtype == torch.float32 and out.dim() == 3 and out.shape[0] == 3 and out.is_contiguous()
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s3 = (ctypes.c_float * 3)(*[float(v) for v in std])
    st = _L.load().brcnn_preprocess_u8(_ptr(src_u8), src_u8.shape[0], src_u8.shape[1], _ptr(out), int(new_h),
                                       int(new_w), out.shape[1], out.shape[2], _FLIP_CODE[flip_direction],
                                       m3, s3, int(bool(to_rgb)), _stream())
    _L.check(st, 'brcnn_preprocess_u8')
    return out


# --------------------------------------------------------------------------- Res2Net / DCNv2
def avgpool_out_size(size, kernel, stride, pad, ceil_mode):
    if ceil_mode:
        o = (size + 2 * pad - kernel + stride - 1) // stride + 1
        if (o - 1) * stride >= size + pad:
            o -= 1
        return o
    return (size + 2 * pad - kernel) // stride + 1


def avgpool_nhwc(x, kernel, stride, pad=0, ceil_mode=False, count_include_pad=True):
    """torch.nn.AvgPool2d on an NHWC fp32 map"""
    _require_gpu(x)
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    n, h, w, c = x.shape
    ho, wo = avgpool_out_size(h, kernel, stride, pad, ceil_mode), avgpool_out_size(w, kernel, stride, pad, ceil_mode)
    y = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
    st = _L.load().brcnn_avgpool_nhwc(_ptr(x), _ptr(y), n, h, w, c, int(kernel), int(stride), int(pad),
                                      int(bool(ceil_mode)), int(bool(count_include_pad)), _stream())
    _L.check(st, 'brcnn_avgpool_nhwc')
    return y


def deform_im2col_nhwc(x, offset_mask, kernel=3, stride=1, pad=1, dilation=1, channels_padded=None):
    """mmcv modulated deformable im2col (deform_groups 1): x (N,H,W,C), offset_mask (N,Ho,Wo,>=27)
    raw conv_offset output -> columns (N*Ho*Wo, k*k*Cp), K order (tap, c), zero in the pad channels"""
    _require_gpu(x, offset_mask)
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    n, h, w, c = x.shape
    cp = c if channels_padded is None else channels_padded
    ho, wo = conv_out_size(h, w, dilation * (kernel - 1) + 1, dilation * (kernel - 1) + 1, stride, pad)
    assert tuple(offset_mask.shape[:3]) == (n, ho, wo) and offset_mask.is_contiguous()
    col = torch.empty((n * ho * wo, kernel * kernel * cp), dtype=torch.float32, device=x.device)
    st = _L.load().brcnn_deform_im2col_nhwc(_ptr(x), _ptr(offset_mask), _ptr(col), n, h, w, c, kernel, kernel,
                                            int(stride), int(pad), int(dilation), offset_mask.shape[3], cp, _stream())
    _L.check(st, 'brcnn_deform_im2col_nhwc')
    return col, (ho, wo)
