"""Batch-level, device-resident restatement of mmcv `batched_nms` for a whole mini-batch.

The reference calls `batched_nms(boxes, scores, idxs, nms_cfg)` once per image from a
Python loop (atss_rpn_head.py:485,756; prob_roi_head.py:263 -> bbox_nms.py:86), after
filtering candidates with boolean masks (host syncs).  Here every image owns a fixed slot of
`T` candidates; the mask filter becomes an order-preserving compaction inside the slot, the
class/level separation uses the same coordinate-offset trick on the same fp32 values
(`boxes + idxs * (max_coordinate + 1)`, max over the image's surviving boxes), and one
segmented NMS launch serves all images.  No host synchronisation.

Valid only below mmcv's `split_thr` (default 10000 candidates per image), where mmcv itself
takes the offset path; callers fall back to `ops.batched_nms` per image above it.
"""
import torch

from . import ops


def batched_nms_images(boxes, scores, ids, valid, iou_threshold, max_keep, offset=0, fused=None):
    """boxes (B,T,4), scores (B,T), ids (B,T) long, valid (B,T) bool.
    Returns dets (B,K,5) zero padded, ids_kept (B,K) long (-1 padded), num (B,) int32, with
    K = max_keep (or T when max_keep <= 0)."""
    B, T = scores.shape
    device = scores.device
    K = max_keep if max_keep > 0 else T
    K = min(K, T)
    if fused is None:
        fused = boxes.is_cuda and boxes.dtype == torch.float32
    if fused:
        # the same steps as below in two launches around the segmented NMS
        c_boxes, c_scores, c_ids, boxes_for_nms, ranges = ops.nms_prepare(boxes, scores, ids, valid)
        keep, num = ops.nms_ranges(boxes_for_nms.view(-1, 4), c_scores.reshape(-1), ranges, T,
                                   iou_threshold, offset, max_keep)
        dets, ids_kept = ops.nms_collect(keep, num, c_boxes, c_scores, c_ids, K)
        return dets, ids_kept.to(ids.dtype), num
    cnt = valid.sum(1)
    dest = torch.cumsum(valid, 1) - 1
    dest = torch.where(valid, dest, torch.full_like(dest, T))   # rejected -> overflow column

    def compact(x):
        out = torch.zeros((B, T + 1) + tuple(x.shape[2:]), dtype=x.dtype, device=device)
        idx = dest.view(B, T, *([1] * (x.dim() - 2))).expand_as(x)
        out.scatter_(1, idx, x)
        return out[:, :T].contiguous()

    c_boxes, c_scores, c_ids = compact(boxes), compact(scores), compact(ids)
    in_range = torch.arange(T, device=device)[None, :] < cnt[:, None]
    lowest = torch.finfo(c_boxes.dtype).min
    max_coord = torch.where(in_range[..., None], c_boxes, c_boxes.new_full((), lowest)).amax((1, 2))
    offs = c_ids.to(c_boxes) * (max_coord + 1)[:, None]
    boxes_for_nms = c_boxes + offs[..., None]
    seg_begin = (torch.arange(B, device=device) * T).to(torch.int32)
    ranges = torch.stack([seg_begin, seg_begin + cnt.to(torch.int32)], 1)
    keep, num = ops.nms_ranges(boxes_for_nms.view(-1, 4), c_scores.reshape(-1), ranges, T,
                               iou_threshold, offset, max_keep)
    keep = keep.view(B, T)[:, :K]
    kmask = torch.arange(K, device=device)[None, :] < num[:, None]
    keep = torch.where(kmask, keep, seg_begin[:, None].long())
    flat = torch.cat([c_boxes.view(-1, 4), c_scores.reshape(-1, 1)], 1)
    dets = flat[keep.reshape(-1)].view(B, K, 5) * kmask[..., None]
    ids_kept = torch.where(kmask, c_ids.reshape(-1)[keep.reshape(-1)].view(B, K),
                           torch.full((B, K), -1, dtype=c_ids.dtype, device=device))
    return dets, ids_kept, num


def batched_nms_images_by_level(boxes, scores, ids, valid, level_sizes, iou_threshold, max_keep, offset=0,
                                return_ids=False, soft=None, fused=None):
    """mmcv `batched_nms` above `split_thr` for a whole mini-batch, no host sync: NMS per id
    (pyramid level) on the offset boxes, survivors re-sorted by score, first `max_keep`
    (mmcv/ops/nms.py batched_nms, the `for id in torch.unique(idxs)` branch).  Column ranges
    of the (B,T) candidate slots belong to the levels in order (`level_sizes`, host ints,
    sum == T), so after the order-preserving compaction every (image, level) is one contiguous
    segment and ONE segmented launch serves all images and levels.
    Returns dets (B,K,5) zero padded and num (B,) int32."""
    B, T = scores.shape
    L = len(level_sizes)
    assert sum(level_sizes) == T
    device = scores.device
    K = min(max_keep, T) if max_keep > 0 else T
    if fused is None:
        fused = boxes.is_cuda and boxes.dtype == torch.float32 and T <= 16384 and L <= 8
    if fused:
        # compaction + per-(image, id) ranges, the segmented (soft-)NMS, the per-image re-sort: 3 entries
        c_boxes, c_scores, c_ids, boxes_for_nms, ranges = ops.nms_prepare(boxes, scores, ids, valid,
                                                                          level_sizes=level_sizes)
        sdets = None
        if soft is not None:
            method = {'naive': 0, 'linear': 1, 'gaussian': 2}[soft.get('method', 'linear')]
            sdets, keep, num = ops.soft_nms_ranges(boxes_for_nms.view(-1, 4), c_scores.reshape(-1), ranges,
                                                   iou_threshold, soft.get('sigma', 0.5), soft.get('min_score', 1e-3),
                                                   method, offset)
        else:
            keep, num = ops.nms_ranges(boxes_for_nms.view(-1, 4), c_scores.reshape(-1), ranges, max(level_sizes),
                                       iou_threshold, offset, -1)
        dets, ids_kept, n_kept = ops.nms_collect_sorted(keep, num, ranges, c_boxes, c_scores, c_ids, K, L, sdets)
        if return_ids:
            return dets, ids_kept.to(ids.dtype), n_kept
        return dets, n_kept
    cnt = valid.sum(1)
    dest = torch.cumsum(valid, 1) - 1
    dest = torch.where(valid, dest, torch.full_like(dest, T))

    def compact(x):
        out = torch.zeros((B, T + 1) + tuple(x.shape[2:]), dtype=x.dtype, device=device)
        idx = dest.view(B, T, *([1] * (x.dim() - 2))).expand_as(x)
        out.scatter_(1, idx, x)
        return out[:, :T].contiguous()

    c_boxes, c_scores, c_ids = compact(boxes), compact(scores), compact(ids)
    pos = torch.arange(T, device=device)[None, :]
    in_range = pos < cnt[:, None]
    lowest = torch.finfo(c_boxes.dtype).min
    max_coord = torch.where(in_range[..., None], c_boxes, c_boxes.new_full((), lowest)).amax((1, 2))
    offs = c_ids.to(c_boxes) * (max_coord + 1)[:, None]
    boxes_for_nms = c_boxes + offs[..., None]
    # per-(image, level) survivor counts of the validity filter -> contiguous segments
    bounds = [0]
    for n in level_sizes:
        bounds.append(bounds[-1] + int(n))
    csum = torch.cat([valid.new_zeros((B, 1), dtype=torch.long), torch.cumsum(valid, 1)], 1)   # (B, T+1)
    ends = csum[:, bounds[1:]]                      # (B, L) within-image end of level l
    begins = csum[:, bounds[:-1]]
    base = (torch.arange(B, device=device) * T)[:, None]
    ranges = torch.stack([(begins + base).reshape(-1), (ends + base).reshape(-1)], 1).to(torch.int32)
    if soft is not None:
        # soft-NMS (mmcv batched_nms with type='soft_nms', per-id branch): every (image, id) segment is
        # decayed independently; the survivors carry their decayed scores
        method = {'naive': 0, 'linear': 1, 'gaussian': 2}[soft.get('method', 'linear')]
        sdets, keep, num = ops.soft_nms_ranges(boxes_for_nms.view(-1, 4), c_scores.reshape(-1), ranges, iou_threshold,
                                               soft.get('sigma', 0.5), soft.get('min_score', 1e-3), method, offset)
    else:
        keep, num = ops.nms_ranges(boxes_for_nms.view(-1, 4), c_scores.reshape(-1), ranges, max(level_sizes),
                                   iou_threshold, offset, -1)
    num = num.view(B, L).long()
    lvl = torch.searchsorted(ends.contiguous(), pos.expand(B, T).contiguous(), right=True).clamp(max=L - 1)
    is_kept_slot = in_range & ((pos - torch.gather(begins, 1, lvl)) < torch.gather(num, 1, lvl))
    kept_idx = torch.where(is_kept_slot, keep.view(B, T), torch.full((), B * T, dtype=torch.long, device=device))
    mask = torch.zeros(B * T + 1, dtype=torch.bool, device=device)
    mask.scatter_(0, kept_idx.reshape(-1), is_kept_slot.reshape(-1))
    mask = mask[:B * T].view(B, T)
    if soft is not None:        # scores_after_nms[kept] = decayed score of the pick
        after = torch.zeros(B * T + 1, dtype=c_scores.dtype, device=device)
        after.scatter_(0, kept_idx.reshape(-1), sdets[:B * T, 4] * is_kept_slot.reshape(-1))
        c_scores = after[:B * T].view(B, T)
    # survivors by descending score (ties: ascending index, the shared tie rule), first K
    key = torch.where(mask, c_scores, c_scores.new_full((), float('-inf')))
    _, order = key.sort(dim=1, descending=True, stable=True)
    order = order[:, :K]
    n_kept = torch.clamp(mask.sum(1), max=K).to(torch.int32)
    kmask = torch.arange(K, device=device)[None, :] < n_kept[:, None]
    dets = torch.cat([torch.gather(c_boxes, 1, order[..., None].expand(B, K, 4)),
                      torch.gather(c_scores, 1, order)[..., None]], 2) * kmask[..., None]
    if return_ids:
        ids_kept = torch.where(kmask, torch.gather(c_ids, 1, order), torch.full((), -1, dtype=c_ids.dtype, device=device))
        return dets, ids_kept, n_kept
    return dets, n_kept
