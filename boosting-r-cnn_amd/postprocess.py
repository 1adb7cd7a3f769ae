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


def batched_nms_images(boxes, scores, ids, valid, iou_threshold, max_keep, offset=0):
    """boxes (B,T,4), scores (B,T), ids (B,T) long, valid (B,T) bool.
    Returns dets (B,K,5) zero padded, ids_kept (B,K) long (-1 padded), num (B,) int32, with
    K = max_keep (or T when max_keep <= 0)."""
    B, T = scores.shape
    device = scores.device
    K = max_keep if max_keep > 0 else T
    K = min(K, T)
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
    offs = c_ids.to(c_boxes) * (max_coord + torch.tensor(1).to(c_boxes))[:, None]
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
