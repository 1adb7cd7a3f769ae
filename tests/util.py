"""Shared helpers of the parity tests (seeded synthetic inputs, tolerant comparisons)."""
import numpy as np
import torch

import brcnn  # noqa: F401


def rand_boxes(n, img_w=1333.0, img_h=800.0, seed=0, min_size=2.0, max_size=400.0):
    g = torch.Generator().manual_seed(seed)
    cx = torch.rand(n, generator=g) * img_w
    cy = torch.rand(n, generator=g) * img_h
    lw = torch.rand(n, generator=g) * (np.log(max_size) - np.log(min_size)) + np.log(min_size)
    lh = lw + (torch.rand(n, generator=g) - 0.5) * 1.4
    w, h = lw.exp(), lh.exp()
    b = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    b[:, 0::2].clamp_(0, img_w)
    b[:, 1::2].clamp_(0, img_h)
    return b.float()


def clustered_boxes(n, n_clusters=40, seed=0, img_w=1333.0, img_h=800.0):
    """many heavily overlapping boxes (what an RPN emits): NMS actually suppresses"""
    g = torch.Generator().manual_seed(seed)
    centers = rand_boxes(n_clusters, img_w, img_h, seed + 1, 16, 300)
    idx = torch.randint(0, n_clusters, (n,), generator=g)
    jit = (torch.rand(n, 4, generator=g) - 0.5) * 0.25
    c = centers[idx]
    wh = torch.stack([c[:, 2] - c[:, 0], c[:, 3] - c[:, 1]], 1).repeat(1, 2)
    b = c + jit * wh
    b[:, 0::2].clamp_(0, img_w)
    b[:, 1::2].clamp_(0, img_h)
    x1 = torch.minimum(b[:, 0], b[:, 2]); x2 = torch.maximum(b[:, 0], b[:, 2])
    y1 = torch.minimum(b[:, 1], b[:, 3]); y2 = torch.maximum(b[:, 1], b[:, 3])
    return torch.stack([x1, y1, x2, y2], 1).float()


def tie_free_scores(n, seed=0):
    """U(0,1) scores made pairwise distinct: a random permutation of a strictly
    increasing fp32 ramp."""
    g = torch.Generator().manual_seed(seed)
    base = (torch.arange(n, dtype=torch.float64) + 0.5) / n
    s = base[torch.randperm(n, generator=g)].float()
    assert torch.unique(s).numel() == n
    return s


def rand_rois(k, batch, img_w=1333.0, img_h=800.0, seed=0, min_size=4.0, max_size=900.0):
    b = rand_boxes(k, img_w, img_h, seed, min_size, max_size)
    g = torch.Generator().manual_seed(seed + 7)
    bi = torch.randint(0, batch, (k, 1), generator=g).float()
    return torch.cat([bi, b], 1)


def ulp_diff(a, b):
    a = a.detach().cpu().float().contiguous().view(torch.int32).long()
    b = b.detach().cpu().float().contiguous().view(torch.int32).long()
    a = torch.where(a < 0, -(a & 0x7fffffff), a)
    b = torch.where(b < 0, -(b & 0x7fffffff), b)
    return (a - b).abs()


from brcnn.synth import seeded_state_dict  # noqa: E402,F401  (product-side helper: bench.py uses it too)


def demo_inputs(batch=2, h=128, w=192, num_classes=4, seed=0, num_gt=5):
    """synthetic batch in the style of tests/test_models/test_forward.py:438-513"""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(batch, 3, h, w, generator=g)
    img_metas = [{'img_shape': (h, w - 3, 3), 'ori_shape': (h, w - 3, 3), 'pad_shape': (h, w, 3),
                  'scale_factor': np.array([1.1, 1.2, 1.1, 1.2], dtype=np.float32), 'flip': False,
                  'filename': '<demo>.png'} for _ in range(batch)]
    gt_bboxes, gt_labels = [], []
    for _ in range(batch):
        cx = torch.rand(num_gt, generator=g) * (w - 3)
        cy = torch.rand(num_gt, generator=g) * h
        bw = torch.rand(num_gt, generator=g) * (w - 3) * 0.5 + 8
        bh = torch.rand(num_gt, generator=g) * h * 0.5 + 8
        b = torch.stack([(cx - bw / 2).clamp(0, w - 3), (cy - bh / 2).clamp(0, h),
                         (cx + bw / 2).clamp(0, w - 3), (cy + bh / 2).clamp(0, h)], 1)
        gt_bboxes.append(b)
        gt_labels.append(torch.randint(0, num_classes, (num_gt,), generator=g))
    return img, img_metas, gt_bboxes, gt_labels


def variant_inputs(num_classes, score_cols, seed):
    """seeded pyramid, proposals (n, 4 + score_cols) near the GTs, GTs for a 2 x 128 x 192 batch"""
    g = torch.Generator().manual_seed(seed)
    sizes = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
    feats = [torch.randn(2, 256, h, w, generator=g) for h, w in sizes]
    _, metas, gts, gls = demo_inputs(2, 128, 192, num_classes=num_classes, seed=seed)
    props = []
    for b in range(2):
        jit = gts[b].repeat(30, 1) + torch.randn(gts[b].shape[0] * 30, 4, generator=g) * 6
        rnd = rand_boxes(150, img_w=192., img_h=128., seed=seed + b, min_size=4., max_size=120.)
        boxes = torch.cat([jit, rnd], 0)
        boxes[:, 0::2] = boxes[:, 0::2].clamp(0, 192)
        boxes[:, 1::2] = boxes[:, 1::2].clamp(0, 128)
        boxes = torch.cat([torch.min(boxes[:, :2], boxes[:, 2:]), torch.max(boxes[:, :2], boxes[:, 2:]) + 1], 1)
        sc = torch.rand(boxes.shape[0], score_cols, generator=g)
        props.append(torch.cat([boxes, sc], 1))
    return feats, metas, gts, gls, props


def fullsize_head_outputs(seed=20, batch=8):
    """seeded synthetic RPN head outputs at the BASELINE map sizes (3 x 800 x 1344 input).  Not stored in
    the fixture g20 (39 MB): the golden generator and the parity test both regenerate them from the same
    CPU generator"""
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    g = torch.Generator().manual_seed(seed)
    cls = [torch.randn(batch, 9, h, w, generator=g) * 2 - 1 for h, w in sizes]
    reg = [torch.randn(batch, 36, h, w, generator=g) * 0.5 for h, w in sizes]
    iou = [torch.randn(batch, 9, h, w, generator=g) * 2 for h, w in sizes]
    return sizes, cls, reg, iou


def fullsize_softnms_candidates(b, n=2000, num_classes=80):
    """image `b` of the soft-NMS stress (BASELINE configs[4]: 2000 proposals x 80 classes, score_thr 1e-4): the inputs
    of `multiclass_nms` -- per-class boxes (n, 4C) and scores (n, C+1).  Built from the CPU generator with +, -, * and /
    only (no exp / softmax / sqrt, whose vectorised implementations differ in the last bit between host CPUs), so the
    golden generator and the test regenerate the SAME bits on any machine.  250 clusters of 8 heavily overlapping
    proposals; every candidate clears the threshold (162 000 per image go through soft-NMS), and a sparse strong
    component shared by a cluster's members of one class puts winners AND their decayed neighbours into the top 200."""
    g = torch.Generator().manual_seed(2100 + b)
    C, m = num_classes, 8
    nc = n // m
    cx, cy = torch.rand(nc, generator=g) * 1333, torch.rand(nc, generator=g) * 800
    u = torch.rand(nc, generator=g)
    w = 16 + 384 * u * u                                          # 16 .. 400 px, mostly small
    h = w * (0.5 + 1.5 * torch.rand(nc, generator=g))
    base = torch.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).repeat_interleave(m, 0)      # (n, 4)
    size = torch.stack([w, h, w, h], 1).repeat_interleave(m, 0)
    base = base + (torch.rand(n, 4, generator=g) - 0.5) * 0.1 * size               # member jitter: IoU ~ 0.8 inside a cluster
    boxes = base[:, None, :] + (torch.rand(n, C, 4, generator=g) - 0.5) * 0.04 * size[:, None, :]
    boxes = torch.stack([boxes[..., 0].clamp(0, 1333), boxes[..., 1].clamp(0, 800),
                         boxes[..., 2].clamp(0, 1333), boxes[..., 3].clamp(0, 800)], -1).reshape(n, 4 * C)
    strong = (torch.randperm(nc * (C + 1), generator=g).float() + 1) / (nc * (C + 1) + 1)
    for _ in range(10):
        strong = strong * strong                                  # r^1024: ~0.3 % of the (cluster, class) pairs stay above 0.05
    strong = strong.view(nc, C + 1).repeat_interleave(m, 0) * (0.5 + 0.5 * torch.rand(n, C + 1, generator=g))
    weak = (torch.randperm(n * (C + 1), generator=g).float() + 1) / (n * (C + 1) + 1)
    scores = 0.0002 + 0.001 * weak.view(n, C + 1) + 0.99 * strong
    return boxes, scores


def fullsize_box_head_outputs(g, n, num_classes=4):
    """seeded box-head outputs of one image's `n` proposals (generator `g` is advanced)"""
    cs = torch.randn(n, num_classes + 1, generator=g) * 2
    cs[:, num_classes] -= 1.0
    bp = torch.randn(n, 4 * num_classes, generator=g) * 0.3
    return cs, bp
