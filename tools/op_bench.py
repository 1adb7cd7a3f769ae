"""Op-level roofline report on MI355X (north_star: achieved HBM GB/s on RoIAlign / NMS):
RoIAlign fwd/bwd at the configs' RoI counts on the real pyramid shapes, NMS on RPN-sized
candidate sets, soft-NMS on per-class segments, focal loss on all anchors.  Algorithmic bytes
follow SURVEY.md section 8(d).  Prints one JSON object."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import brcnn  # noqa: F401
from brcnn import ops
from brcnn.autograd import roi_extract_autograd
from tests import util

DEV = 'cuda'
HBM = 8000.0


def timed(fn, n=20):
    """median over n calls, one HIP-event pair each (an average over the loop let a single stall -- an allocator growth,
    a clock ramp after a host-side pause -- show up as a 10x outlier of one entry in two rounds' reports)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return ts[n // 2] * 1e-3


def roialign_bytes(rois, strides, sizes, C=256, out=7):
    """sum_rois C*4*fp_h*fp_w (footprint clipped to the map) + K*C*49*4 + K*20"""
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lv = torch.floor(torch.log2(scale / 56 + 1e-6)).clamp(0, 4).long()
    tot = 0.0
    for l in range(5):
        r = rois[lv == l]
        if not len(r):
            continue
        s = 1.0 / strides[l]
        h, w = sizes[l]
        x1, y1, x2, y2 = r[:, 1] * s - .5, r[:, 2] * s - .5, r[:, 3] * s - .5, r[:, 4] * s - .5
        fw = (torch.floor(x2).clamp(max=w - 1) + 2 - torch.floor(x1).clamp(min=0)).clamp(min=1, max=w)
        fh = (torch.floor(y2).clamp(max=h - 1) + 2 - torch.floor(y1).clamp(min=0)).clamp(min=1, max=h)
        tot += float((fw * fh).sum()) * C * 4
    return tot + len(rois) * C * out * out * 4 + len(rois) * 20


def main():
    B = 8
    strides = [8, 16, 32, 64, 128]
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    g = torch.Generator().manual_seed(0)
    feats = [torch.randn(B, h, w, 256, generator=g).to(DEV) for h, w in sizes]
    res = {}
    for per_img in (256, 512, 2000):
        K = per_img * B
        # log-uniform scale 16-800 px, aspect 0.5-2, uniform centres
        rois = util.rand_rois(K, B, 1333., 800., seed=per_img, min_size=16., max_size=800.)
        # image by image, as bbox2roi / the padded proposal tensor deliver them (score order inside an image = random here)
        rois = rois[torch.argsort(rois[:, 0], stable=True)].contiguous()
        rg = rois.to(DEV)
        t = timed(lambda: ops.roi_extract(feats, rg, 7, strides, 56, 0))
        by = roialign_bytes(rois, strides, sizes)
        # the read term is bounded by the whole pyramid read once (22 400 px x 256 ch x 4 B / image)
        out_b = K * 256 * 49 * 4 + K * 20
        pyramid = B * sum(h * w for h, w in sizes) * 256 * 4
        capped = out_b + min(by - out_b, pyramid)
        res[f'roialign_fwd_{per_img}x{B}'] = dict(us=t * 1e6, algorithmic_MB=by / 1e6, GBs=by / t / 1e9,
                                                  frac_hbm=by / t / 1e9 / HBM, capped_MB=capped / 1e6,
                                                  capped_GBs=capped / t / 1e9, capped_frac_hbm=capped / t / 1e9 / HBM,
                                                  note='capped_*: read term limited to the pyramid read once '
                                                       '(SURVEY 8d); algorithmic_* sums per-RoI footprints')
        fr = [f.clone().requires_grad_() for f in feats]
        go = torch.randn(K, 7, 7, 256, device=DEV)

        def fb():
            for f in fr:
                f.grad = None          # (else autograd adds every call's gradient to the last: five adds over the pyramid)
            out = roi_extract_autograd(fr, rg, 7, strides, 56, 0)
            out.backward(go)
        tfb = timed(fb, 5)
        # the gradient gather alone (C entry point, no autograd / allocation around it)
        import ctypes
        from brcnn import lib as _lib
        Lb = _lib.load()
        grads = [torch.empty_like(f) for f in feats]
        ptrs = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in grads])
        hs_ = (ctypes.c_int * 5)(*[h for h, _ in sizes]); ws_ = (ctypes.c_int * 5)(*[w for _, w in sizes])
        sc_ = (ctypes.c_float * 5)(*[1.0 / s_ for s_ in strides])
        nb = Lb.brcnn_roi_extract_backward_workspace_bytes_ex(K, B, 256, 5, hs_, ws_)     # (incl. the hit-chunk partials of the coarse levels)
        wsp = torch.empty((nb + 3) // 4, dtype=torch.int32, device=DEV)
        call = lambda: Lb.brcnn_roi_extract_backward_gather(ptrs, hs_, ws_, sc_, 5, rg.data_ptr(), go.data_ptr(), B, 256, K, 7, 7, 0,
                                                            56.0, wsp.data_ptr(), nb, 0, None)
        assert call() == 0
        tg = timed(call, 10)
        res[f'roialign_bwd_gather_{per_img}x{B}'] = dict(us=tg * 1e6, note='record + gather (+ chunk sum) kernels only, fp32')
        res[f'roialign_fwd+bwd_{per_img}x{B}'] = dict(us=tfb * 1e6, GBs=2 * by / tfb / 1e9,
                                                      frac_hbm=2 * by / tfb / 1e9 / HBM)
    # NMS: RPN test (8 x 4693), RPN train level segments (8 x 5 x ~3000), R-CNN (8 x 1024)
    for name, lens, keepn in (('rpn_test_8x4693', [4693] * 8, 256), ('rpn_train_40x3030', [3030] * 40, -1),
                              ('rcnn_8x1024', [1024] * 8, 100), ('train_image_2x10000', [10000] * 2, 2000)):
        n = sum(lens)
        seg = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=DEV)
        boxes = util.clustered_boxes(n, n_clusters=60 * len(lens), seed=3).to(DEV)
        scores = util.tie_free_scores(n, seed=4).to(DEV)
        t = timed(lambda: ops.nms_segments(boxes, scores, seg, max(lens), 0.7, 0, keepn))
        words = (max(lens) + 63) // 64
        by = n * 20 + n * words * 8 * 2
        res['nms_' + name] = dict(us=t * 1e6, boxes_per_s=n / t, algorithmic_MB=by / 1e6, GBs=by / t / 1e9,
                                  frac_hbm=by / t / 1e9 / HBM,
                                  note='latency bound: one wavefront per segment resolves the greedy chain')
    # train-step target / loss kernels at the BASELINE size (8 images, 201 600 anchors, 20 GT each)
    from brcnn import core, train_ops
    gen = core.AnchorGenerator(strides=strides, ratios=[0.5, 1.0, 2.0], octave_base_scale=4, scales_per_octave=3)
    anchors = torch.cat(gen.grid_anchors(sizes, 'cpu'), 0).to(DEV).contiguous()
    gts_l = [util.rand_boxes(20, 1333, 800, seed=40 + b, min_size=16, max_size=600).to(DEV) for b in range(B)]
    gts, _, offs = train_ops.flatten_gts(gts_l)
    t = timed(lambda: train_ops.assign_max_iou(anchors, gts, offs, 0.5, 0.5, 0.0, True, batch=B))
    by = B * anchors.shape[0] * 4 + anchors.numel() * 4
    res['assign_rpn_8x201600x20'] = dict(us=t * 1e6, iou_pairs_per_s=2 * B * anchors.shape[0] * 20 / t,
                                         algorithmic_MB=by / 1e6, GBs=by / t / 1e9,
                                         note='two passes (per-gt maxima, assignment): IoU-evaluation bound, not HBM')
    props = torch.stack([torch.cat([util.rand_boxes(2000, 1333, 800, seed=60 + b), torch.rand(2000, 1)], 1)
                         for b in range(B)]).to(DEV)
    nump = torch.full((B,), 2000, dtype=torch.int32, device=DEV)
    t = timed(lambda: train_ops.assign_max_iou(props, gts, offs, 0.6, 0.6, 0.6, False, num_boxes=nump, want_counts=True))
    res['assign_rcnn_8x2000x20'] = dict(us=t * 1e6, note='launch-latency sized')
    y = torch.randn(B * sum(h * w for h, w in sizes), 64, device=DEV).requires_grad_()
    gi = train_ops.assign_max_iou(anchors, gts, offs, 0.5, 0.5, 0.0, True, batch=B)
    base = [b_.to(DEV) for b_ in gen.base_anchors]
    meta = train_ops.RPNLossMeta(B, sizes, gen.strides, base, 9, offs, 2.0, 0.25, -1, 2.0, (0., 0., 0., 0.),
                                 (1., 1., 1., 1.), 16 / 1000, True, 1.0, 2.0, 2.0, 1.0)
    sc = torch.ones(5, device=DEV, requires_grad=True)

    def rl():
        l3, _, _ = train_ops.rpn_loss(y, sc, gi, gts, meta)
        l3.sum().backward()
    t = timed(rl, 10)
    by = 2 * y.numel() * 4 + 2 * gi.numel() * 4
    res['rpn_loss_fwd+bwd_8x22400px'] = dict(us=t * 1e6, algorithmic_MB=by / 1e6, GBs=by / t / 1e9,
                                             frac_hbm=by / t / 1e9 / HBM,
                                             note='forward reads the 64-channel head rows once, backward writes them once; '
                                                  'includes the autograd / launch overhead of 5 small launches')
    cls = torch.randn(4096, 81, device=DEV).requires_grad_()
    bb = torch.randn(4096, 320, device=DEV).requires_grad_()
    lab = torch.randint(0, 81, (4096,), device=DEV)
    pri, tgt = torch.rand(4096, device=DEV), torch.randn(4096, 4, device=DEV)

    def bl():
        o = train_ops.boost_loss(cls, bb, lab, pri, tgt, 80, 0.5, loss_cls_weight=2.0, loss_bbox_weight=2.0)
        (o[0] + o[1]).backward()
    t = timed(bl, 10)
    res['boost_loss_fwd+bwd_4096x81'] = dict(us=t * 1e6, note='launch-latency sized (3 launches)')
    # soft-NMS: 80 classes x 8 images, 250 boxes per segment (config #5 scale / 8)
    lens = [250] * 640
    n = sum(lens)
    seg = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=DEV)
    boxes = util.clustered_boxes(n, n_clusters=6400, seed=5).to(DEV)
    scores = util.tie_free_scores(n, seed=6).to(DEV)
    t = timed(lambda: ops.soft_nms_segments(boxes, scores, seg, 0.7, 0.5, 0.0, 1, 0), 5)
    res['softnms_640x250'] = dict(us=t * 1e6, boxes_per_s=n / t, updates_per_s=sum(l * l / 2 for l in lens) / t,
                                  note='dependent chain per segment; parallel across segments')
    # focal loss over all anchors of 8 images
    x = torch.randn(201600 * B, 1, device=DEV)
    tg = torch.randint(0, 2, (201600 * B,), device=DEV)
    t = timed(lambda: ops.sigmoid_focal_loss(x, tg, 2.0, 0.25, None, 'none'))
    by = x.numel() * (4 + 4 + 8)
    res['focal_fwd_1.6M'] = dict(us=t * 1e6, GBs=by / t / 1e9, frac_hbm=by / t / 1e9 / HBM)
    # input front door: 1080p uint8 frame -> 750x1333 resized, normalised, padded to 768x1344 fp32 CHW
    import time
    from brcnn import pipelines as P
    frame = np.random.RandomState(0).randint(0, 256, (1080, 1920, 3), dtype=np.uint8)
    src = torch.from_numpy(frame).to(DEV)
    out = torch.empty((3, 768, 1344), device=DEV)
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    t = timed(lambda: ops.preprocess_u8(src, out, 1333, 750, 'horizontal', mean, std, True))
    by = 750 * 1333 * 3 * 4 / (1333 / 1920) ** 2 / 4 + out.numel() * 4     # touched source pixels (u8) + fp32 planes
    t0 = time.perf_counter()
    r = P.imflip(P.imresize_u8(frame, (1333, 750)), 'horizontal')
    P.impad_to_multiple(P.imnormalize(r, np.array(mean, np.float32), np.array(std, np.float32), True), 32)
    t_np = time.perf_counter() - t0
    res['preprocess_1080p_to_768x1344'] = dict(us=t * 1e6, images_per_s=1 / t, algorithmic_MB=by / 1e6,
                                              GBs=by / t / 1e9, frac_hbm=by / t / 1e9 / HBM,
                                              cpu_numpy_chain_ms=t_np * 1e3,
                                              note='one image per launch: 4032 workgroups, launch-latency bound')
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
