"""Result sink: COCO bbox mAP without pycocotools.

The reference's `CocoDataset.evaluate` (mmdet/datasets/coco.py:362-560) hands detections to
pycocotools' `COCOeval` (third party, absent here; the published algorithm of
pycocotools/cocoeval.py, bbox branch, is restated): per (image, category) IoU matrix with
crowd handling, greedy matching by descending score at 10 IoU thresholds x 4 area ranges,
precision envelopes sampled at 101 recall points, the 12 summary statistics.
PARITY UNPINNED against pycocotools itself (not installed); pinned by known-answer cases
(tests/test_eval_cpu.py) derived by hand from the algorithm: IoU thresholds, the 101-point envelope,
crowd regions absorbing detections, `ignore` following `iscrowd`, maxDets truncation, area ranges
inclusive at exactly 32^2 and 96^2.
"""
from collections import defaultdict

import numpy as np


class Params:
    def __init__(self):
        self.imgIds, self.catIds = [], []
        self.iouThrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.recThrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.areaRng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.areaRngLbl = ['all', 'small', 'medium', 'large']
        self.useCats = 1


def bbox_iou_xywh(dt, gt, iscrowd):
    """(D,G) IoU of xywh boxes; a crowd ground truth divides by the detection's area"""
    dt = np.asarray(dt, dtype=np.float64).reshape(-1, 4)
    gt = np.asarray(gt, dtype=np.float64).reshape(-1, 4)
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)))
    da = dt[:, 2] * dt[:, 3]
    ga = gt[:, 2] * gt[:, 3]
    w = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - \
        np.maximum(dt[:, None, 0], gt[None, :, 0])
    h = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - \
        np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.where((w > 0) & (h > 0), w * h, 0.0)
    crowd = np.asarray(iscrowd, dtype=bool)[None, :]
    union = np.where(crowd, da[:, None], da[:, None] + ga[None, :] - inter)
    with np.errstate(divide='ignore', invalid='ignore'):
        out = np.where(inter > 0, inter / union, 0.0)
    return out


class COCOeval:
    def __init__(self, cocoGt, cocoDt, iouType='bbox'):
        assert iouType == 'bbox'
        self.cocoGt, self.cocoDt = cocoGt, cocoDt
        self.params = Params()
        self.params.imgIds = sorted(cocoGt.get_img_ids())
        self.params.catIds = sorted(cocoGt.get_cat_ids())
        self.evalImgs, self.eval, self.stats = [], {}, []

    def _prepare(self):
        p = self.params
        img_set, cat_set = set(p.imgIds), set(p.catIds)
        # useCats == 0 (the 'proposal' metric): every annotation of the images, whatever its category
        gts = [a for i in p.imgIds for a in self.cocoGt.imgToAnns.get(i, []) if not p.useCats or a['category_id'] in cat_set]
        dts = [a for i in p.imgIds for a in self.cocoDt.imgToAnns.get(i, []) if not p.useCats or a['category_id'] in cat_set]
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        for g in gts:
            g['ignore'] = 1 if ('iscrowd' in g and g['iscrowd']) else 0
            self._gts[g['image_id'], g['category_id']].append(g)
        for d in dts:
            if d['image_id'] in img_set:
                self._dts[d['image_id'], d['category_id']].append(d)

    def _of(self, table, imgId, catId):
        """the image's boxes of one category, or (useCats == 0, catId -1) of all categories in category order"""
        if self.params.useCats:
            return table[imgId, catId]
        return [a for c in self.params.catIds for a in table[imgId, c]]

    def computeIoU(self, imgId, catId):
        gt, dt = self._of(self._gts, imgId, catId), self._of(self._dts, imgId, catId)
        if len(gt) == 0 and len(dt) == 0:
            return []
        inds = np.argsort([-d['score'] for d in dt], kind='mergesort')
        dt = [dt[i] for i in inds][:self.params.maxDets[-1]]
        return bbox_iou_xywh([d['bbox'] for d in dt], [g['bbox'] for g in gt], [int(g['iscrowd']) for g in gt])

    def evaluateImg(self, imgId, catId, aRng, maxDet):
        p = self.params
        gt, dt = self._of(self._gts, imgId, catId), self._of(self._dts, imgId, catId)
        if len(gt) == 0 and len(dt) == 0:
            return None
        ig = [1 if (g['ignore'] or g['area'] < aRng[0] or g['area'] > aRng[1]) else 0 for g in gt]
        gtind = np.argsort(ig, kind='mergesort')
        gt = [gt[i] for i in gtind]
        dtind = np.argsort([-d['score'] for d in dt], kind='mergesort')
        dt = [dt[i] for i in dtind[:maxDet]]
        iscrowd = [int(g['iscrowd']) for g in gt]
        ious = self.ious[imgId, catId]
        ious = ious[:, gtind] if len(ious) > 0 else ious
        T, G, D = len(p.iouThrs), len(gt), len(dt)
        gtm, dtm = np.zeros((T, G)), np.zeros((T, D))
        gtIg = np.array([ig[i] for i in gtind])
        dtIg = np.zeros((T, D))
        if len(ious) != 0:
            for tind, t in enumerate(p.iouThrs):
                for dind, d in enumerate(dt):
                    iou = min([t, 1 - 1e-10])
                    m = -1
                    for gind in range(G):
                        if gtm[tind, gind] > 0 and not iscrowd[gind]:
                            continue
                        if m > -1 and gtIg[m] == 0 and gtIg[gind] == 1:
                            break
                        if ious[dind, gind] < iou:
                            continue
                        iou = ious[dind, gind]
                        m = gind
                    if m == -1:
                        continue
                    dtIg[tind, dind] = gtIg[m]
                    dtm[tind, dind] = gt[m]['id']
                    gtm[tind, m] = d['id']
        a = np.array([d['area'] < aRng[0] or d['area'] > aRng[1] for d in dt]).reshape((1, len(dt)))
        dtIg = np.logical_or(dtIg, np.logical_and(dtm == 0, np.repeat(a, T, 0)))
        return dict(image_id=imgId, category_id=catId, aRng=aRng, maxDet=maxDet,
                    dtIds=[d['id'] for d in dt], gtIds=[g['id'] for g in gt], dtMatches=dtm, gtMatches=gtm,
                    dtScores=[d['score'] for d in dt], gtIgnore=gtIg, dtIgnore=dtIg)

    def evaluate(self):
        p = self.params
        p.imgIds = list(np.unique(p.imgIds))
        if p.useCats:
            p.catIds = list(np.unique(p.catIds))
        p.maxDets = sorted(p.maxDets)
        self._prepare()
        cats = p.catIds if p.useCats else [-1]
        self.ious = {(i, c): self.computeIoU(i, c) for i in p.imgIds for c in cats}
        maxDet = p.maxDets[-1]
        self.evalImgs = [self.evaluateImg(i, c, a, maxDet) for c in cats for a in p.areaRng for i in p.imgIds]

    def accumulate(self):
        p = self.params
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(p.catIds) if p.useCats else 1, len(p.areaRng), len(p.maxDets)
        precision = -np.ones((T, R, K, A, M))
        recall = -np.ones((T, K, A, M))
        scores = -np.ones((T, R, K, A, M))
        I0 = len(p.imgIds)
        for k in range(K):
            for a in range(A):
                base = k * A * I0 + a * I0
                E = [e for e in self.evalImgs[base:base + I0] if e is not None]
                if len(E) == 0:
                    continue
                for m, maxDet in enumerate(p.maxDets):
                    dtScores = np.concatenate([e['dtScores'][0:maxDet] for e in E])
                    inds = np.argsort(-dtScores, kind='mergesort')
                    dtScoresSorted = dtScores[inds]
                    dtm = np.concatenate([e['dtMatches'][:, 0:maxDet] for e in E], axis=1)[:, inds]
                    dtIg = np.concatenate([e['dtIgnore'][:, 0:maxDet] for e in E], axis=1)[:, inds]
                    gtIg = np.concatenate([e['gtIgnore'] for e in E])
                    npig = np.count_nonzero(gtIg == 0)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dtIg))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dtIg))
                    tp_sum = np.cumsum(tps, axis=1).astype(dtype=float)
                    fp_sum = np.cumsum(fps, axis=1).astype(dtype=float)
                    for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        q, ss = np.zeros((R,)), np.zeros((R,))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        idx = np.searchsorted(rc, p.recThrs, side='left')
                        for ri, pi in enumerate(idx):
                            if pi >= nd:
                                break
                            q[ri] = pr[pi]
                            ss[ri] = dtScoresSorted[pi]
                        precision[t, :, k, a, m] = q
                        scores[t, :, k, a, m] = ss
        self.eval = dict(params=p, counts=[T, R, K, A, M], precision=precision, recall=recall, scores=scores)

    def _summarize(self, ap=1, iouThr=None, areaRng='all', maxDets=100):
        p = self.params
        aind = [i for i, a in enumerate(p.areaRngLbl) if a == areaRng]
        mind = [i for i, m in enumerate(p.maxDets) if m == maxDets]
        if ap == 1:
            s = self.eval['precision']
            if iouThr is not None:
                s = s[np.where(np.isclose(iouThr, p.iouThrs))[0]]
            s = s[:, :, :, aind, mind]
        else:
            s = self.eval['recall']
            if iouThr is not None:
                s = s[np.where(np.isclose(iouThr, p.iouThrs))[0]]
            s = s[:, :, aind, mind]
        mean_s = -1 if len(s[s > -1]) == 0 else np.mean(s[s > -1])
        title = 'Average Precision' if ap == 1 else 'Average Recall'
        typ = '(AP)' if ap == 1 else '(AR)'
        iou = f'{p.iouThrs[0]:0.2f}:{p.iouThrs[-1]:0.2f}' if iouThr is None else f'{iouThr:0.2f}'
        line = f' {title:<18} {typ} @[ IoU={iou:<9} | area={areaRng:>6s} | maxDets={maxDets:>3d} ] = {mean_s:0.3f}'
        return mean_s, line

    def summarize(self):
        if not self.eval:
            raise Exception('Please run accumulate() first')
        md = self.params.maxDets
        spec = [(1, None, 'all', md[2]), (1, .5, 'all', md[2]), (1, .75, 'all', md[2]),
                (1, None, 'small', md[2]), (1, None, 'medium', md[2]), (1, None, 'large', md[2]),
                (0, None, 'all', md[0]), (0, None, 'all', md[1]), (0, None, 'all', md[2]),
                (0, None, 'small', md[2]), (0, None, 'medium', md[2]), (0, None, 'large', md[2])]
        stats, lines = np.zeros((12,)), []
        for i, (ap, thr, rng, m) in enumerate(spec):
            stats[i], line = self._summarize(ap, thr, rng, m)
            lines.append(line)
        self.stats = stats
        return '\n'.join(lines)


# ------------------------------------------------------------------------------------------
# PASCAL VOC style mAP (mmdet/core/evaluation/mean_ap.py: average_precision:13-60,
# tpfp_default:159-267, get_cls_results:270-294, eval_map:297-420), as `VOCDataset.evaluate`
# calls it (datasets/voc.py:60-76: iou_thr 0.5, legacy +1 box extents, 11-point AP for VOC2007).
# The reference is importable here, so this part is pinned by a golden fixture (g15).
# ------------------------------------------------------------------------------------------
def bbox_overlaps_np(bboxes1, bboxes2, mode='iou', eps=1e-6, use_legacy_coordinate=False):
    """core/evaluation/bbox_overlaps.py:5-65"""
    assert mode in ('iou', 'iof')
    extra = 1. if use_legacy_coordinate else 0.
    bboxes1 = bboxes1.astype(np.float32)
    bboxes2 = bboxes2.astype(np.float32)
    rows, cols = bboxes1.shape[0], bboxes2.shape[0]
    ious = np.zeros((rows, cols), dtype=np.float32)
    if rows * cols == 0:
        return ious
    exchange = False
    if bboxes1.shape[0] > bboxes2.shape[0]:
        bboxes1, bboxes2 = bboxes2, bboxes1
        ious = np.zeros((cols, rows), dtype=np.float32)
        exchange = True
    area1 = (bboxes1[:, 2] - bboxes1[:, 0] + extra) * (bboxes1[:, 3] - bboxes1[:, 1] + extra)
    area2 = (bboxes2[:, 2] - bboxes2[:, 0] + extra) * (bboxes2[:, 3] - bboxes2[:, 1] + extra)
    for i in range(bboxes1.shape[0]):
        x_start = np.maximum(bboxes1[i, 0], bboxes2[:, 0])
        y_start = np.maximum(bboxes1[i, 1], bboxes2[:, 1])
        x_end = np.minimum(bboxes1[i, 2], bboxes2[:, 2])
        y_end = np.minimum(bboxes1[i, 3], bboxes2[:, 3])
        overlap = np.maximum(x_end - x_start + extra, 0) * np.maximum(y_end - y_start + extra, 0)
        union = area1[i] + area2 - overlap if mode == 'iou' else (area1[i] if not exchange else area2)
        union = np.maximum(union, eps)
        ious[i, :] = overlap / union
    return ious.T if exchange else ious


def average_precision(recalls, precisions, mode='area'):
    no_scale = recalls.ndim == 1
    if no_scale:
        recalls, precisions = recalls[np.newaxis, :], precisions[np.newaxis, :]
    assert recalls.shape == precisions.shape and recalls.ndim == 2
    num_scales = recalls.shape[0]
    ap = np.zeros(num_scales, dtype=np.float32)
    if mode == 'area':
        zeros = np.zeros((num_scales, 1), dtype=recalls.dtype)
        ones = np.ones((num_scales, 1), dtype=recalls.dtype)
        mrec = np.hstack((zeros, recalls, ones))
        mpre = np.hstack((zeros, precisions, zeros))
        for i in range(mpre.shape[1] - 1, 0, -1):
            mpre[:, i - 1] = np.maximum(mpre[:, i - 1], mpre[:, i])
        for i in range(num_scales):
            ind = np.where(mrec[i, 1:] != mrec[i, :-1])[0]
            ap[i] = np.sum((mrec[i, ind + 1] - mrec[i, ind]) * mpre[i, ind + 1])
    elif mode == '11points':
        for i in range(num_scales):
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                precs = precisions[i, recalls[i, :] >= thr]
                ap[i] += precs.max() if precs.size > 0 else 0
        ap /= 11
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return ap[0] if no_scale else ap


def tpfp_default(det_bboxes, gt_bboxes, gt_bboxes_ignore=None, iou_thr=0.5, area_ranges=None,
                 use_legacy_coordinate=False):
    extra = 1. if use_legacy_coordinate else 0.
    gt_ignore_inds = np.concatenate((np.zeros(gt_bboxes.shape[0], dtype=bool),
                                     np.ones(gt_bboxes_ignore.shape[0], dtype=bool)))
    gt_bboxes = np.vstack((gt_bboxes, gt_bboxes_ignore))
    num_dets, num_gts = det_bboxes.shape[0], gt_bboxes.shape[0]
    if area_ranges is None:
        area_ranges = [(None, None)]
    tp = np.zeros((len(area_ranges), num_dets), dtype=np.float32)
    fp = np.zeros((len(area_ranges), num_dets), dtype=np.float32)
    if gt_bboxes.shape[0] == 0:
        if area_ranges == [(None, None)]:
            fp[...] = 1
        else:
            det_areas = (det_bboxes[:, 2] - det_bboxes[:, 0] + extra) * (det_bboxes[:, 3] - det_bboxes[:, 1] + extra)
            for i, (lo, hi) in enumerate(area_ranges):
                fp[i, (det_areas >= lo) & (det_areas < hi)] = 1
        return tp, fp
    ious = bbox_overlaps_np(det_bboxes, gt_bboxes, use_legacy_coordinate=use_legacy_coordinate)
    ious_max, ious_argmax = ious.max(axis=1), ious.argmax(axis=1)
    sort_inds = np.argsort(-det_bboxes[:, -1])
    for k, (lo, hi) in enumerate(area_ranges):
        gt_covered = np.zeros(num_gts, dtype=bool)
        if lo is None:
            gt_area_ignore = np.zeros_like(gt_ignore_inds, dtype=bool)
        else:
            gt_areas = (gt_bboxes[:, 2] - gt_bboxes[:, 0] + extra) * (gt_bboxes[:, 3] - gt_bboxes[:, 1] + extra)
            gt_area_ignore = (gt_areas < lo) | (gt_areas >= hi)
        for i in sort_inds:
            if ious_max[i] >= iou_thr:
                m = ious_argmax[i]
                if not (gt_ignore_inds[m] or gt_area_ignore[m]):
                    if not gt_covered[m]:
                        gt_covered[m] = True
                        tp[k, i] = 1
                    else:
                        fp[k, i] = 1
            elif lo is None:
                fp[k, i] = 1
            else:
                b = det_bboxes[i, :4]
                area = (b[2] - b[0] + extra) * (b[3] - b[1] + extra)
                if lo <= area < hi:
                    fp[k, i] = 1
    return tp, fp


def eval_map(det_results, annotations, scale_ranges=None, iou_thr=0.5, dataset=None, logger=None,
             use_legacy_coordinate=False):
    """(mAP, per-class dicts {num_gts, num_dets, recall, precision, ap})"""
    assert len(det_results) == len(annotations)
    extra = 1. if use_legacy_coordinate else 0.
    num_scales = len(scale_ranges) if scale_ranges is not None else 1
    num_classes = len(det_results[0])
    area_ranges = [(r[0] ** 2, r[1] ** 2) for r in scale_ranges] if scale_ranges is not None else None
    eval_results = []
    for c in range(num_classes):
        cls_dets = [img_res[c] for img_res in det_results]
        cls_gts, cls_gts_ignore = [], []
        for ann in annotations:
            cls_gts.append(ann['bboxes'][ann['labels'] == c, :])
            if ann.get('labels_ignore', None) is not None:
                cls_gts_ignore.append(ann['bboxes_ignore'][ann['labels_ignore'] == c, :])
            else:
                cls_gts_ignore.append(np.empty((0, 4), dtype=np.float32))
        tpfp = [tpfp_default(d, g, gi, iou_thr, area_ranges, use_legacy_coordinate)
                for d, g, gi in zip(cls_dets, cls_gts, cls_gts_ignore)]
        tp, fp = tuple(zip(*tpfp))
        num_gts = np.zeros(num_scales, dtype=int)
        for bbox in cls_gts:
            if area_ranges is None:
                num_gts[0] += bbox.shape[0]
            else:
                areas = (bbox[:, 2] - bbox[:, 0] + extra) * (bbox[:, 3] - bbox[:, 1] + extra)
                for k, (lo, hi) in enumerate(area_ranges):
                    num_gts[k] += np.sum((areas >= lo) & (areas < hi))
        cls_dets = np.vstack(cls_dets)
        sort_inds = np.argsort(-cls_dets[:, -1])
        tp = np.cumsum(np.hstack(tp)[:, sort_inds], axis=1)
        fp = np.cumsum(np.hstack(fp)[:, sort_inds], axis=1)
        eps = np.finfo(np.float32).eps
        recalls = tp / np.maximum(num_gts[:, np.newaxis], eps)
        precisions = tp / np.maximum((tp + fp), eps)
        if scale_ranges is None:
            recalls, precisions, n_gt = recalls[0, :], precisions[0, :], num_gts.item()
        else:
            n_gt = num_gts
        ap = average_precision(recalls, precisions, 'area' if dataset != 'voc07' else '11points')
        eval_results.append(dict(num_gts=n_gt, num_dets=cls_dets.shape[0], recall=recalls, precision=precisions, ap=ap))
    if scale_ranges is not None:
        all_ap = np.vstack([r['ap'] for r in eval_results])
        all_gt = np.vstack([r['num_gts'] for r in eval_results])
        mean_ap = [all_ap[all_gt[:, i] > 0, i].mean() if np.any(all_gt[:, i] > 0) else 0.0 for i in range(num_scales)]
    else:
        aps = [r['ap'] for r in eval_results if r['num_gts'] > 0]
        mean_ap = np.array(aps).mean().item() if aps else 0.0
    if logger is not None:
        logger.info(f'mAP@{iou_thr}: {mean_ap}')
    return mean_ap, eval_results



# ------------------------------------------------------------------------------------------
# Proposal recall (mmdet/core/evaluation/recall.py: _recalls:12-44, set_recall_param:47-65, eval_recalls:68-113), the
# 'proposal_fast' metric of CocoDataset.evaluate (datasets/coco.py:311-333,425-434).  The reference is importable here:
# pinned by the golden fixture g22.
# ------------------------------------------------------------------------------------------
def _greedy_gt_ious(ious):
    """the IoU every ground truth ends up with when (gt, proposal) pairs are taken greedily by descending IoU, each
    ground truth and each proposal at most once -- in the order they are taken (rows: ground truths)"""
    ious = np.array(ious, dtype=np.float64)
    g = ious.shape[0]
    out = np.zeros(g)
    if ious.size == 0:
        return out
    for j in range(g):
        # the best remaining proposal of every remaining ground truth, then the best of those (first index on ties)
        best_box = ious.argmax(axis=1)
        best = ious[np.arange(g), best_box]
        gi = int(best.argmax())
        out[j] = best[gi]
        ious[gi, :] = -1
        ious[:, best_box[gi]] = -1
    return out


def eval_recalls(gts, proposals, proposal_nums=None, iou_thrs=0.5, logger=None, use_legacy_coordinate=False):
    """recalls (len(proposal_nums), len(iou_thrs)): the fraction of ground-truth boxes covered at IoU >= thr by the top
    `proposal_num` proposals of their image (descending score when a 5th column carries one), one-to-one greedy matching"""
    assert len(gts) == len(proposals)
    nums = np.array([proposal_nums] if isinstance(proposal_nums, (int, np.integer)) else proposal_nums)
    thrs = np.array([0.5] if iou_thrs is None else
                    ([iou_thrs] if isinstance(iou_thrs, (float, np.floating)) else iou_thrs), dtype=np.float64)
    per_image = []
    for gt, prop in zip(gts, proposals):
        prop = np.asarray(prop)
        if prop.ndim == 2 and prop.shape[1] == 5:
            prop = prop[np.argsort(prop[:, 4])[::-1]]
        keep = min(prop.shape[0], int(nums[-1]))
        if gt is None or len(gt) == 0:
            per_image.append(np.zeros((0, prop.shape[0]), dtype=np.float32))
        else:
            per_image.append(bbox_overlaps_np(np.asarray(gt), prop[:keep, :4], use_legacy_coordinate=use_legacy_coordinate))
    total = sum(m.shape[0] for m in per_image)
    recalls = np.zeros((nums.size, thrs.size))
    for k, n in enumerate(nums):
        covered = np.concatenate([_greedy_gt_ious(m[:, :int(n)]) for m in per_image]) if per_image else np.zeros(0)
        covered = covered.astype(np.float32)            # (the reference collects them in a float32 table)
        for i, t in enumerate(thrs):
            recalls[k, i] = (covered >= t).sum() / float(total) if total else 0.0
    if logger is not None and logger != 'silent' and hasattr(logger, 'info'):
        rows = ['proposals | ' + ' '.join(f'{t:.2f}' for t in thrs)]
        rows += [f'{int(n):9d} | ' + ' '.join(f'{v:.3f}' for v in recalls[k]) for k, n in enumerate(nums)]
        logger.info('\n' + '\n'.join(rows))
    return recalls
