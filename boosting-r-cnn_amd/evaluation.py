"""Result sink: COCO bbox mAP without pycocotools.

The reference's `CocoDataset.evaluate` (mmdet/datasets/coco.py:362-560) hands detections to
pycocotools' `COCOeval` (third party, absent here; the published algorithm of
pycocotools/cocoeval.py, bbox branch, is restated): per (image, category) IoU matrix with
crowd handling, greedy matching by descending score at 10 IoU thresholds x 4 area ranges,
precision envelopes sampled at 101 recall points, the 12 summary statistics.
PARITY UNPINNED against pycocotools itself (not installed); pinned by known-answer cases
(tests/test_eval_cpu.py) derived by hand from the algorithm.
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
        gts = [a for i in p.imgIds for a in self.cocoGt.imgToAnns.get(i, []) if a['category_id'] in cat_set]
        dts = [a for i in p.imgIds for a in self.cocoDt.imgToAnns.get(i, []) if a['category_id'] in cat_set]
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        for g in gts:
            g['ignore'] = 1 if ('iscrowd' in g and g['iscrowd']) else 0
            self._gts[g['image_id'], g['category_id']].append(g)
        for d in dts:
            if d['image_id'] in img_set:
                self._dts[d['image_id'], d['category_id']].append(d)

    def computeIoU(self, imgId, catId):
        gt, dt = self._gts[imgId, catId], self._dts[imgId, catId]
        if len(gt) == 0 and len(dt) == 0:
            return []
        inds = np.argsort([-d['score'] for d in dt], kind='mergesort')
        dt = [dt[i] for i in inds][:self.params.maxDets[-1]]
        return bbox_iou_xywh([d['bbox'] for d in dt], [g['bbox'] for g in gt], [int(g['iscrowd']) for g in gt])

    def evaluateImg(self, imgId, catId, aRng, maxDet):
        p = self.params
        gt, dt = self._gts[imgId, catId], self._dts[imgId, catId]
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
        p.catIds = list(np.unique(p.catIds))
        p.maxDets = sorted(p.maxDets)
        self._prepare()
        self.ious = {(i, c): self.computeIoU(i, c) for i in p.imgIds for c in p.catIds}
        maxDet = p.maxDets[-1]
        self.evalImgs = [self.evaluateImg(i, c, a, maxDet) for c in p.catIds for a in p.areaRng for i in p.imgIds]

    def accumulate(self):
        p = self.params
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(p.catIds), len(p.areaRng), len(p.maxDets)
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
