"""not-gpu: COCO bbox mAP (brcnn.evaluation, restating pycocotools' COCOeval which is not
installed: parity unpinned against the package itself) on known-answer cases worked out by
hand from the algorithm: 101-point interpolated precision, 10 IoU thresholds, crowd / area
range / maxDets handling."""
import numpy as np
import pytest

import brcnn  # noqa: F401
from brcnn.datasets import COCO
from brcnn.evaluation import COCOeval, bbox_iou_xywh


def _gt(images, anns, cats=(1,)):
    c = COCO()
    c.dataset = dict(images=[dict(id=i, width=500, height=500, file_name=f'{i}.npy') for i in images],
                     categories=[dict(id=k, name=f'c{k}') for k in cats],
                     annotations=[dict(id=j + 1, image_id=a[0], category_id=a[1], bbox=list(a[2]),
                                       area=a[2][2] * a[2][3], iscrowd=a[3] if len(a) > 3 else 0)
                                  for j, a in enumerate(anns)])
    c.createIndex()
    return c


def _run(gt, dets, max_dets=(1, 10, 100)):
    dt = gt.loadRes([dict(image_id=d[0], category_id=d[1], bbox=list(d[2]), score=d[3]) for d in dets])
    ev = COCOeval(gt, dt, 'bbox')
    ev.params.maxDets = list(max_dets)
    ev.evaluate()
    ev.accumulate()
    ev.summarize()
    return ev


def test_iou_xywh_and_crowd():
    iou = bbox_iou_xywh([[0, 0, 10, 10]], [[0, 0, 10, 10], [5, 0, 10, 10], [20, 20, 5, 5], [0, 0, 100, 100]],
                        [0, 0, 0, 1])
    assert iou[0, 0] == 1.0 and iou[0, 1] == pytest.approx(50 / 150) and iou[0, 2] == 0.0
    assert iou[0, 3] == 1.0        # crowd: intersection over the DETECTION's area


def test_perfect_detections():
    gt = _gt([1, 2], [(1, 1, (10, 10, 50, 50)), (2, 1, (20, 20, 100, 100)), (2, 1, (200, 200, 20, 20))])
    ev = _run(gt, [(1, 1, (10, 10, 50, 50), .9), (2, 1, (20, 20, 100, 100), .8), (2, 1, (200, 200, 20, 20), .7)])
    assert ev.stats[0] == pytest.approx(1.0) and ev.stats[1] == pytest.approx(1.0) and ev.stats[2] == pytest.approx(1.0)
    assert ev.stats[3] == pytest.approx(1.0)      # small: the 20x20 box
    assert ev.stats[4] == pytest.approx(1.0)      # medium: 50x50 (area 2500)
    assert ev.stats[5] == pytest.approx(1.0)      # large: 100x100
    assert ev.stats[8] == pytest.approx(1.0)
    assert ev.stats[6] == pytest.approx((1 + 0.5) / 2 if False else 2 / 3)   # AR@1: 1 of 1 + 1 of 2 gts -> 2/3


def test_false_positive_ranked_first_halves_precision():
    gt = _gt([1], [(1, 1, (10, 10, 50, 50))])
    ev = _run(gt, [(1, 1, (300, 300, 50, 50), .95), (1, 1, (10, 10, 50, 50), .9)])
    assert ev.stats[0] == pytest.approx(0.5)         # precision 1/2 at every recall level
    ev = _run(gt, [(1, 1, (300, 300, 50, 50), .5), (1, 1, (10, 10, 50, 50), .9)])
    assert ev.stats[0] == pytest.approx(1.0)         # FP after the TP does not touch the envelope


def test_half_recall_gives_51_of_101_points():
    gt = _gt([1], [(1, 1, (10, 10, 50, 50)), (1, 1, (200, 200, 50, 50))])
    ev = _run(gt, [(1, 1, (10, 10, 50, 50), .9)])
    assert ev.stats[0] == pytest.approx(51 / 101)    # recall thresholds 0, .01, ..., .50 reached
    assert ev.stats[8] == pytest.approx(0.5)


def test_iou_thresholds():
    # detection shifted so that IoU = 40*50 / (2*2500 - 2000) = 2/3: TP for .50 .55 .60 .65 only
    gt = _gt([1], [(1, 1, (10, 10, 50, 50))])
    ev = _run(gt, [(1, 1, (20, 10, 50, 50), .9)])
    assert ev.stats[0] == pytest.approx(4 / 10)
    assert ev.stats[1] == pytest.approx(1.0) and ev.stats[2] == 0.0


def test_crowd_and_area_ranges_and_maxdets():
    # a detection on a crowd region is ignored (neither TP nor FP)
    gt = _gt([1], [(1, 1, (10, 10, 50, 50)), (1, 1, (200, 200, 100, 100), 1)])
    ev = _run(gt, [(1, 1, (10, 10, 50, 50), .8), (1, 1, (210, 210, 30, 30), .9), (1, 1, (220, 220, 30, 30), .85)])
    assert ev.stats[0] == pytest.approx(1.0)
    # small gt only counts in the 'small' range; ranges without gt report -1
    gt = _gt([1], [(1, 1, (10, 10, 20, 20))])
    ev = _run(gt, [(1, 1, (10, 10, 20, 20), .9)])
    assert ev.stats[3] == pytest.approx(1.0) and ev.stats[4] == -1 and ev.stats[5] == -1
    # maxDets: with 1 detection per image allowed only the top-scored (a miss) is kept
    gt = _gt([1], [(1, 1, (10, 10, 50, 50))])
    ev = _run(gt, [(1, 1, (300, 300, 50, 50), .95), (1, 1, (10, 10, 50, 50), .9)])
    assert ev.stats[6] == 0.0 and ev.stats[7] == pytest.approx(1.0)
    # two categories are averaged
    gt = _gt([1], [(1, 1, (10, 10, 50, 50)), (1, 2, (100, 100, 50, 50))], cats=(1, 2))
    ev = _run(gt, [(1, 1, (10, 10, 50, 50), .9)])
    assert ev.stats[0] == pytest.approx(0.5)


def test_dataset_evaluate_roundtrip(tmp_path):
    from brcnn.datasets import CocoDataset
    from tests.golden.synth import synthetic_coco
    ann_file, prefix = synthetic_coco(str(tmp_path))
    classes = ('echinus', 'starfish', 'holothurian', 'scallop')
    ds = CocoDataset(ann_file=ann_file, pipeline=[], classes=classes, img_prefix=prefix, test_mode=True)
    results = []
    for i in range(len(ds)):        # every annotation COCOeval counts as ground truth, detected exactly
        anns = [a for a in ds.coco.imgToAnns.get(ds.img_ids[i], []) if a['category_id'] in ds.cat_ids]
        per_cls = []
        for c in range(4):
            b = [[a['bbox'][0], a['bbox'][1], a['bbox'][0] + a['bbox'][2], a['bbox'][1] + a['bbox'][3], 0.9]
                 for a in anns if ds.cat2label[a['category_id']] == c and not a.get('iscrowd', 0)]
            per_cls.append(np.array(b, dtype=np.float32).reshape(-1, 5))
        results.append(per_cls)
    out = ds.evaluate(results, metric='bbox', classwise=True, jsonfile_prefix=str(tmp_path / 'res'))
    assert out['bbox_mAP'] == pytest.approx(1.0) and out['bbox_mAP_50'] == pytest.approx(1.0)
    assert set(out['bbox_classwise']) == set(classes)
    assert len(out['bbox_mAP_copypaste'].split()) == 6
    assert (tmp_path / 'res.bbox.json').exists()
    empty = [[np.zeros((0, 5), np.float32)] * 4 for _ in range(len(ds))]
    assert ds.evaluate(empty) == {}


def test_area_range_edges_are_inclusive_on_both_sides():
    """COCOeval ignores a ground truth for an area range when `area < lo or area > hi`: a box of exactly 32^2 counts
    as small AND medium, one of exactly 96^2 as medium AND large (hand-derived from the published rule)"""
    gt = _gt([1], [(1, 1, (10, 10, 32, 32))])
    ev = _run(gt, [(1, 1, (10, 10, 32, 32), .9)])
    assert ev.stats[3] == pytest.approx(1.0) and ev.stats[4] == pytest.approx(1.0) and ev.stats[5] == -1
    gt = _gt([1], [(1, 1, (10, 10, 96, 96))])
    ev = _run(gt, [(1, 1, (10, 10, 96, 96), .9)])
    assert ev.stats[3] == -1 and ev.stats[4] == pytest.approx(1.0) and ev.stats[5] == pytest.approx(1.0)
    # one pixel more on either side leaves the range
    gt = _gt([1], [(1, 1, (10, 10, 32, 32.03125))])        # area 1025 > 32^2
    ev = _run(gt, [(1, 1, (10, 10, 32, 32.03125), .9)])
    assert ev.stats[3] == -1 and ev.stats[4] == pytest.approx(1.0)
    # an unmatched detection outside the range is ignored there, inside it is a false positive:
    # small gt found + one large false positive ranked first -> AP_small stays 1, AP_all drops to 1/2
    gt = _gt([1], [(1, 1, (10, 10, 20, 20))])
    ev = _run(gt, [(1, 1, (200, 200, 150, 150), .95), (1, 1, (10, 10, 20, 20), .9)])
    assert ev.stats[3] == pytest.approx(1.0) and ev.stats[0] == pytest.approx(0.5)


def test_maxdets_truncation_and_the_precision_envelope():
    """detections ranked FP(.9), TP(.8), TP(.7) on two ground truths: precision 1/2 at recall 1/2 and 2/3 at recall 1;
    the envelope lifts every recall level to 2/3.  With maxDets = 1 only the false positive survives, with 2 one hit."""
    gt = _gt([1], [(1, 1, (10, 10, 50, 50)), (1, 1, (200, 200, 50, 50))])
    dets = [(1, 1, (400, 400, 40, 40), .9), (1, 1, (10, 10, 50, 50), .8), (1, 1, (200, 200, 50, 50), .7)]
    ev = _run(gt, dets, max_dets=(1, 2, 100))
    assert ev.stats[0] == pytest.approx(2 / 3)
    assert ev.stats[6] == 0.0 and ev.stats[7] == pytest.approx(0.5) and ev.stats[8] == pytest.approx(1.0)
    # order of the input list does not matter, only the scores
    ev2 = _run(gt, dets[::-1], max_dets=(1, 2, 100))
    assert ev2.stats[0] == pytest.approx(2 / 3) and ev2.stats[6] == 0.0


def test_crowd_absorbs_any_number_of_detections_and_ignore_follows_iscrowd():
    """a crowd region matches every detection that overlaps it enough (intersection over the detection's area), all of
    them ignored; a plain ground truth is matched once.  pycocotools overwrites an annotation's `ignore` field with
    `iscrowd` for bbox evaluation: `ignore: 1` alone does not hide a ground truth."""
    gt = _gt([1], [(1, 1, (10, 10, 50, 50)), (1, 1, (200, 200, 200, 200), 1)])
    ev = _run(gt, [(1, 1, (210, 210, 30, 30), .95), (1, 1, (250, 250, 30, 30), .9), (1, 1, (300, 300, 30, 30), .85),
                   (1, 1, (10, 10, 50, 50), .5)])
    assert ev.stats[0] == pytest.approx(1.0) and ev.stats[8] == pytest.approx(1.0)
    # a second detection of the plain ground truth is a false positive (matched once): ranked below, it does not matter;
    # ranked above the true one with a worse IoU it halves the precision at the strict thresholds only
    ev = _run(gt, [(1, 1, (10, 10, 50, 50), .9), (1, 1, (12, 10, 50, 50), .8)])
    assert ev.stats[0] == pytest.approx(1.0)
    # only crowd ground truth: nothing to find -> -1
    gt = _gt([1], [(1, 1, (200, 200, 200, 200), 1)])
    ev = _run(gt, [(1, 1, (210, 210, 30, 30), .95)])
    assert ev.stats[0] == -1
    # `ignore` without iscrowd
    c = _gt([1], [(1, 1, (10, 10, 50, 50))])
    c.dataset['annotations'][0]['ignore'] = 1
    c.createIndex()
    ev = _run(c, [(1, 1, (300, 300, 50, 50), .9)])
    assert ev.stats[0] == 0.0 and ev.stats[8] == 0.0          # the ground truth still counts (and is missed)


# ---- proposal recall: eval_recalls is pinned by the reference itself (golden g22), the 'proposal' metric (COCOeval with
# useCats = 0) by hand-derived cases -------------------------------------------------------------------------------------
def test_eval_recalls_matches_the_reference_golden():
    """brcnn.evaluation.eval_recalls against mmdet/core/evaluation/recall.py run on the same boxes (tests/golden/
    make_golden.py g22_recalls): scored / unscored proposals, several proposal numbers and thresholds, legacy extents"""
    import os
    from brcnn.evaluation import eval_recalls
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g22_recalls.npz'))
    gts = [g[f'gt{i}'] for i in range(5)]
    props = [g[f'prop{i}'] for i in range(5)]
    assert np.array_equal(eval_recalls(gts, props, (10, 30, 100), np.linspace(.5, .95, 10)), g['rec_scored'])
    assert np.array_equal(eval_recalls(gts, props, 50, 0.5), g['rec_single_thr'])
    assert np.array_equal(eval_recalls(gts, [p[:, :4] for p in props], (10, 100), [0.5, 0.75]), g['rec_unscored'])
    assert np.array_equal(eval_recalls(gts, props, (100,), [0.5, 0.7], use_legacy_coordinate=True), g['rec_legacy'])
    assert 0 < g['rec_scored'][0, 0] < g['rec_scored'][2, 0] == 1.0        # (the fixture discriminates)


def test_eval_recalls_hand_derived_ragged_and_empty_images():
    """what the reference cannot be asked under numpy >= 1.24 (it stacks ragged IoU matrices): images with different
    box counts, an image without ground truth, one without proposals; one-to-one greedy matching"""
    from brcnn.evaluation import eval_recalls
    gts = [np.array([[0, 0, 10, 10], [20, 20, 30, 30]], np.float32),     # two boxes
           np.zeros((0, 4), np.float32),                                   # none
           np.array([[0, 0, 10, 10]], np.float32)]                         # one box, no proposal
    props = [np.array([[0, 0, 10, 10, .9],       # IoU 1 with gt 0
                       [0, 0, 10, 5, .8],        # IoU .5 with gt 0: gt 0 is taken, this one is left over
                       [20, 20, 30, 25, .7]],    # IoU .5 with gt 1
                      np.float32),
             np.array([[5, 5, 9, 9, .5]], np.float32),
             np.zeros((0, 5), np.float32)]
    rec = eval_recalls(gts, props, (1, 3), [0.5, 0.75])
    # top-1: only the exact box -> 1 of 3 ground truths at both thresholds; top-3: gt 0 at IoU 1, gt 1 at IoU .5
    assert np.allclose(rec, [[1 / 3, 1 / 3], [2 / 3, 1 / 3]])
    # one proposal cannot serve two ground truths
    one = eval_recalls([np.array([[0, 0, 10, 10], [0, 0, 10, 9]], np.float32)], [np.array([[0, 0, 10, 10]], np.float32)], (5,), [0.5])
    assert np.allclose(one, [[0.5]])


def test_proposal_metric_ignores_categories():
    """'proposal' = COCOeval with useCats = 0: a detection labelled with another category still recalls the box; with
    one category the class-agnostic AR equals the per-class one"""
    gt = _gt([1], [(1, 1, (10, 10, 50, 50)), (1, 2, (200, 200, 60, 60))], cats=(1, 2))
    dets = [(1, 2, (10, 10, 50, 50), .9), (1, 1, (200, 200, 60, 60), .8)]        # labels swapped
    dt = gt.loadRes([dict(image_id=d[0], category_id=d[1], bbox=list(d[2]), score=d[3]) for d in dets])
    for use_cats, want in ((1, 0.0), (0, 1.0)):
        ev = COCOeval(gt, dt, 'bbox')
        ev.params.useCats = use_cats
        ev.evaluate(); ev.accumulate(); ev.summarize()
        assert ev.stats[8] == pytest.approx(want), (use_cats, ev.stats)
    gt1 = _gt([1, 2], [(1, 1, (10, 10, 50, 50)), (2, 1, (20, 20, 100, 100)), (2, 1, (200, 200, 20, 20))])
    d1 = [(1, 1, (10, 10, 50, 50), .9), (2, 1, (20, 20, 100, 100), .8)]
    a = _run(gt1, d1)
    dt1 = gt1.loadRes([dict(image_id=d[0], category_id=d[1], bbox=list(d[2]), score=d[3]) for d in d1])
    b = COCOeval(gt1, dt1, 'bbox')
    b.params.useCats = 0
    b.evaluate(); b.accumulate(); b.summarize()
    assert np.allclose(a.stats, b.stats)


def test_dataset_proposal_metrics(tmp_path):
    """CocoDataset.evaluate(metric='proposal_fast' / 'proposal') (datasets/coco.py:311-333,425-434,488-506) on per-image
    proposals: exact boxes -> AR 1; 'proposal' also accepts per-class detection lists; the json sink"""
    from brcnn.datasets import CocoDataset
    from tests.golden.synth import synthetic_coco
    ann_file, prefix = synthetic_coco(str(tmp_path))
    classes = ('echinus', 'starfish', 'holothurian', 'scallop')
    ds = CocoDataset(ann_file=ann_file, pipeline=[], classes=classes, img_prefix=prefix, test_mode=True)
    props, dets = [], []
    for i in range(len(ds)):
        anns = [a for a in ds.coco.imgToAnns.get(ds.img_ids[i], []) if not a.get('iscrowd', 0)]
        b = np.array([[a['bbox'][0], a['bbox'][1], a['bbox'][0] + a['bbox'][2], a['bbox'][1] + a['bbox'][3], 0.9 - 0.01 * j]
                      for j, a in enumerate(anns)], dtype=np.float32).reshape(-1, 5)
        props.append(b)
        dets.append([b if c == 0 else np.zeros((0, 5), np.float32) for c in range(4)])      # all labelled class 0
    fast = ds.evaluate(props, metric='proposal_fast', proposal_nums=(1, 100))
    assert fast['AR@100'] == pytest.approx(1.0) and 0 < fast['AR@1'] < 1
    half = [p[: max(1, len(p) // 2)] for p in props]
    assert 0 < ds.evaluate(half, metric='proposal_fast', proposal_nums=(100,))['AR@100'] < 1
    out = ds.evaluate(props, metric='proposal', jsonfile_prefix=str(tmp_path / 'p'))
    assert out['AR@1000'] == pytest.approx(1.0) and set(out) == {'AR@100', 'AR@300', 'AR@1000', 'AR_s@1000', 'AR_m@1000', 'AR_l@1000'}
    assert (tmp_path / 'p.proposal.json').exists()
    assert ds.evaluate(dets, metric='proposal')['AR@1000'] == pytest.approx(1.0)       # categories ignored
    assert ds.evaluate(dets, metric=['bbox', 'proposal'], metric_items=None).keys() >= {'bbox_mAP', 'AR@100'}
    with pytest.raises(KeyError):
        ds.evaluate(props, metric='segm')
    with pytest.raises(KeyError):
        ds.evaluate(props, metric='bbox')
