"""not-gpu: the data side: transforms, COCO dataset, samplers, collate against
golden fixtures produced by the imported reference (tests/golden/make_golden.py data), the
numpy image arithmetic against the C oracle, and invariants of the OpenCV-style resize."""
import json
import os

import numpy as np
import pytest
import torch

import brcnn  # noqa: F401
from brcnn import datasets as D
from brcnn import pipelines as P
from oracle import orc
from tests.test_host_cpu import load

MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def test_rescale_size_and_resize_invariants():
    assert P.rescale_size((640, 480), (1333, 800)) == (1067, 800)
    assert P.rescale_size((1920, 1080), (1333, 800)) == (1333, 750)
    assert P.rescale_size((200, 300), (1333, 800)) == (800, 1200)
    assert P.rescale_size((100, 50), 1.5) == (150, 75)
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(P.imresize_u8(img, (53, 37)), img)                      # identity
    const = np.full((20, 30, 3), 137, np.uint8)
    assert (P.imresize_u8(const, (77, 41)) == 137).all()                          # constants survive
    assert (P.imresize_u8(const, (11, 7)) == 137).all()
    # within one grey level of exact (float64) bilinear interpolation at the same sample positions
    for (w, h) in [(80, 60), (25, 19), (53, 80), (120, 37)]:
        got = P.imresize_u8(img, (w, h)).astype(np.float64)
        fx = np.clip((np.arange(w) + 0.5) * (53 / w) - 0.5, 0, 52)
        fy = np.clip((np.arange(h) + 0.5) * (37 / h) - 0.5, 0, 36)
        x0, y0 = np.floor(fx).astype(int), np.floor(fy).astype(int)
        x1, y1 = np.minimum(x0 + 1, 52), np.minimum(y0 + 1, 36)
        ax, ay = (fx - x0)[None, :, None], (fy - y0)[:, None, None]
        f = img.astype(np.float64)
        ref = (f[y0][:, x0] * (1 - ax) + f[y0][:, x1] * ax) * (1 - ay) + (f[y1][:, x0] * (1 - ax) + f[y1][:, x1] * ax) * ay
        assert np.abs(got - ref).max() <= 1.0, (w, h, np.abs(got - ref).max())
    # exact 2x decimation = rounded mean of the 2x2 block
    img2 = rng.randint(0, 256, (40, 60, 3), dtype=np.uint8)
    got = P.imresize_u8(img2, (30, 20)).astype(np.int32)
    blk = img2.astype(np.int32).reshape(20, 2, 30, 2, 3).sum((1, 3))
    assert np.abs(got - (blk + 2) // 4).max() <= 1


def test_numpy_chain_equals_c_oracle():
    """Resize -> flip -> Normalize -> Pad: package numpy code == oracle C, bit for bit"""
    rng = np.random.RandomState(1)
    for (h, w, scale) in [(120, 160, (333, 200)), (270, 480, (333, 200)), (75, 50, (333, 200)), (100, 37, (64, 48))]:
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        nw, nh = P.rescale_size((w, h), scale)
        for flip in (None, 'horizontal', 'vertical', 'diagonal'):
            for to_rgb in (True, False):
                r = P.imresize_u8(img, (nw, nh))
                if flip:
                    r = P.imflip(r, flip)
                n = P.impad_to_multiple(P.imnormalize(r, np.array(MEAN, np.float32), np.array(STD, np.float32), to_rgb), 32)
                ref = torch.from_numpy(np.ascontiguousarray(n.transpose(2, 0, 1)))
                got = orc.preprocess_u8(img, nw, nh, n.shape[0], n.shape[1], flip, MEAN, STD, to_rgb)
                assert torch.equal(ref, got), (h, w, flip, to_rgb)


def _cfg(text):
    """pipeline config from its JSON form: scales are tuples in the python configs"""
    def fix(c):
        if isinstance(c, dict):
            c = {k: fix(v) for k, v in c.items()}
            sc = c.get('img_scale')
            if isinstance(sc, list):
                c['img_scale'] = tuple(sc) if isinstance(sc[0], int) else [tuple(x) for x in sc]
            return c
        if isinstance(c, list):
            return [fix(v) for v in c]
        return c
    return fix(json.loads(text))


def _fresh(g):
    img = g['img']
    return dict(img=img.copy(), img_shape=img.shape, ori_shape=img.shape, img_fields=['img'], filename='x.npy',
                ori_filename='x.npy', gt_bboxes=g['boxes'].copy(), gt_labels=g['labels'].copy(),
                bbox_fields=['gt_bboxes'])


def test_transforms_match_reference_golden():
    g = load('g12_pipeline')
    pipe = P.Compose(_cfg(str(g['train_cfg'])))
    flips = set()
    for s in range(6):
        np.random.seed(40 + s)
        out = pipe(_fresh(g))
        meta = json.loads(str(g[f'tr{s}_meta']))
        m = out['img_metas'].data
        assert list(m['img_shape']) == meta['img_shape'] and list(m['pad_shape']) == meta['pad_shape']
        assert list(m['ori_shape']) == meta['ori_shape']
        assert [float(v) for v in m['scale_factor']] == meta['scale_factor']
        assert bool(m['flip']) == meta['flip'] and m['flip_direction'] == meta['flip_direction']
        flips.add(m['flip_direction'])
        t = out['img'].data
        assert list(t.shape) == list(g[f'tr{s}_img_shape']) and t.dtype == torch.float32
        assert torch.equal(t[:, :24, :24], torch.from_numpy(g[f'tr{s}_img_crop']))
        assert torch.equal(t[:, -40:, -40:], torch.from_numpy(g[f'tr{s}_img_tail']))
        assert np.allclose(t.double().sum((1, 2)).numpy(), g[f'tr{s}_img_sum'], rtol=0, atol=1e-6)
        assert np.array_equal(out['gt_bboxes'].data.numpy(), g[f'tr{s}_boxes'])
        assert np.array_equal(out['gt_labels'].data.numpy(), g[f'tr{s}_labels'])
        assert out['img'].stack and out['img_metas'].cpu_only and not out['gt_bboxes'].stack
    assert len(flips) >= 2          # the seeds cover flipped and unflipped draws
    out = P.Compose(_cfg(str(g['test_cfg'])))(_fresh(g))
    assert isinstance(out['img'], list) and len(out['img']) == 1
    meta = json.loads(str(g['te_meta']))
    m = out['img_metas'][0].data
    assert list(m['img_shape']) == meta['img_shape'] and list(m['pad_shape']) == meta['pad_shape']
    assert [float(v) for v in m['scale_factor']] == meta['scale_factor'] and m['flip'] is False
    assert list(out['img'][0].shape) == list(g['te_img_shape'])
    assert torch.equal(out['img'][0][:, :24, :24], torch.from_numpy(g['te_img_crop']))


def test_reference_dataset_configs_build():
    """the shipped train / test pipelines build from the config unchanged"""
    from brcnn import Config
    cfg = Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                       'configs/boosting_rcnn/boosting_rcnn_r50_pafpn_1x_utdac.py'))
    tr = P.Compose(cfg.data.train.pipeline)
    te = P.Compose(cfg.data.test.pipeline)
    assert [type(t).__name__ for t in tr.transforms] == ['LoadImageFromFile', 'LoadAnnotations', 'Resize',
                                                         'RandomFlip', 'Normalize', 'Pad', 'DefaultFormatBundle',
                                                         'Collect']
    assert type(te.transforms[1]).__name__ == 'MultiScaleFlipAug'
    fused = P.fuse_device_pipeline(cfg.data.train.pipeline)
    assert [c['type'] for c in fused] == ['LoadImageFromFile', 'LoadAnnotations', 'FusedResizeNormalizePad',
                                          'DeviceFormatBundle', 'Collect']
    assert fused[2]['img_scale'] == (1333, 800) and fused[2]['size_divisor'] == 32 and fused[2]['flip_ratio'] == 0.5
    fte = P.fuse_device_pipeline(cfg.data.test.pipeline)
    assert [c['type'] for c in fte[1]['transforms']] == ['FusedResizeNormalizePad', 'ImageToTensor', 'Collect']


def test_samplers_match_reference_golden():
    g = load('g13_samplers')

    class DS:
        flag = g['flag']

        def __len__(self):
            return len(self.flag)
    np.random.seed(11)
    assert np.array_equal(np.array(list(D.GroupSampler(DS(), 2))), g['group_spg2'])
    np.random.seed(12)
    assert np.array_equal(np.array(list(D.GroupSampler(DS(), 3))), g['group_spg3'])
    for world in (2, 4):
        seen = []
        for rank in range(world):
            s = D.DistributedGroupSampler(DS(), 2, world, rank, seed=7)
            s.set_epoch(3)
            idx = np.array(list(s))
            assert np.array_equal(idx, g[f'dgroup_w{world}_r{rank}'])
            # every batch of 2 is from one aspect-ratio group
            assert (DS.flag[idx[0::2]] == DS.flag[idx[1::2]]).all()
            t = np.array(list(D.DistributedSampler(DS(), world, rank)))
            assert np.array_equal(t, g[f'dtest_w{world}_r{rank}'])
            seen.append(t)
        assert set(np.concatenate(seen).tolist()) == set(range(len(DS.flag)))


def _synthetic(tmp_path):
    from tests.golden.synth import synthetic_coco
    return synthetic_coco(str(tmp_path))


def test_coco_dataset_matches_reference_golden(tmp_path):
    g = load('g14_coco_dataset')
    ann_file, prefix = _synthetic(tmp_path)
    classes = ('echinus', 'starfish', 'holothurian', 'scallop')
    for mode in ('train', 'test'):
        ds = D.CocoDataset(ann_file=ann_file, pipeline=[], classes=classes, img_prefix=prefix,
                           test_mode=(mode == 'test'))
        assert len(ds) == int(g[f'{mode}_len']) and ds.img_ids == g[f'{mode}_img_ids'].tolist()
        if mode == 'train':
            assert np.array_equal(ds.flag, g['train_flag'])
        for i in range(len(ds)):
            a = ds.get_ann_info(i)
            assert np.array_equal(a['bboxes'], g[f'{mode}_{i}_bboxes']) and a['bboxes'].dtype == np.float32
            assert np.array_equal(a['labels'], g[f'{mode}_{i}_labels']) and a['labels'].dtype == np.int64
            assert np.array_equal(a['bboxes_ignore'], g[f'{mode}_{i}_ignore'])
    results = [[g[f'res_{i}_{c}'] for c in range(4)] for i in range(len(ds))]
    assert ds._det2json(results) == json.loads(str(g['det_json']))


def test_collate_and_loader(tmp_path):
    ann_file, prefix = _synthetic(tmp_path)
    classes = ('echinus', 'starfish', 'holothurian', 'scallop')
    train = [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
             dict(type='Resize', img_scale=(160, 96), keep_ratio=True), dict(type='RandomFlip', flip_ratio=0.5),
             dict(type='Normalize', mean=MEAN, std=STD, to_rgb=True), dict(type='Pad', size_divisor=32),
             dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])]
    ds = D.build_dataset(dict(type='CocoDataset', ann_file=ann_file, img_prefix=prefix, classes=classes,
                              pipeline=train))
    np.random.seed(0)
    loader = D.build_dataloader(ds, 2, 0, dist=False, seed=0)
    n = 0
    for data in loader:
        img = data['img']
        assert isinstance(img, torch.Tensor) and img.dim() == 4 and img.shape[0] == 2
        assert img.shape[2] % 32 == 0 and img.shape[3] % 32 == 0
        assert isinstance(data['img_metas'], list) and isinstance(data['img_metas'][0], dict)
        for b in range(2):
            ph, pw = data['img_metas'][b]['pad_shape'][:2]
            assert (img[b, :, ph:, :] == 0).all() and (img[b, :, :, pw:] == 0).all()   # collate padding
            assert data['gt_bboxes'][b].shape[1] == 4 and len(data['gt_labels'][b]) == len(data['gt_bboxes'][b])
        n += 1
    assert n == len(loader) and n >= 2
    test = [dict(type='LoadImageFromFile'),
            dict(type='MultiScaleFlipAug', img_scale=(160, 96), flip=False,
                 transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                             dict(type='Normalize', mean=MEAN, std=STD, to_rgb=True), dict(type='Pad', size_divisor=32),
                             dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])])]
    dt = D.build_dataset(dict(type='CocoDataset', ann_file=ann_file, img_prefix=prefix, classes=classes,
                              pipeline=test, test_mode=True))
    assert len(dt) == 7
    data = next(iter(D.build_dataloader(dt, 1, 0, dist=False, shuffle=False)))
    assert isinstance(data['img'], list) and data['img'][0].shape[0] == 1
    assert isinstance(data['img_metas'], list) and isinstance(data['img_metas'][0], list)
    assert data['img_metas'][0][0]['ori_filename'] == '000.npy'


def test_loader_keeps_device_transforms_in_the_main_process_without_touching_the_dataset(tmp_path):
    """a pipeline with a device transform: the workers stop in front of it (HostPartView), the main process applies the
    rest; the dataset object keeps its whole pipeline (a second loader and `dataset[i]` still work), also under a
    RepeatDataset wrapper (datasets.build_dataloader)"""
    import os as _os
    PIPELINES = P.PIPELINES

    if 'MarkMainProcess' not in PIPELINES.module_dict:
        @PIPELINES.register_module()
        class MarkMainProcess:
            runs_on_device = True      # stands for FusedResizeNormalizePad: must never run in a forked worker

            def __call__(self, results):
                results['img_info']['ran_in_pid'] = _os.getpid()
                return results

            def __repr__(self):
                return 'MarkMainProcess()'

    ann_file, prefix = _synthetic(tmp_path)
    classes = ('echinus', 'starfish', 'holothurian', 'scallop')
    train = [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True), dict(type='MarkMainProcess'),
             dict(type='Resize', img_scale=(160, 96), keep_ratio=True), dict(type='RandomFlip', flip_ratio=0.0),
             dict(type='Normalize', mean=MEAN, std=STD, to_rgb=True), dict(type='Pad', size_divisor=32),
             dict(type='DefaultFormatBundle'),
             dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'], meta_keys=('img_shape', 'pad_shape'))]
    for wrap in (False, True):
        ds = D.build_dataset(dict(type='CocoDataset', ann_file=ann_file, img_prefix=prefix, classes=classes, pipeline=train))
        n_tf = len(ds.pipeline.transforms)
        whole = D.RepeatDataset(ds, 2) if wrap else ds
        for rep in range(2):                                   # a second loader over the same object
            loader = D.build_dataloader(whole, 2, 2, dist=False, seed=0)
            assert isinstance(loader, D.MainProcessTail) and loader.dataset is whole
            data = next(iter(loader))
            assert data['img'].dim() == 4 and data['img'].shape[0] == 2
            assert len(ds.pipeline.transforms) == n_tf
        sample = whole[0]                                      # the dataset itself still runs the whole pipeline
        assert 'img' in sample and 'img_metas' in sample


# --------------------------------------------------------------------------- VOC recipe (g15)
def test_voc_dataset_and_eval_map_match_reference_golden(tmp_path):
    from brcnn.evaluation import eval_map
    from tests.golden.synth import synthetic_voc
    g = load('g15_voc')
    lst, prefix = synthetic_voc(str(tmp_path))
    ds = D.VOCDataset(ann_file=lst, img_prefix=prefix, pipeline=[], test_mode=True)
    assert len(ds) == int(g['voc_len']) and ds.year == 2007
    for i in range(len(ds)):
        a = ds.get_ann_info(i)
        for k in ('bboxes', 'labels', 'bboxes_ignore', 'labels_ignore'):
            assert np.array_equal(a[k], g[f'voc_{i}_{k}']) and a[k].dtype == g[f'voc_{i}_{k}'].dtype, (i, k)
    results = [[g[f'res_{i}_{c}'] for c in range(20)] for i in range(len(ds))]
    anns = [ds.get_ann_info(i) for i in range(len(ds))]
    for name, kw in (('voc07', dict(dataset='voc07', use_legacy_coordinate=True)),
                     ('area', dict(dataset=None, use_legacy_coordinate=False)),
                     ('thr75', dict(dataset='voc07', use_legacy_coordinate=True, iou_thr=0.75))):
        m, res = eval_map(results, anns, **kw)
        assert m == pytest.approx(float(g[f'map_{name}']), abs=1e-7)
        assert np.allclose(np.array([r['ap'] for r in res]), g[f'aps_{name}'], atol=1e-7)
    assert ds.evaluate(results, metric='mAP') == pytest.approx(json.loads(str(g['voc_eval'])))
    # 'recall' (voc.py:91-106): the ground-truth boxes themselves as proposals recall everything; half of them, less
    props = [np.concatenate([a['bboxes'], np.full((len(a['bboxes']), 1), 0.5, np.float32)], axis=1) for a in anns]
    rec = ds.evaluate(props, metric='recall', proposal_nums=(100,), iou_thr=[0.5, 0.75])
    assert rec['recall@100@0.5'] == 1.0 and rec['recall@100@0.75'] == 1.0 and rec['AR@100'] == 1.0
    some = ds.evaluate([p[: (len(p) + 1) // 2] for p in props], metric=['recall'], proposal_nums=(100,), iou_thr=0.5)
    assert 0 < some['recall@100@0.5'] < 1 and 'AR@100' not in some
    tr = D.VOCDataset(ann_file=lst, img_prefix=prefix, pipeline=[], min_size=8)
    assert len(tr) < len(ds)                       # empty-gt and too-small images are filtered
    rep = D.build_dataset(dict(type='RepeatDataset', times=3,
                               dataset=dict(type='VOCDataset', ann_file=lst, img_prefix=prefix, pipeline=[])))
    assert len(rep) == 3 * len(tr) and len(rep.flag) == len(rep)
    cat = D.build_dataset(dict(type='VOCDataset', ann_file=[lst, lst], img_prefix=[prefix, prefix], pipeline=[]))
    assert len(cat) == 2 * len(tr) and len(cat.flag) == len(cat)


def test_varifocal_loss_and_voc_rpn_loss_match_reference_golden():
    from brcnn import Config
    from brcnn.losses import VarifocalLoss
    from tests import util
    g = load('g15_voc')
    pred = torch.from_numpy(g['vfl_pred']).requires_grad_()
    target = torch.from_numpy(g['vfl_target'])
    for iw in (True, False):
        loss = VarifocalLoss(use_sigmoid=True, alpha=0.75, gamma=2.0, iou_weighted=iw, loss_weight=1.5)(
            pred, target, avg_factor=37.0)
        grad, = torch.autograd.grad(loss, pred)
        assert torch.allclose(loss.detach(), torch.from_numpy(g[f'vfl_{int(iw)}']), rtol=1e-6, atol=1e-7)
        assert torch.allclose(grad, torch.from_numpy(g[f'vfl_grad_{int(iw)}']), rtol=1e-5, atol=1e-8)
    cfg = Config.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                       'configs/boosting_rcnn/boosting_rcnn_r50_pafpn_1x_voc.py'))
    c = cfg.model.rpn_head.copy()
    c.update(train_cfg=cfg.model.train_cfg.rpn, test_cfg=cfg.model.test_cfg.rpn)
    head = brcnn.build_head(c)
    cls = [torch.from_numpy(g[f'rpn_cls{i}']) for i in range(5)]
    reg = [torch.from_numpy(g[f'rpn_reg{i}']) for i in range(5)]
    iou = [torch.from_numpy(g[f'rpn_iou{i}']) for i in range(5)]
    _, metas, gts, _ = util.demo_inputs(2, 128, 192, seed=15, num_gt=4)
    losses = head.loss(cls, reg, iou, gts, metas)
    for k, v in losses.items():
        assert torch.allclose(torch.stack(v), torch.from_numpy(g['rpn_' + k]), rtol=1e-5, atol=1e-6), k


def test_voc_box_head_structure():
    g = load('g15_voc')
    head = brcnn.build_head(json.loads(str(g['head_cfg'])))
    assert sorted(head.state_dict().keys()) == g['head_keys'].tolist()
    assert not head._simple and head.fc_reg.in_features == 256 * 49 and head.fc_cls.in_features == 1024


def test_autoaugment_random_crop_match_reference_golden():
    g = load('g16_resnext_autoaug')
    pipe = P.Compose(_cfg(str(g['pipe_cfg'])))
    seen = set()
    for s in range(8):
        np.random.seed(70 + s)
        out = pipe(dict(img=g['img'].copy(), img_shape=g['img'].shape, ori_shape=g['img'].shape, img_fields=['img'],
                        filename='x', ori_filename='x', gt_bboxes=g['boxes'].copy(), gt_labels=g['labels'].copy(),
                        bbox_fields=['gt_bboxes']))
        t = out['img'].data
        assert list(t.shape) == g[f'aa{s}_shape'].tolist()
        assert np.allclose(t.double().sum((1, 2)).numpy(), g[f'aa{s}_sum'], rtol=0, atol=1e-6)
        assert np.array_equal(out['gt_bboxes'].data.numpy(), g[f'aa{s}_boxes'])
        assert np.array_equal(out['gt_labels'].data.numpy(), g[f'aa{s}_labels'])
        meta = json.loads(str(g[f'aa{s}_meta']))
        m = out['img_metas'].data
        assert list(m['img_shape']) == meta['img_shape'] and bool(m['flip']) == meta['flip']
        assert [float(v) for v in m['scale_factor']] == meta['scale_factor']
        seen.add(len(out['gt_labels'].data))
    assert len(seen) > 1       # some draws took the crop policy and lost boxes


def test_resize_hand_derived_vectors():
    """cv2.resize(INTER_LINEAR, uint8) cases small enough to check on paper (tests/golden/kat_mmcv_ops.json,
    `resize_hand`; hand-derived from OpenCV's published rule, NOT cv2 outputs: cv2 is not in the image, so the
    resize restatement stays "unpinned against cv2" beyond these) through the host restatement and the C oracle"""
    import json
    import os
    from oracle import orc
    kat = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'kat_mmcv_ops.json')))
    assert len(kat['resize_hand']) >= 7
    for case in kat['resize_hand']:
        src = np.array(case['src'], dtype=np.uint8)[:, :, None].repeat(3, 2)
        w, h = case['size_wh']
        out = P.imresize_u8(src, (w, h))
        assert out.shape == (h, w, 3) and np.array_equal(out[:, :, 0], np.array(case['out'])), case['derivation']
        ref = orc.preprocess_u8(src, w, h, h, w, None, [0., 0., 0.], [1., 1., 1.], False)
        assert np.array_equal(ref.numpy()[0], np.array(case['out'], dtype=np.float32))


def test_preprocess_hand_derived_vectors():
    """Resize -> RandomFlip -> Normalize (to_rgb) -> Pad cases derived on paper (kat_mmcv_ops.json `preprocess_hand`):
    flip directions, the BGR -> RGB swap with mean / std in RGB order, zero padding -- through the host chain
    (pipelines.imresize_u8 / imflip / imnormalize / impad) and the C oracle of the fused device front door"""
    import json
    import os
    from oracle import orc
    kat = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'kat_mmcv_ops.json')))
    assert len(kat['preprocess_hand']) >= 3
    for case in kat['preprocess_hand']:
        src = np.array(case['src_bgr'], dtype=np.uint8)
        (w, h), (ph, pw) = case['new_wh'], case['pad_hw']
        want = np.array(case['out_chw'], dtype=np.float32)
        img = P.imresize_u8(src, (w, h))
        if case['flip']:
            img = P.imflip(img, case['flip'])
        img = P.imnormalize(img, np.array(case['mean'], np.float32), np.array(case['std'], np.float32), case['to_rgb'])
        img = P.impad(img, shape=(ph, pw))
        assert np.array_equal(img.transpose(2, 0, 1), want), case['derivation']
        ref = orc.preprocess_u8(src, w, h, ph, pw, case['flip'], case['mean'], case['std'], case['to_rgb'])
        assert np.array_equal(ref.numpy(), want), case['derivation']
