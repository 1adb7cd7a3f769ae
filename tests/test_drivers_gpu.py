"""-m gpu: the input front door and the drivers on the device -- the fused
Resize/Flip/Normalize/Pad HIP kernel against the C oracle (bit-exact), the device pipeline
against the host pipeline, and tools/train.py + tools/test.py end to end on a synthetic
COCO-format dataset."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import brcnn  # noqa: F401
from brcnn import ops
from brcnn import pipelines as P
from oracle import orc
from tests.test_drivers_cpu import CLASSES, _tiny_cfg

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(f'brcnn_tool_{name}', os.path.join(ROOT, 'tools', f'{name}.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_preprocess_kernel_bit_exact_vs_oracle():
    rng = np.random.RandomState(2)
    cases = [(1080, 1920, (1333, 800)), (480, 640, (1333, 800)), (300, 200, (1333, 800)), (75, 113, (160, 96)),
             (800, 1333, (1333, 800)), (33, 17, (64, 48))]
    for (h, w, scale) in cases:
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        nw, nh = P.rescale_size((w, h), scale)
        ph, pw = -(-nh // 32) * 32, -(-nw // 32) * 32
        src = torch.from_numpy(img).to(DEV)
        for flip in (None, 'horizontal', 'vertical', 'diagonal'):
            for to_rgb in (True, False):
                out = torch.full((3, ph, pw), 7.0, device=DEV)
                ops.preprocess_u8(src, out, nw, nh, flip, MEAN, STD, to_rgb)
                ref = orc.preprocess_u8(img, nw, nh, ph, pw, flip, MEAN, STD, to_rgb)
                assert torch.equal(out.cpu(), ref), (h, w, flip, to_rgb)
    with pytest.raises(Exception):
        ops.preprocess_u8(src, torch.empty((3, 8, 8), device=DEV), 64, 48, None, MEAN, STD)      # pad < new size
    with pytest.raises(Exception):
        ops.preprocess_u8(src.cpu(), torch.empty((3, 64, 64), device=DEV), 64, 48, None, MEAN, STD)


def test_device_pipeline_equals_host_pipeline():
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, (75, 113, 3), dtype=np.uint8)
    boxes = np.array([[10, 12, 60, 50], [0, 0, 113, 75], [100, 60, 112.5, 74.2]], dtype=np.float32)
    host_cfg = [dict(type='Resize', img_scale=[(160, 96), (200, 128)], multiscale_mode='range', keep_ratio=True),
                dict(type='RandomFlip', flip_ratio=0.5, direction=['horizontal', 'vertical']),
                dict(type='Normalize', mean=MEAN, std=STD, to_rgb=True), dict(type='Pad', size_divisor=32),
                dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])]
    host, dev = P.Compose(host_cfg), P.Compose(P.fuse_device_pipeline(host_cfg, DEV))

    def fresh():
        return dict(img=img.copy(), img_shape=img.shape, ori_shape=img.shape, img_fields=['img'], filename='x',
                    ori_filename='x', gt_bboxes=boxes.copy(), gt_labels=np.array([0, 3, 1]), bbox_fields=['gt_bboxes'])
    for s in range(8):
        np.random.seed(60 + s)
        a = host(fresh())
        np.random.seed(60 + s)
        b = dev(fresh())
        assert b['img'].data.is_cuda and torch.equal(a['img'].data, b['img'].data.cpu())
        assert torch.equal(a['gt_bboxes'].data, b['gt_bboxes'].data)
        ma, mb = a['img_metas'].data, b['img_metas'].data
        for k in ('img_shape', 'pad_shape', 'flip', 'flip_direction'):
            assert tuple(ma[k]) == tuple(mb[k]) if isinstance(ma[k], tuple) else ma[k] == mb[k]
        assert np.array_equal(ma['scale_factor'], mb['scale_factor'])


def test_train_and_test_tools_end_to_end(tmp_path):
    cfg = _tiny_cfg(tmp_path, max_epochs=1)
    cfg_path = str(tmp_path / 'tiny_cfg.py')
    cfg.dump(cfg_path)
    train, test = _tool('train'), _tool('test')
    work = str(tmp_path / 'work_gpu')
    # forked DataLoader workers + the device front door: the workers decode, the main process runs the HIP transform
    runner = train.main([cfg_path, '--work-dir', work, '--seed', '0', '--device-preprocess', '--allow-random-init',
                         '--cfg-options', 'data.workers_per_gpu=2'])
    assert runner.epoch == 1 and os.path.exists(os.path.join(work, 'epoch_1.pth'))
    assert all(np.isfinite(v) for row in runner.history for v in row[3].values())
    assert len(runner.eval_history) == 1
    # resume for one more epoch through --cfg-options / --resume-from
    runner2 = train.main([cfg_path, '--work-dir', work, '--seed', '0', '--no-validate', '--allow-random-init', '--resume-from',
                          os.path.join(work, 'epoch_1.pth'), '--cfg-options', 'runner.max_epochs=2'])
    assert runner2.epoch == 2 and runner2.iter == 2 * runner.iter
    ckpt = os.path.join(work, 'epoch_2.pth')
    out_pkl = str(tmp_path / 'res.pkl')
    metric = test.main([cfg_path, ckpt, '--eval', 'bbox', '--out', out_pkl])
    assert os.path.exists(out_pkl)
    assert metric == {} or 0.0 <= metric['bbox_mAP'] <= 1.0
    import pickle
    res = pickle.load(open(out_pkl, 'rb'))
    assert len(res) == 7 and len(res[0]) == len(CLASSES) and res[0][0].shape[1] == 5
    # the device front door gives the same detections as the host pipeline
    res_dev = test.main([cfg_path, ckpt, '--out', str(tmp_path / 'res2.pkl'), '--device-preprocess', '--cfg-options',
                         'data.workers_per_gpu=2'])
    for a, b in zip(res, res_dev):
        for x, y in zip(a, b):
            assert x.shape == y.shape and np.allclose(x, y, atol=1e-4)
    test.main([cfg_path, ckpt, '--format-only', '--eval-options', f'jsonfile_prefix={tmp_path}/fmt'])
    assert os.path.exists(str(tmp_path / 'fmt.bbox.json'))


def test_preprocess_kernel_on_hand_derived_vectors():
    """the fused front-door kernel on the paper cases of kat_mmcv_ops.json (`resize_hand`: OpenCV's 8-bit
    INTER_LINEAR rule incl. non-integer down-scaling, 1-pixel sources, round-half-up; `preprocess_hand`: flips,
    BGR -> RGB, mean / std in RGB order, zero padding)"""
    import json
    import os
    import numpy as np
    kat = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'kat_mmcv_ops.json')))
    for case in kat['resize_hand']:
        src = np.array(case['src'], dtype=np.uint8)[:, :, None].repeat(3, 2)
        w, h = case['size_wh']
        out = torch.empty((3, h, w), device=DEV)
        ops.preprocess_u8(torch.from_numpy(np.ascontiguousarray(src)).to(DEV), out, w, h, None, [0., 0., 0.], [1., 1., 1.], False)
        assert np.array_equal(out[0].cpu().numpy(), np.array(case['out'], dtype=np.float32)), case['derivation']
    for case in kat['preprocess_hand']:
        src = np.array(case['src_bgr'], dtype=np.uint8)
        (w, h), (ph, pw) = case['new_wh'], case['pad_hw']
        out = torch.empty((3, ph, pw), device=DEV)
        ops.preprocess_u8(torch.from_numpy(np.ascontiguousarray(src)).to(DEV), out, w, h, case['flip'], case['mean'], case['std'],
                          case['to_rgb'])
        assert np.array_equal(out.cpu().numpy(), np.array(case['out_chw'], dtype=np.float32)), case['derivation']
