"""not-gpu: train / test drivers -- LR schedule, optimizer construction,
checkpoint layout / resume, `--cfg-options` parsing, and one short epoch of `train_detector`
+ evaluation on the CPU oracle pipeline (oracle/cpu_pipeline.patched swaps the HIP ops for
their CPU restatements; the product path itself refuses CPU tensors)."""
import os

import numpy as np
import pytest
import torch

import brcnn  # noqa: F401
from brcnn import Config, apis, build_detector
from brcnn.config import DictAction
from brcnn.datasets import build_dataset
from tests.test_host_cpu import CFG

CLASSES = ('echinus', 'starfish', 'holothurian', 'scallop')


def test_step_lr_with_linear_warmup():
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.02)
    up = apis.StepLrUpdater(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=0.001, step=[8, 11])
    up.before_run(opt)
    assert up.lr_at(0, 0)[0] == pytest.approx(0.02 * 0.001)
    assert up.lr_at(0, 250)[0] == pytest.approx(0.02 * (1 - 0.5 * 0.999))
    assert up.lr_at(0, 499)[0] == pytest.approx(0.02 * (1 - (1 / 500) * 0.999))
    assert up.lr_at(0, 500)[0] == 0.02 and up.lr_at(7, 10 ** 5)[0] == 0.02
    assert up.lr_at(8, 10 ** 5)[0] == pytest.approx(0.002) and up.lr_at(10, 10 ** 5)[0] == pytest.approx(0.002)
    assert up.lr_at(11, 10 ** 5)[0] == pytest.approx(0.0002)
    up.apply(opt, 8, 10 ** 5)
    assert opt.param_groups[0]['lr'] == pytest.approx(0.002) and opt.param_groups[0]['initial_lr'] == 0.02
    exp = apis.StepLrUpdater(policy='step', warmup='exp', warmup_iters=100, warmup_ratio=0.01, step=3)
    exp.before_run(opt)
    assert exp.lr_at(0, 50)[0] == pytest.approx(0.02 * 0.01 ** 0.5)


def test_build_optimizer_paramwise():
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Linear(4, 2))
    m[0].weight.requires_grad_(False)
    opt = apis.build_optimizer(m, dict(type='SGD', lr=0.02, momentum=0.9, weight_decay=1e-4))
    assert len(opt.param_groups) == 1 and len(opt.param_groups[0]['params']) == 5
    opt = apis.build_optimizer(m, dict(type='SGD', lr=0.02, momentum=0.9, weight_decay=1e-4,
                                       paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0., norm_decay_mult=0.,
                                                          custom_keys={'2.weight': dict(lr_mult=0.1)})))
    by = {id(g['params'][0]): g for g in opt.param_groups}
    assert by[id(m[0].bias)]['lr'] == pytest.approx(0.04) and by[id(m[0].bias)]['weight_decay'] == 0.
    assert by[id(m[1].weight)]['weight_decay'] == 0. and by[id(m[1].weight)]['lr'] == 0.02
    assert by[id(m[2].weight)]['lr'] == pytest.approx(0.002)
    assert id(m[0].weight) not in by


def test_cfg_options_action():
    import argparse
    p = argparse.ArgumentParser()
    p.add_argument('--cfg-options', nargs='+', action=DictAction)
    a = p.parse_args(['--cfg-options', 'optimizer.lr=0.01', 'data.samples_per_gpu=4', 'lr_config.step=[8,11]',
                      'model.test_cfg.rpn.nms_pre=2000', 'a.b=x,y', 'flag=True', 'scale=(1333,800)'])
    assert a.cfg_options == {'optimizer.lr': 0.01, 'data.samples_per_gpu': 4, 'lr_config.step': [8, 11],
                             'model.test_cfg.rpn.nms_pre': 2000, 'a.b': ['x', 'y'], 'flag': True,
                             'scale': (1333, 800)}
    cfg = Config.fromfile(CFG)
    cfg.merge_from_dict(a.cfg_options)
    assert cfg.optimizer.lr == 0.01 and cfg.lr_config.step == [8, 11] and cfg.model.test_cfg.rpn.nms_pre == 2000
    assert cfg.optimizer.momentum == 0.9


def test_checkpoint_layout_roundtrip(tmp_path):
    m = build_detector(Config.fromfile(CFG).model)
    m.CLASSES = CLASSES
    opt = apis.build_optimizer(m, dict(type='SGD', lr=0.02, momentum=0.9, weight_decay=1e-4))
    path = str(tmp_path / 'w' / 'epoch_3.pth')
    apis.save_checkpoint(m, path, optimizer=opt, meta=dict(epoch=3, iter=77))
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert set(ck) == {'meta', 'state_dict', 'optimizer'}
    assert ck['meta']['epoch'] == 3 and ck['meta']['iter'] == 77 and ck['meta']['CLASSES'] == CLASSES
    assert list(ck['state_dict'].keys()) == list(m.state_dict().keys())      # the reference's names
    assert 'backbone.layer1.0.conv1.weight' in ck['state_dict'] and 'roi_head.bbox_head.fc_cls.weight' in ck['state_dict']
    # a checkpoint written from a DDP-wrapped model ('module.' prefix) loads too
    torch.save(dict(state_dict={'module.' + k: v + 1 for k, v in ck['state_dict'].items()}), str(tmp_path / 'ddp.pth'))
    m2 = build_detector(Config.fromfile(CFG).model)
    apis.load_checkpoint(m2, str(tmp_path / 'ddp.pth'), strict=True)
    k = 'rpn_head.rpn_cls.weight'
    assert torch.equal(m2.state_dict()[k], ck['state_dict'][k] + 1)


def test_pretrained_backbone_init(tmp_path, monkeypatch):
    """init_cfg=dict(type='Pretrained', checkpoint='torchvision://resnet50') (the recipes' backbone entry):
    resolved against $BRCNN_PRETRAINED_DIR, loaded non-strictly (a torchvision file carries fc.*), and a
    missing file is an error unless random init was asked for"""
    from brcnn.backbones import ResNet
    cfg = Config.fromfile(CFG)
    assert cfg.model.backbone.init_cfg == dict(type='Pretrained', checkpoint='torchvision://resnet50')
    src = ResNet(depth=50)
    sd = {k: torch.randn_like(v) if v.is_floating_point() else v for k, v in src.state_dict().items()}
    sd['fc.weight'], sd['fc.bias'] = torch.zeros(1000, 2048), torch.zeros(1000)       # as in a torchvision file
    monkeypatch.delenv('BRCNN_ALLOW_RANDOM_INIT', raising=False)
    monkeypatch.setenv('BRCNN_PRETRAINED_DIR', str(tmp_path / 'nothing_here'))
    m = build_detector(cfg.model)
    with pytest.raises(FileNotFoundError, match='BRCNN_PRETRAINED_DIR'):
        m.init_weights()
    monkeypatch.setenv('BRCNN_ALLOW_RANDOM_INIT', '1')
    m.init_weights()                                                                   # explicit opt-in: warns, continues
    monkeypatch.delenv('BRCNN_ALLOW_RANDOM_INIT')
    os.makedirs(tmp_path / 'models')
    torch.save(sd, str(tmp_path / 'models' / 'resnet50.pth'))
    monkeypatch.setenv('BRCNN_PRETRAINED_DIR', str(tmp_path / 'models'))
    m.init_weights()
    got = m.backbone.state_dict()
    for k in ('conv1.weight', 'layer1.0.bn3.weight', 'layer4.2.conv3.weight', 'bn1.running_var'):
        assert torch.equal(got[k], sd[k]), k
    # a plain path and a full-detector checkpoint ('backbone.' prefix, 'state_dict' wrapper) work too
    torch.save(dict(state_dict={'backbone.' + k: v + 1 for k, v in sd.items() if not k.startswith('fc.')}),
               str(tmp_path / 'det.pth'))
    m.backbone.init_cfg = dict(type='Pretrained', checkpoint=str(tmp_path / 'det.pth'))
    m.init_weights()
    assert torch.equal(m.backbone.state_dict()['layer2.1.conv2.weight'], sd['layer2.1.conv2.weight'] + 1)


def _tiny_cfg(tmp_path, max_epochs):
    from tests.golden.synth import synthetic_coco
    ann_file, prefix = synthetic_coco(str(tmp_path / 'data'), n_img=7)
    cfg = Config.fromfile(CFG)
    norm = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    train_pipe = [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
                  dict(type='Resize', img_scale=(160, 96), keep_ratio=True), dict(type='RandomFlip', flip_ratio=0.5),
                  dict(type='Normalize', **norm), dict(type='Pad', size_divisor=32), dict(type='DefaultFormatBundle'),
                  dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])]
    test_pipe = [dict(type='LoadImageFromFile'),
                 dict(type='MultiScaleFlipAug', img_scale=(160, 96), flip=False,
                      transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                                  dict(type='Normalize', **norm), dict(type='Pad', size_divisor=32),
                                  dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])])]
    common = dict(type='CocoDataset', ann_file=ann_file, img_prefix=prefix, classes=CLASSES)
    cfg.data = dict(samples_per_gpu=2, workers_per_gpu=0, train=dict(common, pipeline=train_pipe),
                    val=dict(common, pipeline=test_pipe), test=dict(common, pipeline=test_pipe))
    cfg.runner = dict(type='EpochBasedRunner', max_epochs=max_epochs)
    cfg.lr_config = dict(policy='step', warmup='linear', warmup_iters=4, warmup_ratio=0.1, step=[1])
    cfg.optimizer = dict(type='SGD', lr=0.001, momentum=0.9, weight_decay=0.0001)
    cfg.optimizer_config = dict(grad_clip=dict(max_norm=35, norm_type=2))
    cfg.log_config = dict(interval=1, hooks=[dict(type='TextLoggerHook')])
    cfg.work_dir = str(tmp_path / 'work')
    cfg.seed = 0
    cfg.gpu_ids = [0]
    return cfg


def test_train_detector_epoch_checkpoint_resume_and_eval(tmp_path):
    from oracle import cpu_pipeline
    torch.set_num_threads(cpu_pipeline.available_cpus())
    cfg = _tiny_cfg(tmp_path, max_epochs=1)
    apis.set_random_seed(0)
    with cpu_pipeline.patched():
        model = build_detector(cfg.model)
        ds = build_dataset(cfg.data.train)
        model.CLASSES = ds.CLASSES
        n_iter = len(ds) // 2 + (len(ds) % 2 > 0)
        runner = apis.train_detector(model, ds, cfg, distributed=False, validate=True, device=torch.device('cpu'))
        assert runner.epoch == 1 and runner.iter >= n_iter
        assert os.path.exists(os.path.join(cfg.work_dir, 'epoch_1.pth'))
        assert os.path.exists(os.path.join(cfg.work_dir, 'latest.pth'))
        # text-logger rows: warm-up LR ramps linearly from 0.1 * lr, losses are finite
        lrs = [row[2] for row in runner.history]
        assert lrs[0] == pytest.approx(0.001 * 0.1) and lrs[1] == pytest.approx(0.001 * (1 - 0.75 * 0.9))
        for _, _, _, vals in runner.history:
            assert {'loss_rpn_cls', 'loss_rpn_bbox', 'loss_rpn_iou', 'loss_cls', 'loss_bbox', 'loss', 'grad_norm'} \
                <= set(vals) and all(np.isfinite(v) for v in vals.values())
        # the evaluation hook ran on the val split
        assert len(runner.eval_history) == 1 and runner.eval_history[0][0] == 1
        res = runner.eval_history[0][1]
        assert res == {} or 0.0 <= res['bbox_mAP'] <= 1.0
        # resume: epoch / iter / optimizer state continue, LR drops at epoch 1 (step=[1])
        cfg2 = _tiny_cfg(tmp_path, max_epochs=2)
        cfg2.resume_from = os.path.join(cfg.work_dir, 'epoch_1.pth')
        model2 = build_detector(cfg2.model)
        runner2 = apis.train_detector(model2, build_dataset(cfg2.data.train), cfg2, distributed=False,
                                      validate=False, device=torch.device('cpu'))
        assert runner2.epoch == 2 and runner2.iter == 2 * runner.iter
        assert runner2.history[0][0] == 2 and runner2.history[-1][2] == pytest.approx(0.0001)
        assert any('momentum_buffer' in s for s in runner2.optimizer.state_dict()['state'].values())
        # single_gpu_test result structure: per image a list of (k,5) arrays per class
        from brcnn.datasets import build_dataloader
        dt = build_dataset(cfg.data.test, dict(test_mode=True))
        out = apis.single_gpu_test(model2, build_dataloader(dt, 1, 0, dist=False, shuffle=False))
        assert len(out) == len(dt) and len(out[0]) == 4 and out[0][0].shape[1] == 5
    with pytest.raises(AssertionError):
        apis._check_num_classes(model, type('D', (), {'CLASSES': ('a', 'b')})(), apis.get_root_logger())


def test_runner_holds_the_cyclic_collector_off_inside_steps_and_restores_it(tmp_path):
    """bench.py times the train step with Python's cyclic collector off, claiming the runner does the same.
    It does now: EpochBasedRunner.train disables the collector for the epoch's steps, collects at the log interval, and
    restores the caller's state on the way out -- also when a step raises."""
    import gc
    import logging
    seen = []

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(3))

        def train_step(self, data, optimizer):
            seen.append(gc.isenabled())
            if data == 'boom':
                raise ValueError('boom')
            loss = (self.w * 2).sum()
            return dict(loss=loss, log_vars={'loss': float(loss)}, num_samples=1)

    net = Net()
    runner = apis.EpochBasedRunner(net, torch.optim.SGD(net.parameters(), lr=0.1), str(tmp_path), logging.getLogger('t'), 1)
    runner.register_training_hooks(dict(policy='step', step=[1]), log_config=dict(interval=2))
    runner.lr_updater.before_run(runner.optimizer)
    assert gc.isenabled()
    collected = []
    cb = lambda phase, info: collected.append(phase) if phase == 'stop' else None
    gc.callbacks.append(cb)
    try:
        runner.train([0, 1, 2, 3])
    finally:
        gc.callbacks.remove(cb)
    assert seen == [False] * 4 and gc.isenabled()
    assert len(collected) >= 3                  # one before the epoch, one per log interval (2 of them)
    # the caller's collector comes back when a step raises
    seen.clear()
    with pytest.raises(ValueError):
        runner.train([0, 'boom', 2])
    assert seen == [False, False] and gc.isenabled()
    # manual_gc off (BRCNN_RUNNER_GC=1 / cfg.manual_gc = False): the collector is left alone
    seen.clear()
    runner.manual_gc = False
    runner.train([0])
    assert seen == [True]
