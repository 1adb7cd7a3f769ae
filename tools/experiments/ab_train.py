"""interleaved A/B of the bf16 train step on ONE box: python tools/experiments/ab_train.py name=hook:arg[,hook:arg] ...
hooks: wgrad (brcnn_conv_set_tile_wgrad_bf16), conv (brcnn_conv_set_tile_bf16), f32 (brcnn_conv_set_tile(-2, arg)),
side (autograd.WGRAD_SIDE_STREAM), early (model.early_rpn_backward), bn3 (autograd.FUSE_RESIDUAL_BN_BACKWARD).  Example: tools/experiments/ab_train.py atomics=wgrad:10 slabs=wgrad:11"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import bench
from brcnn import blocks, lib, autograd as A
from brcnn.optim import FusedSGD
L = lib.load()
dev = torch.device('cuda', 0)
model, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_coco.py', dev)
model = model.train()
model.set_compute_dtype(os.environ.get('BRCNN_DTYPE', 'bf16'))
blocks.conv_weights_channels_last(model)
params = [p for p in model.parameters() if p.requires_grad]
opt = FusedSGD(params, lr=2e-5, momentum=0.9, weight_decay=1e-4)
opt.register_conv_weights(model, blocks.compute_dtype())
img, metas = bench.synthetic_batch(8, dev)
gtb, gtl = bench.synthetic_gt(8, dev, 80)


def step():
    opt.zero_grad(set_to_none=True)
    losses = model(img=img, img_metas=metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
    loss, _ = model._parse_losses(losses)
    loss.backward()
    opt.step(max_norm=35)


def apply(spec):
    for item in spec.split(','):
        hook, arg = item.split(':')
        arg = int(arg)
        if hook == 'wgrad': assert L.brcnn_conv_set_tile_wgrad_bf16(arg) == 0
        elif hook == 'conv': assert L.brcnn_conv_set_tile_bf16(arg) == 0
        elif hook == 'f32': assert L.brcnn_conv_set_tile(-2, arg) == 0
        elif hook == 'side': A.WGRAD_SIDE_STREAM = bool(arg)
        elif hook == 'early': model.early_rpn_backward = bool(arg)
        elif hook == 'bn3': A.FUSE_RESIDUAL_BN_BACKWARD = bool(arg)
        else: raise SystemExit(f'unknown hook {hook}')


variants = [a.split('=') for a in sys.argv[1:]]
for _ in range(4):
    step()
res = {n: [] for n, _ in variants}
for rnd in range(4):
    for name, spec in variants:
        apply(spec)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 8 * 1e3)
for name, v in res.items():
    print(f'{name:24s} ms/step per round: ' + ' '.join(f'{x:6.2f}' for x in v) + f'   median {sorted(v)[len(v) // 2]:.2f}')
