"""train step against the HIP priorities of the main and the weight-gradient stream (one variant per process:
python tools/experiments/prio_try.py <side priority or 'none'> <main: 'default' or priority>)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
side, mainp = sys.argv[1], sys.argv[2]
if side != 'none':
    os.environ['BRCNN_WGRAD_PRIO'] = side
import torch
import bench
from brcnn import blocks
from brcnn.optim import FusedSGD
dev = torch.device('cuda', 0)
print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else '?')
model, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_coco.py', dev)
model = model.train()
model.set_compute_dtype('bf16')
blocks.conv_weights_channels_last(model)
params = [p for p in model.parameters() if p.requires_grad]
opt = FusedSGD(params, lr=1e-5, momentum=0.9, weight_decay=1e-4)
opt.register_conv_weights(model, blocks.compute_dtype())
model.early_rpn_backward = True
img, metas = bench.synthetic_batch(8, dev)
gtb, gtl = bench.synthetic_gt(8, dev, 80)


def step():
    opt.zero_grad(set_to_none=True)
    losses = model(img=img, img_metas=metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
    loss, _ = model._parse_losses(losses)
    loss.backward()
    opt.step(max_norm=35)


def run():
    for _ in range(5): step()
    res = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): step()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 8 * 1e3)
    print(f'side {side:5s} main {mainp:8s}: ' + ' '.join(f'{v:6.2f}' for v in res), flush=True)


if mainp == 'default':
    run()
else:
    ms = torch.cuda.Stream(dev, priority=int(mainp))
    with torch.cuda.stream(ms):
        run()
