"""bf16 train step of bench.py with the weight gradients of the conv / FC layers switched off (their weights frozen: the
data-gradient chain and everything else still runs) -- what the step would cost if the weight-gradient stream were free.
    python tools/experiments/r06/no_wgrad.py [freeze|train]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import bench
from brcnn import blocks
from brcnn.optim import FusedSGD
mode = sys.argv[1] if len(sys.argv) > 1 else 'freeze'
dev = torch.device('cuda', 0)
model, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_coco.py', dev)
model = model.train()
model.set_compute_dtype('bf16')
blocks.conv_weights_channels_last(model)
if mode == 'freeze':
    for n, p in model.named_parameters():
        if p.dim() >= 2:
            p.requires_grad_(False)
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


for _ in range(6):
    step()
res = []
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 10 * 1e3)
print(f'{mode:7s} wgrad_stream={os.environ.get("BRCNN_WGRAD_STREAM", "1")}: ' + ' '.join(f'{v:6.2f}' for v in res), flush=True)
