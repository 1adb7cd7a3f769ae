"""torch.profiler view of ONE bf16 train step of bench.py: every aten / runtime kernel (not brcnn_*) with its input
shapes and the repo's call sites.    python tools/experiments/r06/aten_train.py > gpurun_out/r06/aten_train.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import bench
from brcnn import blocks
from brcnn.optim import FusedSGD
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
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


for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [r for r in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12) if r.self_device_time_total > 0 and r.key.startswith('aten::')]
rows.sort(key=lambda r: -r.self_device_time_total)
tot = sum(r.self_device_time_total for r in rows)
n = sum(r.count for r in rows)
print(f'aten ops with device time: {n} calls, {tot / 1e3:.3f} ms')
for r in rows:
    st = [f.split('/')[-1] for f in r.stack if ('boosting' in f or 'bench' in f or 'brcnn' in f)][:4]
    shp = str(r.input_shapes)[:70]
    print(f'{r.count:3d} {r.self_device_time_total:8.1f} us  {r.key:28s} {shp:70s} <- ' + ' <- '.join(st))
