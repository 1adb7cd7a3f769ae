"""Which aten copies / fills / adds the train step issues, by shape and GPU time (torch profiler)."""
import sys, os, torch, collections
sys.path.insert(0, os.getcwd())
import bench
from torch.profiler import profile, ProfilerActivity
from brcnn import Config, build_detector
from tests import util
model, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_coco.py', torch.device('cuda', 0))
model = model.train()
model.set_compute_dtype(os.environ.get('BRCNN_DTYPE', 'bf16'))
from brcnn import blocks
from brcnn.optim import FusedSGD
blocks.conv_weights_channels_last(model)
params = [p for p in model.parameters() if p.requires_grad]
opt = FusedSGD(params, lr=2e-5, momentum=0.9, weight_decay=1e-4)
opt.register_conv_weights(model, blocks.compute_dtype())
model.early_rpn_backward = os.environ.get('BRCNN_EARLY_RPN_BWD', '1') != '0'
img, metas = bench.synthetic_batch(8, 'cuda', seed=0)
gtb, gtl = bench.synthetic_gt(8, 'cuda', 80, seed=0)
def step():
    opt.zero_grad(set_to_none=True)
    losses = model(img=img, img_metas=metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
    loss, lv = model._parse_losses(losses)
    loss.backward()
    opt.step(max_norm=35)
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
c = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name.startswith('aten::') and e.device_time > 0 and e.name not in ('aten::zeros', 'aten::clone', 'aten::to', 'aten::_to_copy', 'aten::contiguous', 'aten::zeros_like', 'aten::full', 'aten::ones', 'aten::float', 'aten::linear'):
        st = [f for f in (e.stack or []) if 'boosting-r-cnn_amd' in f or 'bench.py' in f]
        k = (e.name, str(e.input_shapes)[:70], (st[0].split('boosting-r-cnn_amd/')[-1][:60] if st else '?'))
        c[k][0] += 1
        c[k][1] += e.device_time
tot = sum(v[1] for v in c.values())
print('listed aten ops: %.2f ms device time' % (tot / 1e3))
for k, v in sorted(c.items(), key=lambda kv: -kv[1][1])[:70]:
    print('%4d %8.1f us  %s %s  <- %s' % (v[0], v[1], k[0], k[1], k[2]))
