"""Which aten copies / fills / adds the train step issues, by shape and GPU time (torch profiler)."""
import sys, os, torch, collections
sys.path.insert(0, os.getcwd())
import bench
from torch.profiler import profile, ProfilerActivity
from brcnn import Config, build_detector
from tests import util
cfg = Config.fromfile('configs/boosting_rcnn/boosting_rcnn_r50_pafpn_1x_coco.py')
model = build_detector(cfg.model)
model.load_state_dict(util.seeded_state_dict(model, seed=0))
model = model.cuda().train()
model.set_compute_dtype(os.environ.get('BRCNN_DTYPE', 'bf16'))
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=1e-5, momentum=0.9, weight_decay=1e-4)
img, metas = bench.synthetic_batch(8, 'cuda', seed=0)
gtb, gtl = bench.synthetic_gt(8, 'cuda', 80, seed=0)
def step():
    opt.zero_grad(set_to_none=True)
    losses = model(img=img, img_metas=metas, return_loss=True, gt_bboxes=gtb, gt_labels=gtl)
    loss, lv = model._parse_losses(losses)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, max_norm=35, norm_type=2)
    opt.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
c = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name.startswith('aten::') and e.device_time > 0 and e.name in (
            'aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::add', 'aten::add_', 'aten::mul', 'aten::div',
            'aten::cat', 'aten::threshold_backward', 'aten::index', 'aten::sum', 'aten::where', 'aten::mul_'):
        k = (e.name, str(e.input_shapes)[:90])
        c[k][0] += 1
        c[k][1] += e.device_time
tot = sum(v[1] for v in c.values())
print('listed aten ops: %.2f ms device time' % (tot / 1e3))
for k, v in sorted(c.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%4d %8.1f us  %s %s' % (v[0], v[1], k[0], k[1]))
