"""host-side profile of the train step: where the Python / launch overhead goes"""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.getcwd())
import torch
import bench
import brcnn
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
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(60); print(s.getvalue()[:12000])
