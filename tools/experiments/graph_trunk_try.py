"""capture the trunk graphs on a small input and replay them a few times: does the capture survive (ROCm hipGraph) with /
without the weight-gradient side stream inside it?   graph_trunk_try.py <bf16|f32> <side 0|1> [batch h w]"""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import brcnn  # noqa: F401
from brcnn import Config, build_detector, blocks
from brcnn import autograd as A
from brcnn.graphs import GraphedTrunk
from tests import util

dtype, side = sys.argv[1], sys.argv[2] == '1'
b, h, w = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (2, 128, 192)
A.WGRAD_SIDE_STREAM = side
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = Config.fromfile(os.path.join(root, 'configs', 'boosting_rcnn', 'boosting_rcnn_r50_pafpn_1x_coco.py'))
m = build_detector(cfg.model)
m.load_state_dict(util.seeded_state_dict(m, seed=10))
m = m.cuda().train()
blocks.conv_weights_channels_last(m)
m.set_compute_dtype(dtype)
img = torch.randn(b, 3, h, w, device='cuda')
gt = GraphedTrunk(m)
# eager reference
feats = m.extract_feat_nhwc(img)
gouts = [torch.randn_like(f) for f in feats]
params = [p for p in gt._params() if p.requires_grad]
A.grad_arena.new_step()
ref = torch.autograd.grad(feats, params, gouts, allow_unused=True)
A.join_side_streams()
torch.cuda.synchronize()
ref = [None if g is None else g.clone() for g in ref]
ref_feats = [f.detach().clone() for f in feats]
print('eager ok', flush=True)
gt.seen[gt._key(img)] = 5
for it in range(4):
    for p in params:
        p.grad = None
    out = gt(img)
    assert out is not None
    torch.autograd.backward(out, gouts)
    torch.cuda.synchronize()
    ok_f = all(torch.equal(a, b_) for a, b_ in zip(out, ref_feats))
    worst = 0.0
    for p, g in zip(params, ref):
        if g is None:
            assert p.grad is None
            continue
        worst = max(worst, (p.grad.float() - g.float()).abs().max().item() / (g.float().abs().max().item() + 1e-12))
    print(f'replay {it}: feats equal {ok_f}, worst relative gradient deviation {worst:.2e}, captures {gt.captures}', flush=True)
print('GRAPH_OK', flush=True)
