import sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
m, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_utdac.py', torch.device('cuda', 0))
m = m.eval().freeze_for_inference()
m.set_compute_dtype(os.environ.get('BRCNN_DTYPE', 'f32'))
img, metas = bench.synthetic_batch(8, 'cuda')
for _ in range(4):
    with torch.no_grad():
        out = m.simple_test_device(img, metas, rescale=True)
    out[2].cpu()
torch.cuda.synchronize()
