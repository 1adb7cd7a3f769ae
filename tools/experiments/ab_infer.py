"""interleaved A/B of the bf16 / fp16 inference step on ONE box under conv tile hooks:
python tools/experiments/ab_infer.py name=hook[,hook] ...   (hooks: brcnn_conv_set_tile_bf16 codes); BRCNN_DTYPE=bf16|f16"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from brcnn import lib
L = lib.load()
dev = torch.device('cuda', 0)
model, cfg = bench.build_model('boosting_rcnn_r50_pafpn_1x_utdac.py', dev)
model = model.eval()
model.set_compute_dtype(os.environ.get('BRCNN_DTYPE', 'bf16'))
img, metas = bench.synthetic_batch(8, dev)


def step():
    with torch.no_grad():
        return model(return_loss=False, rescale=True, img=[img], img_metas=[metas])


variants = [a.split('=') for a in sys.argv[1:]]
for _ in range(5): step()
res = {n: [] for n, _ in variants}
for rnd in range(4):
    for name, spec in variants:
        for h in spec.split(','):
            assert L.brcnn_conv_set_tile_bf16(int(h)) == 0
        for _ in range(3): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 20 * 1e3)
for name, v in res.items():
    print(f'{name:24s} ms/step per round: ' + ' '.join(f'{x:6.3f}' for x in v) + f'   median {sorted(v)[len(v) // 2]:.3f}')
