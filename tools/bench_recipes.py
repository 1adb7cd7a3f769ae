"""inference throughput of every shipped recipe (seeded synthetic weights, batch B x 3x800x1344, fp32)"""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
import brcnn
from brcnn import Config, build_detector
from tests import util
B = int(os.environ.get('B', '8'))
DTYPE = os.environ.get('BRCNN_DTYPE', 'f32')
out = {}
d = 'configs/boosting_rcnn'
for f in sorted(os.listdir(d)):
    if not f.endswith('.py'):
        continue
    cfg = Config.fromfile(os.path.join(d, f))
    m = build_detector(cfg.model)
    m.load_state_dict(util.seeded_state_dict(m, seed=0))
    m = m.cuda().eval()
    m.set_compute_dtype(DTYPE)
    img, metas = bench.synthetic_batch(B, 'cuda', seed=0)
    try:
      with torch.no_grad():
        for _ in range(3):
            r = m.simple_test(img, metas, rescale=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        n = 10
        for _ in range(n):
            r = m.simple_test(img, metas, rescale=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / n
    except NotImplementedError as e:
        print(f, 'not available in', DTYPE, '-', str(e)[:60], flush=True)
        m.set_compute_dtype('f32')
        continue
    out[f] = dict(ms_per_batch=dt * 1e3, img_per_s=B / dt, params_M=sum(p.numel() for p in m.parameters()) / 1e6,
                  backbone=type(m.backbone).__name__, neck=type(m.neck).__name__,
                  device_resident=bool(m._device_path_ok()), dets=int(sum(len(c) for c in r[0])))
    print(f, json.dumps(out[f]), flush=True)
    del m
    torch.cuda.empty_cache()
from brcnn.blocks import set_compute_dtype
set_compute_dtype('f32')
json.dump(out, open('gpurun_out/recipes.json' if DTYPE == 'f32' else f'gpurun_out/recipes_{DTYPE}.json', 'w'), indent=1)
