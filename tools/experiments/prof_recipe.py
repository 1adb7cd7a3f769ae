"""wall vs GPU time of one recipe's inference step (RECIPE env: config file name)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
import brcnn
from brcnn import Config, build_detector
from tests import util
f = os.environ.get('RECIPE', 'boosting_rcnn_r101_pafpn_softnms_coco.py')
cfg = Config.fromfile(os.path.join('configs/boosting_rcnn', f))
m = build_detector(cfg.model)
m.load_state_dict(util.seeded_state_dict(m, seed=0))
m = m.cuda().eval()
img, metas = bench.synthetic_batch(8, 'cuda', seed=0)
with torch.no_grad():
    for _ in range(2):
        r = m.simple_test(img, metas, rescale=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(int(os.environ.get('N', '4'))):
        r = m.simple_test(img, metas, rescale=True)
    torch.cuda.synchronize()
    print('wall ms/step', (time.perf_counter() - t) / int(os.environ.get('N', '4')) * 1e3)
