import json, numpy as np, torch, sys, os
sys.path.insert(0, os.getcwd())
import brcnn
from brcnn import Config
from brcnn.config import ConfigDict
from tests.test_host_cpu import CFG, T, load
from tests import util
g = load('g5_rpn_get_bboxes')
cfg = Config.fromfile(CFG)
c = cfg.model.rpn_head.copy()
c.update(train_cfg=cfg.model.train_cfg.rpn, test_cfg=cfg.model.test_cfg.rpn)
head = brcnn.build_head(c).to('cuda')
cls = [T(g[f'cls{i}']).cuda() for i in range(5)]
reg = [T(g[f'reg{i}']).cuda() for i in range(5)]
iou = [T(g[f'iou{i}']).cuda() for i in range(5)]
metas = [dict(img_shape=(320, 509, 3), scale_factor=np.ones(4, np.float32), pad_shape=(320, 512, 3)) for _ in range(2)]
for name in ('test','train','small'):
    pc = ConfigDict(json.loads(str(g[name + '_cfg'])))
    res = head.get_bboxes(cls, reg, iou, metas, cfg=pc)
    for b in range(2):
        ref = T(g[f'{name}_props{b}']); got = res[b].cpu()
        d = (got-ref).abs()
        bad = (d[:, :4].max(1)[0] > 2e-4).nonzero().flatten()
        print(name, b, got.shape, 'maxdiff box', d[:, :4].max().item(), 'score', d[:,4].max().item(), 'bad rows', bad.tolist()[:10])
        for r in bad.tolist()[:3]:
            print('  ref', ref[r].tolist(), '\n  got', got[r].tolist())
        # compare pre-nms candidates
        # score ulp
import os
print('affinity', len(os.sched_getaffinity(0)), os.cpu_count())
