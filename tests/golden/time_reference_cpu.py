"""Sanity anchor: wall time of the shimmed REFERENCE detector on
the authoring container's CPU cores for BASELINE configs[0] (UTDAC config, 2 images
800x1344, forward_test / simple_test), beside this repo's CPU restatement of the same pipeline
(oracle/cpu_pipeline.py) on the same inputs and weights.  Authoring container only: reads
/root/reference.  Run:  python tests/golden/time_reference_cpu.py"""
import copy
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _mmcv_shim  # noqa: E402

_mmcv_shim.install()
from tests import util  # noqa: E402
import brcnn  # noqa: E402,F401
from brcnn.config import Config  # noqa: E402
from make_golden import REF_CFG, cfgdict  # noqa: E402


def main():
    from mmdet.models import build_detector
    from oracle import cpu_pipeline
    threads = cpu_pipeline.available_cpus()
    torch.set_num_threads(threads)
    cfg = Config.fromfile(REF_CFG)
    ref = build_detector(cfgdict(copy.deepcopy(cfg.model.to_dict())))
    sd = util.seeded_state_dict(ref, seed=10)
    ref.load_state_dict(sd)
    ref.eval()
    img, metas, _, _ = util.demo_inputs(2, 800, 1344, seed=0)
    out = {}
    with torch.no_grad():
        ref.simple_test(img, metas, rescale=True)
        t = time.perf_counter()
        for _ in range(3):
            r_ref = ref.simple_test(img, metas, rescale=True)
        out['reference_s_per_batch2'] = (time.perf_counter() - t) / 3
    with cpu_pipeline.patched(), torch.no_grad():
        mine = brcnn.build_detector(Config.fromfile(os.path.join(
            ROOT, 'configs/boosting_rcnn/boosting_rcnn_r50_pafpn_1x_utdac.py')).model)
        mine.load_state_dict(sd)
        mine.eval()
        mine.simple_test(img, metas, rescale=True)
        t = time.perf_counter()
        for _ in range(3):
            r_mine = mine.simple_test(img, metas, rescale=True)
        out['restatement_s_per_batch2'] = (time.perf_counter() - t) / 3
    out['threads'] = threads
    out['reference_img_per_s'] = 2 / out['reference_s_per_batch2']
    out['restatement_img_per_s'] = 2 / out['restatement_s_per_batch2']
    out['detections'] = [int(sum(len(c) for c in r)) for r in r_ref], [int(sum(len(c) for c in r)) for r in r_mine]
    print(json.dumps(out))


if __name__ == '__main__':
    main()
