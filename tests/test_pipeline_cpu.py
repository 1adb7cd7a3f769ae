"""CPU plumbing test (BASELINE.json configs[0]): the UTDAC config, 2 images, CPU-only
`forward_test` through the registry-built detector -- with the operator namespace swapped for
the oracle's CPU restatements (oracle.cpu_pipeline.patched, test infrastructure) -- against
the end-to-end golden of the imported reference."""
import numpy as np
import torch

import brcnn  # noqa: F401
from brcnn import Config, build_detector
from oracle import cpu_pipeline
from tests import util
from tests.test_host_cpu import CFG, load


def _match(got, ref, box_tol=1e-2, score_tol=1e-3):
    if len(ref) == 0:
        return len(got) == 0
    if len(got) == 0:
        return False
    d = np.abs(ref[:, None, :4] - got[None, :, :4]).max(-1)
    s = np.abs(ref[:, None, 4] - got[None, :, 4])
    return ((d < box_tol) & (s < score_tol)).any(1).mean() >= 0.99


def test_forward_test_two_images_cpu_matches_reference_golden():
    g = load('g10_model')
    cfg = Config.fromfile(CFG)
    torch.set_num_threads(8)
    with cpu_pipeline.patched():
        m = build_detector(cfg.model)
        m.load_state_dict(util.seeded_state_dict(m, seed=10))
        m.eval()
        img, metas, _, _ = util.demo_inputs(2, 128, 192, seed=10)
        with torch.no_grad():
            res = m(return_loss=False, rescale=True, img=[img], img_metas=[metas])
            feats = m.extract_feat(img)
    assert metas[0]['batch_input_shape'] == (128, 192)
    assert len(res) == 2 and all(len(r) == 4 for r in res)
    for i, f in enumerate(feats):
        assert torch.allclose(f[:, :16], torch.from_numpy(g[f'p{i}']), rtol=1e-3, atol=1e-4)
    for b in range(2):
        for c in range(4):
            assert res[b][c].dtype == np.float32
            assert _match(res[b][c], g[f'res{b}_{c}']), (b, c)


def test_product_ops_restored_after_patch():
    from brcnn import ops
    import pytest
    with pytest.raises(RuntimeError):
        ops.rpn_score(torch.zeros(4), torch.zeros(4))


def test_boosting_variant_heads_cpu_match_reference_golden(monkeypatch):
    """BoostRoIHead / DyProbRoIHead host logic (priors, boosted label weights, Dynamic R-CNN
    schedule) on the CPU oracle pipeline against golden g18; the device run of the same cases is
    tests/test_golden_gpu.py"""
    import tests.test_golden_gpu as tg
    monkeypatch.setattr(tg, 'DEV', 'cpu')
    torch.set_num_threads(8)
    with cpu_pipeline.patched():
        tg.test_boost_roi_head_train_golden('q', dict(boost=True, quality=True, iou_gamma=0.5, gamma=0.5))
        tg.test_boost_roi_head_train_golden('p', dict(boost=True, quality=False, gamma=0.5, alpha=0.75))
        tg.test_boost_roi_head_test_golden()
        tg.test_dyprob_roi_head_schedule_golden()
