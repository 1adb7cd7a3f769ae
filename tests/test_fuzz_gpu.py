"""-m gpu: a short randomised sweep of conv shapes (forward with scale / shift / residual / ReLU and the
autograd path, fp32 and bf16) against float64 torch -- tools/fuzz_conv.py with a fixed seed; the long
sweeps (900 cases) are run by hand after kernel changes."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_conv_shapes_match_float64():
    spec = importlib.util.spec_from_file_location('brcnn_fuzz_conv', os.path.join(ROOT, 'tools', 'fuzz_conv.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(60, seed=11) == 0
