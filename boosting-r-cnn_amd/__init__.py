"""MI355X-native Boosting R-CNN hot path (gfx950 HIP kernels behind a C ABI + a host-side
mirror of the reference's registry/config/operator interface).

The directory name `boosting-r-cnn_amd` is not a Python identifier; import it as `brcnn`
(`brcnn.py` at the repo root aliases this package).
"""
__version__ = '0.1.0'

import os as _os
# more hardware queues than HIP's default four, so that the weight-gradient stream, the gradient reducer's
# communication stream and RCCL's own streams do not share one with the main stream (effective when this package is
# imported before the first device call; bench.py and the tools set it themselves)
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

from .registry import Registry, build_from_cfg  # noqa: F401
from .config import Config, ConfigDict  # noqa: F401

# registering the hot-path classes under the reference's names
from . import core, losses, backbones, necks, dense_heads, roi_heads, detectors  # noqa: E402,F401
from .registry import (MODELS, BACKBONES, NECKS, HEADS, LOSSES, DETECTORS, ROI_EXTRACTORS,  # noqa: E402,F401
                       BBOX_ASSIGNERS, BBOX_SAMPLERS, BBOX_CODERS, PRIOR_GENERATORS,
                       IOU_CALCULATORS, build_detector, build_backbone, build_neck, build_head,
                       build_loss)

# data side and drivers: registries PIPELINES / DATASETS, apis
from . import pipelines, datasets, evaluation, apis  # noqa: E402,F401
from .pipelines import PIPELINES, Compose  # noqa: E402,F401
from .datasets import DATASETS, build_dataset, build_dataloader  # noqa: E402,F401
