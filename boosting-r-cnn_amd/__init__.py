"""MI355X-native Boosting R-CNN hot path (gfx950 HIP kernels behind a C ABI + a host-side
mirror of the reference's registry/config/operator interface).

The directory name `boosting-r-cnn_amd` is not a Python identifier; import it as `brcnn`
(`brcnn.py` at the repo root aliases this package).
"""
__version__ = '0.1.0'

from .registry import Registry, build_from_cfg  # noqa: F401
from .config import Config, ConfigDict  # noqa: F401
