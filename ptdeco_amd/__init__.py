"""MI355X-native implementation of ptdeco's covariance / eigenvector / rank-search hot
path behind ptdeco's own API (``dwain.decompose_in_place``, ``falor.decompose_in_place``,
``utils.apply_decompose_config_in_place``).  All arithmetic of the path runs in
hand-written gfx950 kernels (libptdeco_hip.so, C ABI in include/ptdeco_hip.h); there is
no CPU fallback."""

from . import dwain  # noqa: F401
from . import falor  # noqa: F401
from . import utils  # noqa: F401
from .lowrank import LowRankConv1x1, LowRankLinear  # noqa: F401

__version__ = "0.1.0"
