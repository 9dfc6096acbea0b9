from . import common, losses_primitives, modconfig
from .common import *  # noqa: F401,F403
from .losses_primitives import *  # noqa: F401,F403
from .modconfig import *  # noqa: F401,F403

__all__ = common.__all__ + losses_primitives.__all__ + modconfig.__all__
