from .decomposition import *  # noqa: F401,F403
from .decomposition import __all__  # noqa: F401
