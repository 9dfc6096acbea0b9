"""falor: feature-aware low-rank decomposition -- per-layer rank bisection on NSR / KL (MI355X path).

Public surface, same names as ``ptdeco.falor``:

    decompose_in_place(*, module, device, data_iterator, ...) -> decompose_config
    is_decomposeable_module(module) -> bool
"""

from . import decomposition as _impl

decompose_in_place = _impl.decompose_in_place
is_decomposeable_module = _impl.is_decomposeable_module

__all__ = ["decompose_in_place", "is_decomposeable_module"]
