"""dwain: data-driven weight analysis -- per-layer rank search on perplexity / NSR (MI355X path).

Public surface, same names as ``ptdeco.dwain``:

    decompose_in_place(*, module, device, data_iterator, loss_fn, ...) -> decompose_config
    is_decomposeable_module(module) -> bool
"""

from . import decomposition as _impl

decompose_in_place = _impl.decompose_in_place
is_decomposeable_module = _impl.is_decomposeable_module

__all__ = ["decompose_in_place", "is_decomposeable_module"]
