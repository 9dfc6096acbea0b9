"""The decomposed layer: a rank-r pair whose forward is two back-to-back GEMMs on the
matrix cores (ptd_lowrank_forward).

Both classes ARE ``torch.nn.Sequential`` containers of two ``nn.Linear`` / 1x1
``nn.Conv2d`` children, exactly what the reference builds (dwain.py:69-85, 121-144;
falor.py:79-95, 131-153), so ``get_module_config``, ``state_dict`` keys
('0.weight' [r, n_in], '1.weight' [n_out, r], '1.bias') and ``load_state_dict`` are
unchanged.  Only ``forward`` differs: under ``torch.no_grad()`` on a ROCm device it
calls the HIP kernel; when autograd is recording (a user ``finetune_fn``) it runs the
two children so gradients flow -- backward kernels are a later scope row.
"""

from __future__ import annotations

import torch

from . import ops

_HIP_DTYPES = (torch.float32, torch.bfloat16)


def _use_hip(x: torch.Tensor, w: torch.Tensor) -> bool:
    if not x.is_cuda or x.dtype not in _HIP_DTYPES or x.dtype != w.dtype:
        return False
    return not (torch.is_grad_enabled() and (x.requires_grad or w.requires_grad))


class LowRankLinear(torch.nn.Sequential):
    def forward(self, x: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        first, second = self[0], self[1]
        if not _use_hip(x, first.weight):
            return second(first(x))
        y = ops.lowrank_forward(x.reshape(-1, first.in_features), first.weight, second.weight, second.bias)
        return y.reshape(*x.shape[:-1], second.out_features)


class LowRankConv1x1(torch.nn.Sequential):
    def forward(self, x: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        first, second = self[0], self[1]
        if not _use_hip(x, first.weight):
            return second(first(x))
        b, c, h, w = x.shape
        rows = x.permute(0, 2, 3, 1).reshape(-1, c)  # NHWC rows; free for channels_last inputs
        y = ops.lowrank_forward(rows, first.weight[:, :, 0, 0], second.weight[:, :, 0, 0], second.bias)
        return y.reshape(b, h, w, second.out_channels).permute(0, 3, 1, 2)


def _is_plain_1x1(m: torch.nn.Module) -> bool:
    return (isinstance(m, torch.nn.Conv2d) and tuple(m.kernel_size) == (1, 1) and m.groups == 1
            and tuple(m.stride) == (1, 1) and tuple(m.padding) in ((0, 0),) and tuple(m.dilation) == (1, 1))


def fuse_pair(seq: torch.nn.Sequential) -> torch.nn.Sequential:
    """Re-class a two-child Sequential describing a rank-r pair; anything else is returned as is."""
    kids = list(seq.children())
    if len(kids) != 2 or list(dict(seq.named_children()).keys()) != ["0", "1"]:
        return seq
    a, b = kids
    if isinstance(a, torch.nn.Linear) and isinstance(b, torch.nn.Linear) and a.bias is None \
            and a.out_features == b.in_features:
        seq.__class__ = LowRankLinear
    elif _is_plain_1x1(a) and _is_plain_1x1(b) and a.bias is None and a.out_channels == b.in_channels:
        seq.__class__ = LowRankConv1x1
    return seq
