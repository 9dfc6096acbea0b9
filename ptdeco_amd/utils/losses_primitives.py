"""Rank-selection metrics (interface of ptdeco.utils.losses_primitives, reference
src/ptdeco/utils/losses_primitives.py:3-7) computed by the HIP reductions
ptd_nsr / ptd_sym_kl: one pass over the model outputs, f64 accumulation."""

from __future__ import annotations

import torch

from .. import ops

__all__ = [
    "calc_per_channel_noise_to_signal_ratio",
    "calc_kl_divergence",
    "calc_kl_loss",
]


def _channels_last_view(t: torch.Tensor, non_channel_dim: tuple[int, ...]) -> tuple[torch.Tensor, int]:
    """Move the reduced dims to the front; returns (tensor, number of channels)."""
    nd = t.dim()
    red = sorted(d % nd for d in non_channel_dim)
    keep = [d for d in range(nd) if d not in red]
    if red != list(range(len(red))):  # reduced dims are not the leading ones: permute
        t = t.permute(*red, *keep)
    chan = 1
    for d in keep:
        chan *= t.shape[len(red) + keep.index(d)] if red != list(range(len(red))) else t.shape[d]
    return t, chan


def calc_per_channel_noise_to_signal_ratio(
    x: torch.Tensor,
    y: torch.Tensor,
    non_channel_dim: tuple[int, ...] = (0, 2, 3),
    epsilon: float = 1e-3,
    mode: str = "mean",
) -> torch.Tensor:
    """mean_c[ mean((x-y)^2) / (std(y)^2 + eps) ] with the unbiased std
    (losses_primitives.py:10-22; ``mode`` is ignored there too).  Returns a 0-d f64 tensor."""
    xv, chan = _channels_last_view(x, tuple(non_channel_dim))
    yv, _ = _channels_last_view(y, tuple(non_channel_dim))
    return ops.nsr(xv, yv, chan, epsilon)


def calc_kl_divergence(q_logits: torch.Tensor, p_logits: torch.Tensor) -> torch.Tensor:
    """Per-row KL(p || q) = sum_c p log(p / q) over softmax(dim=-1) (losses_primitives.py:48-54).
    Logits [batch, classes]; returns a [batch] f64 tensor (one wave per row, f64 accumulation)."""
    if q_logits.dim() != 2 or q_logits.shape != p_logits.shape:
        raise ValueError("calc_kl_divergence expects two logit tensors of the same shape [batch, classes]")
    return ops.kl_rows(q_logits, p_logits)


def calc_kl_loss(student_logits: torch.Tensor, teacher_logits: torch.Tensor) -> torch.Tensor:
    """mean(max(KL(t||s), KL(s||t))) (losses_primitives.py:57-63).  Returns a 0-d f64 tensor."""
    if student_logits.dim() != 2:
        raise ValueError("calc_kl_loss expects logits of shape [batch, classes]")
    return ops.sym_kl(student_logits, teacher_logits)
