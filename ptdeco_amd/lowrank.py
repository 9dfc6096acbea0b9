"""The decomposed layer: a rank-r pair whose forward is two back-to-back GEMMs on the
matrix cores (ptd_lowrank_forward).

Both classes ARE ``torch.nn.Sequential`` containers of two ``nn.Linear`` / 1x1
``nn.Conv2d`` children, exactly what the reference builds (dwain.py:69-85, 121-144;
falor.py:79-95, 131-153), so ``get_module_config``, ``state_dict`` keys
('0.weight' [r, n_in], '1.weight' [n_out, r], '1.bias') and ``load_state_dict`` are
unchanged.  Only ``forward`` differs: on a ROCm device it calls the HIP kernels, also when
autograd is recording (a user ``finetune_fn``, dwain.py:779-786): ``_LowRankFunction`` forms
dx, dA, dB with the same strided GEMM entry (``ptd_gemm``), so fine-tuning trains the fused
pair.
"""

from __future__ import annotations

import logging

import torch

from . import ops

logger = logging.getLogger(__name__)

_HIP_DTYPES = (torch.float32, torch.bfloat16)
_warned: set = set()


def warn_once(key: str, message: str) -> None:
    """One WARNING per process and reason whenever something leaves the HIP kernels."""
    if key not in _warned:
        _warned.add(key)
        logger.warning(message)


def _use_hip(x: torch.Tensor, w: torch.Tensor, who: str) -> bool:
    """The pair runs on the HIP kernels for f32 / bf16 tensors on a ROCm device.  Anything else (a CPU
    copy of the model, fp16, an autocast dtype mismatch) is evaluated by the container's two torch
    layers -- the module IS an nn.Sequential -- and says so once, at WARNING."""
    if x.is_cuda and x.dtype in _HIP_DTYPES and x.dtype == w.dtype:
        return True
    why = "a CPU tensor" if not x.is_cuda else f"input dtype {x.dtype} with weight dtype {w.dtype}"
    warn_once(f"{who}:{why}", f"ptdeco_amd.{who}: {why} is not served by the HIP low-rank kernels (f32 / bf16 on a "
                              "ROCm device); running the two torch layers of the pair instead")
    return False


class _LowRankFunction(torch.autograd.Function):
    """y = (x A^T) B^T + bias with x [T, n_i], A [r, n_i], B [n_o, r]; every product on ptd_gemm.

    backward:  dh = dy B,  dx = dh A,  dB = dy^T h,  dA = dh^T x,  dbias = sum_t dy   (h = x A^T is
    recomputed: one [T, r] product instead of keeping it alive between forward and backward)."""

    @staticmethod
    def forward(ctx, x2d, a, b, bias):
        ctx.save_for_backward(x2d, a, b)
        ctx.has_bias = bias is not None
        return ops.lowrank_forward(x2d, a, b, bias)

    @staticmethod
    def backward(ctx, dy):
        x2d, a, b = ctx.saved_tensors
        dy = dy.contiguous()
        need_x, need_a, need_b, need_bias = ctx.needs_input_grad
        dx = da = db = dbias = None
        dh = ops.matmul(dy, b) if (need_x or need_a) else None
        if need_x:
            dx = ops.matmul(dh, a)
        if need_a:
            da = ops.matmul(dh.T, x2d)
        if need_b:
            db = ops.matmul(dy.T, ops.matmul(x2d, a.T))
        if need_bias and ctx.has_bias:
            dbias = dy.sum(dim=0)
        return dx, da, db, dbias


def _pair_forward(x2d: torch.Tensor, a: torch.Tensor, b: torch.Tensor, bias) -> torch.Tensor:
    if torch.is_grad_enabled() and (x2d.requires_grad or a.requires_grad or b.requires_grad
                                    or (bias is not None and bias.requires_grad)):
        return _LowRankFunction.apply(x2d, a, b, bias)
    return ops.lowrank_forward(x2d, a, b, bias)


class LowRankLinear(torch.nn.Sequential):
    def forward(self, x: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        first, second = self[0], self[1]
        if not _use_hip(x, first.weight, "LowRankLinear"):
            return second(first(x))
        x2d = x.reshape(-1, first.in_features)
        y = _pair_forward(x2d, first.weight, second.weight, second.bias)
        return y.reshape(*x.shape[:-1], second.out_features)


class LowRankConv1x1(torch.nn.Sequential):
    def forward(self, x: torch.Tensor) -> torch.Tensor:  # type: ignore[override]
        first, second = self[0], self[1]
        if not _use_hip(x, first.weight, "LowRankConv1x1"):
            return second(first(x))
        b, c, h, w = x.shape
        wa, wb, bias = first.weight[:, :, 0, 0], second.weight[:, :, 0, 0], second.bias
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or wa.requires_grad or wb.requires_grad
                                                  or (bias is not None and bias.requires_grad))
        if x.is_contiguous() and not needs_grad and h * w > 1:
            # NCHW as it lies: per image x_b is a [C, H W] matrix with the pixels contiguous, y_b = B (A x_b) + bias
            # goes straight into NCHW -- no NHWC copy (the reference's permute, dwain.py:116)
            return ops.lowrank_forward_nchw(x, wa, wb, bias)
        rows = x.permute(0, 2, 3, 1).reshape(-1, c)  # NHWC rows: a view for channels_last inputs
        y = _pair_forward(rows, wa, wb, bias)
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
