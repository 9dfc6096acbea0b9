"""TEST-ONLY stand-in for ptdeco_amd.ops on a box without a GPU.

The product has a single backend (the HIP library) and refuses CPU devices.  To test the
HOST logic of the drivers here -- candidate schedules, decisions, data-consumption order,
config format, multi-rank sharding over gloo -- the tests swap the functions of
``ptdeco_amd.ops`` for the CPU oracle's arithmetic via monkeypatch.  Nothing under
``ptdeco_amd/`` imports this file.
"""

from __future__ import annotations

import contextlib

import torch

import ptdeco_oracle as orc


def syrk_accumulate(E, y, scale):
    upd = torch.einsum("bp,bq->pq", y, y).to(E.dtype) * scale if scale != 1.0 / y.shape[0] else \
        (torch.einsum("bp,bq->pq", y, y) / y.shape[0]).to(E.dtype)
    E += torch.tril(upd)


def colsum_accumulate(ey, y, scale):
    ey += (y.sum(dim=0) * scale).to(ey.dtype) if scale != 1.0 / y.shape[0] else y.mean(dim=0).to(ey.dtype)


def cov_finalize(E, steps, damp_factor, ey=None):
    full = (torch.tril(E) + torch.tril(E, -1).T).double() / steps
    if ey is not None:
        m = ey.double() / steps
        full = full - torch.outer(m, m)
    idx = torch.arange(full.shape[0])
    full[idx, idx] += damp_factor * torch.diag(full).mean()
    return full


def eigh(A, k=None, all_values=True):
    w, v = torch.linalg.eigh(A)
    return (w, v) if k is None else (w, v[:, v.shape[1] - max(1, min(int(k), v.shape[1])):])


def eigh_factored(W, Ex, k):
    """Same contract as ops.eigh_factored, through the same algebra the kernel uses
    (G = W^T W = L L^T, B = L^T Ex L, u = W L^-T s) in f64 LAPACK."""
    w = W.double()
    ell = torch.linalg.cholesky(w.T @ w)
    b = ell.T @ Ex @ ell
    lam, s = torch.linalg.eigh(0.5 * (b + b.T))
    k = max(1, min(int(k), s.shape[1]))
    t = torch.linalg.solve_triangular(ell.T, s[:, s.shape[1] - k:], upper=True)
    return lam[lam.shape[0] - k:], w @ t


class _Factored:
    """ops.FactoredProblem in f64 LAPACK (same algebra: matrix = L^T Ex L, finish: u = W L^-T s)."""

    def __init__(self, W, Ex, k):
        self.w = W.double()
        self.ell = torch.linalg.cholesky(self.w.T @ self.w)
        b = self.ell.T @ Ex @ self.ell
        self.matrix = 0.5 * (b + b.T)
        self.k = max(1, min(int(k), self.matrix.shape[0]))

    def finish(self, evals, S):
        t = torch.linalg.solve_triangular(self.ell.T, S, upper=True)
        return evals[evals.shape[0] - self.k:], self.w @ t


def eigh_factored_prepare(W, Ex, k):
    return _Factored(W, Ex, k)


def eigh_batched(mats, k=None, all_values=False, direct=False):
    return [eigh(a, k, all_values) for a in mats]


def matmul(a, b, bias=None, alpha=1.0, out_dtype=None):
    c = (a @ b) * alpha if alpha != 1.0 else a @ b
    if bias is not None:
        c = c + bias
    return c.to(out_dtype) if out_dtype is not None else c


def lowrank_forward(x2d, A, B, bias):
    y = (x2d @ A.T) @ B.T
    return y + bias if bias is not None else y


def nsr(x, y, channels, eps=1e-3):
    xr, yr = x.reshape(-1, channels), y.reshape(-1, channels)
    return orc.nsr(x=xr.double(), y=yr.double(), non_channel_dim=(0,), eps=eps)


def sym_kl(s, t):
    return orc.kl_loss(s.double(), t.double())


@contextlib.contextmanager
def installed(monkeypatch):
    """Route ptdeco_amd.ops to the CPU oracle arithmetic and let the drivers accept CPU devices."""
    import ptdeco_amd
    from ptdeco_amd import _engine, ops

    for name in ("syrk_accumulate", "colsum_accumulate", "cov_finalize", "eigh", "eigh_factored", "eigh_factored_prepare",
                 "eigh_batched", "matmul", "lowrank_forward", "nsr", "sym_kl"):
        monkeypatch.setattr(ops, name, globals()[name])
    monkeypatch.setattr(_engine, "require_device", lambda d: torch.device(d))
    yield ptdeco_amd
