"""Tensor-level front end of the HIP kernels (device memory and streams come from
PyTorch-ROCm; all arithmetic happens in libptdeco_hip.so)."""

from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _hip

_DT = {torch.float32: _hip.F32, torch.float64: _hip.F64, torch.bfloat16: _hip.BF16}


def _code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}") from None


def _dev(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("ptdeco_amd ops need tensors on a ROCm device (no CPU fallback)")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _rows2d(t: torch.Tensor) -> torch.Tensor:
    """Row-major 2-D view with unit inner stride (copy only if needed)."""
    assert t.dim() == 2
    return t if t.stride(1) == 1 and t.stride(0) >= max(1, t.shape[1]) else t.contiguous()


def _strides(t: torch.Tensor) -> tuple[torch.Tensor, int, int]:
    """2-D operand with exactly one unit stride -> (tensor, stride0, stride1)."""
    assert t.dim() == 2
    s0, s1 = t.stride()
    if t.shape[0] == 1 or t.shape[1] == 1 or (s0 == 1) == (s1 == 1):
        t = t.contiguous()
        s0, s1 = t.stride()
        if s1 != 1:  # shape (k, 1) contiguous has strides (1, 1)
            s0, s1 = t.shape[1], 1
    return t, s0, s1


def syrk_accumulate(E: torch.Tensor, y: torch.Tensor, scale: float) -> None:
    """E[i, j] += scale * (y^T y)[i, j] for i >= j  (lower triangle only)."""
    _dev(E, y)
    y = _rows2d(y)
    n = y.shape[1]
    assert E.shape == (n, n) and E.stride(1) == 1
    with torch.cuda.device(E.device):
        rc = _hip.load().ptd_syrk_accumulate(y.data_ptr(), y.shape[0], n, y.stride(0), _code(y), E.data_ptr(),
                                             E.stride(0), _code(E), float(scale), _stream(E))
    _hip.check(rc, "ptd_syrk_accumulate")


def syrk_accumulate_multi(E: torch.Tensor, ys, scale: float) -> None:
    """E[i, j] += sum_s scale * (ys[s]^T ys[s])[i, j] for i >= j: the covariance sums of several calibration steps in
    one pass over the accumulator (ptd_syrk_accumulate_multi).  The matrices share shape, dtype and row pitch; those
    that do not are added one by one."""
    ys = [_rows2d(y) for y in ys]
    if not ys:
        return
    _dev(E, *ys)
    n = ys[0].shape[1]
    assert E.shape == (n, n) and E.stride(1) == 1
    first = ys[0]
    same = all(y.shape == first.shape and y.dtype == first.dtype and y.stride() == first.stride() for y in ys)
    if not same or len(ys) == 1:
        for y in ys:
            syrk_accumulate(E, y, scale)
        return
    import ctypes

    ptrs = (ctypes.c_void_p * len(ys))(*[y.data_ptr() for y in ys])
    with torch.cuda.device(E.device):
        rc = _hip.load().ptd_syrk_accumulate_multi(ptrs, len(ys), first.shape[0], n, first.stride(0), _code(first),
                                                   E.data_ptr(), E.stride(0), _code(E), float(scale), _stream(E))
    _hip.check(rc, "ptd_syrk_accumulate_multi")


def colsum_accumulate(ey: torch.Tensor, y: torch.Tensor, scale: float) -> None:
    _dev(ey, y)
    y = _rows2d(y)
    assert ey.shape == (y.shape[1],) and ey.is_contiguous()
    with torch.cuda.device(ey.device):
        rc = _hip.load().ptd_colsum_accumulate(y.data_ptr(), y.shape[0], y.shape[1], y.stride(0), _code(y),
                                               ey.data_ptr(), _code(ey), float(scale), _stream(ey))
    _hip.check(rc, "ptd_colsum_accumulate")


def cov_finalize(E: torch.Tensor, steps: int, damp_factor: float, ey: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f64 full symmetric  sym(E)/steps - mean term + damping  (E holds the lower triangle)."""
    _dev(E, ey)
    n = E.shape[0]
    lib = _hip.load()
    C = torch.empty((n, n), dtype=torch.float64, device=E.device)
    ws = torch.empty(lib.ptd_cov_finalize_workspace_bytes(n), dtype=torch.uint8, device=E.device)
    with torch.cuda.device(E.device):
        rc = lib.ptd_cov_finalize(E.data_ptr(), E.stride(0), _code(E), _ptr(ey), _code(ey) if ey is not None else 0,
                                  n, float(steps), float(damp_factor), C.data_ptr(), n, ws.data_ptr(), ws.numel(),
                                  _stream(E))
    _hip.check(rc, "ptd_cov_finalize")
    return C


# When a list, every eigh call appends a dict of per-phase device timings (bench.py's roofline lines).
EIGH_PROFILE: Optional[list] = None


def eigh(A: torch.Tensor, k: Optional[int] = None, all_values: bool = True) -> tuple[torch.Tensor, torch.Tensor]:
    """(eigenvalues [n] ascending, eigenvectors in columns) of a symmetric PSD f64 (or f32: f32 results, f64 arithmetic) matrix.

    With ``k`` only the eigenvectors of the k largest eigenvalues are formed: the second result is
    [n, k] and its column c belongs to eigenvalue n - k + c, so ``v[:, v.shape[1] - r:]`` is the
    top-r block for every r <= k exactly as with the full matrix.  ``all_values=False`` lets the
    solver skip the eigenvalues below the k + 1 largest (those entries of the first result are NaN
    then, unless the Jacobi route ran); the decomposition drivers never read eigenvalues."""
    _dev(A)
    assert A.dtype in (torch.float64, torch.float32) and A.dim() == 2 and A.shape[0] == A.shape[1] and A.stride(1) == 1
    n = A.shape[0]
    k = n if k is None else max(1, min(int(k), n))
    lib = _hip.load()
    if A.dtype == torch.float32:
        # f32 in, f32 out, f64 arithmetic in between (ptd_eigh_topk_f32: the face the reference's f32 eigh call takes)
        w = torch.empty(n, dtype=torch.float32, device=A.device)
        v = torch.empty((n, k), dtype=torch.float32, device=A.device)
        ws = torch.empty(lib.ptd_eigh_f32_workspace_bytes(n, k), dtype=torch.uint8, device=A.device)
        with torch.cuda.device(A.device):
            rc = lib.ptd_eigh_topk_f32(A.data_ptr(), A.stride(0), n, k, int(all_values), w.data_ptr(), v.data_ptr(), k,
                                       ws.data_ptr(), ws.numel(), None, _stream(A))
        _hip.check(rc, "ptd_eigh_topk_f32")
        return w, v
    w = torch.empty(n, dtype=torch.float64, device=A.device)
    v = torch.empty((n, k), dtype=torch.float64, device=A.device)
    ws = torch.empty(lib.ptd_eigh_workspace_bytes(n), dtype=torch.uint8, device=A.device)
    with torch.cuda.device(A.device):
        if EIGH_PROFILE is None:
            rc = lib.ptd_eigh_topk(A.data_ptr(), A.stride(0), n, k, int(all_values), w.data_ptr(), v.data_ptr(), k,
                                   ws.data_ptr(), ws.numel(), None, _stream(A))
        else:
            st = _hip.EighStats()
            rc = lib.ptd_eigh_profiled(A.data_ptr(), A.stride(0), n, k, int(all_values), w.data_ptr(), v.data_ptr(), k,
                                       ws.data_ptr(), ws.numel(), ctypes.byref(st), _stream(A))
            EIGH_PROFILE.append({"n": n, "k": k, "method": st.method, "sweeps": st.sweeps,
                                 "launches": list(st.launches), "ms": list(st.ms), "total_ms": st.total_ms,
                                 "work": list(st.work)})
    _hip.check(rc, "ptd_eigh")
    return w, v


def eigh_factored(W: torch.Tensor, Ex: torch.Tensor, k: int) -> Optional[tuple[torch.Tensor, torch.Tensor]]:
    """Top-k eigenpairs (eigenvalues [k] ascending, eigenvectors [n_o, k] f64) of W Ex W^T for
    W [n_o, n_i] with n_o > n_i and Ex [n_i, n_i] f64 symmetric.  Returns None when W^T W is not
    numerically positive definite (the caller then works on the n_o x n_o matrix)."""
    _dev(W, Ex)
    assert W.dim() == 2 and Ex.dtype == torch.float64 and Ex.shape == (W.shape[1], W.shape[1])
    W, Ex = _rows2d(W), _rows2d(Ex)
    n_o, n_i = W.shape
    k = max(1, min(int(k), n_i))
    lib = _hip.load()
    w = torch.empty(k, dtype=torch.float64, device=W.device)
    u = torch.empty((n_o, k), dtype=torch.float64, device=W.device)
    ws = torch.empty(lib.ptd_eigh_factored_workspace_bytes(n_o, n_i, k), dtype=torch.uint8, device=W.device)
    with torch.cuda.device(W.device):
        rc = lib.ptd_eigh_factored(W.data_ptr(), W.stride(0), _code(W), n_o, n_i, Ex.data_ptr(), Ex.stride(0), k,
                                   w.data_ptr(), u.data_ptr(), k, ws.data_ptr(), ws.numel(), _stream(W))
    if rc == -2:  # PTD_ERR_UNSUPPORTED
        return None
    _hip.check(rc, "ptd_eigh_factored")
    return w, u


def eigh_batched(mats, k: Optional[int] = None, all_values: bool = False, direct: bool = False) -> list:
    """ops.eigh for several f64 matrices of ONE order in one ptd_eigh_topk_batched call: a list of (eigenvalues [n],
    eigenvectors [n, k]) in the order of `mats`.  The loop of torch.linalg.eigh calls of dwain's precompute pass
    (dwain.py:580-633, :162) -- the matrices advance through the tridiagonalisation in lockstep, every launch serves all of
    them (one stream, one host thread).  One matrix, or a request the filtered route serves: solved one by one by the
    library, exactly as ops.eigh would -- unless `direct` (PTD_EIGH_FLAG_DIRECT): then the matrices take the direct
    reduction together also where the filtered route applies."""
    mats = list(mats)
    assert mats, "eigh_batched: no matrix"
    a0 = mats[0]
    n = a0.shape[0]
    for a in mats:
        _dev(a)
        assert a.dtype == torch.float64 and a.dim() == 2 and a.shape == (n, n) and a.stride(1) == 1 \
            and a.stride(0) == a0.stride(0) and a.device == a0.device
    k = n if k is None else max(1, min(int(k), n))
    lib = _hip.load()
    count = len(mats)
    ws_out = [torch.empty(n, dtype=torch.float64, device=a0.device) for _ in mats]
    vs = [torch.empty((n, k), dtype=torch.float64, device=a0.device) for _ in mats]
    ws = torch.empty(lib.ptd_eigh_batched_workspace_bytes(n, k, count), dtype=torch.uint8, device=a0.device)
    arr = ctypes.c_void_p * count
    a_ptrs = arr(*[a.data_ptr() for a in mats])
    w_ptrs = arr(*[w.data_ptr() for w in ws_out])
    v_ptrs = arr(*[v.data_ptr() for v in vs])
    st = _hip.EighStats() if EIGH_PROFILE is not None else None
    with torch.cuda.device(a0.device):
        rc = lib.ptd_eigh_topk_batched(a_ptrs, a0.stride(0), count, n, k, int(all_values) | (2 if direct else 0), w_ptrs, v_ptrs, k,
                                       ws.data_ptr(), ws.numel(), ctypes.byref(st) if st is not None else None,
                                       _stream(a0))
    _hip.check(rc, "ptd_eigh_topk_batched")
    if st is not None:
        EIGH_PROFILE.append({"n": n, "k": k, "count": count, "method": st.method, "sweeps": st.sweeps,
                             "launches": list(st.launches), "ms": list(st.ms), "total_ms": st.total_ms,
                             "work": list(st.work)})
    return list(zip(ws_out, vs))


class FactoredProblem:
    """The n_i-sized eigenproblem behind the top-k eigenvectors of W Ex W^T (ops.eigh_factored), exposed so that the
    caller can solve `matrix` (B = L^T Ex L, [np, np] f64, a view into the workspace this object owns) together with
    other matrices of the same order -- ops.eigh_batched -- and hand the eigenpairs back to `finish`."""

    def __init__(self, W: torch.Tensor, ws: torch.Tensor, matrix: torch.Tensor, k: int):
        self.n_o, self.n_i = W.shape
        self.k, self.ws, self.matrix, self.device = k, ws, matrix, W.device

    def finish(self, evals: torch.Tensor, S: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """evals [np] ascending, S [np, k] eigenvectors of `matrix` -> (eigenvalues [k], U [n_o, k]) of W Ex W^T."""
        lib = _hip.load()
        assert S.dtype == torch.float64 and S.shape == (self.matrix.shape[0], self.k) and S.stride(1) == 1
        w = torch.empty(self.k, dtype=torch.float64, device=self.device)
        u = torch.empty((self.n_o, self.k), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            rc = lib.ptd_eigh_factored_finish(self.n_o, self.n_i, self.k, evals.data_ptr(), S.data_ptr(), S.stride(0),
                                              w.data_ptr(), u.data_ptr(), self.k, self.ws.data_ptr(), self.ws.numel(),
                                              torch.cuda.current_stream(self.device).cuda_stream)
        _hip.check(rc, "ptd_eigh_factored_finish")
        return w, u


def eigh_factored_prepare(W: torch.Tensor, Ex: torch.Tensor, k: int) -> Optional[FactoredProblem]:
    """First half of ops.eigh_factored (G = W^T W = L L^T, B = L^T Ex L).  None when W^T W is not numerically positive
    definite (the caller then works on the n_o x n_o matrix)."""
    _dev(W, Ex)
    assert W.dim() == 2 and Ex.dtype == torch.float64 and Ex.shape == (W.shape[1], W.shape[1])
    W, Ex = _rows2d(W), _rows2d(Ex)
    n_o, n_i = W.shape
    k = max(1, min(int(k), n_i))
    lib = _hip.load()
    ws = torch.empty(lib.ptd_eigh_factored_workspace_bytes(n_o, n_i, k), dtype=torch.uint8, device=W.device)
    b_ptr, np_ = ctypes.c_void_p(0), ctypes.c_int64(0)
    with torch.cuda.device(W.device):
        rc = lib.ptd_eigh_factored_prepare(W.data_ptr(), W.stride(0), _code(W), n_o, n_i, Ex.data_ptr(), Ex.stride(0), k,
                                           ws.data_ptr(), ws.numel(), ctypes.byref(b_ptr), ctypes.byref(np_), _stream(W))
    if rc == -2:  # PTD_ERR_UNSUPPORTED
        return None
    _hip.check(rc, "ptd_eigh_factored_prepare")
    npad = int(np_.value)
    off = int(b_ptr.value) - ws.data_ptr()
    assert off >= 0 and off % 8 == 0 and off + npad * npad * 8 <= ws.numel()
    matrix = ws[off:off + npad * npad * 8].view(torch.float64).view(npad, npad)
    return FactoredProblem(W, ws, matrix, k)


def chol_inverse(G: torch.Tensor) -> Optional[torch.Tensor]:
    """Diagnostic: W = L^-T (upper triangular, f64) with G = L L^T for a symmetric positive definite [m, m] f64
    matrix, m a multiple of 64 (the Cholesky sweep of the filtered eigensolver's orthonormalisation passes).
    None when a pivot is not positive.  G is not modified (the kernel works on a copy)."""
    _dev(G)
    assert G.dtype == torch.float64 and G.dim() == 2 and G.shape[0] == G.shape[1]
    m = G.shape[0]
    lib = _hip.load()
    work = G.contiguous().clone()
    W = torch.empty((m, m), dtype=torch.float64, device=G.device)
    ws = torch.empty(lib.ptd_chol_inverse_workspace_bytes(m), dtype=torch.uint8, device=G.device)
    with torch.cuda.device(G.device):
        rc = lib.ptd_chol_inverse(work.data_ptr(), m, W.data_ptr(), ws.data_ptr(), ws.numel(), _stream(G))
    if rc == -2:  # PTD_ERR_UNSUPPORTED
        return None
    _hip.check(rc, "ptd_chol_inverse")
    return W


def tridiagonalize(A: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Diagnostic: (d, e, eigenvalues of T) of the Householder tridiagonalisation of symmetric f64 A."""
    _dev(A)
    assert A.dtype == torch.float64 and A.dim() == 2 and A.shape[0] == A.shape[1] and A.stride(1) == 1
    n = A.shape[0]
    lib = _hip.load()
    d, e, w = (torch.empty(n, dtype=torch.float64, device=A.device) for _ in range(3))
    ws = torch.empty(lib.ptd_tridiagonalize_workspace_bytes(n), dtype=torch.uint8, device=A.device)
    with torch.cuda.device(A.device):
        rc = lib.ptd_tridiagonalize(A.data_ptr(), A.stride(0), n, d.data_ptr(), e.data_ptr(), w.data_ptr(),
                                    ws.data_ptr(), ws.numel(), _stream(A))
    _hip.check(rc, "ptd_tridiagonalize")
    return d, e, w


_GEMM_WS = os.environ.get("PTD_GEMM_WS", "1") != "0"


def matmul(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, alpha: float = 1.0,
           out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """a [M, K] @ b [K, N] (+ bias[N]); a and b may be transposed views (no copies are made
    for row-major or column-major operands)."""
    _dev(a, b, bias)
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] == b.shape[0] and a.dtype == b.dtype
    a, sam, sak = _strides(a)
    b, sbk, sbn = _strides(b)
    M, K = a.shape
    N = b.shape[1]
    out_dtype = out_dtype or a.dtype
    c = torch.empty((M, N), dtype=out_dtype, device=a.device)
    if bias is not None:
        bias = bias.to(a.dtype).contiguous()
    lib = _hip.load()
    # few output tiles (a ViT batch's 1576 rows against 768 columns): the K range is split through a workspace
    ws_bytes = lib.ptd_gemm_workspace_bytes(M, N, K, _code(a), _DT[out_dtype]) if _GEMM_WS else 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device) if ws_bytes else None
    with torch.cuda.device(a.device):
        rc = lib.ptd_gemm_ws(a.data_ptr(), sam, sak, b.data_ptr(), sbk, sbn, c.data_ptr(), N, M, N, K, _code(a),
                             _DT[out_dtype], float(alpha), _ptr(bias), _ptr(ws), ws_bytes, _stream(a))
    _hip.check(rc, "ptd_gemm")
    return c


def lowrank_forward(x2d: torch.Tensor, A: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """(x2d @ A^T) @ B^T + bias with A [r, n_i], B [n_o, r]."""
    _dev(x2d, A, B, bias)
    x2d, A, B = _rows2d(x2d), _rows2d(A), _rows2d(B)
    T, n_i = x2d.shape
    r, n_o = A.shape[0], B.shape[0]
    assert A.shape[1] == n_i and B.shape[1] == r and x2d.dtype == A.dtype == B.dtype
    y = torch.empty((T, n_o), dtype=x2d.dtype, device=x2d.device)
    lib = _hip.load()
    ws_bytes = lib.ptd_lowrank_forward_workspace_bytes(T, n_i, r, _code(x2d))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x2d.device)
    if bias is not None:
        bias = bias.to(x2d.dtype).contiguous()
    with torch.cuda.device(x2d.device):
        rc = lib.ptd_lowrank_forward(x2d.data_ptr(), x2d.stride(0), T, n_i, A.data_ptr(), A.stride(0), r,
                                     B.data_ptr(), B.stride(0), n_o, _ptr(bias), y.data_ptr(), n_o,
                                     ws.data_ptr(), ws_bytes, _code(x2d), _stream(x2d))
    _hip.check(rc, "ptd_lowrank_forward")
    return y


def lowrank_forward_nchw(x: torch.Tensor, A: torch.Tensor, B: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """The rank-r 1x1-convolution pair on a contiguous NCHW input, no layout copy: per image
    y_b = B (A x_b) + bias[:, None] with x_b viewed [n_i, H W]; y is contiguous NCHW."""
    _dev(x, A, B, bias)
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == A.dtype == B.dtype
    A, B = _rows2d(A), _rows2d(B)
    b, n_i, h, w = x.shape
    r, n_o = A.shape[0], B.shape[0]
    assert A.shape[1] == n_i and B.shape[1] == r
    y = torch.empty((b, n_o, h, w), dtype=x.dtype, device=x.device)
    lib = _hip.load()
    ws_bytes = lib.ptd_lowrank_forward_nchw_workspace_bytes(b, h * w, r, _code(x))
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.device)
    if bias is not None:
        bias = bias.to(x.dtype).contiguous()
    with torch.cuda.device(x.device):
        rc = lib.ptd_lowrank_forward_nchw(x.data_ptr(), b, n_i, h * w, A.data_ptr(), A.stride(0), r, B.data_ptr(),
                                          B.stride(0), n_o, _ptr(bias), y.data_ptr(), ws.data_ptr(), ws_bytes,
                                          _code(x), _stream(x))
    _hip.check(rc, "ptd_lowrank_forward_nchw")
    return y


# ptd_nsr workspaces: initialised once (ptd_nsr_workspace_init) and reused -- the kernel leaves its arrival counters
# zeroed.  One per (device, stream): calls on one stream are ordered, calls on different streams may overlap.
_NSR_WS: dict = {}


def _nsr_workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream_of(device))
    ws = _NSR_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            _hip.check(_hip.load().ptd_nsr_workspace_init(ws.data_ptr(), ws.numel(), _stream_of(device)),
                       "ptd_nsr_workspace_init")
        if len(_NSR_WS) > 64:   # (streams come and go: do not keep every workspace ever made)
            _NSR_WS.clear()
        _NSR_WS[key] = ws
    return ws


def _stream_of(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def nsr(x: torch.Tensor, y: torch.Tensor, channels: int, eps: float = 1e-3) -> torch.Tensor:
    """Scalar (f64, on device): mean over the `channels` trailing-dim channels of
    mean((x-y)^2) / (var(y) + eps), x and y viewed as [-1, channels]."""
    _dev(x, y)
    assert x.shape == y.shape and x.dtype == y.dtype
    x, y = x.contiguous(), y.contiguous()
    C = int(channels)
    R = x.numel() // C
    lib = _hip.load()
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    ws = _nsr_workspace(x.device, lib.ptd_nsr_workspace_bytes(R, C))
    with torch.cuda.device(x.device):
        rc = lib.ptd_nsr(x.data_ptr(), y.data_ptr(), R, C, _code(x), float(eps), out.data_ptr(), ws.data_ptr(),
                         ws.numel(), _stream(x))
    if rc != 0:
        #  a launch that failed midway may have left the workspace's ticket non-zero: every later call on this
        # stream would then return the NaN the partial kernel presets -- drop the cached workspaces, the next call
        # initialises a fresh one
        _NSR_WS.clear()
    _hip.check(rc, "ptd_nsr")
    return out[0]


def drop_cached_workspaces() -> None:
    """Forget the per-(device, stream) ptd_nsr workspaces (they are re-made and re-initialised on the next call).  The
    drivers call this when a metric comes back non-finite: a kernel that was aborted between its partial and its final
    launch leaves the workspace's ticket non-zero, and every later ptd_nsr on that stream would return NaN."""
    _NSR_WS.clear()


def sym_kl(s: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """Scalar (f64, on device): mean_b max(KL(t||s), KL(s||t)) for logits [B, C]."""
    _dev(s, t)
    assert s.shape == t.shape and s.dim() == 2 and s.dtype == t.dtype
    s, t = s.contiguous(), t.contiguous()
    B, C = s.shape
    lib = _hip.load()
    out = torch.empty(1, dtype=torch.float64, device=s.device)
    ws = torch.empty(lib.ptd_sym_kl_workspace_bytes(B), dtype=torch.uint8, device=s.device)
    with torch.cuda.device(s.device):
        rc = lib.ptd_sym_kl(s.data_ptr(), t.data_ptr(), B, C, _code(s), out.data_ptr(), ws.data_ptr(), ws.numel(),
                            _stream(s))
    _hip.check(rc, "ptd_sym_kl")
    return out[0]


def kl_rows(q: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """[B] f64: KL(p_b || q_b) over softmax(dim=-1) of logits q, p [B, C]."""
    _dev(q, p)
    assert q.shape == p.shape and q.dim() == 2 and q.dtype == p.dtype
    q, p = q.contiguous(), p.contiguous()
    B, C = q.shape
    rows = torch.empty(B, dtype=torch.float64, device=q.device)
    with torch.cuda.device(q.device):
        rc = _hip.load().ptd_kl_rows(q.data_ptr(), p.data_ptr(), B, C, _code(q), rows.data_ptr(), _stream(q))
    _hip.check(rc, "ptd_kl_rows")
    return rows
