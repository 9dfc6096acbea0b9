"""The C-ABI library builds, loads, and exports every symbol include/ptdeco_hip.h declares
(no compute calls: runs without a GPU)."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ptdeco_hip.h")
LIB = os.path.join(ROOT, "ptdeco_amd", "libptdeco_hip.so")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ptd_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ptdeco_hip.h but not exported"


def test_binding_covers_header():
    from ptdeco_amd import _hip
    assert sorted(_hip.SIGNATURES) == _declared()
    assert _hip.load().ptd_version() == _hip.ABI_VERSION


def test_workspace_queries_and_argument_errors(lib):
    from ptdeco_amd import _hip
    l = _hip.load()
    assert l.ptd_eigh_workspace_bytes(4096) >= 4096 * 4096 * 8
    assert l.ptd_eigh_workspace_bytes(1) >= 64 * 64 * 8
    assert l.ptd_nsr_workspace_bytes(4096, 4096) > 0
    assert l.ptd_sym_kl_workspace_bytes(5) >= 40
    # null pointers are rejected before anything is launched
    rc = l.ptd_syrk_accumulate(None, 4, 4, 4, _hip.F32, None, 4, _hip.F64, 1.0, None)
    assert rc == -1 and b"null" in l.ptd_last_error()
    rc = l.ptd_gemm(None, 1, 1, None, 1, 1, None, 1, 1, 1, 1, _hip.F32, _hip.F32, 1.0, None, None)
    assert rc == -1


def test_eigh_route_query(monkeypatch):
    """ptd_eigh_route is a host-side function of (n, k, all_values) and the environment: the filtered subspace iteration
    for at most 2/7 of the spectrum of a large matrix whose eigenvalues below are not wanted, the direct reduction for
    everything else from order 256 on, Jacobi for tiny matrices."""
    from ptdeco_amd import _hip
    l = _hip.load()
    monkeypatch.delenv("PTD_EIGH_FILTERED", raising=False)
    monkeypatch.delenv("PTD_EIGH_METHOD", raising=False)
    assert l.ptd_eigh_route(4096, 1024, 0) == 3
    assert l.ptd_eigh_route(4096, 1170, 0) == 3 and l.ptd_eigh_route(4096, 1365, 0) == 1     # the cut: 7 k <= 2 n
    assert l.ptd_eigh_route(4096, 2048, 0) == 1 and l.ptd_eigh_route(4096, 1024, 1) == 1
    assert l.ptd_eigh_route(1024, 256, 0) == 1 and l.ptd_eigh_route(4000, 1000, 0) == 1      # small / not a multiple of 128
    assert l.ptd_eigh_route(128, 32, 0) == 0
    monkeypatch.setenv("PTD_EIGH_FILTERED", "0")
    assert l.ptd_eigh_route(4096, 1024, 0) == 1
    monkeypatch.setenv("PTD_EIGH_METHOD", "jacobi")
    assert l.ptd_eigh_route(4096, 1024, 0) == 0
    # the f32 face needs room for the converted matrix, the f64 results and the f64 solver
    assert l.ptd_eigh_f32_workspace_bytes(512, 128) >= 512 * 512 * 8 + 512 * 128 * 8 + l.ptd_eigh_workspace_bytes(512)


def test_no_cpu_path():
    import torch
    import ptdeco_amd
    with pytest.raises(ValueError, match="no CPU"):
        ptdeco_amd.falor.decompose_in_place(
            module=torch.nn.Linear(4, 4), device=torch.device("cpu"), data_iterator=iter([]),
            proportion_threshold=0.9, nsr_final_threshold=0.1, kl_final_threshold=0.1, num_data_steps=1,
            num_metric_steps=1, use_float64=True, use_mean=False, use_damping=True)
    with pytest.raises(ValueError, match="ROCm device"):
        ptdeco_amd.ops.matmul(torch.zeros(2, 2), torch.zeros(2, 2))
