"""ctypes binding of libptdeco_hip.so (C ABI: include/ptdeco_hip.h).

The library is the only compute backend of this package: if it cannot be loaded
every operation fails loudly -- there is no CPU or eager-PyTorch fallback.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptdeco_hip.so")

F32, F64, BF16 = 0, 1, 2
ABI_VERSION = 4

# name -> (restype, argtypes); must list every symbol include/ptdeco_hip.h declares
SIGNATURES = {
    "ptd_version": (c_int, []),
    "ptd_last_error": (c_char_p, []),
    "ptd_set_concurrent_chains": (c_int, [c_int]),
    "ptd_stream_pair_wall_us": (c_int, [c_void_p, c_void_p, c_int, ctypes.POINTER(c_double)]),
    "ptd_streams_wall_us": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_double)]),
    "ptd_stream_create_dedicated": (c_int, [c_int, c_int, ctypes.POINTER(c_void_p)]),
    "ptd_stream_destroy": (c_int, [c_void_p]),
    "ptd_syrk_accumulate": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_int,
                                    c_double, c_void_p]),
    "ptd_syrk_accumulate_multi": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int, c_void_p, c_int64, c_int,
                                          c_double, c_void_p]),
    "ptd_colsum_accumulate": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int, c_double,
                                      c_void_p]),
    "ptd_cov_finalize_workspace_bytes": (c_size_t, [c_int64]),
    "ptd_cov_finalize": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_double, c_double, c_void_p,
                                 c_int64, c_void_p, c_size_t, c_void_p]),
    "ptd_eigh_workspace_bytes": (c_size_t, [c_int64]),
    "ptd_eigh": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_size_t,
                         ctypes.POINTER(c_int), c_void_p]),
    "ptd_eigh_topk": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p,
                              c_size_t, ctypes.POINTER(c_int), c_void_p]),
    "ptd_eigh_route": (c_int, [c_int64, c_int64, c_int]),
    "ptd_eigh_forget_declines": (None, []),
    "ptd_eigh_f32_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "ptd_eigh_topk_f32": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p,
                                  c_size_t, ctypes.POINTER(c_int), c_void_p]),
    "ptd_eigh_profiled": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64, c_void_p,
                                  c_size_t, c_void_p, c_void_p]),
    "ptd_chol_inverse_workspace_bytes": (c_size_t, [c_int64]),
    "ptd_chol_inverse": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ptd_tridiagonalize_workspace_bytes": (c_size_t, [c_int64]),
    "ptd_tridiagonalize": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    "ptd_eigh_batched_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "ptd_eigh_topk_batched": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64,
                                      c_void_p, c_size_t, c_void_p, c_void_p]),
    "ptd_eigh_factored_prepare": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                          c_size_t, ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), c_void_p]),
    "ptd_eigh_factored_finish": (c_int, [c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                         c_void_p, c_size_t, c_void_p]),
    "ptd_eigh_factored_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "ptd_eigh_factored": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                  c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "ptd_gemm": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                         c_int64, c_int64, c_int, c_int, c_double, c_void_p, c_void_p]),
    "ptd_gemm_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int, c_int]),
    "ptd_gemm_ws": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64,
                            c_int64, c_int64, c_int, c_int, c_double, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ptd_lowrank_forward_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int]),
    "ptd_lowrank_forward": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                    c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_int,
                                    c_void_p]),
    "ptd_lowrank_forward_nchw_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int64, c_int]),
    "ptd_lowrank_forward_nchw": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                         c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "ptd_nsr_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "ptd_nsr_workspace_init": (c_int, [c_void_p, c_size_t, c_void_p]),
    "ptd_nsr": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_double, c_void_p, c_void_p, c_size_t,
                        c_void_p]),
    "ptd_sym_kl_workspace_bytes": (c_size_t, [c_int64]),
    "ptd_sym_kl": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ptd_kl_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
}


class EighStats(ctypes.Structure):
    """ptd_eigh_stats of include/ptdeco_hip.h."""
    _fields_ = [("method", c_int), ("sweeps", c_int), ("launches", c_int * 4), ("ms", ctypes.c_float * 4),
                ("work", c_double * 4), ("total_ms", ctypes.c_float)]


def source_sha16(*names: str) -> str:
    """sha256 (first 16 hex digits) of the named kernel sources under csrc/: ties a committed counter summary
    (profiles/pmc_*.json) to the code that produced it, without needing a git checkout on the GPU box."""
    import hashlib

    h = hashlib.sha256()
    for name in names:
        with open(os.path.join(_HERE, "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


class HipLibraryError(RuntimeError):
    pass


_lib = None


def load() -> ctypes.CDLL:
    """Load the shared library once and bind every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        # a fresh checkout: build the library in tree (hipcc cross-compiles gfx950 without a GPU)
        import shutil
        import subprocess

        if shutil.which("make") and shutil.which(os.environ.get("HIPCC", "hipcc")):
            subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], check=False,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `make -C ptdeco_amd/csrc` (or __graft_entry__.build()). "
            "ptdeco_amd has no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64 etc.
        raise HipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.ptd_version() != ABI_VERSION:
        raise HipLibraryError(f"ABI version mismatch: library {lib.ptd_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ptd_last_error().decode(errors="replace")
        raise HipLibraryError(f"{what} failed (status {rc}): {msg}")
