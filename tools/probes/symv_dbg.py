"""What a blocked column's SYMV launch (sytrd_symv2_kernel, trailing orders 4096 .. 3073) spends its time on: the probe
build (make -C ptdeco_amd/csrc prof) with parts of the kernel switched off -- results are wrong, only times count."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptdeco_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libptdeco_prof.so")
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
lib = _hip.load()
raw = ctypes.CDLL(_hip.LIB_PATH)
n = 4096
g = torch.Generator(device=dev).manual_seed(3)
scale = torch.logspace(0, -2, n, device=dev)
e = torch.zeros(n, n, dtype=torch.float64, device=dev)
for _ in range(2):
    y = torch.randn(4096, n, generator=g, device=dev) * scale
    ops.syrk_accumulate(e, y, 1.0 / 4096)
c = ops.cov_finalize(e, 2, 0.01)
os.environ["PTD_SYTRD_RESIDENT"] = "0"       # a broken reduction must not meet the resident kernels' spins
for dbg, label in ((0, "whole kernel"), (1, "launch and dispatch only"), (2, "tile loads only")):
    raw.ptd_debug_symv(dbg)
    torch.cuda.synchronize()
    ops.EIGH_PROFILE = []
    try:
        ops.eigh(c, 2048, all_values=False)
    except Exception as exc:
        print("   (", type(exc).__name__, ")")
    torch.cuda.synchronize()
    p, ops.EIGH_PROFILE = (ops.EIGH_PROFILE[0] if ops.EIGH_PROFILE else None), None
    if p:
        print(f"{label:28s} SYMV launches {p['launches'][0]}: {p['ms'][0] / max(p['launches'][0], 1) * 1e3:.2f} us each")
raw.ptd_debug_symv(0)
