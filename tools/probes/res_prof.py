"""Where a column of the chip-resident tridiagonalisation kernels spends its time (probe build of the library with
-DPTD_RES_PROF: workgroup 17 accumulates 100-MHz clock ticks per phase).  Build beforehand: make -C ptdeco_amd/csrc prof"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ptdeco_amd import _hip
_hip.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libptdeco_prof.so")
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
lib = _hip.load()
raw = ctypes.CDLL(_hip.LIB_PATH)
raw.ptd_debug_res_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
g = torch.Generator(device=dev).manual_seed(3)
n = 4096
scale = torch.logspace(0, -2, n, device=dev)
e = torch.zeros(n, n, dtype=torch.float64, device=dev)
for _ in range(2):
    y = torch.randn(4096, n, generator=g, device=dev) * scale
    ops.syrk_accumulate(e, y, 1.0 / 4096)
c = ops.cov_finalize(e, 2, 0.01)
ops.eigh(c, 2048, all_values=False); torch.cuda.synchronize()
raw.ptd_debug_res_prof(None, 1)
reps = 3
for _ in range(reps):
    ops.eigh(c, 2048, all_values=False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 80)()
raw.ptd_debug_res_prof(buf, 0)
names = {0: "reflector (to the barrier behind vs)", 1: "update + product pass (+ part barrier in r3)", 6: "p, b stores (r3)",
         2: "publish (drain, barrier, number)", 3: "wait for the 256 numbers", 4: "load p, b + dot, barrier",
         5: "w, next x, barrier", 7: "loop tail (vo copy, wrow)"}
for k, (label, cols) in enumerate((("resident<256,2048> 2048 -> 1024", 1024), ("resident3 3072 -> 2048", 1024),
                                  ("resident<32,1024> (one XCD) 1024 -> 768", 256), ("resident<32,768> (one XCD) 768 -> 1", 767),
                                  ("resident4 (quarter rows, four waves) 3840 -> 3584 -> 3328 -> 3072 together", 768))):
    tot = sum(buf[16 * k + i] for i in range(16))
    print(f"{label}: {tot * 0.01 / reps / cols:.2f} us per column")
    for i in (0, 1, 6, 2, 3, 4, 5, 7):
        v = buf[16 * k + i]
        if v:
            print(f"   {names[i]:48s} {v * 0.01 / reps / cols:6.2f} us")
