"""A/B in one process: the bf16 8-phase kernel on v_mfma_f32_32x32x16_bf16 vs 16x16x32 (PTD_GEMM_8PH_MFMA), interleaved
rounds, dense layer and the two products of the rank-512 / 1024 pair at T = 16384; torch (hipBLASLt) beside them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops
dev = torch.device("cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T = 16384
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(T, 4096, device=dev, generator=g).bfloat16()
w = (torch.randn(4096, 4096, device=dev, generator=g) / 64).bfloat16()
cases = {"dense 16384x4096x4096": (x, w)}
for r in (512, 1024):
    a = (torch.randn(r, 4096, device=dev, generator=g) / 64).bfloat16()
    b = (torch.randn(4096, r, device=dev, generator=g) / r ** 0.5).bfloat16()
    h = ops.matmul(x, a.T)
    cases[f"x A^T r={r}"] = (x, a)
    cases[f"h B^T r={r}"] = (h, b)
F = torch.nn.functional
for name, (p, q) in cases.items():
    # exactness of the new form on integers first
    pi = torch.randint(-4, 5, p.shape, device=dev, generator=g).bfloat16(); qi = torch.randint(-4, 5, q.shape, device=dev, generator=g).bfloat16()
    os.environ["PTD_GEMM_8PH_MFMA"] = "32"; r32 = ops.matmul(pi, qi.T, out_dtype=torch.float32)
    os.environ["PTD_GEMM_8PH_MFMA"] = "16"; r16 = ops.matmul(pi, qi.T, out_dtype=torch.float32)
    exact = torch.equal(r32, r16) and torch.equal(r16, (pi.float() @ qi.float().T))
    res = {"32": [], "16": [], "lib": []}
    for rnd in range(3):
        for mf in ("32", "16"):
            os.environ["PTD_GEMM_8PH_MFMA"] = mf
            res[mf].append(t(lambda: ops.matmul(p, q.T)))
        res["lib"].append(t(lambda: F.linear(p, q)))
    fl = 2.0 * p.shape[0] * p.shape[1] * q.shape[0]
    print(f"{name:26s} exact={exact}  32x32x16 {min(res['32']):7.1f} us ({fl/min(res['32'])/1e6:6.0f} TF)   16x16x32 {min(res['16']):7.1f} us ({fl/min(res['16'])/1e6:6.0f} TF)   hipBLASLt {min(res['lib']):7.1f} us ({fl/min(res['lib'])/1e6:6.0f} TF)")
os.environ.pop("PTD_GEMM_8PH_MFMA")
