"""Device time of the Cholesky sweep (ptd_chol_inverse) at the filtered route's block sizes, both tile forms."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops, _hip
dev = torch.device("cuda", 0)
lib = _hip.load()
for m in (640, 1280, 2560):
    x = torch.randn(4096, m, dtype=torch.float64, device=dev)
    g = x.T @ x
    W = torch.empty_like(g)
    ws = torch.empty(lib.ptd_chol_inverse_workspace_bytes(m), dtype=torch.uint8, device=dev)
    for leaf in ("0", "1"):
        os.environ["PTD_EIGH_FILTER_LEAF"] = leaf
        ts = []
        for it in range(6):
            work = g.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.ptd_chol_inverse(work.data_ptr(), m, W.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
            e1.record(); torch.cuda.synchronize()
            assert rc == 0
            ts.append(e0.elapsed_time(e1))
        err = ((W.T @ g @ W) - torch.eye(m, dtype=torch.float64, device=dev)).abs().max().item()
        print(f"m={m} leaf={leaf}: {min(ts[1:]):.3f} ms per sweep ({m // 64} panels, {min(ts[1:]) / (m // 64) * 1e3:.1f} us per panel), |W^T G W - I| = {err:.2e}", flush=True)
