"""GPU check of the two-stage reduction: eigenvalues of T vs LAPACK, eigenvector residuals, timings."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops  # noqa: E402

DEV = torch.device("cuda")


def spd(n, seed):
    g = torch.Generator().manual_seed(seed)
    y = torch.randn(2 * n + 3, n, generator=g, dtype=torch.float64) * torch.logspace(0, -2, n, dtype=torch.float64)
    a = y.T @ y / y.shape[0]
    return a + torch.eye(n, dtype=torch.float64) * (0.01 * torch.diag(a).mean())


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--json")]
    out_json = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--json=")]
    runs = []
    sizes = [int(x) for x in args] or [128, 256, 1024]
    for n in sizes:
        a = spd(n, n)
        w_ref = torch.linalg.eigvalsh(a)
        ad = a.to(DEV)
        d, e, w = ops.tridiagonalize(ad)
        torch.cuda.synchronize()
        err = (w.cpu() - w_ref).abs().max().item() / w_ref.max().item()
        print(f"n={n} tridiagonalize: eigenvalue err {err:.2e}", flush=True)
        k = n // 4
        ops.EIGH_PROFILE = []
        wk, v = ops.eigh(ad, k, all_values=False)
        torch.cuda.synchronize()
        prof = ops.EIGH_PROFILE[0]
        ops.EIGH_PROFILE = None
        vc = v.cpu()
        res = (a @ vc - vc * wk.cpu()[n - k:]).abs().max().item() / w_ref.max().item()
        orth = (vc.T @ vc - torch.eye(k, dtype=torch.float64)).abs().max().item()
        print(f"n={n} eigh top-{k}: method {prof['method']} residual {res:.2e} orth {orth:.2e} ms {prof['ms']} "
              f"total {prof['total_ms']:.2f} q2_us {prof['launches'][3]}", flush=True)
        runs.append({"n": n, "k": k, "method": prof["method"], "ms": prof["ms"], "total_ms": prof["total_ms"],
                     "q2_us": prof["launches"][3], "residual": res, "orth": orth})
        if n >= 2048:
            for _ in range(2):
                t0 = time.perf_counter()
                ops.eigh(ad, k, all_values=False)
                torch.cuda.synchronize()
                print(f"   wall {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)


    if out_json:
        import json
        json.dump({"command": "PTD_EIGH_STAGES=2 python tools/twostage_check.py " + " ".join(map(str, sizes)),
                   "note": "ms = [stage 1 dense -> band 32, stage 2 bulge chase, eigenpairs of T, back-transformation Q1 Q2 Y]; "
                           "q2_us = the Q2 part of the back-transformation; residual = max |A z - lambda z| / |A|, "
                           "orth = max |Z^T Z - I|", "runs": runs}, open(out_json[0], "w"), indent=1)


if __name__ == "__main__":
    main()
