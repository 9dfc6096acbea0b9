import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ptdeco_amd import ops
import twostage_proto as proto
from twostage_check import spd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
a = spd(n, n)
w_ref = np.linalg.eigvalsh(a.numpy())
b1 = ops.band_reduce(a.cuda(), 1).numpy()
print("stage1: outside band", np.abs(np.tril(b1, -33)).max(), "eig err", np.abs(np.linalg.eigvalsh(b1) - w_ref).max() / w_ref.max())
Bd, refl = proto.dense_to_band(a.numpy(), 32)
print("stage1 vs proto |diff| (abs values)", np.abs(np.abs(b1) - np.abs(Bd)).max())
for blk in range(min(3, n // 32 - 1)):
    sub_g = b1[32 * (blk + 1):32 * (blk + 2), 32 * blk:32 * (blk + 1)]
    sub_p = Bd[32 * (blk + 1):32 * (blk + 2), 32 * blk:32 * (blk + 1)]
    dg = b1[32 * blk:32 * (blk + 1), 32 * blk:32 * (blk + 1)]
    dp = Bd[32 * blk:32 * (blk + 1), 32 * blk:32 * (blk + 1)]
    print(" block", blk, "subdiag diff", np.abs(sub_g - sub_p).max(), "diag diff", np.abs(dg - dp).max(), "lower-part of R", np.abs(np.tril(sub_g, -1)).max())
b2 = ops.band_reduce(a.cuda(), 2).numpy()
print("stage2: outside tridiagonal", np.abs(np.tril(b2, -2)).max(), "eig err", np.abs(np.linalg.eigvalsh(b2) - w_ref).max() / w_ref.max())
d, e, V2, tau2, At = proto.band_to_tridiag(b1, 32)
print("stage2 vs proto on the GPU band: d diff", np.abs(np.diag(b2) - d).max(), "e diff", np.abs(np.abs(np.diag(b2, -1)) - np.abs(e)).max())
dg, eg = np.diag(b2), np.diag(b2, -1)
bad_d = np.nonzero(np.abs(dg - d) > 1e-12)[0]
bad_e = np.nonzero(np.abs(np.abs(eg) - np.abs(e)) > 1e-12)[0]
print("first bad d", bad_d[:5], "first bad e", bad_e[:5], "count", len(bad_d), len(bad_e))
