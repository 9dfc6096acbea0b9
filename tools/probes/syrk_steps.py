"""The bf16 covariance product at the calibration shapes (2048-row steps, 8 per call): us per step.
PTD_SYRK_WAVES=8 python tools/probes/syrk_steps.py   (round 5's eight-wave tile; default: the four-wave 64 x 64 form)"""
import json, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import bench
from ptdeco_amd import ops
dev = torch.device("cuda", 0)
for n in (4096, 14336, 2048, 1024):
    ys = [torch.randn(2048, n, device=dev).bfloat16() for _ in range(8)]
    e = torch.zeros(n, n, dtype=torch.float64, device=dev)
    t8 = bench.time_events(lambda: ops.syrk_accumulate_multi(e, ys, 1.0 / 2048), iters=10) / 8
    t1 = bench.time_events(lambda: ops.syrk_accumulate(e, ys[0], 1.0 / 2048), iters=10)
    bound = 2048 * n * (n + 1) / 2.5e15
    print(json.dumps({"n": n, "waves": os.environ.get("PTD_SYRK_WAVES", "4"), "us_per_step_multi8": round(t8 * 1e6, 1), "us_single": round(t1 * 1e6, 1),
                      "mfma_bound_us": round(bound * 1e6, 2), "frac_multi8": round(bound / t8, 3)}), flush=True)
