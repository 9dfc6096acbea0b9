"""ms per round of 256 tiles of the bf16 NT GEMM (256 x 256 tiles, N = K = 4096) as M grows: 1, 2, 4, 8 rounds, the
persistent kernel against one workgroup per tile (PTD_GEMM_8PH_PERSIST=0) and torch's hipBLASLt.
Usage: python tools/bf16_rounds.py"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ptdeco_amd import ops

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


N = int(os.environ.get("N", "4096"))
K = int(os.environ.get("K", "4096"))
w = (torch.randn(N, K, generator=g, device=dev) / K ** 0.5).to(torch.bfloat16)
for rounds in (1, 2, 3, 4, 8, 16):
    M = rounds * 256 * 256 // (N // 256)
    x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    row = {"M": M, "N": N, "K": K, "rounds": rounds}
    for mode in ("1", "0"):
        os.environ["PTD_GEMM_8PH_PERSIST"] = mode
        ms = timed(lambda: ops.matmul(x, w.T))
        row["persistent" if mode == "1" else "per_tile"] = {"ms": ms, "us_per_round": ms * 1e3 / rounds,
                                                            "tflops": 2.0 * M * N * K / ms / 1e9}
    ms = timed(lambda: torch.nn.functional.linear(x, w))
    row["hipblaslt"] = {"ms": ms, "us_per_round": ms * 1e3 / rounds, "tflops": 2.0 * M * N * K / ms / 1e9}
    print(json.dumps(row), flush=True)
