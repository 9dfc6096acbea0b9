#!/usr/bin/env python3
"""Headline benchmark: dwain layers decomposed per second on MI355X.

    python bench.py --gpus N --steps K --warmup W

ONE fixed workload at every N (strong scaling): dwain.decompose_in_place of a stack of 8 nn.Linear(4096, 4096,
bias=False) -- BASELINE.json configs[1]'s layer, eight of them chained so that the same work can be dealt to 1 / 2 /
4 / 8 GPUs the way configs[3] shards a Llama stack -- f32 model, f64 decomposition, B=4 x S=1024 tokens per batch,
precomputing_covariance_num_splits=1, D = 8 calibration steps, M = 2 metric steps, 6 evaluated candidate ranks per
layer (1024 .. 32; 2048 drops no parameters and is skipped), CE loss over the 4096 outputs, identity finetune_fn.
With N ranks: calibration steps dealt to the ranks, the 8 covariance sums reduced to their owners (packed lower
triangles over RCCL, started asynchronously and completed right before each owner's eigensolve), the 8
eigendecompositions owned round-robin, eigenvectors broadcast, (candidate, metric batch) pairs dealt to the ranks.
One "step" = one full decompose_in_place call on a fresh copy of the stack, every input already resident in HBM;
`value` = 8 layers x K / time, "scaling": "strong".  `python bench.py --gpus N` starts its own N ranks when no launcher
did (children, before any GPU call).

The same line carries
  c2_single_layer BASELINE configs[1] / SURVEY C2 itself (ONE nn.Linear(4096, 4096), D = 4, M = 2), timed with the same
                 K / W protocol at N = 1: the workload the roofline / eigh / phases / cpu_baseline blocks describe
  bf16_stack     the fixed stack timed once more with a bf16 model (SURVEY 8d's throughput configuration)
  weak_family    N > 1 only: the round-3 family (a stack of N layers on N GPUs), kept for comparison
  roofline       the dominant kernel of the eigensolver against its bound: frac on SURVEY 8d's
                 algorithmic work, solver_frac for the whole ptd_eigh call
  phases_ms      device-time split of one step: A accumulate, B eigh, C factors, D metrics, comm
  stack_phases_ms  the same split for one step of the fixed stack (the workload `value` is quoted on), N = 1
  kernels        per-kernel device time / rates from HIP events
  cpu_baseline   the CPU oracle (restatement of the reference, torch-CPU/MKL) on the C2
                 workload on this box's physical host cores, rank 0, N = 1 only
  decomposed_fwd rank-r two-GEMM forward vs dense 4096x4096, bf16 (BASELINE configs[4])
  c4_shapes      dwain on one layer of each Llama-3-8B shape (BASELINE configs[3]), f32 and bf16, three timed
                 steps each, with the eigensolver route and its roofline block
  c4_block       one full-width Llama-3-8B block (7 layers) end to end, f32 and bf16
Counter-derived fields (traffic, MFMA utilisation) are quoted from committed rocprofv3 PMC
summaries; each carries the hash of the kernel source it was measured on and is marked
"stale": true when the source has changed since.
"""

from __future__ import annotations

import argparse
import copy
import itertools
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_FEAT = 4096
BATCH, SEQ = 4, 1024
D_STEPS, M_STEPS = 4, 2
STACK_LAYERS, STACK_D_STEPS = 8, 8   # the fixed stack of the scaling family
PEAK_F64_MFMA = 78.6e12   # MI355X dense f64 matrix peak
PEAK_F32_MFMA = 157.3e12  # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_BF16_MFMA = 2.5e15
PEAK_HBM = 8.0e12


class LinearChain(torch.nn.Module):
    def __init__(self, n_layers: int):
        super().__init__()
        self.layers = torch.nn.ModuleList(torch.nn.Linear(N_FEAT, N_FEAT, bias=False) for _ in range(n_layers))

    def forward(self, d):
        x = d["x"]
        for lin in self.layers:
            x = lin(x)
        return x


def ce_loss(batch, logits):
    return torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), batch["targets"].reshape(-1),
                                             reduction="none")


def pmc_file(pattern: str, sources: tuple):
    """Latest committed counter summary matching `pattern`, with provenance: {"data", "source", "stale"}.
    stale = the kernel sources it names have changed since the pass (or it predates the hash)."""
    import glob

    from ptdeco_amd import _hip

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None
    d = json.load(open(files[-1]))
    want = d.get("source_sha16")
    have = _hip.source_sha16(*sources)
    return {"data": d, "source": "profiles/" + os.path.basename(files[-1]), "stale": want != have,
            "measured_on_source_sha16": want, "current_source_sha16": have}


def make_workload(n_layers: int, device, n_data: int, n_metric: int):
    """Seeded synthetic weights and batches: W ~ N(0, 1/4096) seed 1234,
    x ~ N(0,1) * logspace(0,-2) feature scale, data seed 1, metric seed 2, targets = argmax of the
    original model's logits."""
    g = torch.Generator().manual_seed(1234)
    model = LinearChain(n_layers)
    with torch.no_grad():
        for lin in model.layers:
            lin.weight.copy_(torch.randn(N_FEAT, N_FEAT, generator=g) / N_FEAT**0.5)
    scale = torch.logspace(0, -2, N_FEAT)

    def batches(seed, count):
        gg = torch.Generator().manual_seed(seed)
        return [torch.randn(BATCH, SEQ, N_FEAT, generator=gg) * scale for _ in range(count)]

    data, metric = batches(1, n_data), batches(2, n_metric)
    return model, data, metric


def with_targets(model, xs, device):
    out = []
    with torch.no_grad():
        for x in xs:
            x = x.to(device)
            out.append({"x": x, "targets": model({"x": x}).argmax(dim=-1)})
    return out


DWAIN_KW = dict(num_data_steps=D_STEPS, num_metric_steps=M_STEPS, nsr_final_threshold=1.0, min_rank=32,
                trade_off_factor=0.5, reduction_factor=0.5, max_accepted_ppl_diff=0.1, decompose_in_float64=True)


MIN_LOOP_MS = 40.0


def time_events(fn, iters=20, warm=3, rounds=3, min_loop_ms=MIN_LOOP_MS):
    """Seconds per call: `rounds` timed loops between two HIP events, the MEDIAN loop reported -- one stall of the box
    inside a loop (an 80-ms one made a 0.12-ms line read 8.2 ms in a round-5 run) does not become the figure, and neither
    does the best loop.  A loop holds at least `iters` calls and at least `min_loop_ms` of device work: the card leaves
    its idle clocks over the first milliseconds of a loop, and a 4-ms loop of 0.4-ms launches read 480 us a launch where
    a 40-ms loop of the same launches reads 374 (tools/probes/fwd_protocol.py; the library kernels move alike)."""
    for _ in range(warm):
        fn()

    def loop(count):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(count):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    first = loop(iters)
    if first < min_loop_ms:
        iters = min(int(iters * min_loop_ms / max(first, 1e-3)) + 1, 4000)
    got = [loop(iters) / iters * 1e-3 for _ in range(rounds)]
    return sorted(got)[len(got) // 2]


def kernel_lines(device):
    """Device time of the path's individual kernels at the C2 shapes (HIP events on the launch stream)."""
    from ptdeco_amd import ops

    t_tok = BATCH * SEQ
    g = torch.Generator().manual_seed(3)
    x = torch.randn(t_tok, N_FEAT, generator=g).to(device)
    w = (torch.randn(N_FEAT, N_FEAT, generator=g) / 64).to(device)
    y = ops.matmul(x, w.T)
    e = torch.zeros(N_FEAT, N_FEAT, dtype=torch.float64, device=device)
    lines = {}
    t = time_events(lambda: ops.syrk_accumulate(e, y, 1.0 / t_tok))
    fl = t_tok * N_FEAT * (N_FEAT + 1)
    lines["syrk_f32_f64acc"] = {"ms": t * 1e3, "algorithmic_flops": fl, "tflops": fl / t / 1e12,
                                "frac_of_f32_mfma_peak": fl / t / PEAK_F32_MFMA}
    t = time_events(lambda: ops.matmul(x, w.T))
    fl = 2 * t_tok * N_FEAT * N_FEAT
    lines["gemm_f32_nt"] = {"ms": t * 1e3, "algorithmic_flops": fl, "tflops": fl / t / 1e12,
                            "frac_of_f32_mfma_peak": fl / t / PEAK_F32_MFMA}
    yb, xb, wb = y.bfloat16(), x.bfloat16(), w.bfloat16()
    t = time_events(lambda: ops.syrk_accumulate(e, yb, 1.0 / t_tok))
    fl = t_tok * N_FEAT * (N_FEAT + 1)
    lines["syrk_bf16_f64acc"] = {"ms": t * 1e3, "algorithmic_flops": fl, "tflops": fl / t / 1e12,
                                 "frac_of_bf16_mfma_peak": fl / t / PEAK_BF16_MFMA}
    t = time_events(lambda: ops.matmul(xb, wb.T))
    fl = 2 * t_tok * N_FEAT * N_FEAT
    lines["gemm_bf16_nt"] = {"ms": t * 1e3, "algorithmic_flops": fl, "tflops": fl / t / 1e12,
                             "frac_of_bf16_mfma_peak": fl / t / PEAK_BF16_MFMA}
    # the bf16 covariance product at the Llama calibration shapes (2048 tokens a step): one call per step against the
    # multi-step entry (8 steps in one pass over the f64 accumulator); bounds on 2 T n bytes of activations per step +
    # 8 n (n + 1) bytes of accumulator per CALL and T n (n + 1) flop per step
    cal = {}
    for n_c in (4096, 1024, 14336):
        ys = [torch.randn(2048, n_c, device=device).bfloat16() for _ in range(8)]
        e_c = torch.zeros(n_c, n_c, dtype=torch.float64, device=device)

        def one_by_one():
            for y_c in ys:
                ops.syrk_accumulate(e_c, y_c, 1.0 / 2048)

        t1 = time_events(one_by_one, iters=10) / 8
        t8 = time_events(lambda: ops.syrk_accumulate_multi(e_c, ys, 1.0 / 2048), iters=10) / 8
        by1 = 2 * 2048 * n_c + 8 * n_c * (n_c + 1)
        by8 = 2 * 2048 * n_c + n_c * (n_c + 1)
        fl = 2048 * n_c * (n_c + 1)
        cal[f"n{n_c}"] = {"T": 2048, "us_per_step_single_call": t1 * 1e6, "us_per_step_multi_8": t8 * 1e6,
                          "hbm_bound_us_single": by1 / PEAK_HBM * 1e6, "hbm_bound_us_multi_8": by8 / PEAK_HBM * 1e6,
                          "mfma_bound_us": fl / PEAK_BF16_MFMA * 1e6,
                          "frac_of_bound_single": max(by1 / PEAK_HBM, fl / PEAK_BF16_MFMA) / t1,
                          "frac_of_bound_multi_8": max(by8 / PEAK_HBM, fl / PEAK_BF16_MFMA) / t8}
        del ys, e_c
    lines["syrk_bf16_calibration_shapes"] = cal
    # ptd_nsr through the C ABI with prepared arguments (the Python front end costs as much host time per call as this
    # kernel pair takes on the device): the C2 logits [4 x 1024, 4096] f32 and a vocabulary-sized bf16 case
    from ptdeco_amd import _hip
    lib = _hip.load()

    def nsr_line(rows, chans, dt, code):
        yy = torch.randn(rows, chans, device=device).to(dt)
        xx = (yy.float() + 0.01).to(dt)
        outp = torch.empty(1, dtype=torch.float64, device=device)
        ws = torch.empty(lib.ptd_nsr_workspace_bytes(rows, chans), dtype=torch.uint8, device=device)
        st = torch.cuda.current_stream(device).cuda_stream
        _hip.check(lib.ptd_nsr_workspace_init(ws.data_ptr(), ws.numel(), st), "ptd_nsr_workspace_init")
        args = (xx.data_ptr(), yy.data_ptr(), rows, chans, code, 1e-3, outp.data_ptr(), ws.data_ptr(), ws.numel(), st)
        tt = time_events(lambda: lib.ptd_nsr(*args), iters=50)
        nbytes = 2 * yy.numel() * yy.element_size()
        return {"ms": tt * 1e3, "algorithmic_bytes": nbytes, "gbps": nbytes / tt / 1e9, "frac_of_hbm_peak": nbytes / tt / PEAK_HBM,
                "shape": [rows, chans], "launches": "stream kernel + 64-channel final kernel"}
    lines["nsr_f32"] = nsr_line(t_tok, N_FEAT, torch.float32, _hip.F32)
    lines["nsr_bf16_vocab"] = nsr_line(2048, 128256, torch.bfloat16, _hip.BF16)
    # MFMA utilisation from the committed rocprofv3 PMC pass over the same kernels (tools/pmc_driver mfma)
    pmc = pmc_file("pmc_mfma_r*.json", ("gemm_f32.hip", "gemm_bf16.hip"))
    if pmc:
        kern = pmc["data"]["kernels"]
        for line, key in (("syrk_f32_f64acc", "syrk_f32_mixed_kernel"), ("gemm_f32_nt", "gemm_f32_nt_8ph_kernel"),
                          ("syrk_bf16_f64acc", "syrk_bf16"), ("gemm_bf16_nt", "gemm_bf16_nt_8ph")):
            for name, c in kern.items():
                if name.startswith(key) and "MfmaUtil" in c and line in lines:
                    lines[line]["mfma_util_pmc_percent"] = c["MfmaUtil"]
                    lines[line]["mfma_util_source"] = pmc["source"]
                    lines[line]["mfma_util_stale"] = pmc["stale"]
    return lines


def decomposed_forward_lines(device, t_rows=16384, full=True):
    """BASELINE configs[4]: rank-r two-GEMM forward vs dense 4096x4096, bf16, T = t_rows rows (4096, 16384,
    65536).  full=False: the package's pair, the library pair and the dense layer on one resident input only."""
    from ptdeco_amd import ops

    g = torch.Generator().manual_seed(5)
    x = torch.randn(t_rows, N_FEAT, generator=g).bfloat16().to(device)
    w = (torch.randn(N_FEAT, N_FEAT, generator=g) / 64).bfloat16().to(device)
    dense_t = time_events(lambda: ops.matmul(x, w.T), iters=10)
    # the same dense layer through torch (hipBLASLt): the baseline a user of the reference would have
    lib_t = time_events(lambda: torch.nn.functional.linear(x, w), iters=10)
    out = {"rows": t_rows, "dense_ms": dense_t * 1e3, "dense_tflops": 2 * t_rows * N_FEAT * N_FEAT / dense_t / 1e12,
           "dense_torch_hipblaslt_ms": lib_t * 1e3,
           "dense_torch_hipblaslt_tflops": 2 * t_rows * N_FEAT * N_FEAT / lib_t / 1e12}
    import itertools
    rot = itertools.cycle([x] + [torch.randn(t_rows, N_FEAT, device=device).bfloat16() for _ in range(5 if full else 0)])
    for r in (256, 512, 1024):
        a = (torch.randn(r, N_FEAT, generator=g) / 64).bfloat16().to(device)
        b = (torch.randn(N_FEAT, r, generator=g) / r**0.5).bfloat16().to(device)
        t = time_events(lambda: ops.lowrank_forward(x, a, b, None), iters=10)
        if not full:
            lib_pair = time_events(lambda: torch.nn.functional.linear(torch.nn.functional.linear(x, a), b), iters=10)
            fl = 2 * t_rows * r * 2 * N_FEAT
            by = 2 * (2 * t_rows * N_FEAT + 2 * r * N_FEAT)
            out[f"r{r}"] = {"ms": t * 1e3, "gflops": fl / t / 1e9, "speedup_vs_dense": dense_t / t,
                            "torch_hipblaslt_pair_ms": lib_pair * 1e3, "frac_of_bf16_mfma_peak": fl / t / PEAK_BF16_MFMA,
                            "hbm_gbps_algorithmic": by / t / 1e9, "frac_of_hbm_peak": by / t / PEAK_HBM}
            continue
        # the same with the input rotating over buffers larger than the 256-MB Infinity Cache: x comes from HBM at
        # every launch, as in a forward pass of a model (the single-buffer loop above re-reads a cached x)
        # (time_events reports the median of three loops: the first launches of a shape can hit the caching allocator's
        # first allocation of the 134-MB output blocks)
        t_cold = time_events(lambda: ops.lowrank_forward(next(rot), a, b, None), iters=12)
        lib_cold = time_events(lambda: torch.nn.functional.linear(torch.nn.functional.linear(next(rot), a), b), iters=12)
        fl = 2 * t_rows * r * 2 * N_FEAT
        by = 2 * (2 * t_rows * N_FEAT + 2 * r * N_FEAT)
        # the same pair as two torch.nn.functional.linear calls (hipBLASLt): what apply_decompose_config_in_place's
        # Sequential(Linear, Linear) costs without the fused module
        lib_pair = time_events(lambda: torch.nn.functional.linear(torch.nn.functional.linear(x, a), b), iters=10)
        # the module apply_decompose_config_in_place installs (LowRankLinear): always ptd_lowrank_forward
        from ptdeco_amd.lowrank import fuse_pair
        pair = torch.nn.Sequential(torch.nn.Linear(N_FEAT, r, bias=False), torch.nn.Linear(r, N_FEAT, bias=False))
        pair = fuse_pair(pair).to(device).bfloat16()
        with torch.no_grad():
            pair[0].weight.copy_(a); pair[1].weight.copy_(b)
            mod_t = time_events(lambda: pair(x), iters=10)
        out[f"r{r}"] = {"ms": t * 1e3, "gflops": fl / t / 1e9, "speedup_vs_dense": dense_t / t,
                        "frac_of_hbm_peak": by / t / PEAK_HBM,
                        "speedup_vs_dense_torch_hipblaslt": lib_t / t, "torch_hipblaslt_pair_ms": lib_pair * 1e3,
                        "ms_rotating_inputs": t_cold * 1e3, "torch_hipblaslt_pair_ms_rotating_inputs": lib_cold * 1e3,
                        "module_ms": mod_t * 1e3,
                        "module_runs": "package kernels",
                        "frac_of_bf16_mfma_peak": fl / t / PEAK_BF16_MFMA, "hbm_gbps_algorithmic": by / t / 1e9}
    # MFMA utilisation of the two rank-256 kernels from the committed counter pass (tools/pmc_driver mfma)
    pmc = pmc_file("pmc_mfma_r*.json", ("gemm_f32.hip", "gemm_bf16.hip")) if full else None
    if pmc:
        kern = pmc["data"]["kernels"]
        for label, key in (("x_At", "gemm_bf16_nt_glds_kernel<0, 4>"), ("h_Bt", "gemm_bf16_shortk4_kernel<4>")):
            if key in kern and "MfmaUtil" in kern[key]:
                out["r256"][f"mfma_util_pmc_percent_{label}"] = kern[key]["MfmaUtil"]
        out["r256"]["mfma_util_source"] = pmc["source"]
        out["r256"]["mfma_util_stale"] = pmc["stale"]
    return out


# ---------------------------------------------------------------------------------------------- BASELINE configs[3]
D_MODEL, D_KV, D_FF = 4096, 1024, 14336


class LlamaBlock(torch.nn.Module):
    """RMSNorm -> {q, k, v} -> (q + repeat4(k) + repeat4(v)) -> o -> residual;
    RMSNorm -> down(silu(gate) * up) -> residual, at the Llama-3-8B widths."""

    def __init__(self):
        super().__init__()
        mk = lambda i, o: torch.nn.Linear(i, o, bias=False)  # noqa: E731
        self.q, self.k, self.v, self.o = mk(D_MODEL, D_MODEL), mk(D_MODEL, D_KV), mk(D_MODEL, D_KV), mk(D_MODEL, D_MODEL)
        self.gate, self.up, self.down = mk(D_MODEL, D_FF), mk(D_MODEL, D_FF), mk(D_FF, D_MODEL)

    @staticmethod
    def norm(x):
        return x * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + 1e-6).to(x.dtype)

    def forward(self, x):
        h = self.norm(x)
        rep = D_MODEL // D_KV
        x = x + self.o(self.q(h) + self.k(h).repeat(1, 1, rep) + self.v(h).repeat(1, 1, rep))
        h = self.norm(x)
        return x + self.down(torch.nn.functional.silu(self.gate(h)) * self.up(h))


class LlamaStack(torch.nn.Module):
    def __init__(self, blocks: int):
        super().__init__()
        self.blocks = torch.nn.ModuleList(LlamaBlock() for _ in range(blocks))
        self.head = torch.nn.Linear(D_MODEL, D_MODEL, bias=False)

    def forward(self, b):
        x = b["x"]
        for blk in self.blocks:
            x = blk(x)
        return self.head(x)


class OneLinear(torch.nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.lin = torch.nn.Linear(n_in, n_out, bias=False)

    def forward(self, d):
        return self.lin(d["x"])


def seq_ce(batch, logits):
    return torch.nn.functional.cross_entropy(logits.float().reshape(-1, logits.shape[-1]), batch["targets"].reshape(-1),
                                             reduction="none")


ROUTES = {0: "jacobi", 1: "tridiagonal (direct)", 2: "two-stage tridiagonal", 3: "filtered subspace iteration"}


def eigh_route_block(p):
    """Roofline-style block of one profiled eigendecomposition (ops.EIGH_PROFILE entry), per route."""
    n, k, t = p["n"], p["k"], p["total_ms"] * 1e-3
    algo = 4.0 / 3.0 * n**3 + 2.0 * n * n * k
    out = {"route": ROUTES[p["method"]], "n": n, "k": k, "ms": p["total_ms"],
           "algorithmic_tflops": algo / t / 1e12, "frac_of_f64_mfma_peak_on_algorithmic_flops": algo / t / PEAK_F64_MFMA}
    if p["method"] == 3:
        m_blk, nprod = p["launches"][2], p["launches"][1]
        out.update({"bound": "mfma", "products_with_C": nprod, "subspace": m_blk,
                    "solver_frac": nprod * 2.0 * n * n * m_blk / t / PEAK_F64_MFMA,
                    "phases_ms": {"lanczos": p["ms"][0], "filter_rounds": p["ms"][1], "rayleigh_ritz_eigh": p["ms"][2],
                                  "ritz_products_residuals": p["ms"][3]}})
    elif p["method"] == 1:
        all_bytes = sum(8.0 * (n - j - 1) * (n - j - 2) for j in range(n - 1))
        red_ms = p["ms"][0] + p["ms"][1]
        out.update({"bound": "hbm", "solver_frac": all_bytes / (red_ms * 1e-3) / PEAK_HBM,
                    "phases_ms": {"symv_launches": p["ms"][0], "other_reduction": p["ms"][1],
                                  "eigenpairs_backtransform": p["ms"][3]}})
        if p["launches"][0] > 0 and p["ms"][0] > 0:
            out["symv"] = {"launches": p["launches"][0], "gbps_algorithmic": p["work"][0] / (p["ms"][0] * 1e-3) / 1e9,
                           "frac_of_hbm_peak": p["work"][0] / (p["ms"][0] * 1e-3) / PEAK_HBM}
    return out


def llama_shape_lines(device, steps=3):
    """BASELINE configs[3] per shape: dwain on ONE layer of each Llama-3-8B linear shape, [1, 2048, n_in] tokens per
    step, D = 8, M = 2, f64 decomposition; f32 and bf16 model; `steps` timed steps each (+ one warm-up and one profiled)."""
    import ptdeco_amd
    from ptdeco_amd import ops

    out = {"tokens_per_step": 2048, "D": 8, "M": 2, "timed_steps": steps}
    shapes = (("q_o", 4096, 4096, 64), ("k_v", 4096, 1024, 64), ("gate_up", 4096, 14336, 64), ("down", 14336, 4096, 32))
    for dt in (torch.float32, torch.bfloat16):
        block, total = {}, 0.0
        for name, n_in, n_out, count in shapes:
            g = torch.Generator(device=device).manual_seed(1)
            m0 = OneLinear(n_in, n_out).to(device)
            with torch.no_grad():
                m0.lin.weight.copy_(torch.randn(n_out, n_in, generator=g, device=device) / n_in**0.5)
            m0.to(dt)
            scale = torch.logspace(0, -2, n_in, device=device)
            xs = [(torch.randn(1, 2048, n_in, generator=g, device=device) * scale).to(dt) for _ in range(10)]
            with torch.no_grad():
                bt = [{"x": x, "targets": m0({"x": x}).argmax(-1)} for x in xs]
            kw = dict(num_data_steps=8, num_metric_steps=2, nsr_final_threshold=1.0, decompose_in_float64=True)

            def step():
                m = copy.deepcopy(m0)
                return ptdeco_amd.dwain.decompose_in_place(module=m, device=device, data_iterator=itertools.cycle(bt),
                                                           loss_fn=seq_ce, metric_iterator=itertools.cycle(bt[8:]),
                                                           finetune_fn=lambda mm, d, n: mm, **kw)
            step()
            torch.cuda.synchronize()
            marks = []
            for _ in range(steps):
                t0 = time.perf_counter()
                step()
                torch.cuda.synchronize()
                marks.append((time.perf_counter() - t0) * 1e3)
            ops.EIGH_PROFILE = []
            step()
            torch.cuda.synchronize()
            prof, ops.EIGH_PROFILE = ops.EIGH_PROFILE, None
            med = sorted(marks)[len(marks) // 2]
            line = {"n_in": n_in, "n_out": n_out, "ms_per_layer": med, "step_ms": [round(v, 3) for v in marks],
                    "layers_per_s": 1e3 / med}
            if prof:
                line["eigh"] = eigh_route_block(prof[0])
            else:
                line["eigh"] = {"route": "factored: W Ex W^T through an n_in-sized problem (ptd_eigh_factored); its inner "
                                         "eigensolver is the direct route at n = n_in, k = n_in / 2"}
            block[name] = line
            total += med * 1e-3 * count
            del m0, xs, bt
        block["extrapolated_224_layers_s"] = total
        block["extrapolated_layers_per_s_1gpu"] = 224 / total
        block["note"] = "kernel-side per-shape figure (one layer alone); the end-to-end figure is c4_block / tools/c4_stack.py"
        out["f32" if dt == torch.float32 else "bf16"] = block
    return out


C4_BLOCK_KW = dict(num_data_steps=8, num_metric_steps=2, nsr_final_threshold=1.0, min_rank=32, trade_off_factor=20.0,
                   reduction_factor=0.5, max_accepted_ppl_diff=0.4, decompose_in_float64=True,
                   blacklisted_module_names=["head"], precomputing_covariance_num_splits=1)


def llama_block_lines(device, blocks=1, dtypes=(torch.float32, torch.bfloat16), timed=5):
    """BASELINE configs[3] in small, end to end: dwain on ONE full-width Llama-3-8B block (q, k, v, o, gate, up, down at
    4096 / 1024 / 14336 + the blacklisted head), [1, 2048, 4096] batches, D = 8, M = 2, precompute pass (one split: the
    seven eigendecompositions run as concurrent chains), thresholds under which layers ARE replaced, so that the
    replace -> next-layer-sees-the-changed-model path runs (dwain.py:779-787)."""
    import ptdeco_amd
    from ptdeco_amd import _engine as eng

    out = {}
    # (a layer's share of the parameters shrinks with the depth: the trade-off factor scales with the number of blocks)
    kw = dict(C4_BLOCK_KW, trade_off_factor=C4_BLOCK_KW["trade_off_factor"] * blocks)
    for dt in dtypes:
        g = torch.Generator(device=device).manual_seed(0)
        with torch.device(device):
            model0 = LlamaStack(blocks)
        with torch.no_grad():
            for prm in model0.parameters():
                prm.copy_(torch.randn(prm.shape, generator=g, device=device) / prm.shape[1] ** 0.5)
        model0.to(dt)
        scale = torch.logspace(0, -2, D_MODEL, device=device)
        xs = [(torch.randn(1, 2048, D_MODEL, generator=g, device=device) * scale).to(dt) for _ in range(12)]
        with torch.no_grad():
            bt = [{"x": x, "targets": model0({"x": x}).argmax(-1)} for x in xs]

        def step(trace=None):
            m = copy.deepcopy(model0)
            return ptdeco_amd.dwain.decompose_in_place(module=m, device=device, data_iterator=itertools.cycle(bt),
                                                       loss_fn=seq_ce, metric_iterator=itertools.cycle(bt[8:]),
                                                       finetune_fn=lambda mm, d, n: mm, trace=trace, **kw)
        step()
        torch.cuda.synchronize()
        marks = []
        for _ in range(timed):
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            marks.append((time.perf_counter() - t0) * 1e3)
        eng.PHASES = eng.PhaseTimer()
        trace = []
        t0 = time.perf_counter()
        cfg = step(trace)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        ph, eng.PHASES = eng.PHASES.totals_ms(), None
        ph["other_host_and_gaps"] = max(0.0, wall - sum(ph.values()))
        med = sorted(marks)[len(marks) // 2]
        out["f32" if dt == torch.float32 else "bf16"] = {
            "layers": 7 * blocks, "ms_per_step": med, "ms_per_step_max": max(marks), "ms_per_step_min": min(marks),
            "ms_per_block": med / blocks, "step_ms": [round(v, 1) for v in marks], "layers_per_s": 7e3 * blocks / med,
            "spread": (max(marks) - min(marks)) / med, "timed_steps": timed, "statistic": "median of step_ms",
            "phases_ms": {k: round(v, 1) for k, v in ph.items()}, "candidates_evaluated": len(trace),
            "replaced": {k: v["__meta__"]["proportion"] for k, v in cfg.items()}}
        del model0, xs, bt
    out["workload"] = ("dwain.decompose_in_place, %d Llama-3-8B-width block(s) (%d layers) + blacklisted head, [1, 2048, 4096] "
                       "batches, D = 8, M = 2, precompute pass (1 split), trade_off_factor %g, max_accepted_ppl_diff 0.4; "
                       "layers are replaced as the search goes, so the later layers' metric forwards run the changed model"
                       % (blocks, 7 * blocks, kw["trade_off_factor"]))
    return out


def pmc_traffic(n):
    """roofline.traffic: HBM-side bytes per SYMV launch from the committed rocprofv3 PMC passes
    (profiles/pmc_symv_rNN.json, made by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE
    runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be read
    from inside this process, so the figure is the latest committed pass for the same matrix order."""
    pmc = pmc_file("pmc_symv_r*.json", ("eigh_tridiag.hip",))
    if pmc and pmc["data"].get("n") == n:
        d = pmc["data"]
        return {"traffic": d["traffic_bytes_per_launch"],
                "traffic_source": pmc["source"] + ": (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch, "
                                  "%.3f x the algorithmic bytes" % d["traffic_over_algorithmic"],
                "traffic_stale": pmc["stale"]}
    return {"traffic": None}


def cpu_baseline():
    """The CPU oracle on the same C2 workload (1 layer), host cores of this box."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ptdeco_oracle as orc

    # physical cores of this box: torch defaults to the logical count, which oversubscribes MKL
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        physical = os.cpu_count()
    # ... and a cgroup CPU quota below the visible count throttles a pool sized by it (oracle/cpu_quota.py)
    from cpu_quota import usable_cpus
    usable = usable_cpus()
    cores = max(1, min(physical, usable))
    torch.set_num_threads(cores)
    model, data, metric = make_workload(1, "cpu", D_STEPS, 7 * M_STEPS)
    cpu = torch.device("cpu")
    data, metric = with_targets(model, data, cpu), with_targets(model, metric, cpu)
    t0 = time.perf_counter()
    cfg = orc.dwain_decompose(module=model, data_iterator=itertools.cycle(data), loss_fn=ce_loss,
                              metric_iterator=itertools.cycle(metric), finetune_fn=None, **DWAIN_KW)
    dt = time.perf_counter() - t0
    prop = cfg["layers.0"]["__meta__"]["proportion"] if cfg else 1.0
    return {"value": 1.0 / dt, "unit": "layers/s", "cores": cores, "kind": "port",
            "workload": "c2_single_layer (BASELINE configs[1]: dwain of ONE nn.Linear(4096,4096) f32, D=4, M=2) -- NOT the "
                        "8-layer stack `value` is quoted on; the GPU figure for the same workload is gpu_same_workload",
            "physical_cores": physical, "logical_cpus": os.cpu_count(), "usable_cpus": usable,   # usable = affinity and cgroup quota
            "sample": f"the c2_single_layer workload once = 1 layer ({dt:.1f} s, torch threads = {cores}); one eighth of "
                      f"the layers of the stack `value` times, whose per-layer CPU cost is higher (deeper forwards); "
                      f"chosen proportion {prop}"}


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of this one
    (python -m torch.distributed.run, rendezvous on 127.0.0.1) before anything here has touched the GPU, pass
    their output through (rank 0 prints the JSON line) and return their exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip kernel / forward / cpu side measurements")
    ap.add_argument("--no-bf16-stack", action="store_true", help="skip the second (bf16 model) timed run")
    ap.add_argument("--no-weak-family", action="store_true", help="N > 1: skip the round-3 family (N layers on N GPUs)")
    ap.add_argument("--no-c4", action="store_true", help="skip the Llama-3-8B shape / block extras")
    ap.add_argument("--workload", choices=("stack", "c2"), default="stack",
                    help="c2: time BASELINE configs[1] alone (one layer; for kernel traces); default: the fixed stack")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch.distributed as dist

    # PTD_BENCH_REHEARSE=1: rehearsal of the N > 1 code path on a one-GPU box -- every rank on cuda:0,
    # gloo transport (RCCL refuses two ranks on one device).  Not a measurement.
    rehearse = os.environ.get("PTD_BENCH_REHEARSE") == "1"
    device = torch.device("cuda", 0 if rehearse else local)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import ptdeco_amd
    from ptdeco_amd import ops

    # ONE fixed workload at every N (strong scaling): a stack of 8 nn.Linear(4096, 4096), dwain with the precompute
    # pass (1 split), D = 8 calibration steps.  Beside it at N = 1: BASELINE configs[1] (C2, one layer, D = 4) itself.
    def build(n_layers, d_steps):
        model, data, metric = make_workload(n_layers, device, d_steps, 7 * M_STEPS)
        model.to(device)
        return (model, with_targets(model, data, device), with_targets(model, metric, device),
                dict(DWAIN_KW, num_data_steps=d_steps))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def family(work, model_dtype):
        """one_step of the workload `work` with the model and its activations in `model_dtype`."""
        model32, data32, metric32, kw = work
        if model_dtype == torch.float32:
            model0, data, metric, loss = model32, data32, metric32, ce_loss
        else:
            model0 = copy.deepcopy(model32).to(model_dtype)
            data = [{"x": b["x"].to(model_dtype), "targets": b["targets"]} for b in data32]
            metric = [{"x": b["x"].to(model_dtype), "targets": b["targets"]} for b in metric32]
            loss = lambda b, y: ce_loss(b, y.float())  # noqa: E731

        def one_step():
            model = copy.deepcopy(model0)
            return ptdeco_amd.dwain.decompose_in_place(
                module=model, device=device, data_iterator=itertools.cycle(data), loss_fn=loss,
                metric_iterator=itertools.cycle(metric), finetune_fn=lambda m, d, names: m,
                precomputing_covariance_num_splits=1, **kw)
        return one_step

    def timed(one_step):
        """W untimed steps, then exactly K steps between barrier + synchronize; max over ranks.
        Returns (seconds, last config, per-step host durations in ms)."""
        cfg = None
        for _ in range(args.warmup):
            cfg = one_step()
        barrier()
        t0 = time.perf_counter()
        marks = []
        for _ in range(args.steps):
            cfg = one_step()
            marks.append(time.perf_counter())     # (no synchronisation added: a step ends on the host's last decision)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, cfg, [round((b - a) * 1e3, 3) for a, b in zip([t0] + marks[:-1], marks)]

    def kept(cfg):
        return {k: v["__meta__"]["proportion"] for k, v in cfg.items()}

    if args.workload == "c2":
        # profiling aid: BASELINE configs[1] alone under the same protocol (python bench.py --workload c2 --no-extras)
        c2 = build(1, D_STEPS)
        dt, cfg, marks = timed(family(c2, torch.float32))
        if rank == 0:
            print(json.dumps({"metric": "layers decomposed/sec (incl. covariance+SVD)", "value": args.steps / dt,
                              "unit": "layers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": dt / args.steps * 1e3, "step_ms": marks, "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": "BASELINE configs[1]: dwain decompose_in_place of ONE "
                                                     "nn.Linear(4096,4096) f32 (--workload c2)", "ranks_kept": kept(cfg)}}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    stack = build(STACK_LAYERS, STACK_D_STEPS)
    stack_step = family(stack, torch.float32)
    dt, cfg, marks = timed(stack_step)
    what = ("dwain decompose_in_place of a FIXED stack of %d x nn.Linear(4096,4096) %s at every N (BASELINE configs[1]'s "
            "layer, eight of them chained so that the same work is dealt to 1/2/4/8 GPUs as configs[3] shards its "
            "stack), precompute pass (1 split), T=4x1024 tokens/batch, D=%d, M=2, 6 evaluated candidate ranks per "
            "layer, f64 covariance+eigh%s")
    comm = ("; calibration steps and (candidate, batch) pairs dealt to the ranks, packed-triangle covariance sums "
            "reduced to the layer owners + eigenvector broadcast over RCCL") if world > 1 else ""

    result = {
        "metric": "layers decomposed/sec (incl. covariance+SVD)",
        "value": STACK_LAYERS * args.steps / dt,
        "unit": "layers/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "step_ms": marks,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": what % (STACK_LAYERS, "f32", STACK_D_STEPS, comm),
                   "model_dtype": "float32",
                   "layers_per_step": STACK_LAYERS, "parallelism": f"dp{world}" if world > 1 else "single",
                   "metric_forwards": ("every (candidate, batch) pair runs the stack twice as the reference does; the "
                                       "second run reuses the layer outputs ahead of the analysed layer that the first "
                                       "run of the SAME pair just computed (PTD_PREFIX_MEMO_MB=0 switches that off); "
                                       "nothing is kept across pairs, layers or steps"),
                   "ranks_kept": kept(cfg)},
    }
    c2 = None
    if world == 1:
        # BASELINE configs[1] / SURVEY C2 exactly: ONE layer, D = 4 -- the workload of the roofline / eigh / phases /
        # cpu_baseline blocks below, under the same K / W protocol
        c2 = build(1, D_STEPS)
        dt2, cfg2, marks2 = timed(family(c2, torch.float32))
        result["c2_single_layer"] = {
            "value": args.steps / dt2, "unit": "layers/s", "ms_per_step": dt2 / args.steps * 1e3, "step_ms": marks2,
            "workload": "BASELINE configs[1]: dwain decompose_in_place of ONE nn.Linear(4096,4096) f32, precompute pass "
                        "(1 split), T=4x1024 tokens/batch, D=4, M=2, 6 evaluated candidate ranks, f64 covariance+eigh",
            "ranks_kept": kept(cfg2)}
    if not args.no_bf16_stack:
        # the fixed stack with a bf16 model ("model fp32 for parity, bf16 for throughput"), same K / W and
        # the same barrier protocol: the method's own whole-model forwards (16 GEMMs per (candidate, batch) pair) are
        # small next to the covariance + eigensolver part here
        dt16, cfg16, marks16 = timed(family(stack, torch.bfloat16))
        result["bf16_stack"] = {"value": STACK_LAYERS * args.steps / dt16, "unit": "layers/s",
                                "ms_per_step": dt16 / args.steps * 1e3, "step_ms": marks16, "model_dtype": "bfloat16",
                                "workload": what % (STACK_LAYERS, "bf16", STACK_D_STEPS, comm), "ranks_kept": kept(cfg16)}
    if world > 1 and not args.no_weak_family:
        # the round-3 family (a stack of N layers on N GPUs, D = max(4, N)): its own forwards grow as N^2 per step
        weak = build(world, max(D_STEPS, world))
        dtw, cfgw, marksw = timed(family(weak, torch.float32))
        result["weak_family"] = {"value": world * args.steps / dtw, "unit": "layers/s", "scaling": "weak",
                                 "ms_per_step": dtw / args.steps * 1e3, "step_ms": marksw,
                                 "workload": "stack of %d x nn.Linear(4096,4096) f32, one layer per GPU, D=%d, M=2"
                                             % (world, max(D_STEPS, world)), "ranks_kept": kept(cfgw)}
        del weak

    # the step the phase / eigensolver blocks describe: C2 at N = 1, the fixed stack otherwise
    one_step = family(c2, torch.float32) if c2 is not None else stack_step
    described = "c2_single_layer" if c2 is not None else "the fixed stack (`value`)"
    from ptdeco_amd import _engine as _eng_stats
    result["config"]["chain_streams"] = dict(_eng_stats.CHAIN_STREAM_STATS)     # (run_concurrently's stream checks so far)
    if c2 is not None:
        result["config"]["c2_single_layer_layers_per_s"] = result["c2_single_layer"]["value"]
        result["config"]["c2_single_layer_ms_per_step"] = result["c2_single_layer"]["ms_per_step"]
        result["config"]["blocks_of"] = ("value / ms_per_step: the fixed 8-layer stack; roofline, eigh, phases_ms, kernels and "
                                         "cpu_baseline: c2_single_layer (BASELINE configs[1], one layer) -- each block names "
                                         "its workload in `of` / `workload`")
    prof = []
    if not args.no_extras:
        # two extra, untimed steps.  (1) phase spans of the step on the device timeline (A accumulate, B
        # eigh, C factors, D metrics, comm); (2) per-phase HIP-event timings inside the eigensolver.  Every rank takes
        # part (the steps contain collectives); rank 0 keeps the numbers.
        from ptdeco_amd import _engine as eng
        eng.PHASES = eng.PhaseTimer()
        barrier()
        t0p = time.perf_counter()
        one_step()
        barrier()
        wall_p = (time.perf_counter() - t0p) * 1e3
        ph, eng.PHASES = eng.PHASES.totals_ms(), None
        ph["other_host_and_gaps"] = max(0.0, wall_p - sum(ph.values()))
        ph["step_wall_ms"] = wall_p
        result["phases_ms"] = {k: round(v, 3) for k, v in ph.items()}
        result["phases_ms"]["of"] = described
        if c2 is not None:
            # the same split for the workload `value` is quoted on (one more untimed step of the fixed stack)
            eng.PHASES = eng.PhaseTimer()
            barrier()
            t0p = time.perf_counter()
            stack_step()
            barrier()
            wall_s = (time.perf_counter() - t0p) * 1e3
            ph, eng.PHASES = eng.PHASES.totals_ms(), None
            ph["other_host_and_gaps"] = max(0.0, wall_s - sum(ph.values()))
            ph["step_wall_ms"] = wall_s
            result["stack_phases_ms"] = {k: round(v, 1) for k, v in ph.items()}
            result["stack_phases_ms"]["of"] = "the fixed stack (`value`)"
        ops.EIGH_PROFILE = []
        one_step()
        barrier()
        prof, ops.EIGH_PROFILE = ops.EIGH_PROFILE, None
    if rank == 0 and not args.no_extras:
        if prof:
            p = prof[0]
            n = p["n"]
            t = p["total_ms"] * 1e-3
            k = p["k"]  # eigenvectors formed: the largest candidate rank that is evaluated (n / 4 for a square layer)
            algo_flops = 4.0 / 3.0 * n**3 + 2.0 * n * n * k
            kl = {}
            if p["method"] == 1:
                # tridiagonal route: the dominant kernel is the per-column SYMV, bound by the stream
                # of the trailing matrix (8/3 n^3 bytes per matrix for a one-stage reduction)
                ms, cnt, byts = p["ms"][0], p["launches"][0], p["work"][0]
                tr = pmc_traffic(n)
                hw = {}
                if tr.get("traffic"):
                    # what the memory side actually moved per launch (the symmetric kernel reads one triangle) over the
                    # same launch time: the hardware-side bandwidth fraction, below `frac` by construction
                    hw = {"hw_achieved": tr["traffic"] * cnt / (ms * 1e-3) / 1e9,
                          "hw_frac": tr["traffic"] * cnt / (ms * 1e-3) / PEAK_HBM}
                # the whole reduction against the same bound: 8/3 n^3 bytes at the HBM peak vs the time from the first
                # launch of the reduction to the tridiagonal matrix (SYMV + per-column kernels + rank-2k updates + gaps)
                red_ms = p["ms"][0] + p["ms"][1]
                all_bytes = sum(8.0 * (n - j - 1) * (n - j - 2) for j in range(n - 1))  # every column, resident ones too
                result["roofline"] = {
                    "bound": "hbm", "achieved": byts / (ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                    "frac": byts / (ms * 1e-3) / PEAK_HBM, **tr, **hw,
                    "solver_frac": all_bytes / (red_ms * 1e-3) / PEAK_HBM,
                    "solver_note": "solver_frac = the algorithmic bytes of ALL columns (8/3 n^3) over the WHOLE reduction "
                                   "time (%.1f ms: SYMV launches + per-column kernels + rank-2k updates + the resident "
                                   "kernels + launch gaps); the full ptd_eigh call takes %.1f ms" % (red_ms, p["total_ms"]),
                    "kernel": "sytrd_symv2_kernel / sytrd_symv_kernel (Householder tridiagonalisation, one SYMV launch "
                              "per column down to a trailing order of 3072, symmetric lower-triangle tiles; the last "
                              "3072 columns run in three launches, resident in registers -- every CU down to 768 "
                              "columns, one XCD for the rest -- and read nothing from HBM)",
                    "n": n, "launches": cnt, "avg_launch_us": ms / max(cnt, 1) * 1e3,
                    "algorithmic_bytes_per_launch": byts / max(cnt, 1),
                    "note": "algorithmic bytes = 8 (n-j-1)(n-j-2) per column j (rows j+1.., columns j+2.. of the "
                            "trailing matrix, f64), summed over the columns that have a SYMV launch (all columns: "
                            "8/3 n^3, SURVEY 8d: the stream of a one-stage SYMV); the symmetric kernel reads only the "
                            "lower triangle, so its measured traffic is below that figure"}
                kl["sytrd_symv_kernels"] = {"launches": cnt, "avg_us": ms / max(cnt, 1) * 1e3, "total_ms": ms,
                                           "gbps": byts / (ms * 1e-3) / 1e9}
                kl["sytrd_other_per_column"] = {"total_ms": p["ms"][1],
                                                "note": "alpha kernels + rank-2k updates + resident kernels + launch gaps"}
                kl["eigvals_invit_backtransform"] = {"total_ms": p["ms"][3]}
            elif p["method"] == 0:
                names = ("jac_gram_kernel", "jac_inner_kernel", "jac_update_kernel")
                for i, nm in enumerate(names):
                    ms, cnt, fl = p["ms"][i], p["launches"][i], p["work"][i]
                    kl[nm] = {"launches": cnt, "avg_us": ms / max(cnt, 1) * 1e3, "total_ms": ms}
                    if fl:
                        kl[nm]["executed_tflops"] = fl / (ms * 1e-3) / 1e12
                        kl[nm]["frac_of_f64_mfma_peak"] = fl / (ms * 1e-3) / PEAK_F64_MFMA
                result["roofline"] = {"bound": "mfma", "achieved": algo_flops / t / 1e12, "peak": PEAK_F64_MFMA / 1e12,
                                      "unit": "TFLOP/s", "frac": algo_flops / t / PEAK_F64_MFMA, "traffic": None,
                                      "kernel": "ptd_eigh (one-sided block Jacobi: jac_gram + jac_inner + jac_update)",
                                      "n": n, "sweeps": p["sweeps"], "algorithmic_flops": algo_flops}
            if p["method"] == 3:
                # filtered subspace iteration (the default route for k <= n / 3): every large step is an f64 product on
                # the matrix cores; the dominant kernel is the product C X of the filter (n x n x m), timed here on the
                # same shapes with HIP events on the launch stream
                m_blk = p["launches"][2]
                nprod = p["launches"][1]
                gx = torch.Generator(device=device).manual_seed(11)
                cm = torch.randn(n, n, generator=gx, device=device, dtype=torch.float64)
                xm = torch.randn(n, m_blk, generator=gx, device=device, dtype=torch.float64)
                t_prod = time_events(lambda: ops.matmul(cm, xm), iters=20)
                fl = 2.0 * n * n * m_blk
                del cm, xm
                # HBM-side bytes per launch and the matrix-pipe busy share from the committed counter passes over the
                # same kernel (tools/pmc_driver eigh, tools/pmc_filtered_summary.py); counters cannot be read in-process
                trf = {"traffic": None}
                pmc = pmc_file("pmc_gemm_f64_r*.json", ("gemm_f64.hip", "eigh_filtered.hip"))
                if pmc and pmc["data"].get("traffic_bytes_per_launch"):
                    dd = pmc["data"]
                    trf = {"traffic": dd["traffic_bytes_per_launch"],
                           "traffic_source": pmc["source"] + ": (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch, %.2f x the "
                                             "algorithmic bytes 8 (n^2 + 2 n m)" % dd["traffic_over_algorithmic"],
                           "traffic_stale": pmc["stale"]}
                    if dd.get("mfma_busy_over_cu_busy_x4_percent") is not None:
                        trf["mfma_util_pmc_percent"] = dd["mfma_busy_over_cu_busy_x4_percent"]
                result["roofline"] = {
                    "bound": "mfma", "achieved": fl / t_prod / 1e12, "peak": PEAK_F64_MFMA / 1e12, "unit": "TFLOP/s",
                    "frac": fl / t_prod / PEAK_F64_MFMA, **trf,
                    "kernel": "gemm_f64_glds_kernel<5, false> (C X of the Chebyshev filter: %d x %d x %d f64, "
                              "v_mfma_f64_16x16x4_f64, LDS-DMA staged 128 x 80 tiles)" % (n, n, m_blk),
                    "n": n, "m": m_blk, "launches": nprod, "avg_launch_us": t_prod * 1e6,
                    "algorithmic_flops_per_launch": fl,
                    "solver_frac": nprod * fl / (p["total_ms"] * 1e-3) / PEAK_F64_MFMA,
                    "solver_note": "solver_frac = the flop of the %d products with C over the WHOLE ptd_eigh_topk call "
                                   "(%.1f ms: Lanczos bounds %.1f, filter rounds incl. Cholesky-QR passes %.1f, the %d x %d "
                                   "Rayleigh-Ritz eigenproblem %.1f, Ritz products + residual check %.1f)"
                                   % (nprod, p["total_ms"], p["ms"][0], p["ms"][1], m_blk, m_blk, p["ms"][2], p["ms"][3]),
                    "note": "algorithmic flops of one launch = 2 n^2 m; the solver's own count (SURVEY 8d) is 4/3 n^3 + "
                            "2 n^2 k for a direct reduction -- the filtered route executes more flops (%d products) on "
                            "the matrix cores instead of a latency-bound Householder reduction" % nprod}
                kl = {"lanczos_bounds": {"total_ms": p["ms"][0], "steps": p["launches"][0]},
                      "filter_rounds": {"total_ms": p["ms"][1], "products_with_C": nprod,
                                        "product_ms_each": t_prod * 1e3},
                      "rayleigh_ritz_eigh": {"total_ms": p["ms"][2], "order": m_blk},
                      "ritz_products_and_residuals": {"total_ms": p["ms"][3]}}
            if p["method"] == 2:
                # two-stage route (opt-in): stage 1 is the f64-MFMA-bound kernel family
                result["roofline"] = {"bound": "mfma", "achieved": p["work"][0] / (p["ms"][0] * 1e-3) / 1e12,
                                      "peak": PEAK_F64_MFMA / 1e12, "unit": "TFLOP/s",
                                      "frac": p["work"][0] / (p["ms"][0] * 1e-3) / PEAK_F64_MFMA, "traffic": None,
                                      "kernel": "two-stage reduction, stage 1 (dense -> band 32): 4/3 n^3 flop on the f64 "
                                                "matrix cores", "n": n}
                kl = {"stage1_dense_to_band": {"total_ms": p["ms"][0]}, "stage2_bulge_chase": {"total_ms": p["ms"][1]},
                      "tridiagonal_eigenpairs": {"total_ms": p["ms"][2]},
                      "backtransform_q2_q1": {"total_ms": p["ms"][3], "q2_ms": p["launches"][3] / 1e3}}
            result["eigh"] = {"method": {0: "jacobi", 1: "tridiagonal", 2: "two-stage tridiagonal",
                                         3: "filtered subspace iteration"}[p["method"]], "n": n, "k": k,
                              "ms_per_matrix": p["total_ms"],
                              "algorithmic_tflops": algo_flops / t / 1e12,
                              "frac_of_f64_mfma_peak_on_algorithmic_flops": algo_flops / t / PEAK_F64_MFMA}
            result["kernels"] = kl
            # context, not credit: the reference's own device path is torch.linalg.eigh (dwain.py:162) -- the library
            # eigensolver on the same box and a covariance of the same workload (all n eigenpairs: it has no top-k)
            try:
                e = torch.zeros(n, n, dtype=torch.float64, device=device)
                ref_model, ref_data = (c2 if c2 is not None else stack)[:2]
                for b in ref_data[:D_STEPS]:
                    ops.syrk_accumulate(e, ops.matmul(b["x"].reshape(-1, N_FEAT), ref_model.layers[0].weight.T), 1.0 / (BATCH * SEQ))
                c = ops.cov_finalize(e, D_STEPS, 0.01)
                torch.linalg.eigh(c)
                torch.cuda.synchronize()
                t0l = time.perf_counter()
                torch.linalg.eigh(c)
                torch.cuda.synchronize()
                result["eigh"]["gpu_library_eigh_ms"] = (time.perf_counter() - t0l) * 1e3
                result["eigh"]["gpu_library_eigh_note"] = ("torch.linalg.eigh on the device, same n and a covariance of the "
                                                           "same workload, all eigenpairs (context only)")
                del e, c
            except Exception as exc:  # the library call is context: never fail the bench for it
                result["eigh"]["gpu_library_eigh_ms"] = None
                result["eigh"]["gpu_library_eigh_note"] = f"torch.linalg.eigh failed: {exc}"
        if "roofline" in result:
            result["roofline"]["of"] = ("the dominant kernel of ptd_eigh_topk inside one step of `%s` (NOT of the 8-layer "
                                        "stack `value` is quoted on, whose step is 70 %% model forwards)" % described)
        if "eigh" in result:
            result["eigh"]["of"] = described
        result["kernels"] = {**result.get("kernels", {}), **kernel_lines(device)}
        result["decomposed_fwd"] = decomposed_forward_lines(device)
        # SURVEY 8d C5 lists T = 4096 / 16384 / 65536: the other two row counts, package pair against library pair
        result["decomposed_fwd"]["rows_4096"] = decomposed_forward_lines(device, 4096, full=False)
        result["decomposed_fwd"]["rows_65536"] = decomposed_forward_lines(device, 65536, full=False)
        if world == 1 and not args.no_c4:
            result["c4_shapes"] = llama_shape_lines(device)
            result["c4_block"] = llama_block_lines(device)
            # two full-width blocks end to end: replace -> next-layer-sees-it at depth > 1
            result["c4_stack"] = llama_block_lines(device, blocks=2, dtypes=(torch.bfloat16,), timed=3)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline()
            if c2 is not None:
                result["cpu_baseline"]["gpu_same_workload"] = {"value": result["c2_single_layer"]["value"], "unit": "layers/s",
                                                               "ms_per_step": result["c2_single_layer"]["ms_per_step"]}

    if rank == 0:
        result["config"]["chain_streams_at_exit"] = dict(_eng_stats.CHAIN_STREAM_STATS)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()  # rank 0 may still be in its side measurements
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
