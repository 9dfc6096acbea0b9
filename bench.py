#!/usr/bin/env python3
"""Headline benchmark: dwain layers decomposed per second on MI355X (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W

ONE fixed workload at every N (strong scaling): BASELINE configs[3] at reduced depth -- dwain.decompose_in_place
(reference src/ptdeco/dwain/decomposition.py:677-800) of a stack of TWO Llama-3-8B-width blocks (q, k, v, o, gate, up,
down at 4096 / 1024 / 14336 = 14 layers, + a blacklisted 4096 x 4096 head), bf16 model, f64 covariance + eigh,
[1, 2048, 4096] token batches, D = 8 calibration steps, M = 2 metric steps, precomputing_covariance_num_splits = 1,
thresholds under which layers ARE replaced (the next layer then sees the changed model, dwain.py:779-787), identity
finetune_fn.  One "step" = one full decompose_in_place call on a fresh copy of the stack, every input already
resident in HBM; `value` = 14 layers x K / time.  With N ranks: calibration steps and (candidate, batch) pairs dealt to
the ranks, packed covariance sums reduced to the layer owners over RCCL, eigenvectors broadcast (sharding.py).
`python bench.py --gpus N` starts its own N ranks when no launcher did (children, before any GPU call).

Output: the LAST stdout line is ONE compact JSON object (<= 4096 bytes: the driver keeps the tail of stdout) with the
contract keys, `roofline` (the dominant package kernel of the headline step, measured live with HIP events) and
`cpu_baseline` (the CPU oracle on one layer of the stack's shapes, host cores of this box), plus scalars for the
other BASELINE configs: c2_* (configs[1]: ONE nn.Linear(4096, 4096), D = 4, M = 2), c3_* (configs[2]: falor on a
ViT-B/16-shaped clone, depth 12, batch 8, D = 5, M = 5, use_mean=False, use_damping=True), c1_cpu_s (configs[0]: the
CPU oracle's falor on the resnet18-shaped clone, one batch).  Everything else -- per-kernel lines, the decomposed
forward (configs[4]), per-shape and one-block Llama lines, eigensolver phases -- goes to bench_detail.json beside this
file (and to gpurun_out/bench_detail.json, which travels back from the GPU box); an earlier stdout line names it.
Counter-derived fields (traffic, MFMA utilisation) are quoted from committed rocprofv3 PMC summaries; each carries the
hash of the kernel source it was measured on and is marked "stale": true when the source has changed since.
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
HEADLINE_BLOCKS = 2                  # Llama-3-8B-width blocks of the headline workload (configs[3] at reduced depth)
METRIC = "layers decomposed/sec (incl. covariance+SVD)"
LINE_LIMIT = 4096                    # bytes of the last stdout line (the driver parses the tail of stdout)
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


ROUTES = {0: "jacobi", 1: "tridiagonal (direct)", 3: "filtered subspace iteration"}


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
        count = p.get("count", 1) if p.get("sweeps", 0) > 1 else 1       # matrices per launch (ptd_eigh_topk_batched)
        if count > 1:
            out["matrices_per_launch"] = count
            out["ms_per_matrix"] = p["total_ms"] / count
            out["algorithmic_tflops"] *= count
            out["frac_of_f64_mfma_peak_on_algorithmic_flops"] *= count
        all_bytes = count * sum(8.0 * (n - j - 1) * (n - j - 2) for j in range(n - 1))
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


METRIC_POOL = 16    # metric batches the headline's iterator cycles over: more than the 14 draws of a layer's rank search


def llama_workload(device, blocks, dt, metric_pool=METRIC_POOL):
    """BASELINE configs[3] in small: a stack of `blocks` full-width Llama-3-8B blocks (q, k, v, o, gate, up, down at 4096 /
    1024 / 14336 + the blacklisted head), seeded weights and [1, 2048, 4096] batches already on the device, and
    step(trace=None) -> decompose_config of ONE dwain.decompose_in_place call on a fresh copy (precompute pass, one split).
    Identical on every rank (same seeds).  Returns (step, keyword arguments).

    The metric iterator cycles over `metric_pool` batches.  The default, 16, is more than the 7 candidates x M = 2 draws
    of one layer's rank search, so NO batch comes round again within a layer -- what the reference's trainers feed
    (one infinite iterator over a DataLoader for both arguments, examples/trainer_llm/run_decompose_dwain.py:203-223) --
    and every (candidate, batch) pair runs its candidate and its original forward.  metric_pool = 4 is the side line
    `metric_batches_recurring`: batches recur within a layer, and the engine then runs the model ahead of the layer and the
    original model once per batch and layer (PrefixMemo across candidates) -- never part of `value`."""
    import ptdeco_amd

    # (a layer's share of the parameters shrinks with the depth: the trade-off factor scales with the number of blocks)
    kw = dict(C4_BLOCK_KW, trade_off_factor=C4_BLOCK_KW["trade_off_factor"] * blocks)
    g = torch.Generator(device=device).manual_seed(0)
    with torch.device(device):
        model0 = LlamaStack(blocks)
    with torch.no_grad():
        for prm in model0.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g, device=device) / prm.shape[1] ** 0.5)
    model0.to(dt)
    scale = torch.logspace(0, -2, D_MODEL, device=device)
    xs = [(torch.randn(1, 2048, D_MODEL, generator=g, device=device) * scale).to(dt) for _ in range(8 + metric_pool)]
    with torch.no_grad():
        bt = [{"x": x, "targets": model0({"x": x}).argmax(-1)} for x in xs]

    def step(trace=None):
        m = copy.deepcopy(model0)
        return ptdeco_amd.dwain.decompose_in_place(module=m, device=device, data_iterator=itertools.cycle(bt[:12]),
                                                   loss_fn=seq_ce, metric_iterator=itertools.cycle(bt[8:]),
                                                   finetune_fn=lambda mm, d, n: mm, trace=trace, **kw)
    return step, kw


def llama_workload_text(blocks, kw, dt):
    name = dt if isinstance(dt, str) else ("bf16" if dt == torch.bfloat16 else "f32")
    return ("dwain.decompose_in_place, %d Llama-3-8B-width block(s) (%d layers) + blacklisted head, %s model, f64 "
            "covariance + eigh, [1, 2048, 4096] batches, D = 8, M = 2 (metric iterator over %d batches: none recurs "
            "within a layer's search), precompute pass (1 split), trade_off_factor %g, max_accepted_ppl_diff 0.4; layers "
            "are replaced as the search goes" % (blocks, 7 * blocks, name, METRIC_POOL, kw["trade_off_factor"]))


def phase_split(step):
    """One more, untimed step with the phase spans armed: (phases_ms incl. other_host_and_gaps, trace, config)."""
    from ptdeco_amd import _engine as eng

    eng.PHASES = eng.PhaseTimer()
    trace = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cfg = step(trace)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    ph, eng.PHASES = eng.PHASES.totals_ms(), None
    ph["other_host_and_gaps"] = max(0.0, wall - sum(ph.values()))
    return {k: round(v, 1) for k, v in ph.items()}, trace, cfg


def llama_block_lines(device, blocks=1, dtypes=(torch.float32, torch.bfloat16), timed=5):
    """BASELINE configs[3] in small, end to end: dwain on `blocks` full-width Llama-3-8B block(s), f32 and bf16 model;
    one warm-up step, `timed` timed steps (median reported), one more with the phase spans."""
    out = {}
    kw = None
    for dt in dtypes:
        step, kw = llama_workload(device, blocks, dt)
        step()
        torch.cuda.synchronize()
        marks = []
        for _ in range(timed):
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            marks.append((time.perf_counter() - t0) * 1e3)
        ph, trace, cfg = phase_split(step)
        med = sorted(marks)[len(marks) // 2]
        out["f32" if dt == torch.float32 else "bf16"] = {
            "layers": 7 * blocks, "ms_per_step": med, "ms_per_step_max": max(marks), "ms_per_step_min": min(marks),
            "ms_per_block": med / blocks, "step_ms": [round(v, 1) for v in marks], "layers_per_s": 7e3 * blocks / med,
            "spread": (max(marks) - min(marks)) / med, "timed_steps": timed, "statistic": "median of step_ms",
            "phases_ms": ph, "candidates_evaluated": len(trace),
            "replaced": {k: v["__meta__"]["proportion"] for k, v in cfg.items()}}
        del step
    out["workload"] = llama_workload_text(blocks, kw, "f32 / bf16")
    return out


def pmc_traffic(n):
    """roofline.traffic: HBM-side bytes per SYMV launch from the committed rocprofv3 PMC passes
    (profiles/pmc_symv_rNN.json, made by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE
    runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be read
    from inside this process, so the figure is the latest committed pass for the same matrix order."""
    pmc = pmc_file("pmc_symv_r*.json", ("eigh_tridiag.hip",))
    if pmc and pmc["data"].get("n") == n:
        d = pmc["data"]
        return {"traffic": d["traffic_bytes_per_launch"], "matrices_per_launch": d.get("matrices_per_launch", 1),
                "traffic_source": pmc["source"] + ": (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch, "
                                  "%.3f x the algorithmic bytes" % d["traffic_over_algorithmic"],
                "traffic_stale": pmc["stale"]}
    return {"traffic": None}


def host_cores():
    """(threads to use, physical cores, CPUs the affinity mask and the cgroup quota grant)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        physical = os.cpu_count()
    from cpu_quota import usable_cpus    # a cgroup quota below the visible count throttles a pool sized by it
    usable = usable_cpus()
    return max(1, min(physical, usable)), physical, usable


# what the oracle took per layer on a GPU box's 16 granted cores (profiles/c4_shapes_cpu_r03.json), used only to SIZE the
# live sample below and for the gate / up shape, whose 14336^2 LAPACK eigh (65 s) does not fit a bounded sample
CPU_SHAPE_SECONDS_R03 = {"q_o": 5.5, "k_v": 0.81, "gate_up": 65.0, "down": 10.7}


def cpu_baseline(full=False):
    """The CPU oracle (oracle/ptdeco_oracle.py: the restatement of the reference, torch-CPU / MKL, pinned by the
    reference-generated fixtures of tests/golden) on ONE layer of the headline stack's shapes ("for C4 time one
    layer of each of the 4 distinct shapes"), same synthetic inputs as c4_shapes ([1, 2048, n_in] tokens, D = 8, M = 2,
    f64 decomposition), host cores of this box.  A bounded sample: q/o, k/v and down live (about 17 s); gate/up (one
    14336^2 eigh: 65 s) only with full=True, otherwise taken from the committed run of the same code on the same kind of
    box and labelled so.  value = the 14-layer stack's layers per second from the per-shape seconds (a lower bound on the
    CPU's cost: a layer alone is cheaper than the same layer inside the stack, whose metric forwards run 14 layers)."""
    cores, physical, usable = host_cores()      # (puts oracle/ on the path)
    import ptdeco_oracle as orc

    torch.set_num_threads(cores)
    secs, live = {}, []
    for name, n_in, n_out in (("q_o", 4096, 4096), ("k_v", 4096, 1024), ("down", 14336, 4096), ("gate_up", 4096, 14336)):
        if name == "gate_up" and not full:
            secs[name] = CPU_SHAPE_SECONDS_R03[name]
            continue
        g = torch.Generator().manual_seed(1)
        m = OneLinear(n_in, n_out)
        with torch.no_grad():
            m.lin.weight.copy_(torch.randn(n_out, n_in, generator=g) / n_in**0.5)
        scale = torch.logspace(0, -2, n_in)
        xs = [torch.randn(1, 2048, n_in, generator=g) * scale for _ in range(10)]
        with torch.no_grad():
            bt = [{"x": x, "targets": m({"x": x}).argmax(-1)} for x in xs]
        t0 = time.perf_counter()
        orc.dwain_decompose(module=m, data_iterator=itertools.cycle(bt), loss_fn=seq_ce, metric_iterator=itertools.cycle(bt[8:]),
                            finetune_fn=None, num_data_steps=8, num_metric_steps=2, nsr_final_threshold=1.0,
                            decompose_in_float64=True)
        secs[name] = time.perf_counter() - t0
        live.append(name)
        del m, xs, bt
    per_block = 2 * secs["q_o"] + 2 * secs["k_v"] + 2 * secs["gate_up"] + secs["down"]
    return {"value": 7.0 / per_block, "unit": "layers/s", "cores": cores, "kind": "port",
            "sample": "1 layer each of %s live (%.1f s); %s" % (
                ", ".join(live), sum(secs[n] for n in live),
                "all four shapes live" if full else "gate_up (one 14336^2 LAPACK eigh) 65.0 s from profiles/c4_shapes_cpu_r03.json"),
            "workload": "oracle dwain on one layer of each Llama-3-8B shape, [1,2048,n_in] tokens, D=8, M=2, f64; value = 7 "
                        "layers / (2 q_o + 2 k_v + 2 gate_up + down) seconds",
            "s_per_layer": {k: round(v, 2) for k, v in secs.items()},
            "physical_cores": physical, "usable_cpus": usable}


def c1_cpu_line():
    """BASELINE configs[0]: the CPU oracle's falor on the resnet18-shaped clone (tests/toy_models.ResNet18: torchvision
    layout, 3 stride-2 1x1 downsample convolutions + fc), one fixed batch (5, 3, 224, 224), D = M = 1, use_mean=False,
    use_damping=True, thresholds 0.01, proportion_threshold 0.9.  Seconds of one call."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    cores, _, _ = host_cores()                  # (puts oracle/ on the path)
    import ptdeco_oracle as orc
    import toy_models as tm

    torch.set_num_threads(cores)
    torch.manual_seed(271828)
    model = tm.ResNet18().eval()
    g = torch.Generator().manual_seed(1314159)
    x = torch.rand(5, 3, 224, 224, generator=g)
    t0 = time.perf_counter()
    cfg = orc.falor_decompose(module=model, data_iterator=itertools.repeat(x), proportion_threshold=0.9,
                              nsr_final_threshold=0.01, kl_final_threshold=0.01, num_data_steps=1, num_metric_steps=1,
                              use_float64=True, use_mean=False, use_damping=True)
    return {"seconds": time.perf_counter() - t0, "layers": 4, "decomposed": len(cfg), "cores": cores}


def c3_line(device, depth=12, batch=8, pool_size=64):
    """BASELINE configs[2]: falor.decompose_in_place (reference falor.py:424-511) on the ViT-B/16-shaped clone
    (tests/toy_models.ViT: timm vit_base_patch16_224 layer names and shapes, 48 block Linears + head = 49 layers at depth 12),
    random weights, synthetic [8, 3, 224, 224] images, the reference trainer's settings: D = 5, M = 5
    (decompose_falor.yaml:21-22), thresholds 0.01 (:18-19), proportion_threshold 0.9, use_float64, use_mean=False,
    use_damping=True (run_decompose_falor.py:92-93).  One warm-up call at depth 1, then ONE timed call at full depth.
    The data iterator cycles over `pool_size` batches; a layer draws 5 + 9 x 5 = 50 of them, so with the default 64 no
    batch recurs within a layer (the reference trainer streams a DataLoader); pool_size = 24 is the side line
    `c3_recurring_batches`, where the engine's reuse across candidates engages."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ptdeco_amd
    import toy_models as tm
    from ptdeco_amd import _engine as eng

    kw = dict(proportion_threshold=0.9, nsr_final_threshold=0.01, kl_final_threshold=0.01, num_data_steps=5,
              num_metric_steps=5, use_float64=True, use_mean=False, use_damping=True)
    g = torch.Generator().manual_seed(1)
    pool = [torch.randn(batch, 3, 224, 224, generator=g).to(device) for _ in range(pool_size)]

    def run(d, phases):
        model = tm.ViT(depth=d)
        tm.init_randn(model, 0)
        model.to(device).eval()
        trace = []
        if phases:
            eng.PHASES = eng.PhaseTimer()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cfg = ptdeco_amd.falor.decompose_in_place(module=model, device=device, data_iterator=itertools.cycle(pool),
                                                  trace=trace, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ph = None
        if phases:
            ph, eng.PHASES = {k: round(v, 1) for k, v in eng.PHASES.totals_ms().items()}, None
        return dt, cfg, trace, ph

    run(1, False)
    dt, cfg, trace, ph = run(depth, True)
    layers = 4 * depth + 1
    return {"seconds": dt, "layers": layers, "layers_per_s": layers / dt, "candidates_evaluated": len(trace),
            "decomposed": len(cfg), "phases_ms": ph,
            "workload": "falor.decompose_in_place, ViT-B/16-shaped clone depth %d (%d Linear layers), f32 model, f64 covariance "
                        "+ eigh, [%d,3,224,224] images (iterator over %d batches), D=5, M=5, use_mean=False, use_damping=True, thresholds 0.01 / 0.9"
                        % (depth, layers, batch, pool_size)}


def roofline_from_profile(prof, device):
    """`roofline` of the headline step from the per-call eigensolver profiles of ONE step (ops.EIGH_PROFILE: HIP events on
    the launch stream): the dominant package kernel of the step is the per-column SYMV of the direct tridiagonalisation
    (HBM bound: SURVEY 8d, 8/3 n^3 bytes streamed per matrix) -- its launches of the largest direct problem."""
    direct = [p for p in prof if p["method"] == 1 and p["launches"][0] > 0 and p["ms"][0] > 0]
    if not direct:
        return None
    p = max(direct, key=lambda q: (q["n"], q["k"], q.get("count", 1)))
    n, cnt, ms, byts = p["n"], p["launches"][0], p["ms"][0], p["work"][0]
    mats = p.get("count", 1) if p.get("sweeps", 0) > 1 else 1      # matrices per launch (work[0] counts all of them)
    tr = pmc_traffic(n)
    if tr.get("traffic") and mats != tr.get("matrices_per_launch", 1):
        tr["traffic"] = tr["traffic"] * mats / tr.get("matrices_per_launch", 1)   # (the committed pass had another batch size)
    red_ms = p["ms"][0] + p["ms"][1]
    all_bytes = mats * sum(8.0 * (n - j - 1) * (n - j - 2) for j in range(n - 1))
    out = {"bound": "hbm", "achieved": byts / (ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
           "frac": byts / (ms * 1e-3) / PEAK_HBM, "traffic": tr.get("traffic"),
           "kernel": "sytrd_symv2_kernel (per-column SYMV of the Householder tridiagonalisation, lower-triangle tiles)",
           "n": n, "k": p["k"], "matrices_per_launch": mats, "launches": cnt, "avg_launch_us": ms / cnt * 1e3,
           "algorithmic_bytes_per_launch": byts / cnt,
           "solver_frac": all_bytes / (red_ms * 1e-3) / PEAK_HBM, "eigh_ms": p["total_ms"],
           "of": "the (%d, %d) direct eigendecompositions inside one headline step (%d matrices per launch, beside "
                 "the other lanes' work)" % (n, p["k"], mats)}
    if tr.get("traffic"):
        out["traffic_stale"] = tr.get("traffic_stale")
        out["traffic_source"] = tr.get("traffic_source")
    return out


def launch_ranks(args_list, gpus) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of this one
    (python -m torch.distributed.run, rendezvous on 127.0.0.1) before anything here has touched the GPU, pass
    their output through (rank 0 prints the JSON line) and return their exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(args_list)
    return subprocess.call(cmd)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline: time the gate/up shape live too (+65 s)")
    ap.add_argument("--no-extras", action="store_true", help="only the timed headline (+ C2 at N = 1): no side measurements")
    ap.add_argument("--no-c3", action="store_true", help="skip BASELINE configs[2] (falor ViT-B/16 clone) and configs[0]")
    ap.add_argument("--no-c4", action="store_true", help="skip the per-shape / one-block Llama detail lines")
    ap.add_argument("--blocks", type=int, default=HEADLINE_BLOCKS, help="Llama blocks of the headline stack")
    ap.add_argument("--workload", choices=("c4", "c2"), default="c4",
                    help="c2: time BASELINE configs[1] alone (one layer; for kernel traces); default: the Llama stack")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"))
    return ap.parse_args(argv)


def measure(args):
    """Everything that needs the GPU.  Returns the full result dict on rank 0 (None on the other ranks)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
    from ptdeco_amd import _engine as eng
    from ptdeco_amd import ops

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
        return dt, cfg, [round((b - a) * 1e3, 1) for a, b in zip([t0] + marks[:-1], marks)]

    def kept(cfg):
        return {k: v["__meta__"]["proportion"] for k, v in cfg.items()}

    def c2_family():
        model, data, metric = make_workload(1, device, D_STEPS, 7 * M_STEPS)
        model.to(device)
        data, metric = with_targets(model, data, device), with_targets(model, metric, device)

        def one_step():
            m = copy.deepcopy(model)
            return ptdeco_amd.dwain.decompose_in_place(
                module=m, device=device, data_iterator=itertools.cycle(data), loss_fn=ce_loss,
                metric_iterator=itertools.cycle(metric), finetune_fn=lambda mm, d, names: mm,
                precomputing_covariance_num_splits=1, **DWAIN_KW)
        return one_step

    c2_text = ("BASELINE configs[1]: dwain decompose_in_place of ONE nn.Linear(4096,4096) f32, precompute pass (1 split), "
               "T=4x1024 tokens/batch, D=4, M=2, 6 evaluated candidate ranks, f64 covariance+eigh")
    if args.workload == "c2":
        # profiling aid: BASELINE configs[1] alone under the same protocol (python bench.py --workload c2 --no-extras)
        dt, cfg, marks = timed(c2_family())
        res = {"metric": METRIC, "value": args.steps / dt, "unit": "layers/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "step_ms": marks, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": c2_text + " (--workload c2)", "ranks_kept": kept(cfg)}}
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return res if rank == 0 else None

    # ---- the headline: BASELINE configs[3] at reduced depth, bf16 model, the same work at every N
    step, kw = llama_workload(device, args.blocks, torch.bfloat16)
    layers = 7 * args.blocks
    dt, cfg, marks = timed(step)
    comm = ("; calibration steps and (candidate, batch) pairs dealt to the ranks, packed covariance sums reduced to the layer "
            "owners + eigenvector broadcast over RCCL") if world > 1 else ""
    med = sorted(marks)[len(marks) // 2]
    result = {
        "metric": METRIC, "value": layers * args.steps / dt, "unit": "layers/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3] at reduced depth: " + llama_workload_text(args.blocks, kw, torch.bfloat16) + comm,
                   "model_dtype": "bfloat16", "layers_per_step": layers, "blocks": args.blocks,
                   "parallelism": f"dp{world}" if world > 1 else "single"},
        "spread": (max(marks) - min(marks)) / med,
    }
    detail = {"step_ms": marks, "replaced": kept(cfg),
              "metric_forwards": "every (candidate, batch) pair runs the stack twice as the reference does; the second run "
                                 "reuses the layer outputs ahead of the analysed layer that the first run of the SAME pair just "
                                 "computed (PTD_PREFIX_MEMO_MB=0 switches that off).  The metric iterator cycles over %d "
                                 "batches, a layer's search draws 14: no batch recurs within a layer, so nothing is reused "
                                 "across pairs, layers or steps (the engine's reuse across candidates for recurring batches "
                                 "is measured apart: metric_batches_recurring)" % METRIC_POOL}

    # ---- one more, untimed step with the phase spans (every rank takes part: the step contains collectives)
    if not args.no_extras:
        eng.PHASES = eng.PhaseTimer()
        barrier()
        t0p = time.perf_counter()
        trace = []
        step(trace)
        barrier()
        wall_p = (time.perf_counter() - t0p) * 1e3
        ph, eng.PHASES = eng.PHASES.totals_ms(), None
        ph["other_host_and_gaps"] = max(0.0, wall_p - sum(ph.values()))
        if world > 1:
            # per-rank maxima: a straggling owner shows here, not in rank 0's own spans
            keys = ("A_accumulate", "B_eigh", "C_factors", "D_metrics", "comm")
            t = torch.tensor([ph.get(k, 0.0) for k in keys], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            vals = t.tolist()
            result["comm_ms"] = round(vals[4], 1)
            result["b_eigh_ms_max"] = round(vals[1], 1)
            result["d_metrics_ms_max"] = round(vals[3], 1)
            result["rccl_ranks"] = world if not rehearse else 0
            result["cov_collective"] = os.environ.get("PTD_COV_COLLECTIVE", "reduce")
        result["phases_ms"] = {k: round(v, 1) for k, v in ph.items()}
        detail["candidates_evaluated"] = len(trace)
        # per-call eigensolver profiles of one more step (HIP events on the launch streams)
        ops.EIGH_PROFILE = []
        step()
        barrier()
        prof, ops.EIGH_PROFILE = ops.EIGH_PROFILE, None
        if rank == 0:
            rl = roofline_from_profile(prof, device)
            if rl:
                result["roofline"] = rl
            detail["eigh_calls"] = [eigh_route_block(p) for p in prof]
    del step
    if world > 1:
        if rank == 0:
            detail["chain_streams"] = dict(eng.CHAIN_STREAM_STATS)
            result["detail"] = detail
        dist.barrier()
        dist.destroy_process_group()
        return result if rank == 0 else None

    # ---- N = 1: the other BASELINE configs and the side measurements
    dt2, cfg2, marks2 = timed(c2_family())
    result["c2_layers_per_s"] = args.steps / dt2
    result["c2_ms_per_step"] = dt2 / args.steps * 1e3
    detail["c2_single_layer"] = {"workload": c2_text, "step_ms": marks2, "ranks_kept": kept(cfg2)}
    def side(name, fn):
        """A side measurement must never cost the headline line: its failure is recorded in the detail file."""
        try:
            return fn()
        except Exception as exc:  # noqa: BLE001
            import traceback
            detail[name + "_error"] = "%s: %s" % (type(exc).__name__, exc)
            print("bench.py: side measurement %s failed:\n%s" % (name, traceback.format_exc()), file=sys.stderr, flush=True)
            return None

    if not args.no_extras:
        def recurring():
            # the same stack with a metric iterator over FOUR batches: batches recur within a layer's search and the engine
            # runs the model ahead of the layer and the original model once per batch and layer (never part of `value`)
            from ptdeco_amd._engine import PrefixMemo
            step4, _kw = llama_workload(device, args.blocks, torch.bfloat16, metric_pool=4)
            step4()
            torch.cuda.synchronize()
            marks4 = []
            for _ in range(3):
                t0 = time.perf_counter()
                step4()
                torch.cuda.synchronize()
                marks4.append(round((time.perf_counter() - t0) * 1e3, 1))
            ph4, _trace, _cfg = phase_split(step4)
            return {"metric_pool": 4, "step_ms": marks4, "ms_per_step": sorted(marks4)[1], "phases_ms": ph4,
                    "what": "PrefixMemo across candidates (SURVEY 8f-2): per layer the prefix and the original output of a "
                            "batch are computed at its first visit; PTD_MEMO_ACROSS_CANDIDATES=0 switches it off"}
        rec = side("metric_batches_recurring", recurring)
        if rec:
            detail["metric_batches_recurring"] = rec
            result["recurring_batches_ms_per_step"] = rec["ms_per_step"]
        if not args.no_c3:
            c3 = side("c3", lambda: c3_line(device))
            if c3:
                result["c3_s"] = c3["seconds"]
                result["c3_layers_per_s"] = c3["layers_per_s"]
                detail["c3"] = c3
            c3r = side("c3_recurring", lambda: c3_line(device, pool_size=24))
            if c3r:
                result["c3_recurring_batches_s"] = c3r["seconds"]
                detail["c3_recurring_batches"] = c3r
        kl = side("kernels", lambda: kernel_lines(device))
        if kl:
            detail["kernels"] = kl

        def forward_lines():
            fwd = decomposed_forward_lines(device)
            # SURVEY 8d C5 lists T = 4096 / 16384 / 65536: the other two row counts, package pair against library pair
            fwd["rows_4096"] = decomposed_forward_lines(device, 4096, full=False)
            fwd["rows_65536"] = decomposed_forward_lines(device, 65536, full=False)
            return fwd
        fwd = side("decomposed_fwd", forward_lines)
        if fwd:
            detail["decomposed_fwd"] = fwd
            # BASELINE's second metric (decomposed-fwd GFLOP/s), configs[4] at T = 16384
            result["fwd_gflops"] = {f"r{r}": round(fwd[f"r{r}"]["gflops"]) for r in (256, 512, 1024)}
            result["fwd_vs_lib_pair"] = {f"r{r}": round(fwd[f"r{r}"]["torch_hipblaslt_pair_ms"] / fwd[f"r{r}"]["ms"], 3)
                                         for r in (256, 512, 1024)}
        if not args.no_c4:
            detail["c4_shapes"] = side("c4_shapes", lambda: llama_shape_lines(device))
            detail["c4_block"] = side("c4_block", lambda: llama_block_lines(device))
        if not args.no_cpu_baseline:
            cb = side("cpu_baseline", lambda: cpu_baseline(full=args.cpu_full))
            if cb:
                result["cpu_baseline"] = cb
            if not args.no_c3:
                c1 = side("c1_cpu", c1_cpu_line)
                if c1:
                    result["c1_cpu_s"] = c1["seconds"]
                    detail["c1_cpu"] = c1
    detail["chain_streams"] = dict(eng.CHAIN_STREAM_STATS)
    result["detail"] = detail
    return result


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config")


def compact_line(result: dict, detail_path: str) -> str:
    """The ONE line the driver parses: the contract keys, `roofline`, `cpu_baseline` and the scalar side figures, never
    more than LINE_LIMIT bytes -- optional keys are shortened, then dropped (longest first), until it fits."""
    line = {k: v for k, v in result.items() if k != "detail"}
    line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None

    def rounded(v):
        if isinstance(v, float):
            return float(f"{v:.6g}")
        if isinstance(v, dict):
            return {k: rounded(x) for k, x in v.items()}
        if isinstance(v, list):
            return [rounded(x) for x in v]
        return v
    line = rounded(line)
    text = json.dumps(line)
    if len(text.encode()) > LINE_LIMIT:
        for blk in ("roofline", "cpu_baseline"):    # long provenance strings first
            for key in ("traffic_source", "solver_note", "note", "workload", "s_per_layer"):
                if isinstance(line.get(blk), dict):
                    line[blk].pop(key, None)
        text = json.dumps(line)
    optional = [k for k in line if k not in CONTRACT_KEYS and k not in ("roofline", "cpu_baseline", "detail")]
    while len(text.encode()) > LINE_LIMIT and optional:
        victim = max(optional, key=lambda k: len(json.dumps(line[k])))
        optional.remove(victim)
        del line[victim]
        text = json.dumps(line)
    if len(text.encode()) > LINE_LIMIT:
        line["config"] = {"workload": str(line["config"].get("workload"))[:600]}
        for blk in ("roofline", "cpu_baseline"):
            if isinstance(line.get(blk), dict):
                line[blk] = {k: v for k, v in line[blk].items() if not isinstance(v, str) or len(v) <= 80}
        text = json.dumps(line)
    assert len(text.encode()) <= LINE_LIMIT, len(text.encode())
    return text


def main(argv=None, measure_fn=None) -> int:
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if measure_fn is None and world == 1 and args.gpus > 1:
        return launch_ranks(sys.argv[1:] if argv is None else argv, args.gpus)
    result = (measure_fn or measure)(args)
    if result is None:          # ranks other than 0
        return 0
    detail_path = None
    if "detail" in result:
        full = dict(result)
        detail_path = args.detail
        blob = json.dumps(full, indent=1)
        default_path = os.path.join(ROOT, "bench_detail.json")
        # (the copy under gpurun_out/ travels back from the GPU box; only for the default location -- tests pass their own)
        for path in (detail_path,) + ((os.path.join(ROOT, "gpurun_out", "bench_detail.json"),)
                                      if os.path.abspath(detail_path) == default_path else ()):
            try:
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, "w") as fh:
                    fh.write(blob)
            except OSError as exc:      # a read-only tree must not cost the line
                print(f"bench.py: could not write {path}: {exc}", file=sys.stderr)
        print(f"bench.py: detail blocks written to {detail_path}", flush=True)
    print(compact_line(result, detail_path), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
