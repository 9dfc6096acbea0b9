#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference); the reference itself
never ships.  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py

Outputs (data only -- inputs and expected outputs):
    prim.npz      G1/G5  covariance + eigenvectors of the reference's primitive
                         scenario (reduced spatial size), dwain + falor, Linear +
                         1x1 Conv2d, fp64 / fp32 accumulator / bf16 activations /
                         use_mean variants, full-rank outputs y0, y1
    metrics.npz   G6     NSR and KL primitives
    e2e.npz/.json G3/G4  falor + dwain decompose_in_place end to end on toy
                         models: per-candidate metric samples, decisions, final
                         decompose_config (key order preserved) and state_dict
    bf16.npz/.json       (`--bf16`: written alone, the files above untouched) dwain end
                         to end on a bf16 model with bf16 batches: the reference's own
                         bf16 semantics (SURVEY a-Q 4), bf16 tensors as raw bits
Every tensor is stored (no RNG seeds) so the fixtures do not depend on torch's
generator.
"""

from __future__ import annotations

import copy
import json
import logging
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"
if not os.path.isdir(REF_SRC):
    sys.exit("reference checkout not found: golden vectors can only be generated in the build container")
sys.path.insert(0, REF_SRC)
sys.path.insert(0, os.path.dirname(HERE))

import ptdeco  # noqa: E402  (the reference)
import ptdeco.falor  # noqa: E402
import ptdeco.dwain.decomposition as ref_dwain  # noqa: E402
import ptdeco.falor.decomposition as ref_falor  # noqa: E402

import toy_models as tm  # noqa: E402

torch.set_float32_matmul_precision("highest")
torch.set_num_threads(8)


def npy(t: torch.Tensor) -> np.ndarray:
    t = t.detach().cpu()
    if t.dtype == torch.bfloat16:
        return t.view(torch.int16).numpy().copy()  # raw bits; readers view back as bf16
    return t.contiguous().numpy().copy()


def canon(u: torch.Tensor) -> torch.Tensor:
    idx = u.abs().argmax(dim=0)
    s = torch.sign(u[idx, torch.arange(u.shape[1])])
    s[s == 0] = 1
    return u * s


# ---------------------------------------------------------------------------
# prim.npz
# ---------------------------------------------------------------------------
def init_like_reference_test(mod: torch.nn.Module, gen: torch.Generator) -> None:
    # same initialisation recipe as tests/test_deco_primitives_dwain.py:41-46 / 61-67
    torch.nn.init.kaiming_uniform_(mod.weight, a=5**0.5, generator=gen)
    fan_in, _ = torch.nn.init._calculate_fan_in_and_fan_out(mod.weight)
    torch.nn.init.uniform_(mod.bias, -(fan_in**-0.5), fan_in**-0.5, generator=gen)


def prim_scenarios(out: dict) -> None:
    n_in, n_out, bs, hw, steps = 64, 32, 2, 6, 8
    for kind in ("lin", "conv"):
        gen = torch.Generator().manual_seed(271828)
        net = tm.OneLinear(n_in, n_out) if kind == "lin" else tm.OneConv1x1(n_in, n_out)
        init_like_reference_test(net.mod, gen)
        dgen = torch.Generator().manual_seed(1314159)
        shape = (bs, hw, hw, n_in) if kind == "lin" else (bs, n_in, hw, hw)
        batches = [torch.rand(*shape, generator=dgen) for _ in range(steps + 1)]
        out[f"{kind}.weight"] = npy(net.mod.weight)
        out[f"{kind}.bias"] = npy(net.mod.bias)
        out[f"{kind}.batches"] = npy(torch.stack(batches))

        for method in ("dwain", "falor"):
            for variant in ("f64", "f32", "bf16", "mean", "mean_nodamp"):
                if method == "dwain" and variant.startswith("mean"):
                    continue
                if method == "falor" and variant == "bf16":
                    continue
                model = copy.deepcopy(net)
                data = [b.clone() for b in batches]
                if variant == "bf16":
                    model = model.to(torch.bfloat16)
                    data = [b.to(torch.bfloat16) for b in data]
                tag = f"{kind}.{method}.{variant}"
                x0 = data[0]
                recorded = {}
                with torch.no_grad():
                    if method == "dwain":
                        y0 = model({"inp": x0})
                        ref_dwain._wrap_in_place(model, "mod")
                        w0 = model.mod.get_weight_copy()
                        orig_get = ref_dwain._get_eigenvectors

                        def spy(e, _orig=orig_get, _rec=recorded):
                            _rec["E"] = e.clone()  # Eyyt / steps, before damping
                            return _orig(e)

                        ref_dwain._get_eigenvectors = spy
                        try:
                            u = ref_dwain._compute_covariance_matrix_decomposition(
                                root_module=model, decomposed_submodule_name="mod",
                                data_iterator=iter([{"inp": b} for b in data[1:]]), weight=w0,
                                num_data_steps=steps, device=torch.device("cpu"),
                                decompose_in_float64=(variant != "f32"))
                        finally:
                            ref_dwain._get_eigenvectors = orig_get
                        out[f"{tag}.E"] = npy(recorded["E"])
                    else:
                        y0 = model(x0)
                        ref_falor._wrap_in_place(model, "mod")
                        w0 = model.mod.get_weight_copy()
                        u = ref_falor._compute_decompositon_of_covariance_matrix(
                            root_module=model, decomposed_submodule_name="mod",
                            data_iterator=iter(data[1:]), weight=w0, num_data_steps=steps,
                            device=torch.device("cpu"), use_float64=(variant != "f32"),
                            use_mean=variant.startswith("mean"), use_damping=(variant != "mean_nodamp"))
                    # full-rank pair exactly as the reference test builds it (:101-110)
                    uk = u[:, u.shape[1] - min(n_in, n_out):].to(w0.dtype)
                    big_u, big_v = w0.T @ uk, uk.T
                    pair = model.mod.get_decomposed_module(u=big_u.T, v=big_v.T)
                    pair.to(w0.dtype)  # dwain.py:516 (only matters for the bf16 variant)
                    unwrap = ref_dwain._unwrap_in_place if method == "dwain" else ref_falor._unwrap_in_place
                    unwrap(model, "mod")
                    ptdeco.utils.replace_submodule_in_place(model, "mod", pair)
                    y1 = model({"inp": x0}) if method == "dwain" else model(x0)
                out[f"{tag}.u"] = npy(canon(u))
                out[f"{tag}.y0"] = npy(y0)
                out[f"{tag}.y1"] = npy(y1)
                print(f"prim {tag}: u {tuple(u.shape)} {u.dtype}  max|y0-y1| = "
                      f"{(y0.float() - y1.float()).abs().max().item():.3e}")


# ---------------------------------------------------------------------------
# metrics.npz
# ---------------------------------------------------------------------------
def metric_scenarios(out: dict) -> None:
    g = torch.Generator().manual_seed(7)
    cases = {
        "nsr2d": ((48, 10), (0,)), "nsr2d_01": ((48, 10), (0, 1)), "nsr3d": ((4, 12, 10), (0, 1)),
        "nsr4d": ((3, 6, 5, 5), (0, 2, 3)),
    }
    for name, (shape, dims) in cases.items():
        y = torch.randn(*shape, generator=g) * 2.0 + 0.3
        x = y + 0.1 * torch.randn(*shape, generator=g)
        out[f"{name}.x"], out[f"{name}.y"] = npy(x), npy(y)
        out[f"{name}.out"] = npy(ptdeco.utils.calc_per_channel_noise_to_signal_ratio(
            x=x, y=y, non_channel_dim=dims))
    s = torch.randn(32, 10, generator=g) * 3
    t = s + 0.5 * torch.randn(32, 10, generator=g)
    out["kl.s"], out["kl.t"] = npy(s), npy(t)
    out["kl.div"] = npy(ptdeco.utils.calc_kl_divergence(s, t))
    out["kl.loss"] = npy(ptdeco.utils.calc_kl_loss(s, t))


# ---------------------------------------------------------------------------
# e2e.npz / e2e.json
# ---------------------------------------------------------------------------
class LogTap(logging.Handler):
    def __init__(self):
        super().__init__(level=logging.INFO)
        self.lines: list[str] = []

    def emit(self, record):
        self.lines.append(record.getMessage())


def low_rank_weight(gen, n_out, n_in, rank, scale):
    a = torch.randn(n_out, rank, generator=gen)
    b = torch.randn(rank, n_in, generator=gen)
    sv = torch.logspace(0, -1, rank)
    return (a * sv) @ b * scale


def make_mlp(gen, fc2_rank=None):
    m = tm.MLP3()
    with torch.no_grad():
        for lin in (m.fc1, m.fc2, m.fc3):
            lin.weight.copy_(torch.randn(lin.weight.shape, generator=gen) / lin.in_features**0.5)
            lin.bias.copy_(0.1 * torch.randn(lin.bias.shape, generator=gen))
        if fc2_rank is not None:
            m.fc2.weight.copy_(low_rank_weight(gen, 96, 128, fc2_rank, 1.0 / 128**0.5))
    return m


def make_convnet(gen):
    m = tm.ConvNet()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * (0.3 if p.ndim > 1 else 0.1))
        m.pw2.weight.copy_(low_rank_weight(gen, 40, 48, 6, 0.3)[:, :, None, None])
    return m


def save_model(out, prefix, model):
    for k, v in model.state_dict().items():
        out[prefix + k] = npy(v)


def run_falor(out, meta, name, model, pool, ids, **kw):
    tap = LogTap()
    ref_falor.logger.addHandler(tap)
    ref_falor.logger.setLevel(logging.INFO)
    samples = []
    orig = ref_falor._compute_metrics

    def spy(**kwargs):
        r = orig(**kwargs)
        samples.append([float(r[0]), float(r[1])])
        return r

    ref_falor._compute_metrics = spy
    try:
        cfg = ptdeco.falor.decompose_in_place(
            module=model, device=torch.device("cpu"), data_iterator=tm.cycle_tensors(pool), **kw)
    finally:
        ref_falor._compute_metrics = orig
        ref_falor.logger.removeHandler(tap)
    save_model(out, f"{name}.final.", model)
    with torch.no_grad():
        out[f"{name}.final_out"] = npy(model(pool[0]))
    steps = []
    pat = re.compile(r"Processing (\S+): i=(\d+) rank_width=(\d+) rank_new=(\d+) .* rank_best=(\d+) ")
    for line in tap.lines:
        m = pat.search(line)
        if m:
            steps.append({"layer": m.group(1), "i": int(m.group(2)), "width": int(m.group(3)),
                          "rank": int(m.group(4)), "rank_best_after": int(m.group(5))})
    meta[name] = {"kwargs": kw, "config": cfg, "steps": steps, "metric_samples": samples, **ids}
    print(f"falor {name}: {len(steps)} candidates, decomposed {list(cfg)}")
    for s in steps:
        print("   ", s)


def run_dwain(out, meta, name, model, pool, targets, mpool, mtargets, ids, **kw):
    tap = LogTap()
    ref_dwain.logger.addHandler(tap)
    ref_dwain.logger.setLevel(logging.INFO)
    samples = []
    orig = ref_dwain._compute_metrics

    def spy(**kwargs):
        r = orig(**kwargs)
        samples.append([float(r[0]), float(r[1]), float(r[2])])
        return r

    ref_dwain._compute_metrics = spy
    try:
        cfg = ptdeco.dwain.decompose_in_place(
            module=model, device=torch.device("cpu"),
            data_iterator=tm.cycle_dicts(pool, targets), loss_fn=tm.ce_loss,
            metric_iterator=tm.cycle_dicts(mpool, mtargets),
            finetune_fn=lambda m, device, names: m, **kw)
    finally:
        ref_dwain._compute_metrics = orig
        ref_dwain.logger.removeHandler(tap)
    save_model(out, f"{name}.final.", model)
    with torch.no_grad():
        out[f"{name}.final_out"] = npy(model({"x": pool[0]}))
    steps, layer = [], None
    for line in tap.lines:
        m = re.match(r"PROCESSING (\S+) MODULE", line)
        if m:
            layer = m.group(1)
        m = re.search(r"i=(\d+) (ACCEPTING|REJECTING) rank (\d+)/(\d+)", line)
        if m:
            steps.append({"layer": layer, "i": int(m.group(1)), "rank": int(m.group(3)),
                          "accepted": m.group(2) == "ACCEPTING"})
    meta[name] = {"kwargs": kw, "config": cfg, "steps": steps, "metric_samples": samples, **ids}
    print(f"dwain {name}: {len(steps)} candidates, decomposed {list(cfg)}")
    for s in steps:
        print("   ", s)


def e2e_scenarios(out: dict, meta: dict) -> None:
    """Inputs are stored once under shared keys: ``pool.<id>`` (data batches),
    ``model.<id>.<param>`` (initial weights), ``targets.<model id>.<pool id>``."""
    g = torch.Generator().manual_seed(20240523)
    scale = torch.logspace(0, -1.5, 64)
    pools = {
        "x": [torch.randn(64, 64, generator=g) * scale for _ in range(12)],
        "m": [torch.randn(64, 64, generator=g) * scale for _ in range(6)],
        "c": [torch.randn(4, 3, 8, 8, generator=g) for _ in range(6)],
    }
    for k, v in pools.items():
        out[f"pool.{k}"] = npy(torch.stack(v))

    def model(mid, build):
        m = build()
        save_model(out, f"model.{mid}.", m)
        return m

    falor_kw = dict(proportion_threshold=0.9, nsr_final_threshold=0.02, kl_final_threshold=0.02,
                    num_data_steps=4, num_metric_steps=2, use_float64=True, use_mean=False, use_damping=True)
    for rank in (8, 9):
        mid = f"mlp_r{rank}"
        m = model(mid, lambda: make_mlp(torch.Generator().manual_seed(100 + rank), rank))
        run_falor(out, meta, f"falor_mlp_r{rank}", m, pools["x"], {"model": mid, "arch": "MLP3", "pool": "x"},
                  **falor_kw)
    m = model("mlp_r8", lambda: make_mlp(torch.Generator().manual_seed(108), 8))
    run_falor(out, meta, "falor_mlp_mean32", m, pools["x"], {"model": "mlp_r8", "arch": "MLP3", "pool": "x"},
              **{**falor_kw, "use_float64": False, "use_mean": True, "blacklisted_module_names": ["fc3"]})
    m = model("conv_a", lambda: make_convnet(torch.Generator().manual_seed(55)))
    run_falor(out, meta, "falor_conv", m, pools["c"], {"model": "conv_a", "arch": "ConvNet", "pool": "c"},
              **{**falor_kw, "nsr_final_threshold": 0.05, "kl_final_threshold": 0.05})

    def with_targets(mid, m, pid):
        with torch.no_grad():
            t = [m({"x": b}).argmax(dim=-1) for b in pools[pid]]
        out[f"targets.{mid}.{pid}"] = npy(torch.stack(t))
        return t

    dwain_kw = dict(num_data_steps=4, num_metric_steps=2, nsr_final_threshold=0.1, min_rank=4,
                    trade_off_factor=0.2, reduction_factor=0.5, max_accepted_ppl_diff=0.03,
                    decompose_in_float64=True)
    for tag, extra in (("nosplit", {}), ("split1", {"precomputing_covariance_num_splits": 1}),
                       ("split2", {"precomputing_covariance_num_splits": 2}),
                       ("f32acc", {"decompose_in_float64": False, "blacklisted_module_names": ["fc3"]}),
                       ("loose", {"trade_off_factor": 40.0, "max_accepted_ppl_diff": 0.5,
                                  "nsr_final_threshold": 0.5})):
        m = model("mlp_r12", lambda: make_mlp(torch.Generator().manual_seed(314), 12))
        run_dwain(out, meta, f"dwain_mlp_{tag}", m, pools["x"], with_targets("mlp_r12", m, "x"), pools["m"],
                  with_targets("mlp_r12", m, "m"),
                  {"model": "mlp_r12", "arch": "MLP3", "pool": "x", "mpool": "m"}, **{**dwain_kw, **extra})
    m = model("conv_b", lambda: make_convnet(torch.Generator().manual_seed(56)))
    run_dwain(out, meta, "dwain_conv", m, pools["c"], with_targets("conv_b", m, "c"), pools["c"][:3],
              with_targets("conv_b", m, "c")[:3],
              {"model": "conv_b", "arch": "ConvNet", "pool": "c", "mpool": "c", "mpool_len": 3},
              **{**dwain_kw, "trade_off_factor": 100.0, "max_accepted_ppl_diff": 0.5,
                 "nsr_final_threshold": 0.001})


def bf16_scenarios(out: dict, meta: dict) -> None:
    """SURVEY a-Q 4: with a bf16 model the reference forms uk, U, V, W~ in bf16 (dwain.py:423-429), stores the
    precomputed eigenvectors in bf16 (:208) and -- the part that matters most -- forms every step's covariance product
    einsum(y, y) / T in bf16 before promoting it into the f64 sum (:147-152).  One dwain scenario pins those semantics:
    the MLP3 of the f32 scenarios cast to bf16, bf16 batches, loose thresholds (decisions with margins far above bf16
    noise), per-layer and precomputed covariances."""
    g = torch.Generator().manual_seed(20240524)
    scale = torch.logspace(0, -1.5, 64)
    pools = {"x": [(torch.randn(64, 64, generator=g) * scale).bfloat16() for _ in range(12)],
             "m": [(torch.randn(64, 64, generator=g) * scale).bfloat16() for _ in range(6)]}
    for k, v in pools.items():
        out[f"pool.{k}"] = npy(torch.stack(v))
    kw = dict(num_data_steps=4, num_metric_steps=2, nsr_final_threshold=0.5, min_rank=4, trade_off_factor=40.0,
              reduction_factor=0.5, max_accepted_ppl_diff=0.5, decompose_in_float64=True)
    for tag, extra in (("nosplit", {}), ("split1", {"precomputing_covariance_num_splits": 1})):
        m = make_mlp(torch.Generator().manual_seed(314), 12).bfloat16()
        save_model(out, "model.mlp_r12_bf16.", m)
        with torch.no_grad():
            tx = [m({"x": b}).argmax(dim=-1) for b in pools["x"]]
            tmm = [m({"x": b}).argmax(dim=-1) for b in pools["m"]]
        out["targets.mlp_r12_bf16.x"] = npy(torch.stack(tx))
        out["targets.mlp_r12_bf16.m"] = npy(torch.stack(tmm))
        run_dwain(out, meta, f"dwain_mlp_bf16_{tag}", m, pools["x"], tx, pools["m"], tmm,
                  {"model": "mlp_r12_bf16", "arch": "MLP3", "pool": "x", "mpool": "m", "dtype": "bfloat16"},
                  **{**kw, **extra})


def main() -> None:
    if "--bf16" in sys.argv:
        # only the bf16 scenario, into files of its own (the f32 fixtures stay byte-identical)
        e2e, meta = {}, {}
        bf16_scenarios(e2e, meta)
        np.savez_compressed(os.path.join(HERE, "bf16.npz"), **e2e)
        with open(os.path.join(HERE, "bf16.json"), "wt") as f:
            json.dump({"reference_version": ptdeco.__version__, "torch": torch.__version__, "scenarios": meta},
                      f, indent=1)
        for fn in ("bf16.npz", "bf16.json"):
            print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")
        return
    prim, metrics, e2e, meta = {}, {}, {}, {}
    prim_scenarios(prim)
    metric_scenarios(metrics)
    e2e_scenarios(e2e, meta)
    np.savez_compressed(os.path.join(HERE, "prim.npz"), **prim)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **metrics)
    np.savez_compressed(os.path.join(HERE, "e2e.npz"), **e2e)
    with open(os.path.join(HERE, "e2e.json"), "wt") as f:
        json.dump({"reference_version": ptdeco.__version__, "torch": torch.__version__, "scenarios": meta},
                  f, indent=1)
    for fn in ("prim.npz", "metrics.npz", "e2e.npz", "e2e.json"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")


if __name__ == "__main__":
    main()
