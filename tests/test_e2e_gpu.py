"""End-to-end parity on an MI355X: ptdeco_amd.{dwain,falor}.decompose_in_place against
golden runs of the reference itself (tests/golden/e2e.*) -- identical rank decisions,
metrics / factors / outputs within the stated f32 tolerances."""

import json

import numpy as np
import pytest
import torch

import golden_io as gio
import ptdeco_oracle as orc
import toy_models as tm

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")

REL = 1e-4      # north_star tolerance for factors / metrics (f32 model, f64 decomposition)
ABS_NOISE = 2e-6  # metrics that are pure f32 rounding noise in the reference (e.g. nsr ~ 1e-13, kl ~ 7e-8)


def _to_dev_cycle(it):
    for item in it:
        yield item  # drivers move batches themselves (utils.to_device / .to(device))


def _close(a, b):
    return abs(a - b) <= REL * abs(b) + ABS_NOISE


def _check_config(cfg, want):
    cfg = gio.jsonable(cfg)
    assert list(cfg.keys()) == list(want.keys())
    for name in want:
        got_meta, want_meta = cfg[name].pop("__meta__"), dict(want[name]["__meta__"])
        assert {k: v for k, v in want[name].items() if k != "__meta__"} == cfg[name]
        assert got_meta.keys() == want_meta.keys()
        for k in want_meta:
            if isinstance(want_meta[k], int):
                assert got_meta[k] == want_meta[k], (name, k)
            else:
                assert _close(got_meta[k], want_meta[k]), (name, k, got_meta[k], want_meta[k])


def _check_factors(model, name, decomposed, well_separated):
    want = gio.final_state(name)
    got = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    assert list(got.keys()) == list(want.keys())
    for k in want:
        assert got[k].shape == want[k].shape, k
    for layer in decomposed:
        a_g, b_g = got[f"{layer}.0.weight"].double(), got[f"{layer}.1.weight"].double()
        a_w, b_w = want[f"{layer}.0.weight"].double(), want[f"{layer}.1.weight"].double()
        if a_g.dim() == 4:
            a_g, b_g, a_w, b_w = (t[:, :, 0, 0] for t in (a_g, b_g, a_w, b_w))
        prod_g, prod_w = b_g @ a_g, b_w @ a_w
        assert (prod_g - prod_w).norm().item() <= REL * prod_w.norm().item(), layer
        if layer in well_separated:
            # sign-canonicalise by the eigenvector columns of the second factor
            def canon(a, b):
                idx = b.abs().argmax(dim=0)
                s = torch.sign(b[idx, torch.arange(b.shape[1])])
                return a * s[:, None], b * s
            a_gc, b_gc = canon(a_g, b_g)
            a_wc, b_wc = canon(a_w, b_w)
            assert (b_gc - b_wc).norm().item() <= REL * b_wc.norm().item(), layer
            assert (a_gc - a_wc).norm().item() <= REL * a_wc.norm().item(), layer
    for k in want:
        if not any(k.startswith(layer + ".") for layer in decomposed):
            assert torch.equal(got[k], want[k]), k  # untouched parameters stay bit-identical


FALOR = {"falor_mlp_r8": {"fc1"}, "falor_mlp_r9": {"fc1", "fc3"}, "falor_mlp_mean32": set(), "falor_conv": set()}
DWAIN = {"dwain_mlp_nosplit": {"fc1", "fc2"}, "dwain_mlp_split1": {"fc1", "fc2"}, "dwain_mlp_split2": {"fc1", "fc2"},
         "dwain_mlp_f32acc": set(), "dwain_mlp_loose": {"fc1"}, "dwain_conv": set()}


@pytest.mark.parametrize("name", list(FALOR))
def test_falor_end_to_end(name):
    import ptdeco_amd

    scn = gio.e2e_meta()[name]
    model = gio.build_model(scn).to(DEV)
    trace = []
    cfg = ptdeco_amd.falor.decompose_in_place(
        module=model, device=DEV, data_iterator=tm.cycle_tensors(gio.pool(scn["pool"])), trace=trace,
        **scn["kwargs"])
    assert [(s["layer"], s["rank"]) for s in trace] == [(s["layer"], s["rank"]) for s in scn["steps"]]
    m = scn["kwargs"]["num_metric_steps"]
    want = np.array(scn["metric_samples"]).reshape(len(trace), m, 2).mean(axis=1)
    for s, w in zip(trace, want):
        assert _close(s["nsr"], w[0]) and _close(s["kl"], w[1]), (s, w)
    _check_config(cfg, scn["config"])
    _check_factors(model, name, list(cfg.keys()), FALOR[name])
    with torch.no_grad():
        out = model(gio.pool(scn["pool"])[0].to(DEV)).cpu()
    ref = gio.t(gio.npz("e2e")[f"{name}.final_out"])
    assert (out - ref).abs().max().item() <= REL * ref.abs().max().item()
    assert all(isinstance(model.get_submodule(k), (ptdeco_amd.LowRankLinear, ptdeco_amd.LowRankConv1x1))
               for k in cfg)


@pytest.mark.parametrize("name", list(DWAIN))
def test_dwain_end_to_end(name):
    import ptdeco_amd

    scn = gio.e2e_meta()[name]
    model = gio.build_model(scn).to(DEV)
    data, metric = gio.dwain_streams(scn)
    trace = []
    cfg = ptdeco_amd.dwain.decompose_in_place(
        module=model, device=DEV, data_iterator=data, metric_iterator=metric, loss_fn=tm.ce_loss,
        finetune_fn=lambda m, device, names: m, trace=trace, **scn["kwargs"])
    got = [(s["layer"], s["rank"], s["accepted"]) for s in trace]
    assert got == [(s["layer"], s["rank"], s["accepted"]) for s in scn["steps"]]
    m = scn["kwargs"]["num_metric_steps"]
    samples = np.array(scn["metric_samples"]).reshape(len(trace), m, 3)
    for s, smp in zip(trace, samples):
        assert _close(s["nsr"], smp[:, 0].mean()), (s, smp)
        assert _close(s["ppl_deco"], smp[:, 1].mean()), (s, smp)
        assert _close(s["ppl_diff"], ((smp[:, 1] - smp[:, 2]) / smp[:, 2]).mean()), (s, smp)
    _check_config(cfg, scn["config"])
    _check_factors(model, name, list(cfg.keys()), DWAIN[name])
    with torch.no_grad():
        out = model({"x": gio.pool(scn["pool"])[0].to(DEV)}).cpu()
    ref = gio.t(gio.npz("e2e")[f"{name}.final_out"])
    assert (out - ref).abs().max().item() <= REL * ref.abs().max().item()


@pytest.mark.parametrize("method,name", [("dwain", "dwain_mlp_nosplit"), ("dwain", "dwain_conv"),
                                         ("falor", "falor_mlp_r9"), ("falor", "falor_conv")])
def test_metric_steps_with_and_without_the_prefix_memo_are_bit_identical(method, name, monkeypatch):
    """The second forward of a metric step reuses the products ahead of the analysed layer (_engine.PrefixMemo):
    every reused output is recomputed and compared (CHECK), and the metrics equal the ones of a run without it.
    Round 6: the scenarios' iterators cycle over a few batches, so batches recur within a layer's search and the prefix
    and the original output of a batch are kept across the layer's candidates -- same metrics bit for bit with that
    switched off (PTD_MEMO_ACROSS_CANDIDATES=0) and with no memo at all."""
    import ptdeco_amd
    from ptdeco_amd import _engine as eng

    scn = gio.e2e_meta()[name]
    traces = []
    for env in ({"PTD_PREFIX_MEMO_CHECK": "1"}, {"PTD_PREFIX_MEMO_MB": "0"}, {"PTD_MEMO_ACROSS_CANDIDATES": "0"}, {}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        hits0, orig0 = eng.PrefixMemo.total_hits, eng.PrefixMemo.total_orig_hits
        model = gio.build_model(scn).to(DEV)
        trace = []
        if method == "dwain":
            data, metric = gio.dwain_streams(scn)
            ptdeco_amd.dwain.decompose_in_place(
                module=model, device=DEV, data_iterator=data, metric_iterator=metric, loss_fn=tm.ce_loss,
                finetune_fn=lambda m, device, names: m, trace=trace, **scn["kwargs"])
        else:
            ptdeco_amd.falor.decompose_in_place(
                module=model, device=DEV, data_iterator=tm.cycle_tensors(gio.pool(scn["pool"])), trace=trace,
                **scn["kwargs"])
        traces.append((trace, eng.PrefixMemo.total_hits - hits0, eng.PrefixMemo.total_orig_hits - orig0))
        for k in env:
            monkeypatch.delenv(k)
        assert all("forward" not in m.__dict__ for m in model.modules())
    assert traces[0][0] == traces[1][0] == traces[2][0] == traces[3][0]
    assert traces[1][1] == 0 and traces[1][2] == 0 and traces[2][2] == 0
    if method == "dwain":
        assert traces[3][2] > 0    # the default: second forwards were saved on recurring batches
    if name.endswith("nosplit") or name.endswith("r9"):
        assert traces[0][1] > 0       # the MLPs have layers ahead of fc2 / fc3


def test_falor_on_a_two_batch_iterator_reuses_prefix_and_original_output_across_candidates(monkeypatch):
    """falor's bisection draws M batches per candidate from the data iterator; with an iterator over TWO batches every
    candidate after the first meets batches the layer has seen: the same trace bit for bit as with
    PTD_MEMO_ACROSS_CANDIDATES=0, with second forwards saved."""
    import ptdeco_amd
    from ptdeco_amd import _engine as eng

    scn = gio.e2e_meta()["falor_mlp_r9"]
    runs = []
    for across in ("1", "0"):
        monkeypatch.setenv("PTD_MEMO_ACROSS_CANDIDATES", across)
        orig0 = eng.PrefixMemo.total_orig_hits
        model = gio.build_model(scn).to(DEV)
        trace = []
        ptdeco_amd.falor.decompose_in_place(module=model, device=DEV, trace=trace,
                                            data_iterator=tm.cycle_tensors(list(gio.pool(scn["pool"]))[:2]), **scn["kwargs"])
        runs.append((trace, eng.PrefixMemo.total_orig_hits - orig0))
    assert runs[0][0] == runs[1][0] and len(runs[0][0]) > 2
    assert runs[0][1] > 0 and runs[1][1] == 0


@pytest.mark.parametrize("method,name", [("dwain", "dwain_mlp_nosplit"), ("falor", "falor_mlp_r9"), ("falor", "falor_conv")])
def test_calibration_forwards_cut_at_the_layer_change_nothing(method, name, monkeypatch):
    """Where the covariance is accumulated layer by layer the forwards behind the first stop at the analysed layer
    (LayerTap.calibration_forward): the trace -- every metric of every candidate -- equals the one of whole forwards
    (PTD_CALIBRATION_EARLY_STOP=0) bit for bit."""
    import ptdeco_amd

    scn = gio.e2e_meta()[name]
    traces = []
    for stop in ("1", "0"):
        monkeypatch.setenv("PTD_CALIBRATION_EARLY_STOP", stop)
        model = gio.build_model(scn).to(DEV)
        trace = []
        if method == "dwain":
            data, metric = gio.dwain_streams(scn)
            ptdeco_amd.dwain.decompose_in_place(
                module=model, device=DEV, data_iterator=data, metric_iterator=metric, loss_fn=tm.ce_loss,
                finetune_fn=lambda m, device, names: m, trace=trace, **scn["kwargs"])
        else:
            ptdeco_amd.falor.decompose_in_place(
                module=model, device=DEV, data_iterator=tm.cycle_tensors(gio.pool(scn["pool"])), trace=trace,
                **scn["kwargs"])
        traces.append(trace)
    assert traces[0] == traces[1] and len(traces[0]) > 0


def test_config_round_trip_and_reload():
    """README.md:56-105 of the reference: JSON config + state_dict reload onto a fresh model."""
    import ptdeco_amd

    scn = gio.e2e_meta()["falor_mlp_r9"]
    model = gio.build_model(scn).to(DEV)
    cfg = ptdeco_amd.falor.decompose_in_place(
        module=model, device=DEV, data_iterator=tm.cycle_tensors(gio.pool(scn["pool"])), **scn["kwargs"])
    blob = json.dumps(cfg)
    fresh = gio.build_model(scn).to(DEV)
    ptdeco_amd.utils.apply_decompose_config_in_place(fresh, json.loads(blob))
    fresh.load_state_dict(model.state_dict())
    x = gio.pool(scn["pool"])[1].to(DEV)
    with torch.no_grad():
        assert torch.equal(fresh(x), model(x))
    assert isinstance(fresh.fc2, ptdeco_amd.LowRankLinear)


def _primitive(kind, method):
    """Port of the reference's own primitive tests (tests/test_deco_primitives_dwain.py:75-122,
    test_deco_primitives_falor.py): full-rank factors from the HIP covariance + eigensolver must
    reproduce the layer."""
    from ptdeco_amd import _engine as eng

    n_in, n_out, hw = 64, 32, 16
    gen = torch.Generator().manual_seed(271828)
    net = tm.OneLinear(n_in, n_out) if kind == "lin" else tm.OneConv1x1(n_in, n_out)
    torch.nn.init.kaiming_uniform_(net.mod.weight, a=5**0.5, generator=gen)
    torch.nn.init.uniform_(net.mod.bias, -(n_in**-0.5), n_in**-0.5, generator=gen)
    net.to(DEV)
    dgen = torch.Generator().manual_seed(1314159)
    shape = (8, hw, hw, n_in) if kind == "lin" else (8, n_in, hw, hw)
    batches = [torch.rand(*shape, generator=dgen) for _ in range(9)]
    x = batches[0].to(DEV)
    with torch.no_grad():
        y0 = net(x)
        tap = eng.LayerTap(net, "mod")
        w0 = tap.weight_copy()
        cov = eng.Covariance(n_out, DEV, True, with_mean=(method == "falor"))
        for b in batches[1:]:
            net(b.to(DEV))
            cov.add_inputs(tap.last_input_rows(), w0)
        u = cov.eigenvectors(eng.EIGEN_DAMPEN_FACTOR)
        uk, big_u, _ = eng.build_factors(w0, u, min(n_in, n_out), torch.float32)
        pair = eng.build_pair(tap.layer, big_u, uk, None)
        tap.close()
        net.mod = pair
        y1 = net(x)
    return y0, y1


@pytest.mark.parametrize("method", ["dwain", "falor"])
def test_primitive_linear_gpu(method):
    y0, y1 = _primitive("lin", method)
    assert (y0 - y1).abs().max().item() < 1.0e-6  # test_deco_primitives_dwain.py:174-178


@pytest.mark.parametrize("method", ["dwain", "falor"])
def test_primitive_conv1x1_gpu(method):
    y0, y1 = _primitive("conv", method)
    assert (y0 - y1).abs().max().item() < 9.0e-4  # the reference's own GPU bound (:187-192)


def test_dwain_with_a_training_finetune_fn_tracks_the_oracle():
    """dwain.py:779-786: after each accepted layer the caller's finetune_fn trains the model.  Here it
    takes two SGD steps on the already decomposed layers; on the GPU those are the fused pairs, whose
    backward runs on the HIP GEMMs.  Same decisions as the CPU oracle running the same callback on
    plain nn.Sequential pairs, metrics and final outputs within 1e-4."""
    import copy
    import itertools

    import ptdeco_amd

    scn = gio.e2e_meta()["dwain_mlp_nosplit"]
    pool = gio.pool(scn["pool"])
    targets = {}
    grad_norms = []

    def make_finetune(device):
        def finetune(model, dev, names):
            params = [p for n in names for p in model.get_submodule(n).parameters()]
            if not params:
                return model
            for p in model.parameters():
                p.requires_grad_(False)
            for p in params:
                p.requires_grad_(True)
            opt = torch.optim.SGD(params, lr=0.05)
            model.train()
            with torch.enable_grad():
                for i in range(2):
                    x = pool[i].to(device)
                    if i not in targets:
                        targets[i] = base(pool[i]).argmax(-1)
                    loss = torch.nn.functional.cross_entropy(model({"x": x}), targets[i].to(device))
                    opt.zero_grad()
                    loss.backward()
                    grad_norms.append(sum(float(p.grad.norm()) for p in params))
                    opt.step()
            model.eval()
            return model
        return finetune

    base_model = gio.build_model(scn).eval()

    def base(x):
        with torch.no_grad():
            return base_model({"x": x})

    with torch.no_grad():
        for i in range(2):
            targets[i] = base(pool[i]).argmax(-1)

    ref_model, ref_trace = gio.build_model(scn), []
    data, metric = gio.dwain_streams(scn)
    ref_cfg = orc.dwain_decompose(module=ref_model, data_iterator=data, metric_iterator=metric, loss_fn=tm.ce_loss,
                                  finetune_fn=make_finetune(torch.device("cpu")), trace=ref_trace, **scn["kwargs"])
    model, trace = gio.build_model(scn).to(DEV), []
    data, metric = gio.dwain_streams(scn)
    cfg = ptdeco_amd.dwain.decompose_in_place(module=model, device=DEV, data_iterator=data, metric_iterator=metric,
                                              loss_fn=tm.ce_loss, finetune_fn=make_finetune(DEV), trace=trace,
                                              **scn["kwargs"])
    assert len(cfg) >= 1 and list(cfg.keys()) == list(ref_cfg.keys())
    assert [(s["layer"], s["rank"], s["accepted"]) for s in trace] == \
           [(s["layer"], s["rank"], s["accepted"]) for s in ref_trace]
    for s, r in zip(trace, ref_trace):
        assert _close(s["nsr"], r["nsr"]) and _close(s["ppl_deco"], r["ppl_deco"]), (s, r)
    with torch.no_grad():
        out = model({"x": pool[0].to(DEV)}).cpu()
        ref = ref_model({"x": pool[0]})
    assert (out - ref).abs().max().item() <= REL * ref.abs().max().item()
    # the callback really trained something, on both sides
    assert len(grad_norms) >= 4 and all(g > 0.0 for g in grad_norms)


@pytest.mark.parametrize("name", ["dwain_mlp_bf16_nosplit", "dwain_mlp_bf16_split1"])
def test_dwain_bf16_model_against_the_reference_in_bf16(name):
    """the reference's OWN bf16 semantics, pinned by a run of the imported reference
    (tests/golden/bf16.*: bf16 MLP3, bf16 batches; every step's covariance product rounded to bf16 before the f64 add,
    dwain.py:147-152; factors formed in bf16, :423-429).  The HIP path accumulates that product in f32 on the matrix
    cores and forms the factors from f64 eigenvectors -- MORE accurate than the reference, so the two agree only to
    what bf16 rounding of a covariance entry (2^-8 relative) does to eigenvectors, factors and metrics.  Stated
    tolerances (measured on MI355X, twice the worst deviation seen): identical (layer, rank, accepted) decisions and
    config structure; per-candidate nsr within 6 % relative + 2e-4, ppl_deco within 1 % relative, model outputs
    within 3 % of their range."""
    import ptdeco_amd

    scn = gio.bf16_meta()[name]
    model = gio.bf16_model(scn).to(DEV)
    data, metric, x0 = gio.bf16_streams(scn)
    trace = []
    cfg = ptdeco_amd.dwain.decompose_in_place(
        module=model, device=DEV, data_iterator=data, metric_iterator=metric, loss_fn=tm.ce_loss,
        finetune_fn=lambda m, device, names: m, trace=trace, **scn["kwargs"])
    assert [(s["layer"], s["rank"], s["accepted"]) for s in trace] == \
           [(s["layer"], s["rank"], s["accepted"]) for s in scn["steps"]]
    assert list(cfg.keys()) == list(scn["config"].keys())
    for layer, c in scn["config"].items():
        assert cfg[layer]["modules"] == c["modules"] and cfg[layer]["__meta__"]["proportion"] == c["__meta__"]["proportion"]
    m = scn["kwargs"]["num_metric_steps"]
    samples = np.array(scn["metric_samples"]).reshape(len(trace), m, 3)
    dev_nsr = max(abs(s["nsr"] - smp[:, 0].mean()) / (abs(smp[:, 0].mean()) + 1e-12) for s, smp in zip(trace, samples))
    dev_ppl = max(abs(s["ppl_deco"] - smp[:, 1].mean()) / abs(smp[:, 1].mean()) for s, smp in zip(trace, samples))
    want_state, want_out = gio.bf16_final(name)
    with torch.no_grad():
        out = model({"x": x0.to(DEV)}).float().cpu()
    dev_out = (out - want_out.float()).abs().max().item() / want_out.float().abs().max().item()
    print(f"bf16 scenario {name}: nsr {dev_nsr:.3e} ppl_deco {dev_ppl:.3e} out {dev_out:.3e}")
    for s, smp in zip(trace, samples):
        assert abs(s["nsr"] - smp[:, 0].mean()) <= 0.06 * abs(smp[:, 0].mean()) + 2e-4, (s, smp)
        assert abs(s["ppl_deco"] - smp[:, 1].mean()) <= 0.01 * abs(smp[:, 1].mean()), (s, smp)
    assert dev_out <= 0.03, dev_out
    # the untouched parameters (biases of undecomposed layers do not exist here: every layer was replaced) and the
    # installed modules' dtype
    assert all(p.dtype == torch.bfloat16 for p in model.parameters())
