"""Pins the CPU oracle (oracle/ptdeco_oracle.py) against golden vectors that
were produced by running the reference itself (tests/golden/gen_golden.py)."""


import numpy as np
import pytest
import torch

import golden_io as gio
import ptdeco_oracle as orc
import toy_models as tm

PRIM = [(k, m, v) for k in ("lin", "conv") for m, vs in
        (("dwain", ("f64", "f32", "bf16")), ("falor", ("f64", "f32", "mean", "mean_nodamp"))) for v in vs]


def _prim_inputs(kind, variant):
    z = gio.npz("prim")
    w = gio.t(z[f"{kind}.weight"])
    b = gio.t(z[f"{kind}.bias"])
    batches = gio.t(z[f"{kind}.batches"])
    w2d = w[..., 0, 0] if kind == "conv" else w
    rows = [(x.permute(0, 2, 3, 1) if kind == "conv" else x).reshape(-1, w2d.shape[1]) for x in batches]
    if variant == "bf16":
        w2d, b, rows = w2d.bfloat16(), b.bfloat16(), [r.bfloat16() for r in rows]
    return w2d, b, rows


@pytest.mark.parametrize("kind,method,variant", PRIM)
def test_covariance_and_eigenvectors(kind, method, variant):
    z = gio.npz("prim")
    tag = f"{kind}.{method}.{variant}"
    w2d, b, rows = _prim_inputs(kind, variant)
    if method == "dwain":
        eyyt, _, u = orc.dwain_eigvecs_from_batches(w2d, rows[1:], float64=(variant != "f32"))
        np.testing.assert_array_equal((eyyt / 8).numpy(), z[f"{tag}.E"])
    else:
        _, _, _, u = orc.falor_eigvecs_from_batches(
            w2d, rows[1:], use_float64=(variant != "f32"), use_mean=variant.startswith("mean"),
            use_damping=(variant != "mean_nodamp"))
    u_ref = gio.t(z[f"{tag}.u"])
    assert u.dtype == u_ref.dtype
    tol = 1e-12 if u.dtype == torch.float64 else 1e-5
    assert (orc.canonical_sign(u) - u_ref).abs().max().item() <= tol
    # full-rank pair output, as the reference's own tests check (test_deco_primitives_dwain.py:168-171)
    big_u, big_v, _ = orc.factors(w2d, u, 32, w2d.dtype)
    y1 = (rows[0] @ big_u) @ big_v + b
    y1_ref = gio.t(z[f"{tag}.y1"], bf16=(variant == "bf16")).reshape(-1)
    y0_ref = gio.t(z[f"{tag}.y0"], bf16=(variant == "bf16")).reshape(-1)
    if kind == "conv":  # golden is flattened NCHW; ours is [T, C]
        y1 = y1.reshape(2, 6, 6, 32).permute(0, 3, 1, 2)
    lim = 2e-2 if variant == "bf16" else 2e-6
    assert (y1.reshape(-1).float() - y1_ref.float()).abs().max().item() < lim
    assert (y1.reshape(-1).float() - y0_ref.float()).abs().max().item() < lim


def test_metric_primitives():
    z = gio.npz("metrics")
    for name, dims in (("nsr2d", (0,)), ("nsr2d_01", (0, 1)), ("nsr3d", (0, 1)), ("nsr4d", (0, 2, 3))):
        got = orc.nsr(x=gio.t(z[f"{name}.x"]), y=gio.t(z[f"{name}.y"]), non_channel_dim=dims)
        np.testing.assert_array_equal(got.numpy(), z[f"{name}.out"])
    s, tt = gio.t(z["kl.s"]), gio.t(z["kl.t"])
    np.testing.assert_array_equal(orc.kl_div(s, tt).numpy(), z["kl.div"])
    np.testing.assert_array_equal(orc.kl_loss(s, tt).numpy(), z["kl.loss"])


def test_integer_bookkeeping():
    assert orc.dwain_candidate_ranks(10, 4, 0.5) == [5, 2]  # quirk 3: ends below min_rank
    assert orc.dwain_candidate_ranks(4096, 32, 0.5) == [2048, 1024, 512, 256, 128, 64, 32]
    assert orc.dwain_candidate_ranks(96, 4, 0.5) == [48, 24, 12, 6, 3]
    assert orc.params_for_proportion(1.0, 64, 128) == 64 * 128
    assert orc.params_for_proportion(0.5, 64, 128) == 6144
    assert orc.is_num_params_reduced(0.5, 4096, 4096) is False
    assert orc.is_num_params_reduced(0.25, 4096, 4096) is True
    assert orc.split_chunks(list("abcde"), 2) == [["a", "b"], ["c", "d"], ["e"]]
    assert orc.split_chunks(list("ab"), 5) == [["a"], ["b"]]


def _check_state(model, name, tol):
    want = gio.final_state(name)
    got = model.state_dict()
    assert list(got.keys()) == list(want.keys())
    for k in want:
        assert got[k].shape == want[k].shape, k
        assert (got[k] - want[k]).abs().max().item() <= tol * max(1.0, want[k].abs().max().item()), k


FALOR = ["falor_mlp_r8", "falor_mlp_r9", "falor_mlp_mean32", "falor_conv"]
DWAIN = ["dwain_mlp_nosplit", "dwain_mlp_split1", "dwain_mlp_split2", "dwain_mlp_f32acc", "dwain_mlp_loose",
         "dwain_conv"]


@pytest.mark.parametrize("name", FALOR)
def test_falor_end_to_end(name):
    scn = gio.e2e_meta()[name]
    model = gio.build_model(scn)
    trace = []
    cfg = orc.falor_decompose(module=model, data_iterator=tm.cycle_tensors(gio.pool(scn["pool"])), trace=trace,
                              **scn["kwargs"])
    assert gio.jsonable(cfg) == scn["config"]
    assert list(cfg.keys()) == list(scn["config"].keys())
    assert [(s["layer"], s["rank"]) for s in trace] == [(s["layer"], s["rank"]) for s in scn["steps"]]
    m = scn["kwargs"]["num_metric_steps"]
    samples = np.array(scn["metric_samples"]).reshape(len(trace), m, 2).mean(axis=1)
    for s, want in zip(trace, samples):
        assert s["nsr"] == pytest.approx(want[0], rel=1e-12, abs=1e-18)
        assert s["kl"] == pytest.approx(want[1], rel=1e-12, abs=1e-18)
    _check_state(model, name, 0.0)
    with torch.no_grad():
        out = model(gio.pool(scn["pool"])[0])
    np.testing.assert_array_equal(out.numpy(), gio.npz("e2e")[f"{name}.final_out"])


@pytest.mark.parametrize("name", DWAIN)
def test_dwain_end_to_end(name):
    scn = gio.e2e_meta()[name]
    model = gio.build_model(scn)
    data, metric = gio.dwain_streams(scn)
    trace = []
    cfg = orc.dwain_decompose(module=model, data_iterator=data, metric_iterator=metric, loss_fn=tm.ce_loss,
                              trace=trace, **scn["kwargs"])
    assert gio.jsonable(cfg) == scn["config"]
    assert list(cfg.keys()) == list(scn["config"].keys())  # quirk 9: reverse module order
    got = [(s["layer"], s["rank"], s["accepted"]) for s in trace]
    assert got == [(s["layer"], s["rank"], s["accepted"]) for s in scn["steps"]]
    _check_state(model, name, 0.0)
    with torch.no_grad():
        out = model({"x": gio.pool(scn["pool"])[0]})
    np.testing.assert_array_equal(out.numpy(), gio.npz("e2e")[f"{name}.final_out"])


def test_quirk_last_tried_factors_win():
    """falor quirk 1: fc3 of falor_mlp_r9 reports proportion 8/10 but holds a rank-7 pair."""
    scn = gio.e2e_meta()["falor_mlp_r9"]
    assert scn["config"]["fc3"]["__meta__"]["proportion"] == 0.8
    assert scn["config"]["fc3"]["modules"]["0"]["out_features"] == 7


@pytest.mark.parametrize("name", ["dwain_mlp_bf16_nosplit", "dwain_mlp_bf16_split1"])
def test_dwain_bf16_model_end_to_end(name):
    """SURVEY a-Q 4 pinned by the reference itself (tests/golden/bf16.*, gen_golden.py --bf16): a bf16 model with bf16
    batches -- covariance products formed in bf16 before the f64 add (dwain.py:147-152), uk / U / V / W~ in bf16
    (:423-429), precomputed eigenvectors stored in bf16 (:208).  The oracle reproduces decisions, every metric sample,
    the config and the final weights bit for bit."""
    scn = gio.bf16_meta()[name]
    model = gio.bf16_model(scn)
    data, metric, x0 = gio.bf16_streams(scn)
    trace = []
    cfg = orc.dwain_decompose(module=model, data_iterator=data, metric_iterator=metric, loss_fn=tm.ce_loss,
                              trace=trace, **scn["kwargs"])
    assert gio.jsonable(cfg) == scn["config"]
    assert [(s["layer"], s["rank"], s["accepted"]) for s in trace] == \
           [(s["layer"], s["rank"], s["accepted"]) for s in scn["steps"]]
    want_state, want_out = gio.bf16_final(name)
    got = model.state_dict()
    assert got.keys() == want_state.keys()
    for k in got:
        assert torch.equal(got[k], want_state[k]), k
    with torch.no_grad():
        assert torch.equal(model({"x": x0}), want_out)
