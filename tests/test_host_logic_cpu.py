"""Host logic of the drivers (schedules, decisions, stream order, config format, module
surgery) checked on CPU against the reference's golden runs, with the HIP ops swapped for the
oracle's arithmetic by tests/cpu_shim.py.  The GPU parity tests proper are test_e2e_gpu.py."""

import json

import numpy as np
import pytest
import torch

import cpu_shim
import golden_io as gio
import toy_models as tm

CPU = torch.device("cpu")
FALOR = ["falor_mlp_r8", "falor_mlp_r9", "falor_mlp_mean32", "falor_conv"]
DWAIN = ["dwain_mlp_nosplit", "dwain_mlp_split1", "dwain_mlp_split2", "dwain_mlp_f32acc", "dwain_mlp_loose",
         "dwain_conv"]


def _meta_close(cfg, want, rel=1e-5, abs_=1e-6):
    cfg = gio.jsonable(cfg)
    assert list(cfg.keys()) == list(want.keys())
    for name in want:
        gm, wm = cfg[name].pop("__meta__"), want[name]["__meta__"]
        assert {k: v for k, v in want[name].items() if k != "__meta__"} == cfg[name]
        for k in wm:
            assert gm[k] == pytest.approx(wm[k], rel=rel, abs=abs_), (name, k)


@pytest.mark.parametrize("name", FALOR)
def test_falor_driver(name, monkeypatch):
    scn = gio.e2e_meta()[name]
    with cpu_shim.installed(monkeypatch) as pkg:
        model = gio.build_model(scn)
        trace = []
        cfg = pkg.falor.decompose_in_place(module=model, device=CPU, trace=trace,
                                           data_iterator=tm.cycle_tensors(gio.pool(scn["pool"])), **scn["kwargs"])
    assert [(s["layer"], s["rank"]) for s in trace] == [(s["layer"], s["rank"]) for s in scn["steps"]]
    _meta_close(cfg, scn["config"])
    want = gio.final_state(name)
    got = model.state_dict()
    assert list(got.keys()) == list(want.keys())
    with torch.no_grad():
        out = model(gio.pool(scn["pool"])[0])
    ref = gio.t(gio.npz("e2e")[f"{name}.final_out"])
    assert (out - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("name", DWAIN)
def test_dwain_driver(name, monkeypatch):
    scn = gio.e2e_meta()[name]
    with cpu_shim.installed(monkeypatch) as pkg:
        model = gio.build_model(scn)
        data, metric = gio.dwain_streams(scn)
        trace = []
        cfg = pkg.dwain.decompose_in_place(module=model, device=CPU, data_iterator=data, metric_iterator=metric,
                                           loss_fn=tm.ce_loss, finetune_fn=lambda m, d, n: m, trace=trace,
                                           **scn["kwargs"])
    got = [(s["layer"], s["rank"], s["accepted"]) for s in trace]
    assert got == [(s["layer"], s["rank"], s["accepted"]) for s in scn["steps"]]
    _meta_close(cfg, scn["config"])
    assert list(model.state_dict().keys()) == list(gio.final_state(name).keys())
    with torch.no_grad():
        out = model({"x": gio.pool(scn["pool"])[0]})
    ref = gio.t(gio.npz("e2e")[f"{name}.final_out"])
    assert (out - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_finetune_fn_contract(monkeypatch):
    """finetune_fn is called after every accepted layer with the growing list of names (dwain.py:779-786)."""
    scn = gio.e2e_meta()["dwain_mlp_nosplit"]
    calls = []

    def ft(m, device, names):
        calls.append(list(names))
        return m

    with cpu_shim.installed(monkeypatch) as pkg:
        model = gio.build_model(scn)
        data, metric = gio.dwain_streams(scn)
        pkg.dwain.decompose_in_place(module=model, device=CPU, data_iterator=data, metric_iterator=metric,
                                     loss_fn=tm.ce_loss, finetune_fn=ft, **scn["kwargs"])
    assert calls == [["fc2"], ["fc2", "fc1"]]


def test_non_decomposable_target_raises():
    from ptdeco_amd import _engine
    m = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3))
    with pytest.raises(ValueError, match="Cannot decompose"):
        _engine.LayerTap(m, "0")


def test_utils_surface():
    import ptdeco_amd.utils as u

    assert u.MODCONFIG_META_KEY == "__meta__"
    with pytest.raises(ValueError):
        u.to_device([1, 2], CPU)
    d = u.to_device({"a": torch.zeros(2), "b": 3}, CPU)
    assert d["b"] == 3
    lin = torch.nn.Linear(3, 4)
    tied = torch.nn.ModuleList([lin, lin])
    assert u.get_num_params(tied) == 16
    assert u.split_module_parent_child_name("a.b.c") == ("a.b", "c")
    assert u.split_module_parent_child_name("c") == ("", "c")
    with pytest.raises(ValueError):
        u.build_module_from_config({"type": "Nope"})
    with pytest.raises(ValueError):
        u.get_module_config(torch.nn.ReLU())
    conv = torch.nn.Conv2d(4, 6, 1, stride=2)
    cfg = json.loads(json.dumps(u.get_module_config(conv)))
    back = u.build_module_from_config(cfg)
    assert back.stride == (2, 2) and back.kernel_size == (1, 1)


def test_apply_decompose_config_builds_fused_pairs():
    import ptdeco_amd

    scn = gio.e2e_meta()["falor_conv"]
    model = gio.build_model(scn)
    ptdeco_amd.utils.apply_decompose_config_in_place(model, scn["config"])
    assert isinstance(model.pw2, ptdeco_amd.LowRankConv1x1)
    assert isinstance(model.head, ptdeco_amd.LowRankLinear)
    model.load_state_dict(gio.final_state("falor_conv"))
    with torch.no_grad():
        out = model(gio.pool("c")[0])  # CPU: the pair runs its two children
    np.testing.assert_allclose(out.numpy(), gio.npz("e2e")["falor_conv.final_out"], rtol=1e-5, atol=1e-6)


def test_candidate_schedules():
    from ptdeco_amd.dwain import decomposition as dw
    from ptdeco_amd.falor import decomposition as fa

    assert dw._candidate_ranks(10, 4, 0.5) == [5, 2]
    assert dw._candidate_ranks(4096, 32, 0.5) == [2048, 1024, 512, 256, 128, 64, 32]
    assert dw._get_params_for_proportion(0.5, 64, 128) == 6144
    assert fa._bisection_widths(96) == [48, 24, 12, 6, 3, 1]
    assert fa._bisection_widths(10) == [5, 2, 1]
    assert fa._batches_consumed(torch.nn.Linear(96, 10), 4, 2) == 4 + 2 * 3


def test_run_concurrently_keeps_order_and_raises():
    """Without a GPU the jobs run in the calling thread, in order; an exception propagates."""
    from ptdeco_amd import _engine as eng

    cpu = torch.device("cpu")
    assert eng.run_concurrently([lambda i=i: torch.tensor([i]) for i in range(5)], cpu) == [torch.tensor([i]) for i in range(5)]
    assert eng.run_concurrently([], cpu) == []

    def boom():
        raise RuntimeError("job failed")

    with pytest.raises(RuntimeError, match="job failed"):
        eng.run_concurrently([lambda: torch.zeros(1), boom], cpu)


def test_run_concurrently_takes_the_longest_jobs_first_when_asked(monkeypatch):
    """`costs` + PTD_EIGH_LONGEST_FIRST=1: the jobs are started by descending cost, results stay in job order (on the
    CPU the jobs run one after the other, which makes the start order observable)."""
    from ptdeco_amd import _engine as eng

    started = []

    def job(i):
        def run():
            started.append(i)
            return torch.tensor([float(i)])
        return run

    jobs = [job(i) for i in range(5)]
    costs = [1.0, 5.0, 3.0, 5.0, 2.0]

    # (the CPU path of run_concurrently is sequential and ignores the order; drive the ordering logic itself)
    order = sorted(range(len(jobs)), key=lambda i: -costs[i])
    assert order == [1, 3, 2, 4, 0]                      # stable: equal costs keep the model's order
    out = eng.run_concurrently(jobs, CPU, costs=costs)
    assert [float(t) for t in out] == [0.0, 1.0, 2.0, 3.0, 4.0]


def test_largest_evaluated_rank_drives_top_k():
    """Only candidates that lower the parameter count are evaluated (dwain.py:418-421): a square layer
    skips every rank >= full / 2, a widening layer keeps more; top_k = the largest one that is."""
    from ptdeco_amd.dwain import decomposition as dw

    assert dw._max_candidate_rank(4096, 4096, 32, 0.5) == 1024      # 2048 * 8192 == 4096 * 4096: no drop
    assert dw._max_candidate_rank(4096, 14336, 32, 0.5) == 2048     # 2048 * 18432 < 4096 * 14336
    assert dw._max_candidate_rank(64, 128, 4, 0.5) == 32
    assert dw._max_candidate_rank(10, 10, 4, 0.5) == 2               # schedule 5, 2: 5 * 20 == 100 is skipped
    assert dw._max_candidate_rank(2, 2, 4, 0.5) == 1                 # nothing to evaluate: >= 1 by contract


def test_exhausted_iterator_propagates_stop_iteration(monkeypatch):
    """The reference lets StopIteration escape when the caller's iterator runs dry."""
    scn = gio.e2e_meta()["falor_mlp_r8"]
    model = gio.build_model(scn)
    short = iter(gio.pool(scn["pool"])[:2])
    with cpu_shim.installed(monkeypatch) as pkg, pytest.raises(StopIteration):
        pkg.falor.decompose_in_place(module=model, device=CPU, data_iterator=short, **scn["kwargs"])


def test_precompute_refuses_conv_layers_like_the_reference(monkeypatch):
    """dwain's all-layers precompute pass is nn.Linear only (dwain.py:166-208 forms x @ weight.T)."""
    scn = gio.e2e_meta()["dwain_conv"]
    model = gio.build_model(scn)
    data, metric = gio.dwain_streams(scn)
    kw = dict(scn["kwargs"], precomputing_covariance_num_splits=1)
    with cpu_shim.installed(monkeypatch) as pkg, pytest.raises(RuntimeError, match="nn.Linear only"):
        pkg.dwain.decompose_in_place(module=model, device=CPU, data_iterator=data, metric_iterator=metric,
                                     loss_fn=tm.ce_loss, finetune_fn=lambda m, d, n: m, **kw)


def test_splits_must_cover_all_layers(monkeypatch):
    """Quirk 6: chunks of len // num_splits layers plus one remainder chunk; the assert of dwain.py:673
    fires when that cannot absorb what is left (len % num_splits > len // num_splits)."""
    from ptdeco_amd.dwain import decomposition as dw
    from ptdeco_amd.sharding import Shard

    calls = []

    def fake(**kw):
        calls.append(list(kw["submodule_names"]))
        return {n: None for n in kw["submodule_names"]}

    monkeypatch.setattr(dw, "_precompute_covariance_matrix_decompositions", fake)
    names = ["a", "b", "c", "d", "e"]
    out = dw._precompute_covariance_matrix_decompositions_in_splits(
        module=None, modules_to_decompose=names, num_splits=2, num_data_steps=1, data_iterator=None, device=CPU,
        decompose_in_float64=True, shard=Shard.from_env(None), min_rank=1, reduction_factor=0.5)
    assert calls == [["a", "b"], ["c", "d"], ["e"]] and list(out) == names
    calls.clear()
    with pytest.raises(AssertionError):  # 11 layers, 4 splits: chunk 2, 4 + 1 parts cover only 10 layers
        dw._precompute_covariance_matrix_decompositions_in_splits(
            module=None, modules_to_decompose=list("abcdefghijk"), num_splits=4, num_data_steps=1, data_iterator=None,
            device=CPU, decompose_in_float64=True, shard=Shard.from_env(None), min_rank=1, reduction_factor=0.5)


class _TwoBranchBlock(torch.nn.Module):
    """q / k / v read one tensor, gate / up another (a transformer block's sharing pattern)."""

    def __init__(self, d=48, kv=16, ff=120):
        super().__init__()
        self.q, self.k, self.v = (torch.nn.Linear(d, n, bias=False) for n in (d, kv, kv))
        self.o = torch.nn.Linear(d, d, bias=True)
        self.gate, self.up = torch.nn.Linear(d, ff, bias=False), torch.nn.Linear(d, ff, bias=False)
        self.down = torch.nn.Linear(ff, d, bias=False)
        self.rep = d // kv

    def forward(self, batch):
        x = batch["x"]
        h = x * 1.5
        a = self.q(h) + self.k(h).repeat(1, self.rep) + self.v(h).repeat(1, self.rep)
        x = x + self.o(a)
        h = torch.tanh(x)
        return x + self.down(torch.nn.functional.silu(self.gate(h)) * self.up(h))


@pytest.mark.parametrize("mode,expected_groups,syrks_per_step", [
    ("off", [], 7),                                        # every layer on its own
    ("all", [["q", "k", "v"], ["gate", "up"]], 4),         # 2 shared x^T x + o + down
    ("auto", [["gate", "up"]], 6),                         # D = 3 steps: y^T y is cheaper for q / k / v
])
def test_precompute_pass_shares_one_input_moment_per_group(mode, expected_groups, syrks_per_step, monkeypatch):
    """SURVEY 8f-4 / dwain.py:580-633: layers reading the same tensor accumulate ONE x^T x per calibration
    step; the eigenvectors (W Ex W^T route) equal those of every layer's own y^T y."""
    from ptdeco_amd import _engine as eng
    from ptdeco_amd.dwain import decomposition as dw
    from ptdeco_amd.sharding import Shard

    monkeypatch.setenv("PTD_SHARE_INPUT_COVARIANCE", mode)
    g = torch.Generator().manual_seed(7)
    model = _TwoBranchBlock()
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5)
    batches = [{"x": torch.randn(200, 48, generator=g) * torch.logspace(0, -1, 48)} for _ in range(3)]
    names = ["q", "k", "v", "o", "gate", "up", "down"]
    calls = []
    with cpu_shim.installed(monkeypatch):
        from ptdeco_amd import ops
        real = ops.syrk_accumulate
        monkeypatch.setattr(ops, "syrk_accumulate", lambda E, y, s: (calls.append(tuple(E.shape)), real(E, y, s))[1])
        pools = []
        orig_pool = eng.SharedInputPool
        monkeypatch.setattr(eng, "SharedInputPool", lambda *a, **k: (pools.append(orig_pool(*a, **k)), pools[-1])[1])
        ref_out = model(batches[0])
        u = dw._precompute_covariance_matrix_decompositions(
            module=model, submodule_names=names, num_data_steps=3, data_iterator=iter(batches), device=CPU,
            decompose_in_float64=True, shard=Shard.from_env(None), min_rank=4, reduction_factor=0.5)
    assert pools[0].groups == expected_groups
    assert len(calls) == 3 * syrks_per_step
    assert all(isinstance(model.get_submodule(n), torch.nn.Linear) for n in names)  # originals restored
    assert torch.equal(model(batches[0]), ref_out)
    # every layer's top-k eigenvectors span the same space as those of its own output covariance
    taps = {n: [] for n in names}
    hooks = [model.get_submodule(n).register_forward_hook(lambda m, i, o, n=n: taps[n].append(o.detach()))
             for n in names]
    with torch.no_grad():
        for b in batches:
            model(b)
    for h in hooks:
        h.remove()
    for n in names:
        lay = model.get_submodule(n)
        ys = [y - lay.bias if lay.bias is not None else y for y in taps[n]]
        e = sum((y.double().T @ y.double()) / y.shape[0] for y in ys) / 3
        e = e + torch.eye(e.shape[0], dtype=torch.float64) * (0.01 * torch.diag(e).mean())
        k = u[n].shape[1]
        v = torch.linalg.eigh(e)[1][:, -k:]
        p, p_ref = u[n].double() @ u[n].double().T, v @ v.T
        assert (p - p_ref).norm().item() <= 1e-5 * k ** 0.5, n


def test_shared_input_pool_rejects_sharing_that_changes_between_steps(monkeypatch):
    from ptdeco_amd import _engine as eng

    class M:
        def __init__(self, name, w):
            self.name, self.weight, self.top_k = name, w, 4

    with cpu_shim.installed(monkeypatch):
        pool = eng.SharedInputPool(2, True, CPU, "all")
        a, b = M("a", torch.randn(8, 6)), M("b", torch.randn(8, 6))
        pool.register(a), pool.register(b)
        x = torch.randn(10, 6)
        pool.begin_step()
        pool.observe(a, x, x @ a.weight.T), pool.observe(b, x, x @ b.weight.T)
        pool.end_step()
        assert pool.groups == [["a", "b"]] and a.moment is b.moment and a.moment.steps == 1
        pool.begin_step()
        pool.observe(a, x, x @ a.weight.T)
        with pytest.raises(RuntimeError, match="shared its input"):
            pool.observe(b, x.clone(), x @ b.weight.T)


class _TiedNet(torch.nn.Module):
    """`rec` is applied at two time steps of one forward (a layer reused across time steps); a / b share an input."""

    def __init__(self, d=24):
        super().__init__()
        self.rec = torch.nn.Linear(d, d, bias=False)
        self.a, self.b = torch.nn.Linear(d, 2 * d, bias=False), torch.nn.Linear(d, 2 * d, bias=False)
        self.out = torch.nn.Linear(2 * d, d, bias=False)

    def forward(self, batch):
        h = torch.tanh(self.rec(batch["x"]))
        h = torch.tanh(self.rec(h))
        return self.out(self.a(h) * torch.sigmoid(self.b(h)))


@pytest.mark.parametrize("inference", [False, True])
def test_precompute_pass_with_a_layer_called_twice_and_with_inference_tensors(inference, monkeypatch):
    """dwain.py:166-208: a stand-in accumulates at EVERY call and divides by its call count.  A layer called twice
    per forward keeps its own statistics (it is not put into a sharing group, and it is not an error); under
    torch.inference_mode() tensors have no version counter, so nothing is shared and nothing raises."""
    from ptdeco_amd import _engine as eng
    from ptdeco_amd.dwain import decomposition as dw
    from ptdeco_amd.sharding import Shard

    monkeypatch.setenv("PTD_SHARE_INPUT_COVARIANCE", "all")
    g = torch.Generator().manual_seed(11)
    model = _TiedNet()
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(torch.randn(p.shape, generator=g) / p.shape[-1] ** 0.5)
    batches = [{"x": torch.randn(64, 24, generator=g) * torch.logspace(0, -1, 24)} for _ in range(2)]
    names = ["rec", "a", "b", "out"]
    feats = {n: [] for n in names}
    hooks = [model.get_submodule(n).register_forward_hook(lambda m, i, o, n=n: feats[n].append(o.detach().clone()))
             for n in names]
    with torch.no_grad():
        for b in batches:
            model(b)
    for h in hooks:
        h.remove()
    pools = []
    with cpu_shim.installed(monkeypatch):
        orig_pool = eng.SharedInputPool
        monkeypatch.setattr(eng, "SharedInputPool", lambda *a, **k: (pools.append(orig_pool(*a, **k)), pools[-1])[1])
        ctx = torch.inference_mode() if inference else torch.no_grad()
        with ctx:
            u = dw._precompute_covariance_matrix_decompositions(
                module=model, submodule_names=names, num_data_steps=2, data_iterator=iter(batches), device=CPU,
                decompose_in_float64=True, shard=Shard.from_env(None), min_rank=2, reduction_factor=0.5)
    assert pools[0].groups == ([] if inference else [["a", "b"]])
    for n in names:
        ys = feats[n]
        assert len(ys) == (4 if n == "rec" else 2)
        e = sum((y.double().T @ y.double()) / y.shape[0] for y in ys) / len(ys)   # Eyyt / num calls
        e = e + torch.eye(e.shape[0], dtype=torch.float64) * (0.01 * torch.diag(e).mean())
        k = u[n].shape[1]
        v = torch.linalg.eigh(e)[1][:, -k:]
        p, p_ref = u[n].double() @ u[n].double().T, v @ v.T
        assert (p - p_ref).norm().item() <= 1e-5 * k ** 0.5, n


def test_phase_timer_counts_a_nested_span_of_the_same_name_once(monkeypatch):
    from ptdeco_amd import _engine as eng

    class Ev:
        count = 0

        def __init__(self, enable_timing=True):
            pass

        def record(self):
            Ev.count += 1

        def elapsed_time(self, other):
            return 1.0

    monkeypatch.setattr(torch.cuda, "Event", Ev)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda: None)
    t = eng.PhaseTimer()
    with t.span("comm"):
        with t.span("comm"):
            with t.span("B_eigh"):
                pass
    with t.span("comm"):
        pass
    assert t.totals_ms() == {"B_eigh": 1.0, "comm": 2.0} and Ev.count == 6


# ---------------------------------------------------------------------------------- prefix memo (eng.PrefixMemo)
class _MemoNet(torch.nn.Module):
    """conv -> relu_ (in place on the conv output) -> flatten -> a (called twice) -> tapped -> a again -> head"""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.conv = torch.nn.Conv2d(3, 4, 1)
        self.a = torch.nn.Linear(16, 16)
        self.b = torch.nn.Linear(16, 16)
        self.tapped = torch.nn.Linear(16, 16)
        self.head = torch.nn.Linear(16, 8)
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)

    def forward(self, x):
        h = torch.relu_(self.conv(x)).flatten(1)          # [B, 16]
        h = self.a(self.a(h)) + self.b(h)
        h = self.tapped(h)
        return self.head(self.a(h))


def test_prefix_memo_replays_only_what_ran_before_the_tapped_layer():
    from ptdeco_amd import _engine as eng

    torch.manual_seed(0)
    model = _MemoNet().eval()
    x = torch.randn(5, 3, 2, 2)
    real = []

    def spy(name, mod):
        inner = type(mod).forward

        def fwd(self_, *a, **k):
            real.append(name)
            return inner(self_, *a, **k)
        return fwd

    # count REAL executions through a subclass forward (class level: the memo wraps instance level above it)
    for n, m in list(model.named_children()):
        m.__class__ = type(f"Spy{n}", (type(m),), {"forward": spy(n, m)})
    with torch.no_grad():
        want = model(x)
        tap = eng.LayerTap(model, "tapped")
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30, check=True)
        real.clear()
        y1, y2 = eng.forward_pair(model, tap, x, lambda: None, lambda: None)
        assert torch.equal(y1, want) and torch.equal(y2, want)
        # first forward: everything; second: conv again (its kept output was modified in place by relu_), a's two
        # calls ahead of the tapped layer and b come from the memo (check=True recomputes them: counted twice here)
        assert tap.memo.hits == 3
        assert real[:7] == ["conv", "a", "a", "b", "tapped", "a", "head"]
        tap.memo.check = False
        real.clear()
        y1, y2 = eng.forward_pair(model, tap, x, lambda: None, lambda: None)
        assert torch.equal(y2, want)
        assert real == ["conv", "a", "a", "b", "tapped", "a", "head", "conv", "tapped", "a", "head"]
        # a budget that holds one [5, 16] f32 output only: the first call of `a`, nothing after it
        tap.memo.close()
        tap.memo = eng.PrefixMemo(model, tap.layer, 5 * 16 * 4 + 5 * 4 * 2 * 2 * 4)
        real.clear()
        y1, y2 = eng.forward_pair(model, tap, x, lambda: None, lambda: None)
        assert torch.equal(y2, want)
        assert real[7:] == ["conv", "a", "b", "tapped", "a", "head"]
        assert all(len(kept) == 0 for _, kept in tap.memo._patched)       # nothing outlives the step
        tap.close()
    assert all("forward" not in m.__dict__ for m in model.modules())


class _MemoBlock(torch.nn.Module):
    def __init__(self, g, flavour="plain"):
        super().__init__()
        self.norm = torch.nn.LayerNorm(16)
        self.fc1 = torch.nn.Linear(16, 32)
        self.fc2 = torch.nn.Linear(32, 16)
        self.flavour = flavour
        self.count = 0
        with torch.no_grad():
            for p in self.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)

    def forward(self, x, extra=None):
        if self.flavour == "counting":
            self.count += 1                      # a side effect: the block is not a function of its arguments
        return x + self.fc2(torch.relu(self.fc1(self.norm(x))))


class _MemoStack(torch.nn.Module):
    def __init__(self, flavours=("plain",) * 4):
        super().__init__()
        g = torch.Generator().manual_seed(9)
        self.embed = torch.nn.Linear(8, 16)
        self.blocks = torch.nn.ModuleList(_MemoBlock(g, f) for f in flavours)
        self.head = torch.nn.Linear(16, 4)
        self.extra = object()
        with torch.no_grad():
            for p in list(self.embed.parameters()) + list(self.head.parameters()):
                p.copy_(torch.randn(p.shape, generator=g) * 0.3)

    def forward(self, x):
        h = self.embed(x)
        for b in self.blocks:
            h = b(h, extra=self.extra) if b.flavour == "object_arg" else b(h)
        return self.head(h)


def _spy_on(model, real):
    def spy(name, mod):
        inner = type(mod).forward

        def fwd(self_, *a, **k):
            real.append(name)
            return inner(self_, *a, **k)
        return fwd

    for n, m in list(model.named_modules()):
        if n and not isinstance(m, torch.nn.ModuleList):
            m.__class__ = type("Spy" + n.replace(".", "_"), (type(m),), {"forward": spy(n, m)})


def test_prefix_memo_replays_whole_subtrees_beside_the_path_to_the_tapped_layer():
    from ptdeco_amd import _engine as eng

    model = _MemoStack().eval()
    x = torch.randn(6, 8)
    real = []
    _spy_on(model, real)
    with torch.no_grad():
        want = model(x)
        tap = eng.LayerTap(model, "blocks.2.fc2")
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30, check=False)
        for step in range(2):      # (the first step also establishes that blocks 0 and 1 are functions of their input)
            real.clear()
            y1, y2 = eng.forward_pair(model, tap, x, lambda: None, lambda: None)
            assert torch.equal(y1, want) and torch.equal(y2, want)
            first = ["embed", "blocks.0", "blocks.0.norm", "blocks.0.fc1", "blocks.0.fc2", "blocks.1", "blocks.1.norm",
                     "blocks.1.fc1", "blocks.1.fc2", "blocks.2", "blocks.2.norm", "blocks.2.fc1", "blocks.2.fc2",
                     "blocks.3", "blocks.3.norm", "blocks.3.fc1", "blocks.3.fc2", "head"]
            assert real[:len(first)] == first
            # second forward: embed, blocks 0 and 1 come back whole (nothing inside them runs), block 2 runs with its
            # fc1 handed back, everything from the tapped layer on is computed
            assert real[len(first):] == ["blocks.2", "blocks.2.norm", "blocks.2.fc2", "blocks.3", "blocks.3.norm",
                                         "blocks.3.fc1", "blocks.3.fc2", "head"], real[len(first):]
            assert tap.memo.unit_hits == 2 * (step + 1)
        # products only (PTD_PREFIX_MEMO_UNITS=products): the blocks run, their Linear layers do not
        tap.memo.close()
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30, subtrees=False)
        real.clear()
        y1, y2 = eng.forward_pair(model, tap, x, lambda: None, lambda: None)
        assert torch.equal(y2, want) and tap.memo.unit_hits == 0
        assert real[18:] == ["blocks.0", "blocks.0.norm", "blocks.1", "blocks.1.norm", "blocks.2", "blocks.2.norm",
                             "blocks.2.fc2", "blocks.3", "blocks.3.norm", "blocks.3.fc1", "blocks.3.fc2", "head"]
        tap.close()
    assert all("forward" not in m.__dict__ for m in model.modules())


def test_prefix_memo_takes_a_subtree_only_if_it_is_a_function_of_its_arguments():
    from ptdeco_amd import _engine as eng

    # block 0 counts its calls (an attribute is rebound), block 1 receives an object that is not a tensor nest,
    # block 2 has a forward hook inside; only their products may be handed back, and every side effect happens twice
    model = _MemoStack(("counting", "object_arg", "plain", "plain")).eval()
    fired = []
    model.blocks[2].fc1.register_forward_hook(lambda m, a, o: fired.append(1))
    x = torch.randn(6, 8)
    with torch.no_grad():
        want = model(x)
        c0, f0 = model.blocks[0].count, len(fired)
        tap = eng.LayerTap(model, "blocks.3.fc2")
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30, check=True)
        for step in range(3):
            y1, y2 = eng.forward_pair(model, tap, x, lambda: None, lambda: None)
            assert torch.equal(y1, want) and torch.equal(y2, want)
        assert tap.memo.unit_hits == 0
        assert model.blocks[0].count == c0 + 6 and len(fired) == f0 + 6
        # step 0 finds block 0 out AFTER running it (nothing inside it is kept in that step; block 1 is refused on its
        # arguments, before it runs): embed + 2 + 2 + block 3's fc1 = 6; steps 1 and 2 keep all 8 products one by one
        assert tap.memo.hits == 6 + 8 + 8
        tap.close()


class _SideEffectBlock(torch.nn.Module):
    """A block whose forward is NOT a function of its arguments, in ways the round-4 purity test did not see."""

    def __init__(self, kind):
        super().__init__()
        self.kind = kind
        self.flavour = "plain"
        self.fc = torch.nn.Linear(16, 16)
        self.seen = []                   # mutated in place: the attribute binding never changes
        self.calls = {"n": 0}

    def forward(self, x):
        import random
        if self.kind == "list_append":
            self.seen.append(x.shape[0])
        elif self.kind == "dict_update":
            self.calls["n"] += 1
        elif self.kind == "python_rng":
            x = x * (1.0 + 1e-3 * random.random())
        elif self.kind == "torch_rng":
            x = x + 1e-3 * torch.rand(())
        return x + self.fc(x)


@pytest.mark.parametrize("kind", ["list_append", "dict_update", "python_rng", "torch_rng"])
def test_prefix_memo_treats_mutated_containers_and_random_generators_as_impure(kind):
    """a subtree that appends to a list attribute, updates a dict attribute in place or draws from a
    global random generator is not a function of its arguments: it is never handed back whole (its Linear still is), its
    side effects happen in both forwards, and the values are what two plain forwards give."""
    import random
    from ptdeco_amd import _engine as eng

    model = _MemoStack().eval()
    model.blocks[1] = _SideEffectBlock(kind).eval()
    x = torch.randn(6, 8)
    with torch.no_grad():
        tap = eng.LayerTap(model, "blocks.3.fc2")
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30)
        random.seed(5); torch.manual_seed(5)
        got = [eng.forward_pair(model, tap, x, lambda: None, lambda: None) for _ in range(3)]
        units = tap.memo.unit_hits
        tap.close()
        random.seed(5); torch.manual_seed(5)
        model.blocks[1].seen.clear(); model.blocks[1].calls["n"] = 0
        want = [(model(x), model(x)) for _ in range(3)]
    for (a1, a2), (b1, b2) in zip(got, want):
        assert torch.equal(a1, b1) and torch.equal(a2, b2)
    # a block with side effects only never comes back whole, blocks 0 and 2 do; behind a block that draws random numbers
    # nothing is kept at all (block 2 sees other values in the second forward)
    assert units == (3 * 2 if kind in ("list_append", "dict_update") else 3 * 1)
    if kind == "list_append":
        assert len(model.blocks[1].seen) == 6
    if kind == "dict_update":
        assert model.blocks[1].calls["n"] == 6


def test_prefix_memo_checks_itself_on_the_first_metric_step_and_switches_off(caplog):
    """A block that reads state OUTSIDE the module tree (a module-level counter here) passes every purity test.  The
    first metric step of a layer recomputes what it is about to hand back: the difference is found, the recomputed value
    is used (the result is what two plain forwards give), one WARNING is logged and the memo is off for this model --
    for this layer's later steps and for the memos of its other layers."""
    import logging
    from ptdeco_amd import _engine as eng

    ticks = [0]

    class Sneaky(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.flavour = "plain"
            self.fc = torch.nn.Linear(16, 16)

        def forward(self, x):
            ticks[0] += 1
            return x + self.fc(x) * (1.0 + 0.01 * ticks[0])

    model = _MemoStack().eval()
    model.blocks[1] = Sneaky().eval()
    x = torch.randn(6, 8)
    eng.PrefixMemo.level.pop(id(model), None)
    with torch.no_grad(), caplog.at_level(logging.WARNING, logger="ptdeco_amd._engine"):
        tap = eng.LayerTap(model, "blocks.3.fc2")
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30, self_check=True)
        got = [eng.forward_pair(model, tap, x, lambda: None, lambda: None) for _ in range(3)]
        assert eng.PrefixMemo.level[id(model)] == 2 and tap.memo.disabled
        hits = tap.memo.hits
        tap.close()
        ticks[0] = 0
        want = [(model(x), model(x)) for _ in range(3)]
        for (a1, a2), (b1, b2) in zip(got, want):
            assert torch.equal(a1, b1) and torch.equal(a2, b2)
        assert hits == 2          # embed and block 0, handed back in the first step before the difference showed
        assert sum("does not compute the same values" in r.message for r in caplog.records) == 1
        # the next layer's memo is off from the start
        tap = eng.LayerTap(model, "blocks.2.fc2")
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30, self_check=True)
        eng.forward_pair(model, tap, x, lambda: None, lambda: None)
        assert tap.memo.hits == 0 and tap.memo.disabled
        tap.close()
    eng.PrefixMemo.level.pop(id(model), None)
    # a model that IS a function of its input passes the self-check and keeps its subtree units
    model = _MemoStack().eval()
    with torch.no_grad():
        want = model(x)
        tap = eng.LayerTap(model, "blocks.3.fc2")
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30, self_check=True)
        for step in range(2):
            y1, y2 = eng.forward_pair(model, tap, x, lambda: None, lambda: None)
            assert torch.equal(y1, want) and torch.equal(y2, want)
        assert tap.memo.unit_hits == 2 * 3 and id(model) not in eng.PrefixMemo.level
        tap.close()


def test_prefix_memo_with_a_weight_tied_layer_inside_and_behind_a_replayed_subtree():
    """a Linear that is called once INSIDE a subtree that comes back whole and twice more directly ahead of
    the analysed layer.  Keyed by a per-module call index, the replay (which skips the call inside the subtree) handed
    the second direct call the first one's output.  Keys are positions in the forward now."""
    from ptdeco_amd import _engine as eng

    class Inner(torch.nn.Module):
        def __init__(self, shared):
            super().__init__()
            self.shared = shared
            self.own = torch.nn.Linear(16, 16)

        def forward(self, x):
            return self.own(torch.relu(self.shared(x)))

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(3)
            self.block = Inner(torch.nn.Linear(16, 16))      # (registered first: the block qualifies as a unit)
            self.shared = self.block.shared
            self.tapped = torch.nn.Linear(16, 16)

        def forward(self, x):
            h = self.block(x)
            h = self.shared(h)                  # the shared layer's second call in the forward
            h = self.shared(torch.tanh(h))      # ... and third
            return self.tapped(h)

    model = Net().eval()
    x = torch.randn(5, 16)
    with torch.no_grad():
        want = model(x)
        tap = eng.LayerTap(model, "tapped")
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30)
        for _ in range(3):
            y1, y2 = eng.forward_pair(model, tap, x, lambda: None, lambda: None)
            assert torch.equal(y1, want) and torch.equal(y2, want)
        assert tap.memo.unit_hits == 3 and tap.memo.hits == 3 * 3
        tap.close()


def test_prefix_memo_leaves_subtrees_in_training_mode_alone():
    from ptdeco_amd import _engine as eng

    model = _MemoStack().eval()
    model.blocks[0].train()
    x = torch.randn(6, 8)
    with torch.no_grad():
        want = model(x)
        tap = eng.LayerTap(model, "blocks.2.fc2")
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30)
        for _ in range(2):
            y1, y2 = eng.forward_pair(model, tap, x, lambda: None, lambda: None)
            assert torch.equal(y2, want)
        assert tap.memo.unit_hits == 2      # block 1 only, in both steps
        tap.close()


def test_prefix_memo_keeps_nothing_under_inference_mode():
    from ptdeco_amd import _engine as eng

    model = _MemoNet().eval()
    x = torch.randn(5, 3, 2, 2)
    with torch.no_grad():
        want = model(x)
    with torch.inference_mode():
        tap = eng.LayerTap(model, "tapped")
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30, check=True)
        y1, y2 = eng.forward_pair(model, tap, x, lambda: None, lambda: None)
        assert tap.memo.hits == 0 and torch.equal(y1, want) and torch.equal(y2, want)
        tap.close()


def test_prefix_memo_is_off_with_a_zero_budget(monkeypatch):
    from ptdeco_amd import _engine as eng

    model = _MemoNet().eval()
    monkeypatch.setenv("PTD_PREFIX_MEMO_MB", "0")
    tap = eng.LayerTap(model, "tapped")
    tap.enable_prefix_memo(model)
    assert tap.memo is None
    tap.close()


@pytest.mark.parametrize("name", ["dwain_mlp_nosplit", "dwain_conv"])
def test_dwain_driver_with_the_prefix_memo_checked_and_without_it(name, monkeypatch):
    scn = gio.e2e_meta()[name]
    traces = []
    for env in ({"PTD_PREFIX_MEMO_CHECK": "1"}, {"PTD_PREFIX_MEMO_MB": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with cpu_shim.installed(monkeypatch) as pkg:
            model = gio.build_model(scn)
            data, metric = gio.dwain_streams(scn)
            trace = []
            pkg.dwain.decompose_in_place(module=model, device=CPU, data_iterator=data, metric_iterator=metric,
                                         loss_fn=tm.ce_loss, finetune_fn=lambda m, d, n: m, trace=trace,
                                         **scn["kwargs"])
        traces.append(trace)
        for k in env:
            monkeypatch.delenv(k)
    assert traces[0] == traces[1]     # bit-identical metrics with and without the memo


@pytest.mark.parametrize("use_cache", [False, True])
def test_prefix_memo_on_a_hugging_face_llama(use_cache):
    """The module signatures of a real decoder stack: decoder layers take keyword tensors, a tuple of position
    embeddings and -- with use_cache=True -- a cache OBJECT that they update in place: then no whole layer may be
    handed back (the cache would miss its entries), only the matrix products inside."""
    transformers = pytest.importorskip("transformers")
    from ptdeco_amd import _engine as eng

    cfg = transformers.LlamaConfig(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=3,
                                   num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=64)
    torch.manual_seed(0)
    llama = transformers.LlamaForCausalLM(cfg).eval()

    class Logits(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.m = llama

        def forward(self, ids):
            return self.m(input_ids=ids, use_cache=use_cache).logits

    model = Logits().eval()
    ids = torch.randint(0, 64, (2, 16))
    with torch.no_grad():
        want = model(ids)
        tap = eng.LayerTap(model, "m.model.layers.2.mlp.down_proj")
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30, check=True)
        for _ in range(2):
            y1, y2 = eng.forward_pair(model, tap, ids, lambda: None, lambda: None)
            assert torch.equal(y1, want) and torch.equal(y2, want)
        if use_cache:
            assert tap.memo.unit_hits == 0 and tap.memo.hits == 2 * (2 * 7 + 4 + 2)   # q k v o gate up down, ...
        else:
            assert tap.memo.unit_hits == 2 * 3     # two decoder layers and the attention block of the third, per step
        tap.close()
    assert all("forward" not in m.__dict__ for m in model.modules())


def test_calibration_forwards_stop_at_the_analysed_layer_after_the_first_one(monkeypatch):
    """The forwards that only feed the covariance (dwain.py:236-239, falor.py:189-193) need the model up to the
    analysed layer: the first runs whole and counts the layer's calls, the later ones stop at it -- unless the layer
    is called more than once per forward (the reference takes the LAST call's input)."""
    from ptdeco_amd import _engine as eng

    model = _MemoStack().eval()
    real = []
    _spy_on(model, real)
    x = torch.randn(6, 8)
    with torch.no_grad():
        tap = eng.LayerTap(model, "blocks.1.fc2")
        for step in range(3):
            real.clear()
            tap.calibration_forward(model, x)
            want = ["embed", "blocks.0", "blocks.0.norm", "blocks.0.fc1", "blocks.0.fc2", "blocks.1", "blocks.1.norm",
                    "blocks.1.fc1"]
            if step == 0:
                assert real[:8] == want and real[8] == "blocks.1.fc2" and real[-1] == "head"
            else:
                assert real == want                      # neither the layer itself nor anything behind it ran
            assert tap.last_input_rows().shape == (6, 32)
        monkeypatch.setenv("PTD_CALIBRATION_EARLY_STOP", "0")
        real.clear()
        tap.calibration_forward(model, x)
        assert real[-1] == "head"
        tap.close()

    class Twice(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(8, 8)
            self.head = torch.nn.Linear(8, 2)

        def forward(self, x):
            return self.head(self.a(self.a(x)))

    monkeypatch.delenv("PTD_CALIBRATION_EARLY_STOP")
    twice = Twice().eval()
    seen = []
    twice.head.register_forward_hook(lambda m, a, o: seen.append(1))
    with torch.no_grad():
        tap = eng.LayerTap(twice, "a")
        for _ in range(3):
            tap.calibration_forward(twice, x)
        assert len(seen) == 3 and tap.calls == 2       # every forward ran to the end
        tap.close()

    #  every 8th forward runs whole again and re-counts the calls: a layer that is called once on most
    # batches and twice on some keeps whole forwards from the moment that is seen; and a hook on a module AROUND the
    # layer (its post-forward work would be skipped by the unwinding) keeps every forward whole
    class Sometimes(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(8, 8)
            self.head = torch.nn.Linear(8, 2)
            self.n = 0

        def forward(self, x):
            self.n += 1
            h = self.a(x)
            if self.n >= 8:
                h = self.a(h)
            return self.head(h)

    model = Sometimes().eval()
    ends = []
    model.head.register_forward_hook(lambda m, a, o: ends.append(model.n))
    with torch.no_grad():
        tap = eng.LayerTap(model, "a")
        for _ in range(12):
            tap.calibration_forward(model, x)
        tap.close()
    assert ends == [1, 8, 9, 10, 11, 12]      # forwards 2 .. 7 stopped at the layer; the 8th ran whole and saw two calls
    wrapped = _MemoStack().eval()
    done = []
    wrapped.blocks[1].register_forward_hook(lambda m, a, o: done.append(1))
    with torch.no_grad():
        tap = eng.LayerTap(wrapped, "blocks.1.fc2")
        for _ in range(3):
            tap.calibration_forward(wrapped, x)
        tap.close()
    assert len(done) == 3
    #  a hook registered on an ancestor AFTER the first forwards is seen at the next revalidation forward
    # (every 8th): from then on the forwards stay whole and the hook's post-forward work happens
    late = _MemoStack().eval()
    fired = []
    with torch.no_grad():
        tap = eng.LayerTap(late, "blocks.1.fc2")
        for _ in range(3):
            tap.calibration_forward(late, x)         # 1 whole, 2 and 3 cut at the layer
        late.blocks[1].register_forward_hook(lambda m, a, o: fired.append(tap._cal_forwards))
        for _ in range(9):
            tap.calibration_forward(late, x)         # 4 .. 7 still cut (the look is cached), 8 re-looks: 8 .. 12 whole
        tap.close()
    assert fired == [8, 9, 10, 11, 12]


def test_step_batch_holds_calibration_steps_and_adds_them_in_one_call(monkeypatch):
    """_engine.StepBatch (round 5): bf16 activation matrices of up to PTD_SYRK_STEPS steps reach the accumulator in ONE
    ops.syrk_accumulate_multi call.  A matrix the caller may still write to is copied (changing it afterwards must not
    change the sum), one nobody else holds is kept by reference, everything held is added before E is read, a change of
    shape or a spent byte budget falls back to adding at once, f32 matrices are never held.  (Host logic on the CPU: the
    two ops are replaced by f64 arithmetic and `holdable` by its dtype test.)"""
    from ptdeco_amd import _engine as eng, ops

    calls = []

    def multi(E, ys, scale):
        calls.append(("multi", len(ys)))
        for y in ys:
            E += scale * torch.tril(y.double().T @ y.double())

    def single(E, y, scale):
        calls.append(("single", 1))
        E += scale * torch.tril(y.double().T @ y.double())

    monkeypatch.setattr(ops, "syrk_accumulate_multi", multi)
    monkeypatch.setattr(ops, "syrk_accumulate", single)
    monkeypatch.setattr(eng.StepBatch, "holdable", staticmethod(lambda y: y.dtype == torch.bfloat16 and y.dim() == 2))
    monkeypatch.setenv("PTD_SYRK_STEPS", "4")
    g = torch.Generator().manual_seed(1)
    ys = [torch.randn(16, 8, generator=g).bfloat16() for _ in range(10)]
    want = sum(torch.tril(y.double().T @ y.double()) / 16 for y in ys)
    cov = eng.Covariance(8, torch.device("cpu"), True)
    held0 = eng.StepBatch.held_bytes
    for i, y in enumerate(ys):
        exposed = y.clone()
        cov.add_features(exposed, private=(i % 2 == 1))
        if i % 2 == 0:
            exposed.zero_()                      # the caller reuses its buffer: the batch holds a copy
    assert calls == [("multi", 4), ("multi", 4)] and len(cov.batch.pending) == 2
    assert eng.StepBatch.held_bytes == held0 + 2 * 16 * 8 * 2
    cov.batch.flush()                            # (what eigenvectors / reductions / finalize do first)
    assert calls[-1] == ("multi", 2) and eng.StepBatch.held_bytes == held0 and cov.steps == 10
    assert torch.allclose(cov.E, want, rtol=1e-12, atol=1e-12)
    # a change of shape adds what is held first; f32 goes straight through; so does everything once the budget is spent
    calls.clear()
    cov2 = eng.Covariance(8, torch.device("cpu"), True)
    cov2.add_features(ys[0].clone()); cov2.add_features(ys[1][:8].clone())
    assert calls == [("multi", 1)] and len(cov2.batch.pending) == 1
    cov2.add_features(ys[2].float())
    assert calls == [("multi", 1), ("multi", 1), ("single", 1)] and not cov2.batch.pending
    monkeypatch.setenv("PTD_SYRK_BUFFER_MB", "0")
    cov3 = eng.Covariance(8, torch.device("cpu"), True)
    calls.clear()
    cov3.add_features(ys[0].clone())
    assert calls == [("single", 1)] and not cov3.batch.pending
    # PTD_SYRK_STEPS=1: every step at once, as in round 4
    monkeypatch.setenv("PTD_SYRK_BUFFER_MB", "4096")
    monkeypatch.setenv("PTD_SYRK_STEPS", "1")
    cov4 = eng.Covariance(8, torch.device("cpu"), True)
    calls.clear()
    cov4.add_features(ys[0].clone())
    assert calls == [("single", 1)]


def test_step_batch_dropped_with_pending_steps_gives_its_bytes_back(monkeypatch):
    """StepBatch.held_bytes is process-wide; a Covariance dropped with steps still held (an exception during
    calibration, a layer skipped before its eigenvectors) must not leave its bytes in the counter -- once the leaked
    total passed PTD_SYRK_BUFFER_MB every later batch would silently add step by step."""
    import gc

    from ptdeco_amd import _engine as eng, ops

    monkeypatch.setattr(ops, "syrk_accumulate_multi", lambda E, ys, s: None)
    monkeypatch.setattr(eng.StepBatch, "holdable", staticmethod(lambda y: y.dtype == torch.bfloat16 and y.dim() == 2))
    held0 = eng.StepBatch.held_bytes
    cov = eng.Covariance(8, torch.device("cpu"), True)
    for _ in range(3):
        cov.add_features(torch.zeros(16, 8).bfloat16())
    assert eng.StepBatch.held_bytes == held0 + 3 * 16 * 8 * 2
    del cov
    gc.collect()
    assert eng.StepBatch.held_bytes == held0


class _StackNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.pre = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Tanh(), torch.nn.Linear(8, 8))
        self.tapped = torch.nn.Linear(8, 8)
        self.post = torch.nn.Linear(8, 4)

    def forward(self, x):
        return self.post(torch.relu(self.tapped(self.pre(x))))


def test_prefix_memo_across_candidates_keeps_prefix_and_original_output_per_batch():
    """Round 6: candidates of one layer evaluated on batches that come round again.  Per batch the prefix
    runs once for the whole search and the original model once; the values are those of running everything; a batch
    modified in place (version counter) or refilled behind torch's back (sampled fingerprint of host tensors) is a new
    batch; PTD_MEMO_ACROSS_CANDIDATES=0 keeps nothing beyond a pair."""
    from ptdeco_amd import _engine as eng

    torch.manual_seed(0)
    model = _StackNet().eval()
    batches = [torch.randn(6, 8) for _ in range(2)]
    w_orig = model.tapped.weight.detach().clone()
    cands = [w_orig * s for s in (0.5, 0.25, 0.125)]
    runs = {"pre": 0, "post": 0}
    for name in runs:
        mod = getattr(model, name)
        inner = type(mod).forward

        def fwd(self_, *a, _n=name, _inner=inner, **k):
            runs[_n] += 1
            return _inner(self_, *a, **k)
        mod.__class__ = type(f"Spy{name}", (type(mod),), {"forward": fwd})

    def set_w(w):
        def go():
            with torch.no_grad():
                model.tapped.weight.copy_(w)
        return go

    def search(memo_on, self_check=False):
        tap = eng.LayerTap(model, "tapped")
        tap.memo = eng.PrefixMemo(model, tap.layer, 1 << 30, self_check=self_check)
        tap.memo.across = memo_on
        out = []
        with torch.no_grad():
            for w in cands:
                for b in batches:
                    y1, y2 = eng.forward_pair(model, tap, b, set_w(w), set_w(w_orig), key=eng.batch_key(b), pin=b)
                    out.append((y1.clone(), y2.clone()))
        memo = tap.memo
        stats = (memo.prefix_replays, memo.orig_hits)
        tap.close()
        set_w(w_orig)()
        return out, stats

    with torch.no_grad():
        want = []
        for w in cands:
            for b in batches:
                set_w(w)()
                y1 = model(b)
                set_w(w_orig)()
                want.append((y1, model(b)))
    runs.update(pre=0, post=0)
    got, stats = search(True)
    for (a1, a2), (b1, b2) in zip(got, want):
        assert torch.equal(a1, b1) and torch.equal(a2, b2)
    # 6 pairs: the prefix ran once per batch (2), the original model once per batch (2 of the 8 suffix runs)
    assert runs == {"pre": 2, "post": 6 + 2} and stats == (4, 4)
    runs.update(pre=0, post=0)
    got, stats = search(True, self_check=True)      # the first replay of each kind is recomputed and compared
    assert all(torch.equal(a1, b1) and torch.equal(a2, b2) for (a1, a2), (b1, b2) in zip(got, want))
    assert runs == {"pre": 2 + 2, "post": 6 + 2 + 1} and stats == (4, 3)
    runs.update(pre=0, post=0)
    got, stats = search(False)
    assert all(torch.equal(a1, b1) and torch.equal(a2, b2) for (a1, a2), (b1, b2) in zip(got, want))
    assert runs == {"pre": 6, "post": 12} and stats == (0, 0)
    # a batch changed in place, or refilled through numpy, is another batch
    b = torch.randn(6, 8)
    k0 = eng.batch_key(b)
    b.add_(1.0)
    k1 = eng.batch_key(b)
    b.numpy()[:] = 3.0
    assert k0 != k1 and k1 != eng.batch_key(b)
    assert eng.batch_key({"x": b, "n": 3}) is not None and eng.batch_key({"x": b, "cache": object()}) is None
    # plain values the model may read are part of the identity; a batch without tensors has none
    assert eng.batch_key({"x": b, "n": 3}) != eng.batch_key({"x": b, "n": 4}) and eng.batch_key({"n": 3}) is None
