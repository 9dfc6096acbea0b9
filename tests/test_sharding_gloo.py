"""N > 1 path on CPU: two gloo ranks run the sharded drivers (HIP ops swapped for the oracle
arithmetic, tests/cpu_shim.py) and must reproduce the sequential reference goldens exactly:
same decisions on every rank, same config, same final weights on every rank."""

import json
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Patch:
    """Minimal monkeypatch (the pytest fixture does not exist in the spawned ranks)."""

    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def _worker(rank, world, port, name, outdir, collective="reduce"):
    os.environ["PTD_COV_COLLECTIVE"] = collective
    for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    torch.set_float32_matmul_precision("highest")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cpu_shim
    import golden_io as gio
    import toy_models as tm

    scn = gio.e2e_meta()[name]
    cpu = torch.device("cpu")
    with cpu_shim.installed(_Patch()) as pkg:
        model = gio.build_model(scn)
        trace = []
        if name.startswith("falor"):
            cfg = pkg.falor.decompose_in_place(module=model, device=cpu, trace=trace,
                                               data_iterator=tm.cycle_tensors(gio.pool(scn["pool"])),
                                               **scn["kwargs"])
            out = model(gio.pool(scn["pool"])[0])
        else:
            data, metric = gio.dwain_streams(scn)
            cfg = pkg.dwain.decompose_in_place(module=model, device=cpu, data_iterator=data,
                                               metric_iterator=metric, loss_fn=tm.ce_loss,
                                               finetune_fn=lambda m, d, n: m, trace=trace, **scn["kwargs"])
            out = model({"x": gio.pool(scn["pool"])[0]})
    torch.save({"cfg": json.loads(json.dumps(cfg)), "trace": trace, "out": out.detach(),
                "state": {k: v.detach().clone() for k, v in model.state_dict().items()}},
               os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _run(name, world=2, collective="reduce"):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), name, d, collective), nprocs=world, join=True)
        return [torch.load(os.path.join(d, f"rank{r}.pt"), weights_only=False) for r in range(world)]


@pytest.mark.parametrize("name", ["dwain_mlp_nosplit", "dwain_mlp_split2", "dwain_conv"])
def test_dwain_two_ranks_match_sequential_golden(name):
    sys.path.insert(0, HERE)
    import golden_io as gio

    scn = gio.e2e_meta()[name]
    r0, r1 = _run(name)
    want = [(s["layer"], s["rank"], s["accepted"]) for s in scn["steps"]]
    for r in (r0, r1):
        assert [(s["layer"], s["rank"], s["accepted"]) for s in r["trace"]] == want
        assert list(r["cfg"].keys()) == list(scn["config"].keys())
        for layer, c in scn["config"].items():
            got = r["cfg"][layer]
            assert got["modules"] == c["modules"]
            for k, v in c["__meta__"].items():
                assert got["__meta__"][k] == pytest.approx(v, rel=1e-5, abs=1e-6)
    # replicas stay identical
    assert r0["state"].keys() == r1["state"].keys()
    for k in r0["state"]:
        assert torch.equal(r0["state"][k], r1["state"][k]), k
    ref = gio.t(gio.npz("e2e")[f"{name}.final_out"])
    assert (r0["out"] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("name", ["dwain_mlp_split2", "falor_mlp_r9"])
def test_all_reduce_form_of_the_covariance_exchange_gives_the_same_run(name):
    """PTD_COV_COLLECTIVE=allreduce (north_star's literal collective: every rank receives the covariance sum) against the
    default reduce-to-owner: same decisions, same config, bit-identical final weights on both ranks."""
    a0, a1 = _run(name, collective="allreduce")
    r0, r1 = _run(name, collective="reduce")
    key = (lambda s: (s["layer"], s["rank"], s.get("accepted")))
    assert sorted(map(key, a0["trace"] + a1["trace"])) == sorted(map(key, r0["trace"] + r1["trace"]))
    assert a0["cfg"] == r0["cfg"] and a1["cfg"] == r1["cfg"]
    for k in r0["state"]:
        assert torch.equal(a0["state"][k], r0["state"][k]) and torch.equal(a1["state"][k], r1["state"][k]), k


@pytest.mark.parametrize("name", ["falor_mlp_r9", "falor_conv"])
def test_falor_two_ranks_match_sequential_golden(name):
    sys.path.insert(0, HERE)
    import golden_io as gio

    scn = gio.e2e_meta()[name]
    r0, r1 = _run(name)
    # each rank traced only the layers it owns; together they are the sequential trace
    merged = sorted(r0["trace"] + r1["trace"], key=lambda s: (list(dict.fromkeys(
        t["layer"] for t in scn["steps"])).index(s["layer"]), s["i"]))
    assert [(s["layer"], s["rank"]) for s in merged] == [(s["layer"], s["rank"]) for s in scn["steps"]]
    assert {s["layer"] for s in r0["trace"]}.isdisjoint({s["layer"] for s in r1["trace"]})
    for r in (r0, r1):
        assert list(r["cfg"].keys()) == list(scn["config"].keys())
        for layer, c in scn["config"].items():
            assert r["cfg"][layer]["modules"] == c["modules"]
            for k, v in c["__meta__"].items():
                assert r["cfg"][layer]["__meta__"][k] == pytest.approx(v, rel=1e-5, abs=1e-6)
    for k in r0["state"]:
        assert torch.equal(r0["state"][k], r1["state"][k]), k
    ref = gio.t(gio.npz("e2e")[f"{name}.final_out"])
    assert (r1["out"] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


def test_pack_lower_round_trip_and_traffic():
    """The covariance exchange ships the lower triangle only: slabs E[i0:i1, :i1] (no index tensors)."""
    from ptdeco_amd.sharding import pack_lower, unpack_lower

    for n, block in [(5, 2), (64, 16), (100, 32), (1000, 512)]:
        g = torch.Generator().manual_seed(n)
        e = torch.randn(n, n, generator=g, dtype=torch.float64)
        packed = pack_lower(e, block)
        assert packed.numel() <= n * n // 2 + n * block
        out = torch.full((n, n), 7.0, dtype=torch.float64)
        unpack_lower(packed, out, block)
        assert torch.equal(torch.tril(out), torch.tril(e))
    assert pack_lower(torch.zeros(4096, 4096)).numel() / 4096**2 < 0.57


@pytest.mark.parametrize("name", ["dwain_mlp_f32acc", "dwain_mlp_split2", "falor_mlp_mean32"])
def test_three_ranks_with_a_layer_count_they_do_not_divide(name):
    """World size 3: two decomposable layers (one rank owns none; 4 calibration steps dealt 2 / 1 / 1), and three
    layers precomputed in three one-layer splits (rank 0 owns every eigendecomposition, ranks 1 and 2 only
    contribute partial sums through the packed-triangle reduce and receive the eigenvectors).  Every rank must
    reproduce the sequential golden run and end with identical weights."""
    sys.path.insert(0, HERE)
    import golden_io as gio

    scn = gio.e2e_meta()[name]
    runs = _run(name, world=3)
    if name.startswith("dwain"):
        want = [(s["layer"], s["rank"], s["accepted"]) for s in scn["steps"]]
        for r in runs:
            assert [(s["layer"], s["rank"], s["accepted"]) for s in r["trace"]] == want
    else:
        merged = [s for r in runs for s in r["trace"]]
        order = list(dict.fromkeys(t["layer"] for t in scn["steps"]))
        merged.sort(key=lambda s: (order.index(s["layer"]), s["i"]))
        assert [(s["layer"], s["rank"]) for s in merged] == [(s["layer"], s["rank"]) for s in scn["steps"]]
        assert sum(1 for r in runs if not r["trace"]) == 1   # two layers, three ranks
    for r in runs:
        assert list(r["cfg"].keys()) == list(scn["config"].keys())
        for layer, c in scn["config"].items():
            assert r["cfg"][layer]["modules"] == c["modules"]
            for k, v in c["__meta__"].items():
                assert r["cfg"][layer]["__meta__"][k] == pytest.approx(v, rel=1e-5, abs=1e-6)
    for k in runs[0]["state"]:
        assert torch.equal(runs[0]["state"][k], runs[1]["state"][k]) and torch.equal(runs[0]["state"][k], runs[2]["state"][k]), k
    ref = gio.t(gio.npz("e2e")[f"{name}.final_out"])
    assert (runs[2]["out"] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
