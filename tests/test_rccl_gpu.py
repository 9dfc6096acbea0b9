"""The sharded code path on REAL RCCL work objects without a second GPU: a 1-rank `nccl` process
group in the pytest process, a Shard that reports itself active, and the three-layer precompute split of dwain run
through it -- `dist.reduce(async_op=True)` on the packed lower triangles completed from the worker threads and side
streams of `run_concurrently`, the eigenvector broadcasts, the all-reduce of the metric sums.  gloo on the CPU (the
world-size-2 / -3 tests) cannot exercise the RCCL work objects, `work.wait()` on non-default streams or the allocator
lifetime of the packed buffers; a 1-rank communicator runs the same kernels and stream hand-offs with itself as peer.
The sharded run must take the unsharded run's decisions, with metrics and factors equal to the eigensolver's
run-to-run reproducibility."""

import copy
import itertools
import os
import socket
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda", 0)


@pytest.fixture(scope="module")
def one_rank_nccl():
    import torch.distributed as dist

    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(DEV)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=DEV)
    try:
        yield dist
    finally:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def test_three_layer_split_through_a_one_rank_rccl_group_equals_the_unsharded_run(one_rank_nccl, monkeypatch):
    import bench
    import ptdeco_amd
    from ptdeco_amd import sharding
    from ptdeco_amd.dwain import decomposition as dw

    model, data, metric = bench.make_workload(3, "cpu", bench.D_STEPS, 7 * bench.M_STEPS)
    cpu = torch.device("cpu")
    data_c, metric_c = bench.with_targets(model, data, cpu), bench.with_targets(model, metric, cpu)
    data_g = [{k: v.to(DEV) for k, v in b.items()} for b in data_c]
    metric_g = [{k: v.to(DEV) for k, v in b.items()} for b in metric_c]

    def run():
        m = copy.deepcopy(model).to(DEV)
        trace = []
        cfg = ptdeco_amd.dwain.decompose_in_place(
            module=m, device=DEV, data_iterator=itertools.cycle(data_g), loss_fn=bench.ce_loss,
            metric_iterator=itertools.cycle(metric_g), finetune_fn=lambda mm, d, n: mm, trace=trace,
            precomputing_covariance_num_splits=1, **bench.DWAIN_KW)
        torch.cuda.synchronize()
        return cfg, trace, m

    cfg0, trace0, m0 = run()                       # world size 1: Shard.from_env gives the inactive shard

    calls = {"reduce_async": [], "broadcast": 0, "all_reduce_small": 0, "waits": []}

    class ForcedShard(sharding.Shard):
        """One rank that behaves as a member of a sharded job: every collective is issued (on the 1-rank RCCL group)."""

        @property
        def active(self):
            return True

        def reduce_lower_to_owner_async(self, E, index):
            done = super().reduce_lower_to_owner_async(E, index)
            calls["reduce_async"].append(index)

            def complete():
                calls["waits"].append((index, threading.get_ident(), torch.cuda.current_stream(DEV).cuda_stream))
                done()
            return complete

        def broadcast_from_owner(self, *a, **k):
            calls["broadcast"] += 1
            return super().broadcast_from_owner(*a, **k)

        def all_reduce_small(self, t):
            calls["all_reduce_small"] += 1
            return super().all_reduce_small(t)

    monkeypatch.setattr(dw.Shard, "from_env", classmethod(lambda cls, group=None: ForcedShard(group, 0, 1)))
    cfg1, trace1, m1 = run()

    assert calls["reduce_async"] == [0, 1, 2] and calls["broadcast"] == 3 and calls["all_reduce_small"] == 3
    # the exchanges were completed by the worker threads of run_concurrently, each on its own side stream
    main = threading.get_ident()
    assert sorted(w[0] for w in calls["waits"]) == [0, 1, 2]
    assert all(w[1] != main for w in calls["waits"]) and len({w[2] for w in calls["waits"]}) == 3
    assert torch.cuda.current_stream(DEV).cuda_stream not in {w[2] for w in calls["waits"]}
    # decisions identical; values to the run-to-run reproducibility of the eigensolver (the direct route adds its
    # back-transformation's K-split partial sums with atomics: eigenvectors repeat to ~1e-15), which is
    # what two UNSHARDED runs differ by as well
    assert [(t["layer"], t["rank"], t["accepted"]) for t in trace1] == [(t["layer"], t["rank"], t["accepted"]) for t in trace0]
    # (ppl_diff is a difference of two f32 losses of order one: a last-place flip of either is 6e-8)
    for a, b in zip(trace0, trace1):
        assert abs(a["nsr"] - b["nsr"]) <= 1e-8 * abs(a["nsr"]) and abs(a["ppl_diff"] - b["ppl_diff"]) <= 1e-6 * abs(a["ppl_diff"]) + 3e-7, (a, b)
    assert cfg1.keys() == cfg0.keys()
    for name in cfg0:
        m0_, m1_ = cfg0[name]["__meta__"], cfg1[name]["__meta__"]
        assert m0_["proportion"] == m1_["proportion"] and m0_["drop_in_params"] == m1_["drop_in_params"]
        assert {k: v for k, v in cfg0[name].items() if k != "__meta__"} == {k: v for k, v in cfg1[name].items() if k != "__meta__"}
    for (ka, va), (kb, vb) in zip(m0.state_dict().items(), m1.state_dict().items()):
        assert ka == kb and va.shape == vb.shape
        assert (va.double() - vb.double()).abs().max().item() <= 1e-5 * va.double().abs().max().item(), ka
