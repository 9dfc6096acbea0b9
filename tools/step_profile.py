"""Developer probe: host-side profile of one bench step (cProfile) + phase wall times."""
import cProfile, copy, itertools, pstats, sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench, ptdeco_amd
dev = torch.device("cuda", 0)
model0, data, metric = bench.make_workload(1, dev, bench.D_STEPS, 7 * bench.M_STEPS)
model0.to(dev)
data, metric = bench.with_targets(model0, data, dev), bench.with_targets(model0, metric, dev)
def step():
    model = copy.deepcopy(model0)
    return ptdeco_amd.dwain.decompose_in_place(module=model, device=dev, data_iterator=itertools.cycle(data),
        loss_fn=bench.ce_loss, metric_iterator=itertools.cycle(metric), finetune_fn=lambda m, d, n: m, **bench.DWAIN_KW)
step(); torch.cuda.synchronize()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); print("step wall ms", (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
