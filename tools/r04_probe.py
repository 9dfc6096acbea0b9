"""Round-4 baseline probe (one MI355X): eigensolver phase splits at the shapes the routes serve, covariance SYRK at the
C4 shapes, and the stream count of the concurrent eigendecompositions.  Usage: python tools/r04_probe.py [eigh] [syrk] [streams]"""
import copy, itertools, json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptdeco_amd
from ptdeco_amd import ops

dev = torch.device("cuda", 0)
what = set(sys.argv[1:]) or {"eigh", "syrk", "streams"}
out = {}


def ev_time(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def cov(n, t=4096, seed=3):
    g = torch.Generator(device=dev).manual_seed(seed)
    scale = torch.logspace(0, -2, n, device=dev)
    e = torch.zeros(n, n, dtype=torch.float64, device=dev)
    for _ in range(2):
        y = torch.randn(t, n, generator=g, device=dev) * scale
        ops.syrk_accumulate(e, y, 1.0 / t)
    return ops.cov_finalize(e, 2, 0.01)


if "eigh" in what:
    res = []
    for n, k in ((4096, 1024), (4096, 2048), (4096, 1365), (3072, 1536), (2560, 2048), (2048, 1024), (2048, 2047), (1280, 1024), (1024, 512), (768, 767)):
        c = cov(n)
        ops.eigh(c, k, all_values=False); torch.cuda.synchronize()
        t = ev_time(lambda: ops.eigh(c, k, all_values=False), iters=3, warm=1)
        ops.EIGH_PROFILE = []
        ops.eigh(c, k, all_values=False)
        p, ops.EIGH_PROFILE = ops.EIGH_PROFILE[0], None
        line = {"n": n, "k": k, "ms": t, "method": p["method"], "profile_ms": p["ms"], "launches": p["launches"], "profile_total": p["total_ms"]}
        print(json.dumps(line), file=sys.stderr, flush=True)
        res.append(line)
    out["eigh"] = res

if "syrk" in what:
    res = []
    for n, t in ((4096, 4096), (4096, 2048), (1024, 2048), (14336, 2048), (4096, 16384), (8192, 4096)):
        for dt in (torch.bfloat16, torch.float32):
            y = torch.randn(t, n, device=dev).to(dt)
            e = torch.zeros(n, n, dtype=torch.float64, device=dev)
            ms = ev_time(lambda: ops.syrk_accumulate(e, y, 1.0 / t), iters=20, warm=3)
            fl = t * n * (n + 1)
            by = y.element_size() * t * n + 8 * n * (n + 1)      # y once, the live triangle read + written
            line = {"n": n, "T": t, "dtype": str(dt), "ms": ms, "tflops": fl / ms / 1e9, "algorithmic_gbps": by / ms / 1e6,
                    "mfma_bound_ms": fl / (2.5e15 if dt == torch.bfloat16 else 157.3e12) * 1e3, "hbm_bound_ms": by / 8e12 * 1e3}
            print(json.dumps(line), file=sys.stderr, flush=True)
            res.append(line)
            del y, e
    out["syrk"] = res

if "streams" in what:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import bench
    import fullwidth_cases as fc
    res = []
    # (a) three square layers in one split: three filtered chains
    model, data, metric = bench.make_workload(3, dev, bench.D_STEPS, 7 * bench.M_STEPS)
    model.to(dev)
    data, metric = bench.with_targets(model, data, dev), bench.with_targets(model, metric, dev)
    # (b) the 2-block full-width Llama stack, bf16 (the throughput configuration)
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.device(dev):
        stack = fc.LlamaStack(2).to(torch.bfloat16)
    with torch.no_grad():
        for p in stack.parameters():
            p.copy_((torch.randn(p.shape, generator=g, device=dev) / p.shape[1] ** 0.5).to(torch.bfloat16))
    xs = [torch.randn(1, 2048, 4096, generator=g, device=dev).to(torch.bfloat16) for _ in range(12)]
    with torch.no_grad():
        bt = [{"x": x, "targets": stack({"x": x}).argmax(-1)} for x in xs]
    from ptdeco_amd import _engine as eng
    for streams, by_route in ((1, "0"), (2, "0"), (3, "0"), (3, "1"), (1, "0"), (2, "0"), (3, "0"), (3, "1")):
        os.environ["PTD_EIGH_STREAMS"] = str(streams)
        os.environ["PTD_EIGH_STREAMS_BY_ROUTE"] = by_route
        def chain3():
            m = copy.deepcopy(model)
            return ptdeco_amd.dwain.decompose_in_place(module=m, device=dev, data_iterator=itertools.cycle(data), loss_fn=bench.ce_loss,
                                                       metric_iterator=itertools.cycle(metric), finetune_fn=lambda mm, d, n: mm,
                                                       precomputing_covariance_num_splits=1, **bench.DWAIN_KW)
        def llama2():
            m = copy.deepcopy(stack)
            return ptdeco_amd.dwain.decompose_in_place(module=m, device=dev, data_iterator=itertools.cycle(bt), loss_fn=fc.seq_ce,
                                                       metric_iterator=itertools.cycle(bt[8:]), num_data_steps=8, num_metric_steps=2,
                                                       nsr_final_threshold=1.0, finetune_fn=lambda mm, d, n: mm,
                                                       blacklisted_module_names=["head"], precomputing_covariance_num_splits=4)
        for name, fn in (("chain3_f32", chain3), ("llama2_bf16", llama2)):
            fn(); torch.cuda.synchronize()
            eng.PHASES = eng.PhaseTimer()
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            ph, eng.PHASES = eng.PHASES.totals_ms(), None
            line = {"workload": name, "streams": streams, "filtered_chains_alone": by_route == "1", "seconds": dt, "B_eigh_ms": ph.get("B_eigh"), "A_ms": ph.get("A_accumulate"), "D_ms": ph.get("D_metrics")}
            print(json.dumps(line), file=sys.stderr, flush=True)
            res.append(line)
    out["streams"] = res

print(json.dumps(out))
