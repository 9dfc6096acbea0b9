"""Round-5 probe (one MI355X).  Usage: python tools/r05_probe.py [eigh] [block] [syrk]
  eigh   ptd_eigh_topk at the shapes of the direct route, with the phase split of ptd_eigh_profiled
  block  B_eigh of the full-width Llama block (bf16) over several passes at PTD_EIGH_STREAMS = 4 / 6 / 7, on streams that
         are verified to sit on distinct hardware queues (ptdeco_amd._engine.chain_streams)
  syrk   the covariance product at the calibration shapes: one call per step against the multi-step entry"""
import copy, itertools, json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptdeco_amd
from ptdeco_amd import ops, _engine as eng

dev = torch.device("cuda", 0)
what = set(a for a in sys.argv[1:] if not a.startswith("-")) or {"eigh", "block", "syrk"}
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
    for n, k in ((4096, 2048), (4096, 1024), (3072, 1536), (2048, 1024), (1280, 1024), (1024, 512)):
        c = cov(n)
        ops.eigh(c, k, all_values=False); torch.cuda.synchronize()
        t = ev_time(lambda: ops.eigh(c, k, all_values=False), iters=3, warm=1)
        ops.EIGH_PROFILE = []
        ops.eigh(c, k, all_values=False)
        p, ops.EIGH_PROFILE = ops.EIGH_PROFILE[0], None
        line = {"n": n, "k": k, "ms": round(t, 2), "method": p["method"], "profile_ms": [round(x, 2) for x in p["ms"]],
                "launches": p["launches"], "profile_total": round(p["total_ms"], 2)}
        print(json.dumps(line), file=sys.stderr, flush=True)
        res.append(line)
    out["eigh"] = res

if "block" in what:
    import bench
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.device(dev):
        model0 = bench.LlamaStack(1)
    with torch.no_grad():
        for prm in model0.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g, device=dev) / prm.shape[1] ** 0.5)
    model0.to(torch.bfloat16)
    scale = torch.logspace(0, -2, bench.D_MODEL, device=dev)
    xs = [(torch.randn(1, 2048, bench.D_MODEL, generator=g, device=dev) * scale).to(torch.bfloat16) for _ in range(12)]
    with torch.no_grad():
        bt = [{"x": x, "targets": model0({"x": x}).argmax(-1)} for x in xs]

    def step():
        m = copy.deepcopy(model0)
        eng.PHASES = eng.PhaseTimer()
        t0 = time.perf_counter()
        ptdeco_amd.dwain.decompose_in_place(module=m, device=dev, data_iterator=itertools.cycle(bt), loss_fn=bench.seq_ce,
                                            metric_iterator=itertools.cycle(bt[8:]), finetune_fn=lambda mm, d, n: mm,
                                            **bench.C4_BLOCK_KW)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ph, eng.PHASES = eng.PHASES.totals_ms(), None
        return round(dt * 1e3, 1), round(ph["B_eigh"], 1)

    step()
    res = []
    passes = 5
    for streams, longest in (("4", "0"), ("4", "1"), ("6", "0"), ("6", "1"), ("7", "0"), ("7", "1"), ("4", "0")):
        os.environ["PTD_EIGH_STREAMS"] = streams
        os.environ["PTD_EIGH_LONGEST_FIRST"] = longest
        line = {"streams": int(streams), "longest_first": longest == "1", "step_ms, B_eigh_ms": [step() for _ in range(passes)]}
        print(json.dumps(line), file=sys.stderr, flush=True)
        res.append(line)
    out["block"] = res

if "syrk" in what:
    res = []
    for n, t, steps in ((4096, 2048, 8), (1024, 2048, 8), (14336, 2048, 8), (4096, 4096, 4)):
        for dt in (torch.bfloat16, torch.float32):
            ys = [torch.randn(t, n, device=dev).to(dt) for _ in range(steps)]
            e = torch.zeros(n, n, dtype=torch.float64, device=dev)

            def one_by_one():
                for y in ys:
                    ops.syrk_accumulate(e, y, 1.0 / t)

            ms1 = ev_time(one_by_one, iters=10, warm=2) / steps
            line = {"n": n, "T": t, "steps": steps, "dtype": str(dt), "ms_per_step_single": round(ms1, 4)}
            if hasattr(ops, "syrk_accumulate_multi"):
                ms2 = ev_time(lambda: ops.syrk_accumulate_multi(e, ys, 1.0 / t), iters=10, warm=2) / steps
                line["ms_per_step_multi"] = round(ms2, 4)
            fl = t * n * (n + 1)
            by = ys[0].element_size() * t * n + 8 * n * (n + 1)
            line.update(mfma_bound_ms=round(fl / (2.5e15 if dt == torch.bfloat16 else 157.3e12) * 1e3, 4),
                        hbm_bound_ms_single=round(by / 8e12 * 1e3, 4),
                        hbm_bound_ms_multi=round((ys[0].element_size() * t * n + 8 * n * (n + 1) / steps) / 8e12 * 1e3, 4))
            print(json.dumps(line), file=sys.stderr, flush=True)
            res.append(line)
    out["syrk"] = res

print(json.dumps(out))
