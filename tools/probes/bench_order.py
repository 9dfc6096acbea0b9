"""Probe (round 5): does B_eigh of the Llama block depend on what ran before it in the process?  bench.py measured
243 ms where tools/r05_probe.py block measured 187 ms on the same verified streams."""
import copy, ctypes, itertools, json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, ptdeco_amd
from ptdeco_amd import _engine as eng, _hip
dev = torch.device("cuda", 0)

def block_setup(dt):
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.device(dev):
        model0 = bench.LlamaStack(1)
    with torch.no_grad():
        for prm in model0.parameters():
            prm.copy_(torch.randn(prm.shape, generator=g, device=dev) / prm.shape[1] ** 0.5)
    model0.to(dt)
    scale = torch.logspace(0, -2, bench.D_MODEL, device=dev)
    xs = [(torch.randn(1, 2048, bench.D_MODEL, generator=g, device=dev) * scale).to(dt) for _ in range(12)]
    with torch.no_grad():
        bt = [{"x": x, "targets": model0({"x": x}).argmax(-1)} for x in xs]
    return model0, bt

def block_step(model0, bt):
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

def overlaps():
    st = eng.chain_streams(dev, 4)
    lib, wall = _hip.load(), ctypes.c_double(0.0)
    out = []
    for i in range(4):
        for j in range(i + 1, 4):
            lib.ptd_stream_pair_wall_us(st[i].cuda_stream, st[j].cuda_stream, 150, ctypes.byref(wall))
            out.append(round(wall.value))
    return out

order = sys.argv[1] if len(sys.argv) > 1 else "bf16,f32,bf16"
for what in order.split(","):
    if what in ("bf16", "f32"):
        m0, bt = block_setup(torch.bfloat16 if what == "bf16" else torch.float32)
        block_step(m0, bt)
        print(what, [block_step(m0, bt) for _ in range(3)], "pair walls", overlaps(), flush=True)
        del m0, bt
    elif what == "stack":
        model, data, metric = bench.make_workload(8, dev, 8, 14)
        model.to(dev)
        data, metric = bench.with_targets(model, data, dev), bench.with_targets(model, metric, dev)
        for _ in range(2):
            mm = copy.deepcopy(model)
            ptdeco_amd.dwain.decompose_in_place(module=mm, device=dev, data_iterator=itertools.cycle(data), loss_fn=bench.ce_loss,
                                                metric_iterator=itertools.cycle(metric), finetune_fn=lambda m, d, n: m,
                                                precomputing_covariance_num_splits=1, **dict(bench.DWAIN_KW, num_data_steps=8))
        torch.cuda.synchronize()
        print("stack done; pair walls", overlaps(), flush=True)
        del model, data, metric
    elif what == "kernels":
        bench.kernel_lines(dev); print("kernel_lines done; pair walls", overlaps(), flush=True)
    elif what == "fwd":
        bench.decomposed_forward_lines(dev); print("decomposed_forward_lines done; pair walls", overlaps(), flush=True)
    elif what == "shapes":
        bench.llama_shape_lines(dev); print("llama_shape_lines done; pair walls", overlaps(), flush=True)
    elif what == "libeigh":
        c = torch.randn(4096, 4096, dtype=torch.float64, device=dev); c = c @ c.T
        torch.linalg.eigh(c); torch.cuda.synchronize(); print("torch.linalg.eigh done; pair walls", overlaps(), flush=True)
    elif what == "empty":
        torch.cuda.empty_cache()
        print("cache emptied", flush=True)
