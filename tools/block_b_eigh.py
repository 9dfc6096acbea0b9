"""B_eigh of bench.py's full-width Llama block (bf16) over several passes under the engine's plans (CONFIGS=split,seq,threads).
Usage: python tools/block_b_eigh.py [passes]"""
import copy, itertools, json, os, sys, time, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench, ptdeco_amd
from ptdeco_amd import _engine as eng
dev = torch.device("cuda", 0)
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
g = torch.Generator(device=dev).manual_seed(0)
with torch.device(dev):
    model0 = bench.LlamaStack(int(os.environ.get('BLOCKS', '1')))
with torch.no_grad():
    for prm in model0.parameters():
        prm.copy_(torch.randn(prm.shape, generator=g, device=dev) / prm.shape[1] ** 0.5)
model0.to(torch.bfloat16)
scale = torch.logspace(0, -2, bench.D_MODEL, device=dev)
xs = [(torch.randn(1, 2048, bench.D_MODEL, generator=g, device=dev) * scale).to(torch.bfloat16) for _ in range(12)]
with torch.no_grad():
    bt = [{"x": x, "targets": model0({"x": x}).argmax(-1)} for x in xs]


def step():
    if os.environ.get("EMPTY"):
        torch.cuda.empty_cache()
    m = copy.deepcopy(model0)
    eng.PHASES = eng.PhaseTimer()
    t0 = time.perf_counter()
    ptdeco_amd.dwain.decompose_in_place(module=m, device=dev, data_iterator=itertools.cycle(bt), loss_fn=bench.seq_ce,
                                        metric_iterator=itertools.cycle(bt[8:]), finetune_fn=lambda mm, d, n: mm,
                                        **dict(bench.C4_BLOCK_KW, trade_off_factor=20.0 * int(os.environ.get('BLOCKS', '1'))))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ph, eng.PHASES = eng.PHASES.totals_ms(), None
    return round(dt * 1e3, 1), round(ph["B_eigh"], 1)


import gc
if os.environ.get("NOGC"): gc.disable()
step()
configs = [c for c in os.environ.get("CONFIGS", "2x2,3x2,2x3,2x4,1x4,threads").split(",")]
for cfg in configs:
    # LxC: PTD_EIGH_LANES x PTD_EIGH_BATCH_MAX of the batched engine; threads: round 5's one thread and stream per chain
    os.environ["PTD_EIGH_BATCHED"] = "0" if cfg == "threads" else "1"
    if cfg == "auto":       # the engine's defaults: three lanes, matrices per launch from the pass
        for v in ("PTD_EIGH_LANES", "PTD_EIGH_BATCH_MAX", "PTD_LANE_CUS"):
            os.environ.pop(v, None)
        os.environ["PTD_LANE_STREAMS"] = "dedicated"
    elif cfg != "threads":
        # LxC[:pool | :a,b,c]  -- lanes x batch cap, then the lane streams: torch pool streams (probed), or dedicated streams
        # with that many CUs each
        lc, _, extra = cfg.partition(":")
        os.environ["PTD_EIGH_LANES"], os.environ["PTD_EIGH_BATCH_MAX"] = lc.split("x")
        os.environ["PTD_LANE_STREAMS"] = "pool" if extra == "pool" else "dedicated"
        os.environ["PTD_LANE_CUS"] = extra.replace("-", ",") if extra not in ("", "pool") else ""
    print(json.dumps({"config": cfg, "step_ms, B_eigh_ms": [step() for _ in range(passes)]}), flush=True)
    if os.environ.get("PTD_EIGH_TIMELINE") == "1":
        for row in eng.LAST_TIMELINE:
            print("   ", row, flush=True)
