"""Timeline of the seven eigensolver chains of one full-width Llama block (bf16): PTD_EIGH_JOB_LOG=1 lines of
run_concurrently for the model order and for longest-first, with 4 streams."""
import copy, itertools, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, ptdeco_amd
from ptdeco_amd import _engine as eng
dev = torch.device("cuda", 0)
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
    ptdeco_amd.dwain.decompose_in_place(module=m, device=dev, data_iterator=itertools.cycle(bt), loss_fn=bench.seq_ce,
                                        metric_iterator=itertools.cycle(bt[8:]), finetune_fn=lambda mm, d, n: mm,
                                        **bench.C4_BLOCK_KW)
    torch.cuda.synchronize()
    ph, eng.PHASES = eng.PHASES.totals_ms(), None
    return round(ph["B_eigh"], 1)

step()
for env in sys.argv[1:] or ["PTD_EIGH_LONGEST_FIRST=0", "PTD_EIGH_LONGEST_FIRST=1"]:
    for kv in env.split(","):
        k, v = kv.split("=")
        os.environ[k] = v
    os.environ.pop("PTD_EIGH_JOB_LOG", None)
    quiet = [step() for _ in range(3)]
    os.environ["PTD_EIGH_JOB_LOG"] = "1"
    print(env, "B_eigh without the log", quiet, "with", step(), file=sys.stderr, flush=True)
