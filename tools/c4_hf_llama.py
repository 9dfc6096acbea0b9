"""BASELINE configs[3] on the real model family: dwain.decompose_in_place on a transformers.LlamaForCausalLM with the
Llama-3-8B architecture (hidden 4096, 32 query / 8 KV heads, MLP 14336, vocabulary 128256, RoPE theta 5e5) at `layers`
decoder layers, random weights generated on the device, bf16, synthetic token batches [1, 2048], D = 8, M = 2,
precomputing_covariance_num_splits = 4, lm_head blacklisted, one MI355X.
Usage: python tools/c4_hf_llama.py [layers] [--trade-off X] [--max-ppl Y] [--attn sdpa|eager]"""
import itertools, json, os, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import transformers
import ptdeco_amd

# metric batches the iterator cycles over: 16 > the 14 draws of a layer's search (none recurs within a layer, as with a
# streamed DataLoader); METRIC_POOL=4: batches recur and the engine's reuse across candidates engages
METRIC_POOL = int(os.environ.get("METRIC_POOL", "16"))
from ptdeco_amd import _engine as eng

dev = torch.device("cuda", 0)
layers = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 2


def opt(name, default, cast=float):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


trade_off = opt("--trade-off", 20.0 * layers)
max_ppl = opt("--max-ppl", 0.4)
attn = opt("--attn", "sdpa", str)
cfg_l = transformers.LlamaConfig(vocab_size=128256, hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers,
                                 num_attention_heads=32, num_key_value_heads=8, max_position_embeddings=8192,
                                 rope_theta=500000.0, rms_norm_eps=1e-5, attn_implementation=attn,
                                 tie_word_embeddings=False)
with torch.device("meta"):
    llama = transformers.LlamaForCausalLM(cfg_l)
llama = llama.to(torch.bfloat16).to_empty(device=dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    for name, p in llama.named_parameters():
        if p.ndim == 2:
            p.copy_((torch.randn(p.shape, generator=g, device=dev) / p.shape[1] ** 0.5).to(p.dtype))
        else:
            p.fill_(1.0)      # RMSNorm weights
# (to_empty left the rotary tables uninitialised: build that module again, on the device)
with torch.device(dev):
    llama.model.rotary_emb = type(llama.model.rotary_emb)(config=cfg_l)


class Logits(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.m = llama

    def forward(self, b):
        return self.m(input_ids=b["ids"], use_cache=False).logits


model = Logits().eval()


def ce(b, y):
    return torch.nn.functional.cross_entropy(y.float().reshape(-1, y.shape[-1]), b["targets"].reshape(-1), reduction="none")


ids = [torch.randint(0, 128256, (1, 2048), generator=g, device=dev) for _ in range(8 + METRIC_POOL)]
with torch.no_grad():
    bt = [{"ids": i, "targets": model({"ids": i}).argmax(-1)} for i in ids]
torch.cuda.synchronize()
# sample check (tests/factor_checks.py): a few layers of the first precompute split are armed before the run
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import factor_checks as sample_check
all_names = [n for n, m in model.named_modules() if isinstance(m, torch.nn.Linear) and not n.endswith("lm_head")]
armed = sample_check.arm(model, all_names[:max(1, len(all_names) // 4)], bt[:8])
if os.environ.get("PTD_PHASES"):
    eng.PHASES = eng.PhaseTimer()
trace = []
done = threading.Event()
t0 = time.perf_counter()


def heartbeat():
    while not done.wait(45.0):
        print(f"[c4_hf_llama] {time.perf_counter() - t0:.0f} s, {len(trace)} candidates evaluated", file=sys.stderr, flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
hits0 = eng.PrefixMemo.total_hits
cfg = ptdeco_amd.dwain.decompose_in_place(
    module=model, device=dev, data_iterator=itertools.cycle(bt[:12]), loss_fn=ce, metric_iterator=itertools.cycle(bt[8:]),
    num_data_steps=8, num_metric_steps=2, nsr_final_threshold=1.0, finetune_fn=lambda m, d, n: m,
    trade_off_factor=trade_off, max_accepted_ppl_diff=max_ppl, blacklisted_module_names=["m.lm_head"],
    precomputing_covariance_num_splits=4, trace=trace)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
done.set()
phases = None
if eng.PHASES is not None:
    phases = {k: round(v, 1) for k, v in eng.PHASES.totals_ms().items()}
    phases["other_host_and_gaps"] = round(dt * 1e3 - sum(phases.values()), 1)
n_layers = 7 * layers
checked = sample_check.verify(armed, model, cfg)
print(json.dumps({"sample_check": checked, "metric_pool": METRIC_POOL,
                  "workload": f"dwain.decompose_in_place, transformers.LlamaForCausalLM (transformers {transformers.__version__}), "
                              f"Llama-3-8B architecture at {layers} decoder layers ({n_layers} Linear layers; lm_head 4096 -> 128256 "
                              f"blacklisted), attention = {attn}, random bf16 weights, token batches [1, 2048], D = 8, M = 2, "
                              "precomputing_covariance_num_splits = 4, f64 covariance + eigh, one MI355X",
                  "phases_ms": phases, "decoder_layers": layers, "layers": n_layers, "seconds": dt, "layers_per_s": n_layers / dt,
                  "trade_off_factor": trade_off, "max_accepted_ppl_diff": max_ppl, "candidates_evaluated": len(trace),
                  "layers_replaced": len(cfg), "prefix_memo_hits": eng.PrefixMemo.total_hits - hits0,
                  "decomposed": {k: v["__meta__"]["proportion"] for k, v in cfg.items()},
                  "max_mem_gb": torch.cuda.max_memory_allocated() / 2**30}))
