"""Regenerate profiles/README.md from profiles/bench_rNN.json, rocprofv3_kernel_stats_rNN.csv and
c4_shapes_f32_rNN.json.  Usage: python tools/make_profiles_readme.py 01"""
import csv, json, os, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "01"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
b = json.load(open(os.path.join(root, f"bench_r{rnd}.json")))
rows = list(csv.DictReader(open(os.path.join(root, f"rocprofv3_kernel_stats_r{rnd}.csv"))))


def short(name):
    name = name.replace("ptd::(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:72]


o = [f"# profiles -- round {int(rnd)} (one MI355X, ROCm 7.2, gpurun box)\n\n", "Files:\n\n",
     f"* `bench_r{rnd}.json` -- `python bench.py --steps 5 --warmup 1` (the driver's contract line plus roofline / eigh / kernels / cpu_baseline / decomposed_fwd)\n",
     f"* `rocprofv3_kernel_stats_r{rnd}.csv` -- `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-extras` (4 decompositions)\n",
     f"* `c4_shapes_f32_r{rnd}.json` -- `python tools/c4_shapes.py`: dwain on one layer of each Llama-3-8B shape (BASELINE configs[3]), 224-layer figure extrapolated\n",
     f"* `c4_shapes_bf16_r{rnd}.json`, `c4_stack_2blocks_r{rnd}.json` -- the same in bf16; `python tools/c4_stack.py 2 [bf16]` end to end on a 2-block full-width stack\n",
     f"* `c3_vit_falor_r{rnd}.json` -- `python tools/c3_vit.py`: falor on a ViT-B/16-shaped model (BASELINE configs[2])\n",
     f"* `pmc_symv_r{rnd}.json` / `.csv` -- `tools/pmc_summary.py` over separate `rocprofv3 --pmc` passes of `tools/pmc_driver 4096` (per-launch HBM traffic of the SYMV kernels)\n",
     f"* `gpu_tests_r{rnd}.log` -- `python -m pytest tests -q -m gpu` on the same box\n",
     "* `tools/refresh_profiles.sh` reruns the first four on a GPU box\n\n", "## Headline\n\n"]
cb = b.get("cpu_baseline")
o.append(f"* **{b['value']:.2f} layers/s** ({b['ms_per_step']:.0f} ms per dwain decomposition of a 4096x4096 Linear: f32 model, "
         f"T = 4x1024 tokens per batch, D = 4, M = 2, f64 covariance + eigendecomposition)")
if cb:
    o.append(f" vs the CPU oracle **{cb['value']:.4f} layers/s** on {cb['cores']} host threads ({cb['sample']}): "
             f"{b['value'] / cb['value']:.0f}x.\n")
else:
    o.append(".\n")
r, e = b["roofline"], b.get("eigh", {})
o.append(f"* dominant kernel `{r['kernel'].split(' ')[0]}`: bound {r['bound']}, {r['achieved']:.0f} {r['unit']} = "
         f"**{100 * r['frac']:.0f} %** of the {r['peak']:.0f} {r['unit']} peak over {r.get('launches', '?')} launches "
         f"(avg {r.get('avg_launch_us', 0):.1f} us, dispatch-attached HIP events; rocprofv3's average for the same kernel is in the table below).\n")
sv = [x for x in rows if "sytrd_symv" in x["Name"]]
if sv:
    calls = sum(int(x["Calls"]) for x in sv)
    tot = sum(float(x["TotalDurationNs"]) for x in sv)
    avg_us = tot / calls / 1e3
    gbps = r.get("algorithmic_bytes_per_launch", 0) / (avg_us * 1e-6) / 1e9
    o.append(f"* the same launches in the rocprofv3 trace below ({' + '.join(short(x['Name']) for x in sv)}): weighted average "
             f"{avg_us:.2f} us -> {gbps:.0f} GB/s = **{100 * gbps / r['peak']:.0f} %** of peak on the same algorithmic bytes "
             "(dispatch-attached HIP events read about 1 us more per launch than the profiler's kernel begin / end timestamps, also when only every 8th launch is timed; the bench line quotes the lower fraction).\n")
if e:
    o.append(f"* whole eigensolver ({e['method']}, n = {e['n']}): {e['ms_per_matrix']:.0f} ms per matrix (HIP events around the call; every 8th SYMV launch also carries dispatch-attached events) = "
             f"{e['algorithmic_tflops']:.2f} TFLOP/s on the algorithmic 4/3 n^3 + 2 n^2 k flops "
             f"({100 * e['frac_of_f64_mfma_peak_on_algorithmic_flops']:.1f} % of the f64 MFMA peak: a one-stage reduction is bandwidth-bound).\n")
pmc = os.path.join(root, f"pmc_symv_r{rnd}.json")
if os.path.exists(pmc):
    pm = json.load(open(pmc))
    o.append(f"* `roofline.traffic` (`pmc_symv_r{rnd}.json` / `.csv`, one row per launch): HBM-side bytes of the SYMV kernels from separate "
             "`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over `tools/pmc_driver 4096` (torch-free, counter collection restricted "
             "with `--kernel-include-regex sytrd_symv`; an unrestricted pass over a python process segfaulted or stalled), FETCH_SIZE doubled as the "
             f"microarchitecture guide prescribes for gfx950: **{pm['traffic_bytes_per_launch'] / 1e6:.1f} MB per launch = "
             f"{pm['traffic_over_algorithmic']:.3f} x the algorithmic {pm['algorithmic_bytes_per_launch'] / 1e6:.1f} MB** "
             f"(first launch {pm['first_launch']['read'] / 1e6:.1f} MB read for {pm['first_launch']['algorithmic'] / 1e6:.1f} MB); "
             f"the symmetric kernel reads only the lower triangle of the trailing matrix, hence less than the algorithmic stream of a one-stage SYMV. "
             f"L2 hit rate {pm['l2_hit_rate']:.2f}: the matrix is re-fetched from the memory side every launch, "
             "also when it would fit in L2 (L2 does not keep lines across kernel boundaries).\n\n")
else:
    o.append("* `roofline.traffic` is null: no PMC pass committed for this round.\n\n")
mf = os.path.join(root, f"pmc_mfma_r{rnd}.json")
if os.path.exists(mf):
    mk = json.load(open(mf))["kernels"]
    o.append("* MFMA utilisation from the counters (`pmc_mfma_r%s.json`, `tools/pmc_driver mfma`): " % rnd
             + "; ".join(f"`{k}` MfmaUtil {v.get('MfmaUtil', float('nan')):.0f} % (busy cycles {100 * v.get('busy_fraction', float('nan')):.0f} %)" for k, v in mk.items())
             + " -- in line with the event-timed rates below.\n\n")
o.append("## Top kernels (rocprofv3 --stats)\n\n| kernel | calls | avg us | % of GPU time |\n|---|---|---|---|\n")
for x in rows[:14]:
    o.append(f"| `{short(x['Name'])}` | {x['Calls']} | {float(x['AverageNs']) / 1e3:.1f} | {float(x['Percentage']):.2f} |\n")
o.append("\n`Cijk_...` (if present) is hipBLASLt running the user model's own nn.Linear forward.\n\n")
o.append("## Per-kernel rates (HIP events inside bench.py, C2 shapes)\n\n| kernel | time | rate | fraction of peak |\n|---|---|---|---|\n")
for k, v in b["kernels"].items():
    if "ms" in v and ("tflops" in v or "gbps" in v):
        fr = [f"{100 * x:.0f} % of {kk.replace('frac_of_', '').replace('_', ' ')}" for kk, x in v.items() if kk.startswith("frac_of")]
        rate = f"{v['tflops']:.0f} TFLOP/s" if "tflops" in v else f"{v['gbps']:.0f} GB/s"
        o.append(f"| {k} | {v['ms']:.3f} ms | {rate} | {fr[0] if fr else ''} |\n")
    elif "total_ms" in v:
        extra = f"{v['gbps']:.0f} GB/s" if "gbps" in v else v.get("note", "")
        avg = f" (avg {v['avg_us']:.1f} us x {v['launches']})" if "avg_us" in v else ""
        o.append(f"| {k} | {v['total_ms']:.1f} ms per matrix{avg} | {extra} | |\n")
d = b.get("decomposed_fwd")
if d:
    o.append(f"\n## Decomposed forward (bf16, {d['rows']} rows, 4096 -> r -> 4096; BASELINE configs[4])\n\n"
             "| r | ms | GFLOP/s (2 T r (n_i + n_o)) | speed-up vs dense 4096^2 (own kernel) | vs dense through torch / hipBLASLt | the pair as two torch linears (hipBLASLt), ms |\n|---|---|---|---|---|---|\n")
    for rr in (256, 512, 1024):
        v = d[f"r{rr}"]
        lib = f"{v['speedup_vs_dense_torch_hipblaslt']:.2f}x" if "speedup_vs_dense_torch_hipblaslt" in v else ""
        lp = f"{v['torch_hipblaslt_pair_ms']:.3f}" if "torch_hipblaslt_pair_ms" in v else ""
        o.append(f"| {rr} | {v['ms']:.3f} | {v['gflops']:.0f} | {v['speedup_vs_dense']:.2f}x | {lib} | {lp} |\n")
    o.append(f"\nDense 4096x4096 bf16 on `gemm_bf16_nt_8ph_kernel`: {d['dense_ms']:.3f} ms = {d['dense_tflops']:.0f} TFLOP/s"
             + (f"; the same layer through `torch.nn.functional.linear` (hipBLASLt): {d['dense_torch_hipblaslt_ms']:.3f} ms = "
                f"{d['dense_torch_hipblaslt_tflops']:.0f} TFLOP/s" if "dense_torch_hipblaslt_ms" in d else "") + ".\n")
c4p = os.path.join(root, f"c4_shapes_f32_r{rnd}.json")
if os.path.exists(c4p):
    c4 = json.load(open(c4p))
    o.append("\n## Llama-3-8B layer shapes (C4), one GPU, f32 model, 2048 tokens per step, D = 8, M = 2\n\n"
             "| layer | n_in -> n_out | ms per layer | eigensolver |\n|---|---|---|---|\n")
    for k in ("q_o", "k_v", "gate_up", "down"):
        v = c4[k]
        eg = v["eigh"]
        desc = eg.get("route") or f"{eg['method']} n={eg['n']} k={eg['k']}" + (f", SYMV {eg['symv_gbps']:.0f} GB/s" if "symv_gbps" in eg else "")
        o.append(f"| {k} | {v['n_in']} -> {v['n_out']} | {v['ms_per_layer']:.0f} | {desc} |\n")
    o.append(f"\nExtrapolated to the 224 layers of the 32-block stack: {c4['extrapolated_224_layers_s']:.0f} s on one GPU "
             f"({c4['extrapolated_layers_per_s_1gpu']:.1f} layers/s).\n")
c4b = os.path.join(root, f"c4_shapes_bf16_r{rnd}.json")
if os.path.exists(c4b):
    cb16 = json.load(open(c4b))
    o.append(f"\nSame in bf16 (`c4_shapes_bf16_r{rnd}.json`): " + ", ".join(f"{k} {cb16[k]['ms_per_layer']:.0f} ms" for k in ("q_o", "k_v", "gate_up", "down"))
             + f"; 224 layers extrapolated {cb16['extrapolated_224_layers_s']:.0f} s.\n")
stk = os.path.join(root, f"c4_stack_2blocks_r{rnd}.json")
if os.path.exists(stk):
    st = json.load(open(stk))
    o.append(f"\nEnd to end on a 2-block full-width stack (`c4_stack_2blocks_r{rnd}.json`, `python tools/c4_stack.py 2`): "
             f"f32 {st['f32']['seconds']:.1f} s, bf16 {st['bf16']['seconds']:.1f} s for 14 layers ({st['f32']['candidates_evaluated']} candidates); "
             "here the user model's own forwards (two per metric step through the whole stack, torch / hipBLASLt) dominate, "
             "as SURVEY 3.5 predicts for real LLM configs.\n")
c3 = os.path.join(root, f"c3_vit_falor_r{rnd}.json")
if os.path.exists(c3):
    v = json.loads(open(c3).read().strip().splitlines()[-1])
    o.append(f"\n## ViT-B/16-shaped falor run (C3)\n\n`python tools/c3_vit.py`: {v['layers']} Linear layers, {v['candidates_evaluated']} bisection steps, "
             f"**{v['seconds']:.1f} s = {v['layers_per_s']:.2f} layers/s** on one GPU ({v['decomposed']} layers replaced). The run is dominated by the "
             "user model's own forwards (two per bisection step and metric batch); the widening layers (qkv, fc1, head: rank-deficient feature "
             "covariance) only ask for the eigenvectors the bisection can use, which keeps them on the tridiagonal route (23.6 s before).\n")
open(os.path.join(root, "README.md"), "w").write("".join(o))
print("".join(o))
