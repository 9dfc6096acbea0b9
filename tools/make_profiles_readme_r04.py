"""profiles/README.md of round 4 from the round's files (bench_r04.json, pmc_*_r04.json, roofline_kernel_split_r04.json,
c4_stack_32blocks_bf16_r04.json, streams_r04.json).  Usage: python tools/make_profiles_readme_r04.py"""
import json, os

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
J = lambda n: json.load(open(os.path.join(root, n)))  # noqa: E731
b = J("bench_r04.json")
c2, ro, eg, k, f, cb, ph = b["c2_single_layer"], b["roofline"], b["eigh"], b["kernels"], b["decomposed_fwd"], b["cpu_baseline"], b["phases_ms"]
sy, mf, g64, sp = J("pmc_syrk_r04.json")["shapes"], J("pmc_mfma_r04.json")["kernels"], J("pmc_gemm_f64_r04.json"), J("roofline_kernel_split_r04.json")
c4, blk, s32 = b["c4_shapes"], b["c4_block"], J("c4_stack_32blocks_bf16_r04.json")
lib_eigh = eg.get("gpu_library_eigh_ms")
o = []
o.append("# profiles -- round 4 (one MI355X per call, ROCm 7.2, gpurun boxes)\n\n")
o.append("Every file carries its round in its name; rounds 1-3 stay for history (their README text is in git).  This file is produced by\n"
         "`tools/make_profiles_readme_r04.py` from the round-4 files (`tools/make_profiles_readme.py` knows the bench line of rounds 1-3).  The boxes of\n"
         "the pool differ by a few per cent in the clock they hold under load; numbers from different files may come from different boxes.\n\n")
o.append("Files (round 4):\n\n"
         "* `bench_r04.json` -- `python bench.py --steps 5 --warmup 1` (the driver's contract line + c2_single_layer / bf16_stack / roofline / eigh / phases / kernels / cpu_baseline / decomposed_fwd / c4_shapes / c4_block)\n"
         "* `rocprofv3_kernel_stats_r04.csv`, `roofline_kernel_split_r04.json` -- `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload c2 --steps 3 --warmup 1 --no-extras` (4 decompositions of BASELINE configs[1]); the dominant kernel's launches split by duration\n"
         "* `pmc_gemm_f64_r04.json`, `pmc_symv_r04.json/.csv`, `pmc_mfma_r04.json`, `pmc_syrk_r04.json` -- separate `rocprofv3 --pmc` passes over the torch-free `tools/pmc_driver eigh | 4096 | mfma | syrk`, condensed on the box (`tools/pmc_*_summary.py`); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950\n"
         "* `streams_r04.json` -- `tools/r04_probe.py streams`: PTD_EIGH_STREAMS 1 / 2 / 3 and the opt-in by-route rule on a three-layer chain and on the 2-block Llama stack\n"
         "* `c4_stack_32blocks_bf16_r04.json` -- `PTD_PHASES=1 python tools/c4_stack.py 32 bf16 --trade-off 640 --max-ppl 0.4`: BASELINE configs[3] at full depth on one GPU with thresholds under which layers are replaced\n"
         "* `c4_hf_llama3_8b_r04.json` -- `PTD_PHASES=1 python tools/c4_hf_llama.py 32`: BASELINE configs[3] on the real model family -- a `transformers.LlamaForCausalLM` with the Llama-3-8B architecture (32 decoder layers, 224 Linear layers, vocabulary 128256, lm_head blacklisted), random bf16 weights, token batches [1, 2048], D = 8, M = 2, 4 splits, one GPU: **226 s**, 219 layers replaced; A 1.4 s, B 6.9 s (16.0 s before rank-deficient / outlier-dominated covariances stopped going to the Jacobi solver), D 217 s of whole-model forwards\n"
         "* `c3_vit_falor_phases_r04.json` -- `PTD_PHASES=1 python tools/c3_vit.py`: BASELINE configs[2] (falor on a ViT-B/16-shaped model, 49 layers, 441 bisection steps): 9.8 s -- D metrics 8.5 s, A accumulate 1.0 s (1.45 s before the calibration forwards stopped at the analysed layer), B eigh 0.22 s (0.72 s before clustered covariances stayed on the tridiagonal route) -- against 13.6 s / D 11.3 s without the prefix memo (11.3 / 9.1 with matrix products as its only units)\n"
         "* `bench_r04_rehearsal_2ranks_1gpu.json` -- `PTD_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 2 --warmup 1`: the N = 2 code path (self-launched ranks, dealing, reduce-to-owner, broadcast) with BOTH ranks on the one GPU of the box: a check that the path runs, not a scaling number\n"
         "* `gpu_tests_r04.log` -- `python -m pytest tests -q -m gpu`\n"
         "* `tools/refresh_profiles.sh 04 main|bench|pmc` reruns them on a GPU box\n\n")
o.append("## Headline\n\n")
o.append(f"* `value` = **{b['value']:.2f} layers/s** ({b['ms_per_step']:.0f} ms per step): the FIXED stack of 8 x nn.Linear(4096,4096), f32 model, D = 8, M = 2, one GPU -- the strong-scaling family of `bench.py` (the same work at every N).  "
         + ((lambda q: f"Of a step (`stack_phases_ms`, one extra untimed step: {q['step_wall_ms']:.0f} ms): D metrics {q['D_metrics']:.0f} ms -- the method's own whole-model forwards, 8 layers x 12 (candidate, batch) pairs x two passes over 8 dense f32 GEMMs, the second pass reusing what the first computed ahead of the analysed layer (round 4: `_engine.PrefixMemo`; 1465 ms per step without it) --, B eigh {q['B_eigh']:.0f} ms (eight eigendecompositions on four streams), A accumulate {q['A_accumulate']:.0f} ms, C factors {q['C_factors']:.0f} ms.  ")(b['stack_phases_ms']) if 'stack_phases_ms' in b else
            "Most of a step is the method's own whole-model forwards (8 layers x 12 (candidate, batch) pairs x two passes over 8 dense f32 GEMMs, the second pass reusing what the first computed ahead of the analysed layer: round 4, `_engine.PrefixMemo`; 1465 ms per step without it), then the eight eigendecompositions (~0.2 s on four streams) and the covariance accumulation (~0.1 s).  ")
         +
         f"The same stack with a bf16 model (`bf16_stack`): **{b['bf16_stack']['value']:.1f} layers/s** ({b['bf16_stack']['ms_per_step']:.0f} ms per step).\n")
o.append(f"* `c2_single_layer` = BASELINE configs[1] itself: **{c2['value']:.2f} layers/s** ({c2['ms_per_step']:.1f} ms per dwain decomposition of ONE 4096x4096 Linear, f32 model, T = 4x1024, D = 4, M = 2; steps {c2['step_ms']}) -- round 3: 18.9 / 52.9 ms.  "
         f"CPU oracle on the same workload: **{cb['value']:.3f} layers/s** on {cb['cores']} host threads ({cb['sample']}).\n")
o.append(f"* phases of a C2 step (`phases_ms`): A accumulate {ph['A_accumulate']:.1f} ms, B eigh {ph['B_eigh']:.1f}, C factors {ph['C_factors']:.2f}, D metrics {ph['D_metrics']:.1f}, host and gaps {ph['other_host_and_gaps']:.1f}.\n")
o.append(f"* dominant kernel `gemm_f64_glds_kernel<5, false>` (C X of the Chebyshev filter, 4096 x 4096 x 1280 f64): {ro['achieved']:.1f} TFLOP/s = **{ro['frac']:.2f}** of the 78.6 TFLOP/s f64 MFMA peak, {ro['avg_launch_us']:.0f} us per launch (HIP events inside bench.py; "
         f"rocprofv3 on another box: {sp['long_launches_K4096']} K = 4096 launches average {sp['long_avg_us']:.0f} us, the {sp['short_launches']} shorter X W launches {sp['short_avg_us']:.0f} us), {ro['launches']} launches per eigendecomposition; "
         f"`traffic` {ro['traffic'] / 1e6:.0f} MB per launch = {g64['traffic_over_algorithmic']:.2f} x the algorithmic bytes (X once per XCD), matrix-pipe busy {g64['mfma_busy_over_cu_busy_x4_percent']:.0f} %; `solver_frac` {ro['solver_frac']:.2f}.\n")
o.append(f"* whole eigensolver (filtered route, n = 4096, k = 1024): **{eg['ms_per_matrix']:.1f} ms** (round 3: 29.1): Lanczos {k['lanczos_bounds']['total_ms']:.1f}, filter rounds {k['filter_rounds']['total_ms']:.1f} ({k['filter_rounds']['products_with_C']} products + six Cholesky-QR passes; the sweep's panel launch 39 -> 24 us), "
         f"Rayleigh-Ritz eigenproblem {k['rayleigh_ritz_eigh']['total_ms']:.1f}, Ritz products + residuals {k['ritz_products_and_residuals']['total_ms']:.1f}; {eg['algorithmic_tflops']:.2f} TFLOP/s on the algorithmic 4/3 n^3 + 2 n^2 k"
         + (f"; `torch.linalg.eigh` on the same GPU {lib_eigh:.0f} ms" if lib_eigh else "") + ".\n\n")
o.append("## Kernels (HIP events in bench.py; MfmaUtil from `pmc_mfma_r04.json`)\n\n| line | ms | rate | of peak | MfmaUtil |\n|---|---|---|---|---|\n")


def mu(key):
    for n_, c in mf.items():
        if n_.startswith(key):
            return f"{c.get('MfmaUtil', 0):.0f} %"
    return ""


o.append(f"| syrk_f32_f64acc (n = T = 4096) | {k['syrk_f32_f64acc']['ms']:.3f} | {k['syrk_f32_f64acc']['tflops']:.0f} TFLOP/s | {k['syrk_f32_f64acc']['frac_of_f32_mfma_peak']:.2f} f32 mfma | {mu('syrk_f32')} |\n")
o.append(f"| gemm_f32_nt (4096^3) | {k['gemm_f32_nt']['ms']:.3f} | {k['gemm_f32_nt']['tflops']:.0f} TFLOP/s | {k['gemm_f32_nt']['frac_of_f32_mfma_peak']:.2f} f32 mfma | {mu('gemm_f32_nt_8ph')} |\n")
o.append(f"| syrk_bf16_f64acc (n = T = 4096) | {k['syrk_bf16_f64acc']['ms']:.3f} | {k['syrk_bf16_f64acc']['tflops']:.0f} TFLOP/s | {k['syrk_bf16_f64acc']['frac_of_bf16_mfma_peak']:.2f} bf16 mfma | {mu('syrk_bf16')} |\n")
o.append(f"| gemm_bf16_nt (4096^3) | {k['gemm_bf16_nt']['ms']:.3f} | {k['gemm_bf16_nt']['tflops']:.0f} TFLOP/s | {k['gemm_bf16_nt']['frac_of_bf16_mfma_peak']:.2f} bf16 mfma | {mu('gemm_bf16_nt_8ph')} |\n")
o.append(f"| nsr_f32 ([4096, 4096] C2 logits, 134 MB) | {k['nsr_f32']['ms']:.4f} | {k['nsr_f32']['gbps']:.0f} GB/s | {k['nsr_f32']['frac_of_hbm_peak']:.2f} hbm | |\n")
o.append(f"| nsr_bf16_vocab ([2048, 128256], 1.05 GB) | {k['nsr_bf16_vocab']['ms']:.4f} | {k['nsr_bf16_vocab']['gbps']:.0f} GB/s | {k['nsr_bf16_vocab']['frac_of_hbm_peak']:.2f} hbm | |\n\n")
o.append("`ptd_nsr`: the stream kernel alone takes 25 us on the C2 logits (5.3 TB/s, rocprofv3), the 64-channel final kernel and the boundary make 38; on vocabulary-sized logits the same pair reaches 5.1-6.0 TB/s.\n\n")
o.append("## bf16 covariance product at the calibration shapes (`pmc_syrk_r04.json`; y bf16, f64 accumulator, profiled clocks)\n\n"
         "| n | T | us | MfmaUtil | TFLOP/s (triangle) | of bf16 peak | algorithmic GB/s | of HBM | memory-side bytes / algorithmic | MFMA bound us | HBM bound us |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
for r in sy:
    o.append(f"| {r['n']} | {r['T']} | {r['us']:.1f} | {r['MfmaUtil']:.0f} % | {r['tflops']:.0f} | {r['frac_of_bf16_mfma_peak']:.2f} | {r['algorithmic_gbps']:.0f} | {r['frac_of_hbm_peak']:.2f} | {r['traffic_over_algorithmic']:.2f} | {r['mfma_bound_us']:.1f} | {r['hbm_bound_us']:.1f} |\n")
o.append("\nAt T = 2048 the f64 accumulator's read-modify-write (8 n (n + 1) bytes per call) outweighs the activations 8 : 1: the HBM bound is above the MFMA bound at every Llama shape.  DESIGN 7 (round 4, item 4) has the per-CU model behind the measured times.\n\n")
o.append("## Decomposed forward (bf16, 16384 rows, 4096 -> r -> 4096; BASELINE configs[4])\n\n| r | ms | GFLOP/s | speed-up vs own dense | hipBLASLt pair ms | rotating inputs: ours / library | module ms |\n|---|---|---|---|---|---|---|\n")
for r in ("r256", "r512", "r1024"):
    v = f[r]
    o.append(f"| {r[1:]} | {v['ms']:.3f} | {v['gflops']:.0f} | {v['speedup_vs_dense']:.2f}x | {v['torch_hipblaslt_pair_ms']:.3f} | {v['ms_rotating_inputs']:.3f} / {v['torch_hipblaslt_pair_ms_rotating_inputs']:.3f} | {v['module_ms']:.3f} |\n")
o.append(f"\nDense 4096x4096 bf16: {f['dense_ms']:.3f} ms = {f['dense_tflops']:.0f} TFLOP/s; through `torch.nn.functional.linear` (hipBLASLt): {f['dense_torch_hipblaslt_ms']:.3f} ms = {f['dense_torch_hipblaslt_tflops']:.0f} TFLOP/s.  "
         "The r = 1024 line moved between 0.211 and 0.256 ms across the boxes of the pool this round with the library pair at 0.199-0.217 (`tools/timing_protocol.py`, `tools/fwd_fresh.py`).\n\n")
o.append("## Llama-3-8B layer shapes (C4), one GPU, 2048 tokens per step, D = 8, M = 2 (`bench_r04.json: c4_shapes`, three timed steps each)\n\n| layer | n_in -> n_out | f32 ms per layer | bf16 ms per layer | eigensolver (f32 run) |\n|---|---|---|---|---|\n")
for name in ("q_o", "k_v", "gate_up", "down"):
    a, bb = c4["f32"][name], c4["bf16"][name]
    e = a["eigh"]
    desc = e["route"] + (f", n = {e['n']}, k = {e['k']}, {e['ms']:.1f} ms" if "ms" in e else "")
    o.append(f"| {name} | {a['n_in']} -> {a['n_out']} | {a['ms_per_layer']:.1f} | {bb['ms_per_layer']:.1f} | {desc} |\n")
o.append(f"\nExtrapolated to 224 layers (kernel-side, one layer alone): f32 {c4['f32']['extrapolated_224_layers_s']:.1f} s, bf16 {c4['bf16']['extrapolated_224_layers_s']:.1f} s.\n\n")
o.append(f"One full-width block end to end (`c4_block`; 7 layers, thresholds under which 5 are replaced): f32 **{blk['f32']['ms_per_block']:.0f} ms** (phases {blk['f32']['phases_ms']}), bf16 **{blk['bf16']['ms_per_block']:.0f} ms** (phases {blk['bf16']['phases_ms']}); replaced: {blk['bf16']['replaced']}.\n\n")
o.append(f"Full depth (`c4_stack_32blocks_bf16_r04.json`): 32 blocks, 224 layers, bf16: **{s32['seconds']:.0f} s = {s32['layers_per_s']:.2f} layers/s**, {s32['layers_replaced']} layers replaced, {s32['candidates_evaluated']} candidates; phases {s32['phases_ms']} -- "
         "the run is the user model's own forwards (two whole-model forwards per candidate and batch, in torch / hipBLASLt); the covariance and eigensolver kernels are 8.5 s of it.\n\n")

hf = json.load(open(os.path.join(root, "c4_hf_llama3_8b_r04.json"))) if os.path.exists(os.path.join(root, "c4_hf_llama3_8b_r04.json")) else None
if hf:
    o.append(f"The same on a `transformers.LlamaForCausalLM` with the Llama-3-8B architecture (`c4_hf_llama3_8b_r04.json`): {hf['decoder_layers']} decoder layers, {hf['layers']} Linear layers, vocabulary 128256, bf16: **{hf['seconds']:.0f} s = {hf['layers_per_s']:.2f} layers/s**, {hf['layers_replaced']} layers replaced; phases {hf['phases_ms']}.\n\n")
o.append("## Streams (`streams_r04.json`)\n\nB_eigh in ms, two runs each, interleaved: three 4096^2 layers in one split -- 1 stream 171 / 171, 2 streams 207 / 163, 3 streams 181 / 137, by-route rule 171 / 171; "
         "2-block Llama stack bf16 -- 667 / 668, 596 / 684, 622 / 583, 681 / 668.  Default since the second sweep (DESIGN 3): four streams, every chain on its own.\n")
open(os.path.join(root, "README.md"), "w").write("".join(o))
print("profiles/README.md written,", sum(len(x) for x in o), "bytes")
