"""profiles/README.md of round 5 from the round's files (bench_r05.json, pmc_*_r05.json, roofline_kernel_split_r05.json, the
full-depth runs).  Usage: python tools/make_profiles_readme_r05.py"""
import csv, json, os

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
J = lambda n: json.load(open(os.path.join(root, n)))  # noqa: E731
E = lambda n: os.path.exists(os.path.join(root, n))  # noqa: E731
b = J("bench_r05.json")
c2, ro, eg, k, f, cb, ph = b["c2_single_layer"], b["roofline"], b["eigh"], b["kernels"], b["decomposed_fwd"], b["cpu_baseline"], b["phases_ms"]
sy, mf, g64, sp = J("pmc_syrk_r05.json")["shapes"], J("pmc_mfma_r05.json")["kernels"], J("pmc_gemm_f64_r05.json"), J("roofline_kernel_split_r05.json")
c4, blk, stk = b["c4_shapes"], b["c4_block"], b["c4_stack"]
o = []
o.append("# profiles -- round 5 (one MI355X per call, ROCm 7.2, gpurun boxes)\n\n")
o.append("Every file carries its round in its name; rounds 1-4 stay for history (their README text is in git).  This file is produced by\n"
         "`tools/make_profiles_readme_r05.py` from the round-5 files.  The boxes of the pool differ by a few per cent in the clock they hold under\n"
         "load; numbers from different files may come from different boxes.\n\n")
o.append("Files (round 5):\n\n"
         "* `bench_r05.json` -- `python bench.py` (the driver's default command: the contract line + c2_single_layer / bf16_stack / roofline / eigh / phases / kernels / cpu_baseline / decomposed_fwd / c4_shapes / c4_block / c4_stack; every block names its workload)\n"
         "* `rocprofv3_kernel_stats_default_r05.csv` -- `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline`: the kernel summary of the DEFAULT command (kernels of concurrent chains overlap, so their durations are longer than alone)\n"
         "* `rocprofv3_kernel_stats_r05.csv`, `roofline_kernel_split_r05.json` -- the same for `bench.py --workload c2 --steps 3 --warmup 1 --no-extras` (4 decompositions of BASELINE configs[1]); the dominant kernel's launches split by duration\n"
         "* `rocprofv3_direct_eigh_4096_2048_r05.txt` -- `tools/prof_kernels.sh r05_direct tools/eigh_trace.py 4096 2048`: the per-kernel split of the direct route at k = n / 2 (three calls)\n"
         "* `pmc_gemm_f64_r05.json`, `pmc_symv_r05.json/.csv`, `pmc_mfma_r05.json`, `pmc_syrk_r05.json` -- separate `rocprofv3 --pmc` passes over the torch-free `tools/pmc_driver eigh | 4096 | mfma | syrk`, condensed on the box (`tools/pmc_*_summary.py`); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; `pmc_syrk` now has the multi-step entry (8 steps per launch) beside one call per step\n"
         "* `launch_rate_probe_r05.txt` -- `tools/probes/launch_rate_probe.hip`: which of eight streams share a hardware queue (pairwise 300-us kernels), aggregated launch rates of 1 .. 8 host threads, the same chains as replayed hipGraphs\n"
         "* `streams_r05.json` -- `tools/r05_probe.py block`: B_eigh of the full-width Llama block over five passes at PTD_EIGH_STREAMS 4 / 6 / 7 (high- and normal-priority candidates: the measurement that sent the default to normal priority, four streams)\n"
         "* `c4_stack_32blocks_bf16_r05.json`, `c4_hf_llama3_8b_r05.json` -- the full-depth runs (`tools/c4_stack.py 32 bf16 --trade-off 640 --max-ppl 0.4`, `tools/c4_hf_llama.py 32`), now with a sample check of one replaced layer (`tools/sample_check.py`)\n"
         "* `bench_r05_rehearsal_2ranks_1gpu.json` -- `PTD_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 2 --warmup 1`: the N = 2 code path with BOTH ranks on the one GPU of the box: a check that the path runs, not a scaling number\n"
         "* `fwd_split_r05.txt` -- `tools/fwd_split.py`: the two products of the decomposed forward one by one, package and library, at T = 4096 / 16384 / 65536\n"
         "* `gpu_tests_r05.log` -- `python -m pytest tests -q -m gpu`\n"
         "* `tools/refresh_profiles.sh 05 main|bench|pmc` reruns them on a GPU box\n\n")
o.append("## Headline\n\n")
q = b["stack_phases_ms"]
o.append(f"* `value` = **{b['value']:.2f} layers/s** ({b['ms_per_step']:.0f} ms per step; steps {b['step_ms']}): the FIXED stack of 8 x nn.Linear(4096,4096), f32 model, D = 8, M = 2, one GPU -- the strong-scaling family of `bench.py`.  "
         f"Of a step (`stack_phases_ms`): D metrics {q['D_metrics']:.0f} ms (the method's own whole-model forwards), B eigh {q['B_eigh']:.0f} ms (eight eigendecompositions on four measured streams), A accumulate {q['A_accumulate']:.0f} ms, C factors {q['C_factors']:.0f} ms.  "
         f"The same stack with a bf16 model (`bf16_stack`): **{b['bf16_stack']['value']:.1f} layers/s** ({b['bf16_stack']['ms_per_step']:.0f} ms per step; steps {b['bf16_stack']['step_ms']}).\n")
o.append(f"* `c2_single_layer` = BASELINE configs[1] itself: **{c2['value']:.2f} layers/s** ({c2['ms_per_step']:.1f} ms per dwain decomposition of ONE 4096x4096 Linear, f32 model, T = 4x1024, D = 4, M = 2).  "
         f"CPU oracle on the same workload (`cpu_baseline`, which names it): **{cb['value']:.3f} layers/s** on {cb['cores']} host threads.\n")
o.append(f"* phases of a C2 step (`phases_ms`): A accumulate {ph['A_accumulate']:.1f} ms, B eigh {ph['B_eigh']:.1f}, C factors {ph['C_factors']:.2f}, D metrics {ph['D_metrics']:.1f}, host and gaps {ph['other_host_and_gaps']:.1f}.\n")
o.append(f"* dominant kernel of the C2 step `gemm_f64_glds_kernel<5, false>` (C X of the Chebyshev filter, 4096 x 4096 x 1280 f64): {ro['achieved']:.1f} TFLOP/s = **{ro['frac']:.2f}** of the 78.6 TFLOP/s f64 MFMA peak, {ro['avg_launch_us']:.0f} us per launch (HIP events inside bench.py; "
         f"rocprofv3 on another box: {sp['long_launches_K4096']} K = 4096 launches average {sp['long_avg_us']:.0f} us, the {sp['short_launches']} shorter X W launches {sp['short_avg_us']:.0f} us); "
         f"`traffic` {ro['traffic'] / 1e6:.0f} MB per launch = {g64['traffic_over_algorithmic']:.2f} x the algorithmic bytes (X once per XCD), matrix-pipe busy {g64['mfma_busy_over_cu_busy_x4_percent']:.0f} %; `solver_frac` {ro['solver_frac']:.2f}.\n")
o.append(f"* whole eigensolver (filtered route, n = 4096, k = 1024): **{eg['ms_per_matrix']:.1f} ms**; direct route (n = 4096, k = 2048, `c4_shapes.*.down.eigh`): **{c4['f32']['down']['eigh']['ms']:.1f} ms** (round 4: 59.0; the twisted-factorisation work list on one wave per vector and the resident kernels' pass without predicates, DESIGN section 7 item 1); "
         "three such chains at once on the blocked path: 109.5 ms = 36.5 ms per matrix (`tools/probes/filtered_half.py`).\n\n")
o.append("## Concurrent chains (`c4_block`, `c4_stack`; five / three timed steps, median reported)\n\n| workload | ms per step (median) | steps | spread | B_eigh | A | D |\n|---|---|---|---|---|---|---|\n")
for name, d in (("c4_block f32", blk["f32"]), ("c4_block bf16", blk["bf16"]), ("c4_stack bf16 (2 blocks)", stk["bf16"])):
    p_ = d["phases_ms"]
    o.append(f"| {name} | {d['ms_per_step']:.1f} | {d['step_ms']} | {100 * d['spread']:.1f} % | {p_['B_eigh']:.0f} | {p_['A_accumulate']:.1f} | {p_['D_metrics']:.0f} |\n")
cs = b["config"].get("chain_streams_at_exit", {})
o.append(f"\nStream checks in that process (`config.chain_streams_at_exit`): {cs}.  Round 4's line had 394.5 / 345.5 ms for two steps of `c4_block` bf16 in one process.\n\n")
o.append("## Kernels (HIP events in bench.py, median of three loops of at least 40 ms each; MfmaUtil from `pmc_mfma_r05.json`)\n\n| line | ms | rate | of peak | MfmaUtil |\n|---|---|---|---|---|\n")


def mu(key):
    for n_, c in mf.items():
        if n_.startswith(key):
            return f"{c.get('MfmaUtil', 0):.0f} %"
    return ""


o.append(f"| syrk_f32_f64acc (n = T = 4096) | {k['syrk_f32_f64acc']['ms']:.3f} | {k['syrk_f32_f64acc']['tflops']:.0f} TFLOP/s | {k['syrk_f32_f64acc']['frac_of_f32_mfma_peak']:.2f} f32 mfma | {mu('syrk_f32')} |\n")
o.append(f"| gemm_f32_nt (4096^3) | {k['gemm_f32_nt']['ms']:.3f} | {k['gemm_f32_nt']['tflops']:.0f} TFLOP/s | {k['gemm_f32_nt']['frac_of_f32_mfma_peak']:.2f} f32 mfma | {mu('gemm_f32_nt_8ph')} |\n")
o.append(f"| syrk_bf16_f64acc (n = T = 4096, one call) | {k['syrk_bf16_f64acc']['ms']:.3f} | {k['syrk_bf16_f64acc']['tflops']:.0f} TFLOP/s | {k['syrk_bf16_f64acc']['frac_of_bf16_mfma_peak']:.2f} bf16 mfma | {mu('syrk_bf16')} |\n")
o.append(f"| gemm_bf16_nt (4096^3) | {k['gemm_bf16_nt']['ms']:.3f} | {k['gemm_bf16_nt']['tflops']:.0f} TFLOP/s | {k['gemm_bf16_nt']['frac_of_bf16_mfma_peak']:.2f} bf16 mfma | {mu('gemm_bf16_nt_8ph')} |\n")
o.append(f"| nsr_f32 ([4096, 4096] C2 logits, 134 MB) | {k['nsr_f32']['ms']:.4f} | {k['nsr_f32']['gbps']:.0f} GB/s | {k['nsr_f32']['frac_of_hbm_peak']:.2f} hbm | |\n")
o.append(f"| nsr_bf16_vocab ([2048, 128256], 1.05 GB) | {k['nsr_bf16_vocab']['ms']:.4f} | {k['nsr_bf16_vocab']['gbps']:.0f} GB/s | {k['nsr_bf16_vocab']['frac_of_hbm_peak']:.2f} hbm | |\n\n")
o.append("## bf16 covariance product at the calibration shapes (2048 tokens a step)\n\n`bench_r05.json: kernels.syrk_bf16_calibration_shapes` (HIP events, un-profiled clocks):\n\n"
         "| n | us per step, one call per step | us per step, 8 steps per call | HBM bound us (one call / 8) | MFMA bound us | of the bound (one call / 8) |\n|---|---|---|---|---|---|\n")
for key, v in k["syrk_bf16_calibration_shapes"].items():
    o.append(f"| {key[1:]} | {v['us_per_step_single_call']:.1f} | {v['us_per_step_multi_8']:.1f} | {v['hbm_bound_us_single']:.1f} / {v['hbm_bound_us_multi_8']:.1f} | {v['mfma_bound_us']:.1f} | {v['frac_of_bound_single']:.2f} / {v['frac_of_bound_multi_8']:.2f} |\n")
o.append("\nRound 4 (`pmc_syrk_r04.json`, one call per step): n = 4096 74.4 us, n = 1024 35.2 us, n = 14336 772 us.\n\n"
         "`pmc_syrk_r05.json` (counter passes, profiled clocks; `steps` = calibration steps per launch):\n\n"
         "| n | T | steps | us per launch | us per step | MfmaUtil | of bf16 peak | memory-side bytes / algorithmic | HBM bound us |\n|---|---|---|---|---|---|---|---|---|\n")
for r in sy:
    o.append(f"| {r['n']} | {r['T']} | {r['steps_per_launch']} | {r['us']:.1f} | {r['us_per_step']:.1f} | {r['MfmaUtil']:.0f} % | {r['frac_of_bf16_mfma_peak']:.2f} | {r['traffic_over_algorithmic']:.2f} | {r['hbm_bound_us']:.1f} |\n")
o.append("\nThe memory-side bytes are L2 misses (Infinity-Cache hits included): a step's activations (16.8 MB at n = 4096) do not fit the 4-MB L2 of an XCD, whose 32 workgroups walk an 8 x 8 patch of tiles (16 panels of 0.5 MB per K range), so Y is fetched about once per XCD; the accumulator moves once per launch.  DESIGN 7 (round 5, item 3) has the split of a step into fill / fragment reads / MFMAs.\n\n")
o.append("## Decomposed forward (bf16, 4096 -> r -> 4096; BASELINE configs[4]; ours / hipBLASLt pair, ms)\n\n| r | T = 4096 | T = 16384 | T = 65536 | T = 16384 rotating inputs | of HBM at T = 16384 |\n|---|---|---|---|---|---|\n")
for r in ("r256", "r512", "r1024"):
    a4, a16, a64 = f["rows_4096"][r], f[r], f["rows_65536"][r]
    o.append(f"| {r[1:]} | {a4['ms']:.3f} / {a4['torch_hipblaslt_pair_ms']:.3f} | {a16['ms']:.3f} / {a16['torch_hipblaslt_pair_ms']:.3f} | {a64['ms']:.3f} / {a64['torch_hipblaslt_pair_ms']:.3f} | "
             f"{a16['ms_rotating_inputs']:.3f} / {a16['torch_hipblaslt_pair_ms_rotating_inputs']:.3f} | {a16.get('frac_of_hbm_peak', 0):.2f} |\n")
o.append(f"\nDense 4096x4096 bf16 (ours / `torch.nn.functional.linear`): T = 4096 {f['rows_4096']['dense_ms']:.3f} / {f['rows_4096']['dense_torch_hipblaslt_ms']:.3f} ms, T = 16384 {f['dense_ms']:.3f} / {f['dense_torch_hipblaslt_ms']:.3f} ms "
         f"({f['dense_tflops']:.0f} / {f['dense_torch_hipblaslt_tflops']:.0f} TFLOP/s), T = 65536 {f['rows_65536']['dense_ms']:.3f} / {f['rows_65536']['dense_torch_hipblaslt_ms']:.3f} ms.  The kernels are round 3's.\n\n")
o.append("## Llama-3-8B layer shapes (C4), one GPU, 2048 tokens per step, D = 8, M = 2 (`c4_shapes`, three timed steps each)\n\n| layer | n_in -> n_out | f32 ms per layer | bf16 ms per layer | eigensolver (f32 run) |\n|---|---|---|---|---|\n")
for name in ("q_o", "k_v", "gate_up", "down"):
    a, bb = c4["f32"][name], c4["bf16"][name]
    e = a["eigh"]
    desc = e["route"] + (f", n = {e['n']}, k = {e['k']}, {e['ms']:.1f} ms" if "ms" in e else "")
    o.append(f"| {name} | {a['n_in']} -> {a['n_out']} | {a['ms_per_layer']:.1f} | {bb['ms_per_layer']:.1f} | {desc} |\n")
o.append("\n")
for fn, label in (("c4_stack_32blocks_bf16_r05.json", "Full depth (`tools/c4_stack.py 32 bf16`)"), ("c4_hf_llama3_8b_r05.json", "`transformers.LlamaForCausalLM`, Llama-3-8B architecture (`tools/c4_hf_llama.py 32`)")):
    if E(fn):
        d = J(fn)
        o.append(f"{label}: **{d['seconds']:.0f} s = {d['layers_per_s']:.2f} layers/s**, {d['layers_replaced']} of {d['layers']} layers replaced, {d['candidates_evaluated']} candidates; phases {d['phases_ms']}; sample check: {d.get('sample_check')}.\n\n")
if E("streams_r05.json"):
    o.append("## Streams (`streams_r05.json`: `tools/r05_probe.py block`, B_eigh of the Llama block, five passes each)\n\n")
    for line in J("streams_r05.json")["block"]:
        o.append(f"* streams {line['streams']}, longest first {line['longest_first']}: (step ms, B_eigh ms) {line['step_ms, B_eigh_ms']}\n")
    o.append("\n(That run drew candidates from the high-priority pool first; inside `bench.py`'s process chains on high-priority streams ran at 243 ms, and the default became normal priority, four streams: the table above.)\n")
open(os.path.join(root, "README.md"), "w").write("".join(o))
print("profiles/README.md written,", sum(len(x) for x in o), "bytes")
