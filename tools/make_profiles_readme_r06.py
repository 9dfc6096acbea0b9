"""profiles/README.md of round 6 from the round's files (bench_r06.json = the compact line, bench_detail_r06.json = every side
block, pmc_*_r06.json, the full-depth runs).  Usage: python tools/make_profiles_readme_r06.py"""
import csv, json, os

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def J(n):
    text = open(os.path.join(root, n)).read().strip()
    try:
        return json.loads(text)
    except json.JSONDecodeError:      # (a tool's stdout: the JSON object is its last line)
        return json.loads(text.split("\n")[-1])


def E(n):
    return os.path.exists(os.path.join(root, n))


b = J("bench_r06.json")
d = J("bench_detail_r06.json")
d = d.get("detail", d)
ro, cb, ph = b["roofline"], b["cpu_baseline"], b["phases_ms"]
k, f = d["kernels"], d["decomposed_fwd"]
o = []
o.append("# profiles -- round 6 (one MI355X per call, ROCm 7.2, gpurun boxes)\n\n")
o.append("Every file carries its round in its name; rounds 1-5 stay for history (their README text is in git).  This file is produced by\n"
         "`tools/make_profiles_readme_r06.py` from the round-6 files.  The boxes of the pool differ by a few per cent in the clock they hold under\n"
         "load; numbers from different files may come from different boxes.\n\n")
o.append("Files (round 6):\n\n"
         "* `bench_r06.json` -- the LAST stdout line of `python bench.py --steps 5 --warmup 1` (the compact line the driver parses, <= 4 KB); `bench_detail_r06.json` -- the detail file the same run wrote (`bench_detail.json`: every side block)\n"
         "* `gpu_tests_r06.log` -- `python -m pytest tests -q -m gpu`\n"
         "* `rocprofv3_kernel_stats_default_r06.csv` -- `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline`: the kernel summary of the DEFAULT command (kernels of concurrent lanes overlap, so their durations read longer than alone; the `Cijk_*` / `attn_fwd` kernels are the CALLER's model and the library side of the comparison lines)\n"
         "* `rocprofv3_headline_steps_r06.txt` -- `tools/prof_kernels.sh headline2 bench.py --steps 3 --warmup 1 --no-extras`: the headline steps alone\n"
         "* `rocprofv3_kernel_stats_r06.csv`, `roofline_kernel_split_r06.json` -- the same for `bench.py --workload c2 --steps 3 --warmup 1 --no-extras` (BASELINE configs[1]); the filtered route's dominant kernel split by duration\n"
         "* `pmc_symv_r06.json/.csv` (the batched SYMV, four matrices per launch), `pmc_gemm_f64_r06.json`, `pmc_mfma_r06.json`, `pmc_syrk_r06.json` -- separate `rocprofv3 --pmc` passes over the torch-free `tools/pmc_driver batched 4 | eigh | mfma | syrk`, condensed on the box (`tools/pmc_*_summary.py`); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950; each carries the sha256 of the kernel sources it was measured on\n"
         "* `batched_eigh_*_r06.*`, `lanes_ab_r06.txt` -- `tools/probes/batched_eigh.py`, `tools/block_b_eigh.py`: 1-4 matrices per launch, lockstep / staggered, lanes x batch caps, CU-partitioned lanes, pool streams\n"
         "* `c4_stack_32blocks_bf16_r06.json`, `c4_hf_llama3_8b_r06.json` -- the full-depth runs (`tools/c4_stack.py 32 bf16 --trade-off 640 --max-ppl 0.4`, `tools/c4_hf_llama.py 32`; metric iterator over 16 batches: none recurs within a layer) with a sample check of one replaced layer (`tests/factor_checks.py`); `*_recurring_r06.json` -- the same with `METRIC_POOL=4` (batches recur within a layer: PrefixMemo across candidates)\n"
         "* `bench_r06_rehearsal_2ranks_1gpu.json` -- `PTD_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 2 --warmup 1`: the N = 2 code path with BOTH ranks on the one GPU of the box (gloo): a check that the path runs and what its line carries (`comm_ms`, `b_eigh_ms_max`, `d_metrics_ms_max`, `cov_collective`), not a scaling number\n"
         "* `tools/refresh_profiles.sh 06 main|bench|pmc` reruns them on a GPU box\n\n")
o.append("## Headline (`bench_r06.json`)\n\n")
o.append(f"* `value` = **{b['value']:.2f} layers/s** ({b['ms_per_step']:.0f} ms per step, spread {100 * b['spread']:.1f} %; steps {d['step_ms']}): {b['config']['workload']}.\n")
o.append(f"* phases of one step (`phases_ms`): A accumulate {ph['A_accumulate']:.1f} ms, B eigh {ph['B_eigh']:.1f}, C factors {ph['C_factors']:.1f}, D metrics {ph['D_metrics']:.1f}, host and gaps {ph['other_host_and_gaps']:.1f}; layers replaced: {d['replaced']}.\n")
o.append(f"* `roofline`: `{ro['kernel'].split(' ')[0]}`, {ro['matrices_per_launch']} matrices per launch, {ro['avg_launch_us']:.1f} us per launch over {ro['launches']} columns (dispatch-attached HIP events inside bench.py): "
         f"{ro['achieved']:.0f} GB/s on the algorithmic bytes = **{ro['frac']:.2f}** of 8 TB/s beside the other lanes' work; `traffic` {ro['traffic'] / 1e6:.1f} MB per launch ({ro['traffic_source']}); `solver_frac` {ro['solver_frac']:.2f}; the eigensolver call {ro['eigh_ms']:.0f} ms for the batch.\n")
o.append(f"* `cpu_baseline`: **{cb['value']:.3f} layers/s** on {cb['cores']} host threads ({cb['sample']}); seconds per layer {cb['s_per_layer']}.\n")
o.append(f"* other configs on the line: configs[1] `c2_layers_per_s` **{b['c2_layers_per_s']:.1f}** ({b['c2_ms_per_step']:.1f} ms per step); configs[2] `c3_s` **{b['c3_s']:.1f} s** ({b['c3_layers_per_s']:.2f} layers/s; phases {d['c3']['phases_ms']}); "
         f"configs[0] `c1_cpu_s` {b['c1_cpu_s']:.2f} s; configs[4] `fwd_gflops` {b['fwd_gflops']}, `fwd_vs_lib_pair` {b['fwd_vs_lib_pair']}.\n")
rec = d["metric_batches_recurring"]
o.append(f"* iterators whose batches recur within a layer's search (never part of `value` / `c3_s`): headline stack on a 4-batch metric iterator **{rec['ms_per_step']:.0f} ms** per step (phases {rec['phases_ms']}); "
         f"falor on the ViT clone with a 24-batch iterator **{b['c3_recurring_batches_s']:.1f} s**.\n\n")
o.append("## One block, per shape (`bench_detail_r06.json`: `c4_block`, `c4_shapes`)\n\n| workload | ms per step (median) | steps | spread | A | B_eigh | D |\n|---|---|---|---|---|---|---|\n")
for name in ("f32", "bf16"):
    q = d["c4_block"][name]
    p_ = q["phases_ms"]
    o.append(f"| c4_block {name} | {q['ms_per_step']:.1f} | {q['step_ms']} | {100 * q['spread']:.1f} % | {p_['A_accumulate']:.1f} | {p_['B_eigh']:.0f} | {p_['D_metrics']:.0f} |\n")
o.append("\n| shape | f32 ms per layer | bf16 ms per layer | eigensolver route (f32 run) | eigensolver ms |\n|---|---|---|---|---|\n")
for shape in ("q_o", "k_v", "gate_up", "down"):
    a32, a16 = d["c4_shapes"]["f32"][shape], d["c4_shapes"]["bf16"][shape]
    o.append(f"| {shape} ({a32['n_in']} -> {a32['n_out']}) | {a32['ms_per_layer']:.1f} | {a16['ms_per_layer']:.1f} | {a32['eigh']['route']} (n = {a32['eigh']['n']}, k = {a32['eigh']['k']}) | {a32['eigh']['ms']:.1f} |\n")
o.append("\n## Kernels (HIP events in bench.py, median of three loops; MfmaUtil from `pmc_mfma_r06.json`)\n\n| line | ms | rate | of peak | MfmaUtil |\n|---|---|---|---|---|\n")
for name, unit, key in (("syrk_f32_f64acc", "f32 mfma", "frac_of_f32_mfma_peak"), ("gemm_f32_nt", "f32 mfma", "frac_of_f32_mfma_peak"),
                        ("syrk_bf16_f64acc", "bf16 mfma", "frac_of_bf16_mfma_peak"), ("gemm_bf16_nt", "bf16 mfma", "frac_of_bf16_mfma_peak")):
    q = k[name]
    o.append(f"| {name} (n = T = 4096 / 4096^3) | {q['ms']:.3f} | {q['tflops']:.0f} TFLOP/s | {q[key]:.2f} {unit} | {q.get('mfma_util_pmc_percent', 0):.0f} %{' (stale)' if q.get('mfma_util_stale') else ''} |\n")
for name in ("nsr_f32", "nsr_bf16_vocab"):
    q = k[name]
    o.append(f"| {name} {q['shape']} | {q['ms']:.3f} | {q['gbps']:.0f} GB/s | {q['frac_of_hbm_peak']:.2f} hbm | |\n")
o.append("\nThe bf16 covariance product at the calibration shapes (2048 tokens a step; `kernels.syrk_bf16_calibration_shapes`):\n\n| n | us per step, one call per step | us per step, 8 steps per call | MFMA bound us | fraction (8 per call) |\n|---|---|---|---|---|\n")
for n_, q in k["syrk_bf16_calibration_shapes"].items():
    if isinstance(q, dict) and "us_per_step_multi_8" in q:
        o.append(f"| {n_[1:]} | {q['us_per_step_single_call']:.1f} | {q['us_per_step_multi_8']:.1f} | {q['mfma_bound_us']:.2f} | {q['frac_of_bound_multi_8']:.2f} |\n")
o.append("\n(the tiling's own bound is the L2 -> LDS fill: 2 MB per CU and step at n = 4096 against 66-73 GB/s per CU = 28-31 us; DESIGN section 7 item 3)\n")
o.append("\n## Decomposed forward (configs[4]; `decomposed_fwd`)\n\n| T | r | package ms | library pair ms | package / dense speed-up |\n|---|---|---|---|---|\n")
for T, blk in ((16384, f), (4096, f["rows_4096"]), (65536, f["rows_65536"])):
    for r in (256, 512, 1024):
        q = blk[f"r{r}"]
        o.append(f"| {T} | {r} | {q['ms']:.3f} | {q['torch_hipblaslt_pair_ms']:.3f} | {q['speedup_vs_dense']:.2f} |\n")
    o.append(f"| {T} | dense | {blk['dense_ms']:.3f} | {blk['dense_torch_hipblaslt_ms']:.3f} (library) | |\n")
o.append("\n## Default command, device time by kernel (`rocprofv3_kernel_stats_default_r06.csv`)\n\n")
rows = list(csv.DictReader(open(os.path.join(root, "rocprofv3_kernel_stats_default_r06.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
share = lambda pred: 100 * sum(float(r["TotalDurationNs"]) for r in rows if pred(r["Name"])) / tot  # noqa: E731
o.append(f"total {tot / 1e9:.1f} s of kernels; `sytrd_alpha` + `sytrd_symv2` + `sytrd_symv` **{share(lambda n: 'sytrd_alpha' in n or 'sytrd_symv' in n):.1f} %** "
         f"(VERDICT r5 asked <= 20 %); hipBLASLt `Cijk_*` (the caller's layers and the library side of the comparison lines) {share(lambda n: 'Cijk_' in n):.1f} %; "
         f"the package's bf16 / f32 / f64 GEMM kernels {share(lambda n: 'ptd::' in n and 'gemm_' in n):.1f} %.  Top ten:\n\n| kernel | calls | share |\n|---|---|---|\n")
for r in rows[:10]:
    o.append(f"| `{r['Name'][:90].replace('ptd::(anonymous namespace)::', '').replace('|', '/')}` | {r['Calls']} | {100 * float(r['TotalDurationNs']) / tot:.1f} % |\n")
o.append("\n## Full depth (32 blocks, 224 layers, one MI355X)\n\n| run | metric iterator | seconds | layers/s | replaced | phases ms | sample check |\n|---|---|---|---|---|---|---|\n")
for label, fn in (("Llama-shaped stack", "c4_stack_32blocks_bf16_r06.json"), ("Llama-shaped stack", "c4_stack_32blocks_bf16_recurring_r06.json"),
                  ("transformers.LlamaForCausalLM", "c4_hf_llama3_8b_r06.json"), ("transformers.LlamaForCausalLM", "c4_hf_llama3_8b_recurring_r06.json")):
    if not E(fn):
        continue
    q = J(fn)
    sc = q["sample_check"]
    o.append(f"| {label} (`{fn}`) | {q.get('metric_pool', '?')} batches | **{q['seconds']:.1f}** | {q['layers_per_s']:.2f} | {q['layers_replaced']} of {q['layers']} | {q['phases_ms']} | "
             f"{sc['checked']} r = {sc['rank']}: orthonormality {sc['orthonormality_max_dev']:.1e}, first factor {sc['first_factor_rel_err']:.1e}, energy / optimum {sc['captured_energy_over_optimal']:.5f} |\n")
open(os.path.join(root, "README.md"), "w").write("".join(o))
print("".join(o)[:3000])
