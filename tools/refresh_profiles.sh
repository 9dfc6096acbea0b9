#!/bin/bash
# Regenerates the measurements under profiles/ on a GPU box:  bash tools/refresh_profiles.sh <round, e.g. 02> [part]
# part: all (default) | main (tests + bench + kernel stats) | c4 (= c4gpu + c4cpu) | pmc.  Outputs land in gpurun_out/ and are
# condensed / copied into profiles/ afterwards in the build container (tools/pmc_summary.py, pmc_mfma_summary.py,
# make_profiles_readme.py).
set -o pipefail
R=${1:-02}; PART=${2:-all}
cd $GRAFT_REPO_ROOT
if [ "$PART" = all ] || [ "$PART" = main ]; then
  timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests_r$R.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/gpu_tests_r$R.log
fi
if [ "$PART" = all ] || [ "$PART" = main ] || [ "$PART" = bench ]; then
  timeout -k 10 500 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_r$R.txt 2> gpurun_out/bench_r$R.err; echo "bench rc=$?"
  tail -1 gpurun_out/bench_r$R.txt > gpurun_out/bench_r$R.json; cp bench_detail.json gpurun_out/bench_detail_r$R.json
  cut -c1-200 gpurun_out/bench_r$R.json
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r$R   # (earlier runs leave PID-named files beside the new ones)
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r$R -- python3 $GRAFT_REPO_ROOT/bench.py --workload c2 --steps 3 --warmup 1 --no-extras > $GRAFT_REPO_ROOT/gpurun_out/prof_r$R.log 2>&1); echo "rocprof rc=$?"
  # the dominant kernel's launches split by duration (the long C X products / the shorter X W ones), then drop the raw
  # trace: gpurun copies back at most 64 MiB
  python3 - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/prof_r$R/*/*kernel_trace.csv")
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "gemm_f64_glds_kernel<5, false>" in r["Kernel_Name"]]
    us = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in rows)
    long_ = [u for u in us if u > 400.0]
    short = [u for u in us if u <= 400.0]
    json.dump({"kernel": "gemm_f64_glds_kernel<5, false>", "launches": len(us), "long_launches_K4096": len(long_),
               "long_avg_us": sum(long_) / max(len(long_), 1), "short_launches": len(short),
               "short_avg_us": sum(short) / max(len(short), 1),
               "source": "rocprofv3 --kernel-trace of python bench.py --workload c2 --steps 3 --warmup 1 --no-extras"},
              open("gpurun_out/roofline_kernel_split_r$R.json", "w"), indent=1)
PY
  rm -f gpurun_out/prof_r$R/*/*kernel_trace.csv
  # the same for the DEFAULT command the driver runs (the fixed stack as `value`, every side block; the CPU baseline, which
  # launches no kernels, left out): which kernels the whole bench spends its device time in
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_default_r$R
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_default_r$R -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_default_r$R.log 2>&1); echo "rocprof default rc=$?"
  rm -f gpurun_out/prof_default_r$R/*/*kernel_trace.csv
fi
if [ "$PART" = all ] || [ "$PART" = c4 ] || [ "$PART" = c4gpu ]; then
  timeout -k 10 300 python tools/c4_shapes.py > gpurun_out/c4_f32_r$R.json 2> gpurun_out/c4_f32.err; echo "c4 f32 rc=$?"
  timeout -k 10 300 python tools/c4_shapes.py bf16 > gpurun_out/c4_bf16_r$R.json 2> gpurun_out/c4_bf16.err; echo "c4 bf16 rc=$?"
  timeout -k 10 300 python tools/c4_stack.py 2 > gpurun_out/c4_stack_f32_r$R.json 2> gpurun_out/c4_stack_f32.err; echo "c4 stack f32 rc=$?"
  timeout -k 10 300 python tools/c4_stack.py 2 bf16 > gpurun_out/c4_stack_bf16_r$R.json 2> gpurun_out/c4_stack_bf16.err; echo "c4 stack bf16 rc=$?"
  timeout -k 10 300 python tools/c3_vit.py > gpurun_out/c3_vit_r$R.json 2> gpurun_out/c3_vit.err; echo "c3 rc=$?"
  PTD_PHASES=1 timeout -k 10 300 python tools/c3_vit.py > gpurun_out/c3_phases_r$R.json 2> gpurun_out/c3_phases.err; echo "c3 phases rc=$?"
fi
if [ "$PART" = all ] || [ "$PART" = c4 ] || [ "$PART" = c4cpu ]; then
  # (the torch-CPU oracle on the box's host cores: ~8 minutes; the tool writes a line a minute to gpurun_out/c4_cpu.err)
  timeout -k 10 900 python tools/c4_shapes_cpu.py > gpurun_out/c4_cpu_r$R.json 2> gpurun_out/c4_cpu.err; echo "c4 cpu rc=$?"; tail -5 gpurun_out/c4_cpu.err
fi
if [ "$PART" = all ] || [ "$PART" = pmc ]; then
  hipcc -O2 -o tools/pmc_driver tools/pmc_driver.cpp -Iinclude -Lptdeco_amd -lptdeco_hip -Wl,-rpath,'$ORIGIN/../ptdeco_amd' || exit 1
  cd /tmp && export TMPDIR=/tmp
  G=$GRAFT_REPO_ROOT
  for c in FETCH_SIZE:pmc_fetch WRITE_SIZE:pmc_write "TCC_HIT_sum TCC_MISS_sum":pmc_tcc; do
    ctr=${c%%:*}; dir=${c##*:}
    # (round 6: the unit of the headline step's direct lane -- four (4096, 2048) matrices per launch, every column blocked)
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-include-regex sytrd_symv --kernel-trace --output-format csv -d $G/gpurun_out/$dir -- $G/tools/pmc_driver batched 4 > $G/gpurun_out/$dir.log 2>&1; echo "$dir rc=$?"
  done
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "syrk|gemm" --kernel-trace --output-format csv -d $G/gpurun_out/pmc_mfma -- $G/tools/pmc_driver mfma > $G/gpurun_out/pmc_mfma.log 2>&1; echo "pmc_mfma rc=$?"
  timeout -k 10 300 rocprofv3 --pmc MfmaUtil --kernel-include-regex "syrk|gemm" --kernel-trace --output-format csv -d $G/gpurun_out/pmc_mfma2 -- $G/tools/pmc_driver mfma > $G/gpurun_out/pmc_mfma2.log 2>&1; echo "pmc_mfma2 rc=$?"
  # the bf16 covariance product at n = T = 4096 and at the Llama-3-8B calibration shapes (tools/pmc_syrk_summary.py)
  for ctr in MfmaUtil FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-include-regex "syrk|gemm_bf16" --kernel-trace --output-format csv -d $G/gpurun_out/pmcs_$ctr -- $G/tools/pmc_driver syrk > $G/gpurun_out/pmcs_$ctr.log 2>&1; echo "pmcs_$ctr rc=$?"
  done
  # the dominant kernel of the default eigensolver route (filtered subspace iteration): the product C X
  for c in FETCH_SIZE:pmcf_fetch WRITE_SIZE:pmcf_write "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE":pmcf_mfma; do
    ctr=${c%%:*}; dir=${c##*:}
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-include-regex gemm_f64_glds --kernel-trace --output-format csv -d $G/gpurun_out/$dir -- $G/tools/pmc_driver eigh > $G/gpurun_out/$dir.log 2>&1; echo "$dir rc=$?"
  done
  # condense on the box (gpurun copies back at most 64 MiB: the raw counter tables stay here), summaries -> gpurun_out/
  cd $G
  python3 tools/pmc_summary.py $R 4096 4 > gpurun_out/pmc_summary.log 2>&1; echo "pmc_summary rc=$?"
  python3 tools/pmc_mfma_summary.py $R > gpurun_out/pmc_mfma_summary.log 2>&1; echo "pmc_mfma_summary rc=$?"
  python3 tools/pmc_syrk_summary.py $R > gpurun_out/pmc_syrk_summary.log 2>&1; echo "pmc_syrk_summary rc=$?"
  python3 tools/pmc_filtered_summary.py $R > gpurun_out/pmc_filtered_summary.log 2>&1; echo "pmc_filtered_summary rc=$?"
  mkdir -p gpurun_out/pmc_summaries_r$R && cp profiles/pmc_*_r$R.* gpurun_out/pmc_summaries_r$R/
  rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_tcc gpurun_out/pmc_mfma gpurun_out/pmc_mfma2 gpurun_out/pmcs_* gpurun_out/pmcf_*
fi
