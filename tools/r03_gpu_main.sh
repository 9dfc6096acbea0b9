#!/bin/bash
# Round-3 GPU pass: the -m gpu suite, the bench line, a two-rank rehearsal of `bench.py --gpus 2` started WITHOUT a
# launcher, and the 32-block C4 stack.  A step that hits its time limit ends the call (no further GPU step).
set -o pipefail
cd $GRAFT_REPO_ROOT
step() {  # step <seconds> <label> <command...>
  local limit=$1 label=$2; shift 2
  timeout -k 10 $limit "$@"; local rc=$?
  echo "$label rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$label hit its limit: stopping"; exit $rc; fi
  return 0
}
PARTS=${1:-tests,bench,rehearse,stack32,probe}
if [[ $PARTS == *tests* ]]; then
  step 1000 tests bash -c "python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/gpu_tests_r03.log 2>&1"; tail -25 gpurun_out/gpu_tests_r03.log
fi
if [[ $PARTS == *bench* ]]; then
  step 500 bench bash -c "python bench.py --steps 5 --warmup 1 > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err"; cut -c1-400 gpurun_out/bench_r03.json; tail -3 gpurun_out/bench_r03.err
fi
if [[ $PARTS == *rehearse* ]]; then
  step 300 rehearse bash -c "PTD_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_r03_n2.json 2> gpurun_out/bench_r03_n2.err"; cut -c1-400 gpurun_out/bench_r03_n2.json; tail -3 gpurun_out/bench_r03_n2.err
fi
if [[ $PARTS == *stack32* ]]; then
  step 900 c4stack bash -c "PTD_PHASES=1 python tools/c4_stack.py 32 bf16 > gpurun_out/c4_stack_32blocks_bf16_r03.json 2> gpurun_out/c4_stack_32.err"; cat gpurun_out/c4_stack_32blocks_bf16_r03.json | cut -c1-1500; tail -3 gpurun_out/c4_stack_32.err
fi
if [[ $PARTS == *probe* ]]; then
  step 300 probe bash -c "python tools/chefsi_probe.py > gpurun_out/chefsi_probe.json 2> gpurun_out/chefsi_probe.err"; cat gpurun_out/chefsi_probe.json; tail -3 gpurun_out/chefsi_probe.err
fi
if [[ $PARTS == *filt* ]]; then
  step 300 filt bash -c "PTD_JACOBI_DEBUG=1 python tools/filtered_probe.py > gpurun_out/filtered_probe.json 2> gpurun_out/filtered_probe.err"; cat gpurun_out/filtered_probe.json; grep -v amdgpu.ids gpurun_out/filtered_probe.err | tail -12
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_filt -- python3 $GRAFT_REPO_ROOT/tools/filtered_probe.py > $GRAFT_REPO_ROOT/gpurun_out/prof_filt.log 2>&1); echo "rocprof rc=$?"
  f=$(ls gpurun_out/prof_filt/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-160
fi
if [[ $PARTS == *gemmtest* ]]; then
  step 300 gemmtest bash -c "python -m pytest tests/test_kernels_gpu.py -q -x -k 'gemm_f64 or resident_kernels or two_eigendecompositions or filtered or eigh_topk or eigh_mid' > gpurun_out/gemmtest.log 2>&1"; tail -15 gpurun_out/gemmtest.log
fi
if [[ $PARTS == *determ* ]]; then
  step 200 determ bash -c "python tools/filtered_determinism.py 2> gpurun_out/determinism.err"; grep -v amdgpu.ids gpurun_out/determinism.err | awk '/--- run/{r=$3} /checksum/{print r, $0}' | sort -k4,4 -s | awk '{key=$4" "$5" "$6" "$7; if (key==prev && $NF!=pv) print "DIFF:", $0; prev=key; pv=$NF}' | head; grep -c checksum gpurun_out/determinism.err
fi
if [[ $PARTS == *mfab* ]]; then
  step 300 mfab bash -c "python tools/mfma_shape_ab.py > gpurun_out/mfma_shape_ab.txt 2> gpurun_out/mfma_shape_ab.err"; cat gpurun_out/mfma_shape_ab.txt; grep -v amdgpu.ids gpurun_out/mfma_shape_ab.err | tail -5
fi
if [[ $PARTS == *c4cpu* ]]; then
  step 1000 c4cpu bash -c "python tools/c4_shapes_cpu.py > gpurun_out/c4_cpu_r03.json 2> gpurun_out/c4_cpu.err"; tail -3 gpurun_out/c4_cpu.err; cut -c1-400 gpurun_out/c4_cpu_r03.json
fi
if [[ $PARTS == *c3ab* ]]; then
  step 200 c3a bash -c "PTD_PHASES=1 python tools/c3_vit.py > gpurun_out/c3_phases_r03.json 2> gpurun_out/c3_phases.err"; cut -c1-260 gpurun_out/c3_phases_r03.json
  step 200 c3b bash -c "PTD_PHASES=1 PTD_GEMM_F64_NO_GLDS=1 python tools/c3_vit.py > gpurun_out/c3_phases_noglds.json 2> gpurun_out/c3_phases2.err"; cut -c1-260 gpurun_out/c3_phases_noglds.json
  step 200 c3c bash -c "python tools/c3_vit.py > gpurun_out/c3_plain.json 2> gpurun_out/c3_plain.err"; cut -c1-260 gpurun_out/c3_plain.json
fi
if [[ $PARTS == *oversample* ]]; then
  for f in 0.25 0.375 0.5; do
    step 200 over$f bash -c "PTD_EIGH_FILTER_OVERSAMPLE=$f PTD_JACOBI_DEBUG=1 python tools/filtered_probe.py > gpurun_out/filtered_probe_os$f.json 2> gpurun_out/filtered_probe_os$f.err"
    python - <<PY
import json
d=json.load(open("gpurun_out/filtered_probe_os$f.json"))
print("oversample $f:", round(d["filtered_ms"],2), "ms", [round(x,2) for x in d["filtered_profile"]["ms"]], d["filtered_profile"]["launches"], "resid %.1e orth %.1e dv %.1e" % (d["filtered"]["resid"], d["filtered"]["orth"], d["filtered"]["max_dv"]))
PY
    grep "eigh_filtered\] n=" gpurun_out/filtered_probe_os$f.err | head -1
  done
fi
if [[ $PARTS == *retrydbg* ]]; then
  step 200 retrydbg bash -c "python tools/retry_debug.py 2> gpurun_out/retry_debug.err"; grep -v amdgpu.ids gpurun_out/retry_debug.err | grep -v "eigh_tridiag" | head -40
fi
if [[ $PARTS == *hangdbg* ]]; then
  step 90 hangdbg bash -c "PTD_JACOBI_DEBUG=1 python -m pytest tests/test_kernels_gpu.py -x -q -s -k 'declines_where' > gpurun_out/hangdbg.log 2>&1"; grep -v "amdgpu.ids" gpurun_out/hangdbg.log | tail -40 | cut -c1-220
fi
