set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests_r01.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/gpu_tests_r01.log
timeout -k 10 400 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err; echo "bench rc=$?"
cut -c1-200 gpurun_out/bench_r01.json
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-extras > $GRAFT_REPO_ROOT/gpurun_out/prof_r01b.log 2>&1); echo "rocprof rc=$?"
timeout -k 10 300 python tools/c4_shapes.py > gpurun_out/c4_f32.json 2> gpurun_out/c4_f32.err; echo "c4 f32 rc=$?"
timeout -k 10 300 python tools/c4_shapes.py bf16 > gpurun_out/c4_bf16.json 2> gpurun_out/c4_bf16.err; echo "c4 bf16 rc=$?"
ls gpurun_out/prof_r01b/*/
