set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --kernel-include-regex "shortk" --output-format csv -d $R/gpurun_out/skpmc_$tag -- python3 $R/tools/shortk_probe.py > $R/gpurun_out/skpmc_$tag.log 2>&1 || echo "pass $tag failed"
done
ls $R/gpurun_out/ | grep skpmc
