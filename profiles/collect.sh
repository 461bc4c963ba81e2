#!/bin/bash
# Run ON THE GPU BOX from the repo root (via gpurun): collects the evidence bench.py's numbers are checked against.
#   1. rocprofv3 kernel trace + stats of a bench run
#   2. two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over the threshold+CCL stage AS IT RUNS IN THE FUSED PATH
#      (k_morph, k_ccl_band, k_ccl_open, k_label, k_finalize), per workload (c3 = 1280x1024, c5 = 1920x1200)
#   3. SQ counters of the two matrix-core kernels
# rocprofv3 is always given the program itself after `--`, never a wrapper; counters are collected in their own runs.
set -e
TAG=${1:-r2}
FR=${2:-512}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --frames $FR --batch $FR --steps 2 --warmup 1 --no-cpu-baseline --no-extras --pass-streams 1 > $OUT/prof_$TAG.log 2>&1
echo "collected stats $TAG"
STAGE='k_stage|k_morph|k_ccl|k_label|k_finalize'
for WL in c3 c5; do
  rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$STAGE" --output-format csv -d $OUT/pmc_fetch_${TAG}_$WL -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py $FR 2 $WL > $OUT/pmc_fetch_${TAG}_$WL.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$STAGE" --output-format csv -d $OUT/pmc_write_${TAG}_$WL -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py $FR 2 $WL > $OUT/pmc_write_${TAG}_$WL.log 2>&1
  echo "collected traffic $TAG $WL"
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex "k_blur16|k_blur_mfma|k_ncc_mfma" --output-format csv -d $OUT/pmc_sqa_$TAG -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py $FR 2 > $OUT/pmc_sqa_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-include-regex "k_blur16|k_blur_mfma|k_ncc_mfma" --output-format csv -d $OUT/pmc_sqb_$TAG -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py $FR 2 > $OUT/pmc_sqb_$TAG.log 2>&1
echo "collected sq $TAG"
