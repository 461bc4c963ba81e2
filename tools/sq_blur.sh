#!/bin/bash
# SQ counters of the blur kernel on the product path (k_blur16 or, with BLUR_IMPL=1 in tools/gpu_detect_run.py's environment,
# k_blur_mfma): three PMC passes, program = tools/gpu_detect_run.py.  Run on the GPU box from the repo root.
set -e
TAG=${1:-r3blur}
FR=${2:-512}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_st_a_$TAG $OUT/pmc_st_b_$TAG $OUT/pmc_st_c_$TAG
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-include-regex "k_blur" --output-format csv -d $OUT/pmc_st_a_$TAG -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py $FR 2 > $OUT/pmc_st_a_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "k_blur" --output-format csv -d $OUT/pmc_st_b_$TAG -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py $FR 2 > $OUT/pmc_st_b_$TAG.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES --kernel-include-regex "k_blur" --output-format csv -d $OUT/pmc_st_c_$TAG -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py $FR 2 > $OUT/pmc_st_c_$TAG.log 2>&1 || echo "pass c failed"
echo done
