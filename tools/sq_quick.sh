#!/bin/bash
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_sqa_q $OUT/pmc_sqb_q $OUT/pmc_sqc_q
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex "k_blur16|k_blur_mfma|k_ncc_mfma" --output-format csv -d $OUT/pmc_sqa_q -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py 512 2 > $OUT/pmc_sqa_q.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-include-regex "k_blur16|k_blur_mfma|k_ncc_mfma" --output-format csv -d $OUT/pmc_sqb_q -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py 512 2 > $OUT/pmc_sqb_q.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_LDS SQ_WAVE32_INSTS SQ_LEVEL_WAVES --kernel-include-regex "k_blur16|k_blur_mfma|k_ncc_mfma" --output-format csv -d $OUT/pmc_sqc_q -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py 512 2 > $OUT/pmc_sqc_q.log 2>&1 || echo "sqc failed"
echo done
