#!/bin/bash
# Run ON THE GPU BOX from the repo root (via gpurun): collects the evidence bench.py's numbers are checked against.
#   rocprofv3 kernel trace + stats of a bench run, then two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over the
#   threshold+CCL stage alone (rocprofv3 is given the program itself after `--`, never a wrapper).
set -e
TAG=${1:-r1}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --frames 1024 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $GRAFT_REPO_ROOT/tools/gpu_stage_run.py 512 > $OUT/pmc_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -- python3 $GRAFT_REPO_ROOT/tools/gpu_stage_run.py 512 > $OUT/pmc_write_$TAG.log 2>&1
echo collected $TAG
# SQ counters of the two matrix-core kernels (issue / wait / MFMA-busy cycles), two passes of 8 counters
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex "k_blur_mfma|k_ncc_mfma" --output-format csv -d $OUT/pmc_sqa_$TAG -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py 512 2 > $OUT/pmc_sqa_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-include-regex "k_blur_mfma|k_ncc_mfma" --output-format csv -d $OUT/pmc_sqb_$TAG -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py 512 2 > $OUT/pmc_sqb_$TAG.log 2>&1
echo collected sq $TAG
