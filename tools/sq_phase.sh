#!/bin/bash
# SQ instruction counters of k_ncc_mfma per phase knob (debug library): run on the GPU box from the repo root
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for D in ${PHASES:-3 2 1 0}; do
  rm -rf $OUT/pmc_ph_$D
  VBS_NCC_DBG=$D rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-include-regex "k_ncc_mfma" --output-format csv -d $OUT/pmc_ph_$D -- python3 $GRAFT_REPO_ROOT/tools/gpu_ncc_phase.py 512 child > $OUT/pmc_ph_$D.log 2>&1
done
echo done
