#!/bin/bash
# VALU / SALU instruction counts of k_stage per phase: the debug library's VBS_STAGE_STOP knob under rocprofv3 --pmc.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for D in ${PHASES:-2 10 12 0}; do
  rm -rf $OUT/pmc_stp_$D
  VBS_LIB=dbg VBS_STAGE_STOP=$D rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-include-regex "k_stage" --output-format csv -d $OUT/pmc_stp_$D -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py 512 2 > $OUT/pmc_stp_$D.log 2>&1
done
echo done
