#!/bin/bash
# L1 / TA / L2 counters of the blur kernel on the product path (tools/gpu_detect_run.py): separate PMC passes.
# (Round 3 also asked for "TCP_TAGRAM0_REQ_sum TCP_LFIFO_STALL_CYCLES_sum": that pair cannot be collected together on gfx950 -
#  rocprofv3 aborts ("Request exceeds the capabilities of the hardware to collect") inside the first launch and leaves the
#  process hung until gpurun kills it for silence.  Removed; each pass now also runs under its own timeout.)
set -e
TAG=${1:-r3blurmem}
FR=${2:-256}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $OUT/pmc_mem_${i}_$TAG
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-include-regex "k_blur" --output-format csv -d $OUT/pmc_mem_${i}_$TAG -- python3 $GRAFT_REPO_ROOT/tools/gpu_detect_run.py $FR 2 > $OUT/pmc_mem_${i}_$TAG.log 2>&1 || echo "pass $i failed"
done
echo done
