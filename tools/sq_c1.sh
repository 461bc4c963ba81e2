#!/bin/bash
# SQ counters of k_ncc_mfma / k_stage on the reference's real configuration (c1: 480x450 crop of 640x480 BGR, small branch), per
# build of the library (SFX="'' _x"): run on the GPU box from the repo root.  KERNEL=k_ncc_mfma|k_stage
OUT=$GRAFT_REPO_ROOT/gpurun_out
K=${KERNEL:-k_ncc_mfma}
cd /tmp && export TMPDIR=/tmp
for S in ${SFX:-""}; do
  T=${S:-prod}
  i=0
  for SET in "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1)); D=$OUT/pmc_c1_${T}_$i
    rm -rf $D
    timeout -k 10 200 rocprofv3 --pmc $SET --kernel-include-regex "$K" --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/gpu_lib_ab_c1.py child "$S" 512 > $D.log 2>&1 || echo "set failed: $T $i"
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for d in sorted(glob.glob(out + "/pmc_c1_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(os.path.basename(d), {k: round(v / max(n[k], 1) / 1e6, 3) for k, v in acc.items()})
PY
