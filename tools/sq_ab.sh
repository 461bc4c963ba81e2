#!/bin/bash
# SQ counters of k_ncc_mfma under several builds of the library (SFX="_r4 _abc ''"): run on the GPU box from the repo root.
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for S in ${SFX:-_r4 ""}; do
  T=${S:-prod}
  i=0
  for SET in "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    i=$((i+1)); D=$OUT/pmc_ab_${T}_$i
    rm -rf $D
    timeout -k 10 200 rocprofv3 --pmc $SET --kernel-include-regex "k_ncc_mfma" --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/gpu_lib_ab.py child "$S" 512 > $D.log 2>&1 || echo "set failed: $T $i"
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for d in sorted(glob.glob(out + "/pmc_ab_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(os.path.basename(d), {k: round(v / max(n[k], 1) / 1e6, 2) for k, v in acc.items()})
PY
