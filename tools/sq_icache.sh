#!/bin/bash
# Instruction-fetch counters of k_ncc_mfma under two builds of the library (suffixes in SFX, default "_r4 ''"): is the
# kernel waiting for its instructions?  Run on the GPU box from the repo root.
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL|SQ_WAIT|SQ_BUSY|SQC_" | head -80 > $OUT/icache_avail.log
for S in ${SFX:-_r4 ""}; do
  T=${S:-prod}
  for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN GRBM_GUI_ACTIVE"; do
    D=$OUT/pmc_ic_${T}_$(echo $SET | cut -c1-12 | tr ' ' _)
    rm -rf $D
    timeout -k 10 200 rocprofv3 --pmc $SET --kernel-include-regex "k_ncc_mfma" --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/tools/gpu_lib_ab.py child "$S" 512 > $D.log 2>&1 || echo "set failed: $T $SET"
  done
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
for d in sorted(glob.glob(out + "/pmc_ic_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(os.path.basename(d), {k: (round(v / max(n[k], 1)), n[k]) for k, v in acc.items()})
PY
