#!/usr/bin/env python3
"""SQ counters of the threshold+CCL stage's kernels (tools/sq_stage.sh passes a, b, c) -> profiles/<tag>_sq_counters_stage.json.
usage: python profiles/summarize_stage_sq.py <tag> [frames_per_launch=512]"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 512
G = os.path.join(ROOT, "gpurun_out")


def kname(s):
    k = s.split("(")[0].replace("void ", "")
    if k.startswith("k_ccl<"):
        return "k_ccl_band" if k.startswith("k_ccl<0") else "k_ccl_open"
    return k.split("<")[0]


class RetryNamer:
    """`k_stage<.., 768>` dispatched right behind a `k_stage<.., 256>` is the second-chance launch (workgroups that leave at once on
    marker frames): its own name, so that "the last launch of k_stage" stays the one that did the work."""
    def __init__(self):
        self.small = -2

    def __call__(self, row):
        name, did = row["Kernel_Name"], int(row["Dispatch_Id"])
        k = kname(name)
        if k == "k_stage":
            inst = name.split("(")[0]
            if ", 256>" in inst:
                self.small = did
            elif did == self.small + 1:
                return "k_stage_retry"
        return k


sq = {}
for p in "abc":
    f = sorted(glob.glob(os.path.join(G, f"pmc_st_{p}_{tag}/**/*counter_collection.csv"), recursive=True))
    if not f:
        continue
    namer = RetryNamer()
    for r in sorted(csv.DictReader(open(f[-1])), key=lambda r: int(r["Dispatch_Id"])):
        sq.setdefault(namer(r), {})[r["Counter_Name"]] = float(r["Counter_Value"])       # last launch wins
for k, c in sq.items():
    w = c.get("SQ_WAVE_CYCLES")
    if w:
        c["frac_wait_any"] = round(c.get("SQ_WAIT_ANY", 0) / w, 3)
        c["frac_wait_inst"] = round(c.get("SQ_WAIT_INST_ANY", 0) / w, 3)
        c["frac_active"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / w, 3)
        c["frac_active_valu"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) / w, 3)
    if c.get("SQ_WAVES"):
        c["valu_per_wave"] = round(c.get("SQ_INSTS_VALU", 0) / c["SQ_WAVES"], 1)
        c["salu_per_wave"] = round(c.get("SQ_INSTS_SALU", 0) / c["SQ_WAVES"], 1)
        c["lds_per_wave"] = round(c.get("SQ_INSTS_LDS", 0) / c["SQ_WAVES"], 1)
        c["vmem_rd_per_wave"] = round(c.get("SQ_INSTS_VMEM_RD", 0) / c["SQ_WAVES"], 1)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        c["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"], 3)
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
        # SQ_BUSY_CYCLES is summed over the SQs (one per CU-pair group); ACTIVE_INST_VALU in quad-cycles over all waves
        c["valu_quadcycles_per_frame"] = round(c["SQ_ACTIVE_INST_VALU"] / frames)
import hashlib
_h = hashlib.sha256()
for _n in ("k_stage.hip", "stage_common.h", "ccl_common.h"):        # (bench.py: _sources_sha16 - the kernel these counters are of)
    _h.update(open(os.path.join(ROOT, "vision-basedsensor_amd", "csrc", _n), "rb").read())
json.dump({"tag": tag, "frames_per_launch": frames, "src_sha16": _h.hexdigest()[:16],
           "note": "one launch over `frames_per_launch` frames of the fused path (tools/gpu_detect_run.py); SQ_WAVE_CYCLES / "
                   "SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles summed over all waves; *_per_wave = instructions per wave",
           "kernels": sq}, open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters_stage.json"), "w"), indent=1)
for k, c in sorted(sq.items()):
    print(k, {x: c[x] for x in c if x.startswith(("frac", "valu_per", "salu_per", "lds_per", "vmem", "lds_conf"))}, "waves", c.get("SQ_WAVES"))
