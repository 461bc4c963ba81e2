#!/usr/bin/env python3
"""Turn the raw rocprofv3 output merged back under gpurun_out/ into the small files kept in profiles/:
   <tag>_rocprofv3_kernel_stats.csv  rows of this repo's kernels from `--kernel-trace --stats`
   <tag>_pmc_traffic.json            HBM traffic per launch of the threshold+CCL stage from the FETCH_SIZE / WRITE_SIZE passes
usage: python profiles/summarize.py <tag> [frames_per_launch=256]"""
import csv, glob, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 256
G = os.path.join(ROOT, "gpurun_out")

def one(pattern):
    f = sorted(glob.glob(os.path.join(G, pattern), recursive=True))
    if not f:
        raise SystemExit(f"missing {pattern}")
    return f[-1]

rows = list(csv.reader(open(one(f"prof_{tag}/**/*kernel_stats.csv"))))
with open(os.path.join(ROOT, "profiles", f"{tag}_rocprofv3_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        if r[0].startswith(("void k_", "k_")):
            w.writerow(r)

def counter(pattern):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(one(pattern))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("k_"):
            agg[k].append(float(r["Counter_Value"]))
    return agg

fetch, write = counter(f"pmc_fetch_{tag}/**/*counter_collection.csv"), counter(f"pmc_write_{tag}/**/*counter_collection.csv")
stage = ("k_threshold", "k_morph", "k_label", "k_finalize")
per_kernel = {}
total = 0.0
for k in stage:
    n_launch = max(1, len(fetch[k]) // max(1, len(fetch["k_threshold"])))   # launches per stage call (k_morph 2,
                                                     # k_label 3 since the hole-fill / relabel modes were added)
    # steady-state launches only (the script runs the stage 4x after one find_markers pass): take the last ones
    fk = sum(fetch[k][-n_launch:]) * 1024.0          # FETCH_SIZE / WRITE_SIZE are in KiB
    wk = sum(write[k][-n_launch:]) * 1024.0
    corr = 2.0 if k == "k_threshold" else 1.0         # gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide
    per_kernel[k] = {"fetch_bytes_raw": fk, "fetch_correction": corr, "write_bytes": wk,
                     "bytes": fk * corr + wk}         # (16 B/lane) streaming loads; calibrated for k_threshold only
    total += fk * corr + wk
out = {"tag": tag, "frames_per_launch": frames, "stage": "+".join(stage),
       "traffic_bytes_per_launch": total, "traffic_bytes_per_frame": total / frames,
       "per_kernel": per_kernel,
       "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE); k_threshold's FETCH_SIZE doubled per MI355X_MICROARCH.md "
               "(verified: 2 x raw == 2*H*W*frames exactly); the other kernels' narrow / LDS-staged accesses are uncalibrated and "
               "taken at face value"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out)[:400])

# SQ counters of the matrix-core kernels (optional passes of collect.sh): last launch of each kernel
sq = {}
for pat in (f"pmc_sqa_{tag}/**/*counter_collection.csv", f"pmc_sqb_{tag}/**/*counter_collection.csv"):
    f = sorted(glob.glob(os.path.join(G, pat), recursive=True))
    if not f:
        continue
    for r in csv.DictReader(open(f[-1])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        sq.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
if sq:
    for k, c in sq.items():
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            w = c["SQ_WAVE_CYCLES"]                     # quad-cycles, like the WAIT / ACTIVE counters
            c["frac_wait_any"] = round(c.get("SQ_WAIT_ANY", 0) / w, 3)
            c["frac_wait_inst"] = round(c.get("SQ_WAIT_INST_ANY", 0) / w, 3)
            c["frac_active"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / w, 3)
    json.dump({"tag": tag, "frames_per_launch": frames, "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles, "
               "SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the 1024 SIMDs; one launch over `frames_per_launch` frames",
               "kernels": sq}, open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters.json"), "w"), indent=1)
