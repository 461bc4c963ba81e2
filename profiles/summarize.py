#!/usr/bin/env python3
"""Turn the raw rocprofv3 output merged back under gpurun_out/ into the small files kept in profiles/:
   <tag>_rocprofv3_kernel_stats.csv     rows of this repo's kernels from `--kernel-trace --stats` (rocprofv3's own table: it
                                        averages EVERY call of a name, one-frame launches included)
   <tag>_kernel_trace_by_shape.json     the same kernels from the per-dispatch trace, averaged over launches of the benchmark
                                        shape only (the numbers DESIGN.md quotes)
   <tag>_pmc_traffic_<workload>.json    HBM traffic per launch of the threshold+CCL stage of the FUSED path from the
                                        FETCH_SIZE / WRITE_SIZE passes (profiles/collect.sh), workload c3 / c5
   <tag>_sq_counters.json               SQ counters of the matrix-core kernels
usage: python profiles/summarize.py <tag> [frames_per_launch=512]"""
import csv, glob, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 512
G = os.path.join(ROOT, "gpurun_out")


def one(pattern, required=True):
    f = sorted(glob.glob(os.path.join(G, pattern), recursive=True))
    if not f:
        if required:
            raise SystemExit(f"missing {pattern}")
        return None
    return f[-1]


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


def kfull(s):
    """The kernel with its template arguments (`k_blur16<false, false, false>`): the by-shape table keys on THIS, so that
    the product instantiation is not averaged with the uint8-output one of the staged entry (same grid, 9 % slower: round
    4's table quoted 1.27 / 2.11 us for kernels that ran at 1.17 / 2.02)."""
    k = s.split("(")[0].replace("void ", "").strip()
    if k.startswith("k_ccl<"):
        return "k_ccl_band" if k.startswith("k_ccl<0") else "k_ccl_open"
    return k


rows = list(csv.reader(open(one(f"prof_{tag}/**/*kernel_stats.csv"))))
with open(os.path.join(ROOT, "profiles", f"{tag}_rocprofv3_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:]:
        if r[0].startswith(("void k_", "k_")):
            w.writerow(r)

# Per-kernel averages over launches of ONE shape only.  rocprofv3's kernel_stats.csv averages every call of a name: the bench
# also launches each kernel once on a single frame (frame 0's reference table), and that 15 us call in an average of eleven
# 500 us calls made round 3's quoted per-frame times 9 % too low.  From the kernel TRACE: calls are
# grouped by grid size, the group with the largest total duration is the benchmark shape, and `us_per_frame` divides ITS
# average by the frames one such launch covers.
trace = one(f"prof_{tag}/**/*kernel_trace.csv", required=False)
if trace:
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(trace)):
        k = kfull(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        grid = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        by[k][grid].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    shapes = {}
    for k, groups in sorted(by.items()):
        grid, d = max(groups.items(), key=lambda kv: sum(kv[1]))
        d_sorted = sorted(d)
        shapes[k] = {"grid": list(grid), "calls": len(d), "avg_ns": round(sum(d) / len(d), 1), "min_ns": d_sorted[0],
                     "median_ns": d_sorted[len(d) // 2], "max_ns": d_sorted[-1],
                     "us_per_frame": round(sum(d) / len(d) / frames / 1e3, 4),
                     "other_shapes": {"x".join(map(str, g)): {"calls": len(v), "avg_ns": round(sum(v) / len(v), 1)}
                                      for g, v in groups.items() if g != grid},
                     "all_calls_avg_ns": round(sum(sum(v) for v in groups.values()) / sum(len(v) for v in groups.values()), 1)}
    json.dump({"tag": tag, "frames_per_launch": frames, "source": "rocprofv3 --kernel-trace, calls grouped by grid size; the "
               "group with the largest total duration per kernel INSTANTIATION (template arguments kept) = the benchmark shape",
               "kernels": shapes},
              open(os.path.join(ROOT, "profiles", f"{tag}_kernel_trace_by_shape.json"), "w"), indent=1)
    for k, v in shapes.items():
        print(f"{k:44s} {v['calls']:4d} launches of grid {v['grid']}: {v['us_per_frame']:.3f} us/frame "
              f"(all {sum(len(x) for x in by[k].values())} calls averaged: {v['all_calls_avg_ns'] / frames / 1e3:.3f})")


def counter(pattern):
    agg = collections.defaultdict(list)
    p = one(pattern, required=False)
    if p is None:
        return None
    namer = RetryNamer()
    for r in sorted(csv.DictReader(open(p)), key=lambda r: int(r["Dispatch_Id"])):
        k = namer(r)
        if k.startswith("k_"):
            agg[k].append(float(r["Counter_Value"]))
    return agg


alg = {"c3": 1280 * 1024 + 24 * 169, "c5": 1920 * 1200 + 24 * 441}
for wl in ("c3", "c5"):
    fetch, write = counter(f"pmc_fetch_{tag}_{wl}/**/*counter_collection.csv"), counter(f"pmc_write_{tag}_{wl}/**/*counter_collection.csv")
    if not fetch or not write:
        continue
    per_kernel, total = {}, 0.0
    for k in sorted(fetch):
        # the program runs the fused path twice over `frames` frames in ONE internal pass: the last launch of every kernel
        fk = fetch[k][-1] * 1024.0                      # FETCH_SIZE / WRITE_SIZE are in KiB
        wk = write[k][-1] * 1024.0
        per_kernel[k] = {"fetch_bytes": fk, "write_bytes": wk, "bytes": fk + wk}
        total += fk + wk
    out = {"tag": tag, "workload": wl, "frames_per_launch": frames, "stage": "+".join(sorted(per_kernel)),
           "traffic_bytes_per_launch": total, "traffic_bytes_per_frame": total / frames,
           "algorithmic_bytes_per_frame": alg[wl], "traffic_over_algorithmic": round(total / frames / alg[wl], 3),
           "per_kernel": per_kernel,
           "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over the fused path, stage kernels only. gfx950 counts a wide "
                   "(16 B / lane) streaming read at half its bytes (MI355X_MICROARCH.md); these kernels read the bit images "
                   "with 8-byte loads, an access width the guide leaves uncalibrated: the counters are taken at face value, "
                   "and doubling every FETCH_SIZE gives the upper bound"}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_{wl}.json"), "w"), indent=1)
    print(wl, "traffic/frame", round(total / frames), "=", out["traffic_over_algorithmic"], "x algorithmic")

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
