#!/usr/bin/env python3
"""Benchmark: frames/sec (track -> 3D) at 1280x1024, 169 markers (BASELINE.json).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE config 3/4): `--frames` (default 4096) synthetic 1280x1024 gray uint8 frames PER
GPU, 13x13 dots, resident in HBM before the timed region.  One step = one pass of the hot path over
the rank's batch: fused frames -> [frames, 169, 10] table (blur/DoG -> NCC -> band + open -> CCL +
moments -> ellipse -> match -> track -> undistort + 3-D solve), ONE all-gather of the tables over
RCCL (N > 1), then the last-seen displacement pass on the gathered table.  value = frames of all
ranks / max-over-ranks time.

Extra objects on the JSON line:
  (dtype "u8+f64": uint8 / int32 integer work in the blur (int8 matrix cores, exact) and labelling kernels, float64
   decisions in the NCC (behind a float16-operand / float32-accumulate matrix-core filter), the ellipse fit and the
   3-D solve; tables are stored as float32)
  roofline      threshold+CCL stage (`vbs_marker_center` on uint8 mask + area_mask: k_threshold,
                k_morph x2, k_label, k_finalize), timed live with HIP events on the launch stream
                inside libvbs.  achieved = algorithmic bytes / stage time, algorithmic bytes per
                frame = 2*H*W (the two uint8 images it thresholds) + 24 B per component.
  roofline_mfma k_blur_mfma (int8) and k_ncc_mfma (float16) against the dense matrix-core peaks: algorithmic
                operations of the separable filters / live kernel time (same HIP events)
  kernels       live average ms per launch of every kernel of the fused path (one launch = `batch` frames)
  cpu_baseline  the NumPy/SciPy oracle (oracle/stages.py, a port: the reference needs OpenCV) on the
                box's host cores over a bounded sample of the same frames (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)


def _cpu_worker(args):
    """Oracle over a few frames in one process (cpu_baseline leg only)."""
    import numpy as np
    from oracle import stages as O
    frames, ref, cam = args
    K, dist, R, T = cam
    t0 = time.perf_counter()
    rows = []
    for fc, fr in enumerate(frames):
        mask, area = O.find_markers(fr)
        markers = O.marker_center(mask, area)
        rows.extend(O.track_markers(ref, markers, fc + 1, 20))
    uv = O.undistort_points(np.array([[r["Cx"], r["Cy"]] for r in rows]), K, dist)
    for r, (u, v) in zip(rows, uv):
        try:
            O.calculate_3d_position(np.float64(u), np.float64(v), np.float64(r["major_axis"]), K, R, T)
        except ValueError:
            pass
    return time.perf_counter() - t0, len(frames)


def cpu_baseline(spec, seed, cam, workers, per_worker):
    import numpy as np
    import multiprocessing as mp
    import vbs_amd.synth as S
    from oracle import stages as O
    f0 = S.make_frames(spec, [0], seed=seed)[0]
    m0, a0 = O.find_markers(f0)
    ref = O.process_first_frame(O.marker_center(m0, a0), 5, "full", "optimal")
    frames = S.make_frames(spec, range(1, 1 + workers * per_worker), seed=seed)
    t_single, n_single = _cpu_worker((frames[:2], ref, cam))
    single_fps = n_single / t_single
    chunks = [(frames[i * per_worker:(i + 1) * per_worker], ref, cam) for i in range(workers)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.map(_cpu_worker, chunks)
    wall = time.perf_counter() - t0
    busy = max(r[0] for r in res)
    total = sum(r[1] for r in res)
    return {"value": round(total / busy, 3), "unit": "frames/s", "cores": workers, "kind": "port",
            "sample": f"{total} of the benchmark's 1280x1024 frames, {workers} worker processes x {per_worker} "
                      f"frames, oracle/stages.py end to end (find_markers+marker_center+track+3D); "
                      f"single process: {single_fps:.3f} frames/s; pool wall incl. spawn {wall:.1f}s; "
                      f"host has {os.cpu_count()} logical CPUs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU per step")
    ap.add_argument("--batch", type=int, default=512, help="frames per internal pass (workspace size)")
    ap.add_argument("--roofline-frames", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-workers", type=int, default=8)
    ap.add_argument("--cpu-frames-per-worker", type=int, default=3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--workload", default="c3", choices=["c3", "c5"],
                    help="c3 = BASELINE config 3/4 (1280x1024, 13x13; the headline metric); c5 = config 5 (1920x1200, 21x21, "
                         "adds the plane-fit pose per frame; the JSON line then names that workload)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the real thing) | gloo (rehearsal of N>1 on one GPU)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as td
    import vbs_amd.synth as S
    from vbs_amd import _lib as L
    from vbs_amd import dist as D
    from vbs_amd.engine import Engine
    from vbs_amd.pipeline import reference_from_frame0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()       # rehearsal: ranks may share a device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            td.init_process_group("gloo", rank=rank, world_size=world)
        else:
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    spec = S.config2() if args.workload == "c3" else S.config5()
    H, W, M = spec.height, spec.width, spec.n_markers
    n_local, n_total = args.frames, args.frames * world
    K, dist, R, T = S.default_camera(spec)
    cam = L.make_camera(K, dist, R, T, 2.0)
    eng = Engine(H, W, max_markers=512 if args.workload == "c3" else 1024, max_batch=args.batch, device=local_rank)

    # synthetic frames of this rank's contiguous block, rendered on the device (same bytes as NumPy)
    a, b = D.shard_bounds(n_total, world, rank)
    frames = S.make_frames_torch(spec, range(a, b), seed=args.seed, device=dev, chunk=16)
    f0 = frames[:1] if rank == 0 else None
    ids = xy = None
    if rank == 0:
        ids, xy = reference_from_frame0(eng, f0, 5, "full", "optimal")
    ids, xy = D.broadcast_reference(ids, xy, dev)
    assert len(ids) == M, f"frame 0 gave {len(ids)} IDs, expected {M}"

    def step():
        table, _, counts = eng.track_to_3d(frames, xy, 20.0, cam, 5.0)
        table = D.gather_tables(table, n_total)
        disp = eng.displacement(table, 0, 5.0, 50.0, frame_range=(a, b))    # this rank's frames of the gathered table
        if args.workload == "c5":
            eng.plane_fit(table[a:b])
        return table, disp, counts

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend != "gloo" else "cpu")
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
    table, disp, counts = out
    tracked = int((table[..., 0].int() & 1).sum().item())
    solved = int(((table[..., 0].int() & 2) > 0).sum().item())
    assert int(counts.min().item()) >= 0, "a frame reported a device status"
    assert tracked == n_total * M, f"tracked {tracked} of {n_total * M} marker observations"

    result = None
    if rank == 0:
        fps = n_total * args.steps / elapsed
        result = {
            "metric": f"frames/sec (track->3D) at {W}x{H}, {M} markers", "value": round(fps, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8+f64", "data": "synthetic",
            "config": {"workload": (f"BASELINE config 3/4: {args.frames} synthetic 1280x1024 gray uint8 frames per GPU "
                                    f"(13x13 dots, seeded jitter+noise)" if args.workload == "c3" else
                                    f"BASELINE config 5: {args.frames} synthetic 1920x1200 gray uint8 frames per GPU "
                                    f"(21x21 dots), plus plane-fit pose") +
                                   f", resident in HBM; fused track->3D table + "
                                   f"{'RCCL all-gather + ' if world > 1 else ''}last-seen displacement",
                       "frames_per_gpu": args.frames, "internal_batch": args.batch, "markers": M,
                       "tracked_observations": tracked, "xyz_solved": solved,
                       "us_per_frame_per_gpu": round(1e6 * elapsed / args.steps / args.frames, 2),
                       "whole_path_hbm_frac": round(fps / world * (H * W + M * 40) / 1e9 / HBM_PEAK_GBS, 6)},
        }

    # ---- live per-kernel timing + the threshold+CCL roofline (rank 0; other ranks idle at the barrier) ----
    if rank == 0:
        nk = min(args.roofline_frames, n_local)
        eng.profile(True)
        eng.track_to_3d(frames[:nk], xy, 20.0, cam, 5.0)
        prof = eng.profile_read()
        launches_per = {k: v[0] for k, v in prof.items()}
        kernels = {k: {"launches": c, "avg_ms": round(ms / c, 4), "us_per_frame": round(1e3 * ms / nk, 3)}
                   for k, (c, ms) in prof.items()}
        # stage on uint8 images (the reference's `_marker_center(mask, area_mask)` interface)
        mask, area = eng.find_markers(frames[:nk])
        torch.cuda.synchronize()
        eng.marker_center(mask, area)                       # warm
        eng.profile(True)
        reps = 3
        for _ in range(reps):
            det, cnts = eng.marker_center(mask, area)
        sp = eng.profile_read()
        eng.profile(False)
        stage = ("k_threshold", "k_morph", "k_label", "k_finalize")
        stage_ms = sum(sp[k][1] for k in stage) / reps                      # per nk frames
        n_passes = sp["k_label"][0] / reps                                   # launches of each kernel per call
        ncomp = 2 * M
        alg_bytes_frame = 2 * H * W + ncomp * 24
        achieved = alg_bytes_frame * nk / (stage_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process):
        # the newest profiles/*_pmc_traffic.json, scaled to this run's frames per launch
        traffic = None
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
        if pm:
            pj = json.load(open(pm[-1]))
            traffic = round(pj["traffic_bytes_per_frame"] * nk / n_passes)
        result["roofline"] = {
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "kernel": "+".join(stage), "stage": "threshold+CCL (vbs_marker_center on uint8 mask+area_mask)",
            "algorithmic_bytes_per_frame": alg_bytes_frame, "frames_per_launch": round(nk / n_passes, 1),
            "stage_ms_per_launch": round(stage_ms / n_passes, 4),
            "per_kernel_avg_ms": {k: round(sp[k][1] / sp[k][0], 4) for k in stage},
            "us_per_frame": round(1e3 * stage_ms / nk, 3)}
        result["kernels"] = kernels
        # the two matrix-core kernels of the front end against the dense MFMA peaks (MI355X_MICROARCH.md: bf16/f16
        # ~2.5 PFLOP/s, int8 2x that); algorithmic operations = the separable filters as written in the reference
        # (Toeplitz padding, hi/lo splits and the count product are overhead, not counted)
        small = H <= 480
        ta, tb, ln = (21, 35, 33) if small else (39, 101, 80)
        mf = []
        for name, ops_px, peak, dt in (("k_blur_mfma", 2 * 2 * (ta + tb), 5000.0, "i8"),
                                       ("k_ncc_mfma", 2 * 2 * ln, 2500.0, "f16")):
            if name in prof:
                c, ms = prof[name]
                ach = ops_px * H * W * nk / (ms * 1e-3) / 1e12
                mf.append({"kernel": name, "bound": "mfma", "dtype": dt, "achieved": round(ach, 2), "peak": peak,
                           "unit": "TOP/s" if dt == "i8" else "TFLOP/s", "frac": round(ach / peak, 5),
                           "algorithmic_ops_per_pixel": ops_px, "avg_ms": round(ms / c, 4),
                           "frames_per_launch": round(nk / c, 1)})
        result["roofline_mfma"] = mf
        del launches_per
    if world > 1:
        td.barrier()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(spec, args.seed, (K, dist, R, T), args.cpu_workers,
                                                  args.cpu_frames_per_worker)
        except Exception as e:                                  # the baseline must not sink the GPU number
            result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port",
                                      "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
