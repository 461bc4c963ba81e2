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
  roofline      threshold+CCL stage AS IT RUNS ON THE TIMED PATH (bit-packed masks in -> detections out: k_morph,
                k_ccl_band, k_ccl_open, k_slow_list, k_label*, k_probe_slow, k_finalize), timed live with HIP events
                on the launch stream inside libvbs.  achieved = SURVEY 8(d) algorithmic bytes (H*W + 24 B per label,
                per frame) x frames per launch / sum of the kernels' average launch durations.  `staged_u8` repeats it
                for the staged entry `vbs_marker_center` on two uint8 images (k_threshold + the same kernels), with the
                one-image (8d) and the two-image numerators.  `traffic` = HBM bytes per launch from the newest
                profiles/*_pmc_traffic_<workload>.json (separate rocprofv3 --pmc passes), null when there is none.
                `band_stage` = the 8(d) stage proper (band half of the kernel, from the committed phase log), `valu` = the
                vector-issue share of the SIMDs' cycles in k_stage (committed SQ counters): the yardstick of a kernel that
                reads 0.3 x its algorithmic bytes.
                The per-kernel leg runs every pass on ONE stream with an event pair around each launch (>= 8 launches per
                kernel after a warm-up call), the timed region runs the passes on TWO streams: the sum of the kernels' times
                per frame may therefore exceed `us_per_frame_per_gpu`, and `roofline.frac` is the pessimistic reading of the
                two.  `band_stage` is MEASURED IN THIS RUN (`measured_in_this_run`): a child process (fresh interpreter, the
                debug library with its stop-after-the-band-half knob, 512 frames) started once the parent's GPU work is
                done.  `valu` comes from committed counters and says whether the kernel's sources are still the ones they
                were taken on (`same_sources`).
  roofline_mfma k_blur16 / k_blur_mfma (int8) and k_ncc_mfma (float16) against the dense matrix-core peaks: algorithmic
                operations of the separable filters / live kernel time (same HIP events)
  kernels       live average ms per launch of every kernel of the fused path (one launch = `batch` frames)
  cpu_baseline  BASELINE.md 3: the NumPy/SciPy oracle (oracle/stages.py, a port: the reference needs OpenCV) on the
                box's host cores over a bounded sample of the same frames (rank 0, N = 1 only): single process over
                >= 32 frames after 2 warm-ups, and one worker process per core this job can really keep busy: the
                smallest of physical cores, affinity mask, cgroup quota / cpuset and a MEASURED count (1, 2, 4, ... forked
                busy loops until the aggregate rate stops growing: a shared box schedules fewer CPUs than its mask shows).
                `cores` = the workers that ran; `oversubscribed` = a worker ran below half the single-process rate.
  (A utilisation sampler with a period of seconds can show the GPU idle throughout: the timed region is ~0.3 s of a run
   that spends two minutes rendering frames, in the CPU baseline and in the host-path leg.)
  config        N > 1 self-validation: device_uuids (all-gathered, one per rank) / distinct_devices, rccl_version,
                gathered_table_checksum (every rank's gathered table reduced to two integers, all-gathered and ASSERTED
                equal on every rank), pipelined_gather; frame0_ids = the device ID assignment and its host checker.
                also: bgr_fps (the same workload fed as 3-channel BGR frames, the reference's input format),
                host_path_fps (NumPy frames in host memory -> CSV on disk through MarkerTracker), avi_path_fps (a 640x480
                Motion-JPEG AVI file -> CSV through MarkerTracker.process(): native decoder and Pillow), the NCC decision
                counters of the timed batch, world size / backend as torch.distributed reports them;
                single_frame_us (BASELINE config 2, the way MarkerTracker.process calls: ONE resident frame per call, track ->
                3-D, call + stream synchronise; `eager` and `graph_replay` = the same call captured once into a HIP graph;
                `per_kernel_us` by HIP events).
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)
STAGE = ("k_stage", "k_stage_retry", "k_morph", "k_ccl_band", "k_ccl_open", "k_slow_list", "k_label", "k_label_fill", "k_label_redo", "k_probe_slow",
         "k_finalize")


def _newest(pattern):
    """profiles/ files are named <round tag>_...: r1a < r1b < ... < r3pre < r3a < ... < r4a; the newest one matching."""
    import re

    def key(path):
        m = re.match(r"r(\d+)([a-z]*)_", os.path.basename(path))
        if not m:
            return (-1, 0, "")
        return (int(m.group(1)), 0 if m.group(2).startswith("pre") else 1, m.group(2))
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), key=key)
    return files[-1] if files else None


def _sources_sha16(names=("k_stage.hip", "stage_common.h", "ccl_common.h")):
    """Fingerprint of the labelling kernel's sources: committed counter files carry it, so that the line can say whether a
    number taken from them belongs to the kernel that ran."""
    import hashlib
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, "vision-basedsensor_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _band_stage_live(workload, frames=512, timeout=240):
    """us per frame of k_stage stopped after the band half's sums are out (band + 4-connected labels + count /
    sum x / sum y per label), measured NOW: tools/gpu_stage_phase.py as a child process (a fresh interpreter that loads
    libvbs_dbg.so and reads VBS_STAGE_STOP=10; the product library has no such knob).  None + reason when it cannot run."""
    import subprocess
    tool = os.path.join(ROOT, "tools", "gpu_stage_phase.py")
    dbg = os.path.join(ROOT, "vision-basedsensor_amd", "csrc", "libvbs_dbg.so")
    if not os.path.exists(dbg):
        return None, "libvbs_dbg.so is not built (__graft_entry__.build())"
    out = {}
    for stop in (10, 0):
        try:
            r = subprocess.run([sys.executable, tool, str(frames), workload, "child"], env=dict(os.environ, VBS_STAGE_STOP=str(stop)),
                               capture_output=True, text=True, timeout=timeout)
        except Exception as e:
            return None, f"{type(e).__name__}: {e}"
        if r.returncode != 0:
            return None, (r.stderr or r.stdout).strip()[-300:]
        try:
            out[stop] = json.loads(r.stdout.strip().splitlines()[-1])["k_stage"]
        except Exception as e:
            return None, f"unparsable child output: {r.stdout[-200:]!r}"
    return out, None


def _cpu_worker(args):
    """Oracle over a few frames in one process (cpu_baseline leg only); the first `warm` frames are not timed."""
    import numpy as np
    from oracle import stages as O
    frames, ref, cam, warm = args
    K, dist, R, T = cam
    t0 = time.perf_counter()
    rows, n = [], 0
    for fc, fr in enumerate(frames):
        if fc == warm:
            t0 = time.perf_counter()
            rows = []
        mask, area = O.find_markers(fr)
        markers = O.marker_center(mask, area)
        rows.extend(O.track_markers(ref, markers, fc + 1, 20))
        n += fc >= warm
    uv = O.undistort_points(np.array([[r["Cx"], r["Cy"]] for r in rows]), K, dist)
    for r, (u, v) in zip(rows, uv):
        try:
            O.calculate_3d_position(np.float64(u), np.float64(v), np.float64(r["major_axis"]), K, R, T)
        except ValueError:
            pass
    return time.perf_counter() - t0, n


def _host_cpus():
    """(logical CPUs, physical cores, CPUs this process may run on, model name) of the host."""
    phys, model = set(), ""
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return os.cpu_count() or 1, len(phys) or (os.cpu_count() or 1), usable, model


def _cgroup_quota_cpus():
    """CPUs the cgroup CPU controller grants this process (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1), and
    the size of its effective cpuset; None where nothing is set or readable."""
    quota = cpuset = None
    paths = {}
    try:
        for line in open("/proc/self/cgroup"):
            _, ctrl, path = line.rstrip("\n").split(":", 2)
            for c in ctrl.split(","):
                paths[c] = path
    except OSError:
        pass

    def first(cands):
        for c in cands:
            try:
                return open(c).read().strip()
            except OSError:
                continue
        return None

    v2 = paths.get("", "/")
    t = first([f"/sys/fs/cgroup{v2.rstrip('/')}/cpu.max", "/sys/fs/cgroup/cpu.max"])
    if t:
        q, _, per = t.partition(" ")
        if q != "max" and per:
            quota = int(q) / int(per)
    if quota is None:
        v1 = paths.get("cpu", "/").rstrip("/")
        q = first([f"/sys/fs/cgroup/cpu{v1}/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"])
        per = first([f"/sys/fs/cgroup/cpu{v1}/cpu.cfs_period_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"])
        if q and per and int(q) > 0:
            quota = int(q) / int(per)
    cs = paths.get("cpuset", "/").rstrip("/")
    t = first([f"/sys/fs/cgroup{v2.rstrip('/')}/cpuset.cpus.effective", "/sys/fs/cgroup/cpuset.cpus.effective",
               f"/sys/fs/cgroup/cpuset{cs}/cpuset.effective_cpus", "/sys/fs/cgroup/cpuset/cpuset.effective_cpus"])
    if t:
        try:
            cpuset = sum((int(b) - int(a) + 1) for a, _, b in (r.partition("-") for r in (x if "-" in x else f"{x}-{x}"
                                                                                      for x in t.split(","))))
        except ValueError:
            cpuset = None
    return quota, cpuset


def _spin(seconds):
    t_end = time.perf_counter() + seconds
    n = 0
    while time.perf_counter() < t_end:
        for _ in range(2000):
            n += 1
    return n


def probe_parallel_cpus(limit, seconds=0.25):
    """How many CPUs this process can really keep busy at once: fork 1, 2, 4, ... busy loops for `seconds` each and take
    the best aggregate rate over the one-process rate.  A box that shows 128 CPUs in its affinity mask but schedules 16 of
    them (a quota the cgroup files do not always show) saturates here at 16.  Called BEFORE anything touches the GPU
    (fork), takes about 2 s."""
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    rates, n = {}, 1
    while n <= max(1, limit):
        with ctx.Pool(n) as pool:
            rates[n] = sum(pool.map(_spin, [seconds] * n)) / seconds
        if n > 1 and rates[n] < 1.10 * rates[n // 2] and rates[n // 2] < 1.10 * rates.get(n // 4, 0.0):
            break                                          # two doublings without gain: saturated
        n *= 2
    best = max(rates.values())
    return {"effective_cpus": round(best / rates[1], 1), "rate_vs_one_process": {str(k): round(v / rates[1], 2) for k, v in rates.items()}}


def cpu_baseline(spec, seed, cam, single_frames, max_workers, per_worker, warm=2, probe=None):
    """(i) single process, default FFT/BLAS threading, fps over >= 32 frames after 2 warm-ups;
    (ii) one worker process per physical core over disjoint frame ranges: min(physical cores, CPUs this process may run
    on) workers; `max_workers` > 0 caps that (a shared box that grants fewer CPUs than its affinity mask shows), and the
    record then says how many of the available cores were used."""
    import multiprocessing as mp
    import vbs_amd.synth as S
    from oracle import stages as O
    logical, physical, usable, model = _host_cpus()
    quota, cpuset = _cgroup_quota_cpus()
    limits = {"physical_cores": physical, "affinity_mask": usable}
    if quota:
        limits["cgroup_quota"] = max(1, int(quota))
    if cpuset:
        limits["cgroup_cpuset"] = cpuset
    if probe:
        limits["measured_parallel_cpus"] = max(1, int(probe["effective_cpus"] + 0.5))
    available = max(1, min(limits.values()))
    workers = min(available, max_workers) if max_workers > 0 else available
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):       # one worker = one core
        os.environ.setdefault(var, "1")
    f0 = S.make_frames(spec, [0], seed=seed)[0]
    m0, a0 = O.find_markers(f0)
    ref = O.process_first_frame(O.marker_center(m0, a0), 5, "full", "optimal")
    nfr = max(single_frames + warm, workers * (per_worker + warm))
    frames = S.make_frames(spec, range(1, 1 + nfr), seed=seed)
    t_single, n_single = _cpu_worker((frames[:single_frames + warm], ref, cam, warm))
    single_fps = n_single / t_single
    step = per_worker + warm
    chunks = [(frames[i * step:(i + 1) * step], ref, cam, warm) for i in range(workers)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.map(_cpu_worker, chunks)
    wall = time.perf_counter() - t0
    busy = max(r[0] for r in res)
    total = sum(r[1] for r in res)
    value = total / busy
    per_worker_fps = value / workers
    return {"value": round(value, 3), "unit": "frames/s", "cores": workers, "cores_available": available, "kind": "port",
            "per_worker_fps": round(per_worker_fps, 3),
            # a worker that runs at less than half the single-process rate did not have a core to itself
            "oversubscribed": bool(per_worker_fps < 0.5 * single_fps),
            "quota_cpus": quota, "cpu_limits": limits, "parallel_probe": probe,
            "single_process": {"value": round(single_fps, 3), "frames": n_single, "warmup_frames": warm,
                               "threads": "default FFT/BLAS threading"},
            "host": {"logical_cpus": logical, "physical_cores": physical, "usable_cpus": usable, "model": model},
            "sample": f"oracle/stages.py end to end (find_markers+marker_center+track+3D) on the benchmark's "
                      f"{spec.width}x{spec.height} frames: single process {n_single} frames after {warm} warm-ups = "
                      f"{single_fps:.3f} frames/s; {workers} single-threaded worker processes = the smallest of "
                      f"{limits} ({logical} logical CPUs"
                      f"{'; capped by --cpu-workers' if workers < available else ''}) x {per_worker} frames after "
                      f"{warm} warm-ups each = value; pool wall incl. spawn {wall:.1f}s"}


def single_frame_us(spec, seed, cam_params, reps=200):
    """One frame per call (marker_detection.py:434-453): latency of vbs_track_to_3d on one resident frame, stream synchronised."""
    import torch
    import vbs_amd.synth as S
    from vbs_amd import _lib as L
    from vbs_amd.engine import Engine
    from vbs_amd.pipeline import reference_from_frame0
    eng = Engine(spec.height, spec.width, max_markers=1024 if spec.n_markers > 400 else 512, max_batch=1)
    ft = S.make_frames_torch(spec, range(8), seed=seed, device="cuda")
    cam = L.make_camera(*cam_params, 2.0)
    ids, xy = reference_from_frame0(eng, ft[:1], 5, "full", "optimal")
    xy_d = torch.as_tensor(xy, dtype=torch.float64, device="cuda")
    for i in range(5):
        eng.track_to_3d(ft[i % 8:i % 8 + 1], xy_d, 20.0, cam, 5.0)
    torch.cuda.synchronize()
    ts = []
    for i in range(reps):
        t0 = time.perf_counter()
        table, _, _ = eng.track_to_3d(ft[i % 8:i % 8 + 1], xy_d, 20.0, cam, 5.0)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    tracked = int((table[..., 0].int() & 1).sum())
    eng.profile(True)
    for i in range(20):
        eng.track_to_3d(ft[i % 8:i % 8 + 1], xy_d, 20.0, cam, 5.0)
    torch.cuda.synchronize()
    per = {k: round(1e3 * v[1] / v[0], 1) for k, v in eng.profile_read().items()}
    eng.profile(False)
    out = {"eager": round(1e6 * sorted(ts)[len(ts) // 2], 1), "eager_mean": round(1e6 * sum(ts) / len(ts), 1), "per_kernel_us": per,
           "tracked": tracked, "markers": len(ids), "frame": f"{spec.width}x{spec.height}"}
    try:                                                    # the same call as a HIP graph: copy the frame in, replay, synchronise
        buf = ft[:1].clone()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                gt, _, _ = eng.track_to_3d(buf, xy_d, 20.0, cam, 5.0)
        torch.cuda.current_stream().wait_stream(side)
        for i in range(5):
            buf.copy_(ft[i % 8:i % 8 + 1]); g.replay()
        torch.cuda.synchronize()
        tg = []
        for i in range(reps):
            t0 = time.perf_counter()
            buf.copy_(ft[i % 8:i % 8 + 1]); g.replay()
            torch.cuda.synchronize()
            tg.append(time.perf_counter() - t0)
        ref, _, _ = eng.track_to_3d(ft[(reps - 1) % 8:(reps - 1) % 8 + 1], xy_d, 20.0, cam, 5.0)
        torch.cuda.synchronize()
        out["graph_replay"] = round(1e6 * sorted(tg)[len(tg) // 2], 1)
        out["graph_equals_eager"] = bool(torch.equal(ref, gt))
    except Exception as e:
        out["graph_replay"] = None
        out["graph_error"] = f"{type(e).__name__}: {e}"
    eng.close()
    return out


def host_path_fps(spec, n, seed, batch):
    """The drop-in's own rate from frames in HOST memory to the CSV on disk (upload over PCIe, fused kernels, row
    building, CSV writer): `MarkerTracker.process_frames` + `_save_results`.  Twice: frames in page-locked memory
    (`marker_detection.pinned_frames`, what a decoder that feeds this path should fill: batch k + 1 is uploaded by DMA
    under batch k's kernels and row building) = `fps`, and the same frames in ordinary pageable memory = `pageable`."""
    import tempfile
    import vbs_amd.synth as S
    from vbs_amd.marker_detection import MarkerTracker, pinned_frames
    import contextlib
    frames = S.make_frames_torch(spec, range(n), seed=seed, device="cuda", chunk=64).cpu().numpy()    # (same bytes as make_frames)
    pinned = pinned_frames(frames.shape)
    pinned[:] = frames
    # (the drop-in prints progress lines like the reference does: keep stdout for the ONE JSON line)
    with tempfile.TemporaryDirectory() as td_, contextlib.redirect_stdout(sys.stderr):
        clip = os.path.join(td_, "clip.npy")
        open(clip, "wb").close()                           # `video_path` must exist; frames are passed in memory
        out, texts = {}, {}
        for name, src, reps in (("pageable", frames, 2), ("pinned", pinned, 3)):
            for rep in range(reps):                        # first repetition warms the engine / allocator
                trk = MarkerTracker({"video_path": clip, "output_dir": os.path.join(td_, f"o{name}{rep}"),
                                     "crop_ratios": (0, 0, 0, 0), "id_mode": "full", "batch": batch})
                t0 = time.perf_counter()
                rows = trk.process_frames(src)
                t1 = time.perf_counter()
                trk._save_results(rows)
                t2 = time.perf_counter()
                out[name] = {"fps": round(n / (t2 - t0), 1), "fps_without_csv_write": round(n / (t1 - t0), 1)}
                texts[name] = open(trk.output_csv, "rb").read()
        assert texts["pinned"] == texts["pageable"], "the CSV must not depend on where the frames lie"
    return {"frames": n, "fps": out["pinned"]["fps"], "fps_without_csv_write": out["pinned"]["fps_without_csv_write"],
            "frames_memory": "page-locked (marker_detection.pinned_frames); upload of batch k+1 overlaps batch k",
            "pageable": out["pageable"], "csv_rows": len(rows), "csv_bytes": len(texts["pinned"]),
            "csv_identical_pinned_vs_pageable": True, "batch": batch,
            "pcie_bound_fps": "about 42 000 gray 1280x1024 frames/s at ~55 GB/s"}


def avi_path_fps(n, seed, batch=256):
    """Row f4: the reference's real input - a Motion-JPEG AVI in the camera's format (640x480, quality 70,
    collecting.py:100) - from the FILE to the CSV through `MarkerTracker.process()` with the reference's default crop.
    Twice: the native decoder (Huffman on C++ threads, IDCT + colour on the device) and Pillow's thread pool; the two CSV
    files must be the same bytes.  Encoding the clip (Pillow, ~2 ms per frame) is outside the timed calls."""
    import contextlib
    import tempfile
    import numpy as np
    import vbs_amd.synth as S
    from vbs_amd.marker_detection import MarkerTracker
    from vbs_amd.video_io import write_avi
    spec = S.config1()
    base = S.make_frames(spec, range(min(n, 64)), seed=seed, channels=3)
    frames = np.concatenate([base] * ((n + len(base) - 1) // len(base)))[:n]
    with tempfile.TemporaryDirectory() as td_, contextlib.redirect_stdout(sys.stderr):
        path = os.path.join(td_, "clip.avi")
        write_avi(path, frames, fps=12.0, codec="MJPG", quality=70)
        short = os.path.join(td_, "short.avi")             # (the slow path on a prefix of the clip: same rate, less waiting)
        write_avi(short, frames[:min(n, 512)], fps=12.0, codec="MJPG", quality=70)
        out, texts = {"frames": n, "frame": "640x480 BGR, MJPG quality 70", "bytes_per_frame": round(os.path.getsize(path) / n),
                      "batch": batch, "crop": "(1/8, 1/8, 1/16, 0)"}, {}
        for name, on_dev, clip, nn in (("device", True, path, n), ("pillow", False, short, min(n, 512))):
            for rep in range(2):                           # first repetition warms the engine / allocator
                trk = MarkerTracker({"video_path": clip, "output_dir": os.path.join(td_, f"o{name}{rep}"),
                                     "crop_ratios": (1 / 8, 1 / 8, 1 / 16, 0), "id_mode": "full", "batch": batch,
                                     "mjpeg_on_device": on_dev})
                t0 = time.perf_counter()
                trk.process()
                dt = time.perf_counter() - t0
            assert trk.decode_path == name, (trk.decode_path, name)
            out[name + "_fps"] = round(nn / dt, 1)
            texts[name] = open(trk.output_csv, "rb").read()
        assert texts["device"].startswith(texts["pillow"]), "the CSV must not depend on the decoder"
        out["fps"] = out["device_fps"]
        out["csv_identical_device_vs_pillow"] = True
        out["csv_bytes"] = len(texts["device"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU per step")
    ap.add_argument("--stage-impl", type=int, default=0, help="VBS_OPT_STAGE_IMPL for the timed engine (A/B of the labelling kernel's shapes: 3 = 768 threads per frame always, 4 = 256 wherever the geometry allows; results identical)")
    ap.add_argument("--batch", type=int, default=1536, help="frames per internal pass (workspace size).  1536 = six frames per CU for the labelling kernel's 256-thread instance (three workgroups per CU): 512 / 768 / 1024 / 1536 / 2048 ran at 290.6 / 291.4 / 293.1 / 294.0 / 292.7 k frames/s in one sweep (profiles/r5o_stage256_large_ab3.log); rounds 2-5 quoted 512")
    ap.add_argument("--roofline-frames", type=int, default=0, help="frames of the per-kernel leg; 0 = two internal passes (2 x --batch): every kernel is timed at the pass size of the timed region")
    ap.add_argument("--pass-streams", type=int, default=2, choices=[1, 2],
                    help="VBS_OPT_PASS_STREAMS: 2 = odd internal passes on a second workspace and stream (the library's default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the BGR and host-path side measurements")
    ap.add_argument("--cpu-single-frames", type=int, default=32)
    ap.add_argument("--cpu-workers", type=int, default=0,
                    help="0 (default) = one worker per physical core in the affinity mask (BASELINE.md 3); > 0 caps the workers")
    ap.add_argument("--cpu-frames-per-worker", type=int, default=3)
    ap.add_argument("--host-frames", type=int, default=1024)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--workload", default="c3", choices=["c3", "c5", "c1", "real"],
                    help="c3 = BASELINE config 3/4 (1280x1024, 13x13; the headline metric); c5 = config 5 (1920x1200, 21x21, "
                         "adds the plane-fit pose per frame; the JSON line then names that workload); c1 = the reference's REAL "
                         "configuration: 640x480 BGR frames (collecting.py:29-31) through its default crop (1/8, 1/8, 1/16, 0) "
                         "(marker_detection.py:481) = 480 wide x 450 high, small branch, 7x7 dots; --frames defaults to 16384; "
                         "real = the reference's REAL LAYOUT AND TEXTURE: its one published frame (img/raw_markers.png -> "
                         "tests/golden/raw_markers_bgr.npz, 467x437 BGR, 65 dots of ~27 px at a pitch of 35-42 px; put back into the "
                         "480x450 crop frame it was cut from by repeating its edge pixels) expanded on "
                         "the device to --frames (default 16384) frames by seeded shifts of +-3 px and noise sigma 2; the line "
                         "adds the labelling path every frame took (`slow_path_frames`, reasons) and the rate of the general "
                         "labelling kernel on the same frames")
    ap.add_argument("--channels", type=int, default=1, choices=[1, 3],
                    help="1 = gray frames (headline); 3 = the whole benchmark on BGR frames (the line then says so)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the real thing) | gloo (rehearsal of N>1 on one GPU)")
    ap.add_argument("--pipelined", type=int, default=1, choices=[0, 1],
                    help="N > 1: 1 = one all-gather per internal pass pair, overlapping the next passes (dist.TableGather); "
                         "0 = every pass first, then the SINGLE all-gather of SURVEY 8(e) (dist.gather_tables)")
    args = ap.parse_args()

    # the parallel-CPU probe of the cpu_baseline leg forks: it runs before anything touches the GPU
    cpu_probe = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_cpu_baseline:
        try:
            cpu_probe = probe_parallel_cpus(_host_cpus()[2])
        except Exception as e:
            cpu_probe = None
            print(f"parallel-CPU probe failed: {type(e).__name__}: {e}", file=sys.stderr)

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

    real_bgr = None
    if args.workload == "real":
        real_bgr = np.load(os.path.join(ROOT, "tests", "golden", "raw_markers_bgr.npz"))["bgr"]
        # the still is a 467x437 window of the reference's 480x450 crop frame (marker_detection.py:481), 11 px from its left
        # and 12 px from its top edge (the similarity fit of its markers to the reference's figure, tests/figure_check.py):
        # put back into that frame by repeating its edge pixels, so that the workload has the camera's real geometry
        real_bgr = np.pad(real_bgr, ((12, 1), (11, 2), (0, 0)), mode="edge")
        assert real_bgr.shape == (450, 480, 3)
        spec = S.FrameSpec(int(real_bgr.shape[1]), int(real_bgr.shape[0]), np.zeros((65, 2), np.int64), 27 * 16, name="real")
        args.channels = 3
        if args.frames == 4096:
            args.frames = 16384
    else:
        spec = {"c3": S.config2, "c5": S.config5, "c1": S.config1}[args.workload]()
    crop = None
    if args.workload == "c1":
        from vbs_amd.marker_detection import _crop_box
        crop = _crop_box(spec.width, spec.height, (1 / 8, 1 / 8, 1 / 16, 0))         # left, right, top, bottom
        args.channels = 3
        if args.frames == 4096:
            args.frames = 16384
    H, W, M = spec.height, spec.width, spec.n_markers
    if crop:
        H, W = crop[3] - crop[2], crop[1] - crop[0]
    n_local, n_total = args.frames, args.frames * world
    K, dist, R, T = S.default_camera(spec)
    cam = L.make_camera(K, dist, R, T, 2.0)
    eng = Engine(H, W, max_markers={"c3": 512, "c5": 1024, "c1": 256, "real": 256}[args.workload], max_batch=args.batch, device=local_rank)
    eng.set_option(L.OPT_PASS_STREAMS, args.pass_streams)
    if args.stage_impl:
        eng.set_option(L.OPT_STAGE_IMPL, args.stage_impl)

    # synthetic frames of this rank's contiguous block, rendered on the device (same bytes as NumPy)
    a, b = D.shard_bounds(n_total, world, rank)
    if real_bgr is not None:
        # this rank's block of the sequence (frame 0 = the still itself, on rank 0)
        allf, shifts = S.jittered_copies_torch(real_bgr, b - a, seed=args.seed, device=dev, start=a, total=n_total)
        gray = None
    else:
        gray = S.make_frames_torch(spec, range(a, b), seed=args.seed, device=dev, chunk=16)

    def as_bgr(g):
        return g.unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()

    if real_bgr is not None:
        frames = allf
    elif crop:
        # rendered as BGR in slices (the full-size gray copy and its BGR copy would not both be needed at once); what the
        # engine sees is the reference's cropped view: a pointer offset and the full frames' strides, no copy
        full = torch.empty((gray.shape[0], spec.height, spec.width, 3), dtype=torch.uint8, device=dev)
        for s0 in range(0, gray.shape[0], 1024):
            full[s0:s0 + 1024] = gray[s0:s0 + 1024].unsqueeze(-1)
        del gray
        frames = full[:, crop[2]:crop[3], crop[0]:crop[1], :]
    else:
        frames = gray if args.channels == 1 else as_bgr(gray)
        if args.channels == 3:
            del gray
    ids = xy = None
    id_check = None
    if rank == 0:
        # frame-0 identities on the device (vbs_assign_ids) with the host assignment as the checker: see pipeline.py
        import vbs_amd.pipeline as P
        ids, xy = reference_from_frame0(eng, frames[:1], 5, "full", "optimal", ids_on_device=True)
        id_check = dict(P.ID_CHECK)
    ids, xy = D.broadcast_reference(ids, xy, dev)
    assert len(ids) == M, f"frame 0 gave {len(ids)} IDs, expected {M}"

    # N > 1: the run validates itself.  Every rank reports the device it computes on (N distinct devices or the line says
    # so), and after the timed region the gathered table of every rank is reduced to a checksum that all ranks must share.
    dev_ids, rccl = None, None
    if world > 1:
        pr = torch.cuda.get_device_properties(local_rank)
        mine = str(getattr(pr, "uuid", "")) or f"{getattr(pr, 'pci_bus_id', '?')}:{getattr(pr, 'pci_device_id', '?')}"
        mine = f"{mine}|{pr.name}|local_rank {local_rank}"
        got = [None] * world
        td.all_gather_object(got, mine)
        dev_ids = got
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:                                 # (not fatal: the line then says what failed)
            rccl = f"unavailable: {type(e).__name__}"

    from vbs_amd.pipeline import track_and_gather

    def step(fr):
        _, counts, table = track_and_gather(eng, fr, n_total, xy, 20.0, cam, 5.0,     # all-gather per internal pass pair
                                            pipelined=bool(args.pipelined))
        disp = eng.displacement(table, 0, 5.0, 50.0, frame_range=(a, b))    # this rank's frames of the gathered table
        if args.workload == "c5":
            eng.plane_fit(table[a:b])
        return table, disp, counts

    def barrier():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    def timed(fr, steps, warmup):
        for _ in range(warmup):
            out = step(fr)
        barrier()
        eng.ncc_counters(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step(fr)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev if args.backend != "gloo" else "cpu")
            td.all_reduce(t, op=td.ReduceOp.MAX)
            el = float(t.item())
        return el, out

    elapsed, out = timed(frames, args.steps, args.warmup)
    ncc_ctr = eng.ncc_counters()
    table, disp, counts = out
    tracked = int((table[..., 0].int() & 1).sum().item())
    solved = int(((table[..., 0].int() & 2) > 0).sum().item())
    assert int(counts.min().item()) >= 0, "a frame reported a device status"
    if args.workload == "real":
        # noise on real texture: a frame may miss a dot; every frame must still be found almost whole
        assert tracked >= 0.999 * n_total * M, f"tracked {tracked} of {n_total * M} marker observations"
    else:
        assert tracked == n_total * M, f"tracked {tracked} of {n_total * M} marker observations"
    table_sums = None
    if world > 1:
        # the same gathered table on every rank: (sum, xor-fold) of its bits, exchanged and compared everywhere
        bits = table.contiguous().view(torch.int32).to(torch.int64).view(-1, M * 10)
        wcol = torch.arange(1, M * 10 + 1, dtype=torch.int64, device=bits.device)
        wrow = torch.arange(1, bits.shape[0] + 1, dtype=torch.int64, device=bits.device)
        # (int64 arithmetic wraps: a position-weighted sum, so that two tables with the same rows in another order differ)
        mine = torch.stack([bits.sum(), ((bits * wcol).sum(dim=1) * wrow).sum()]).to(dev if args.backend != "gloo" else "cpu")
        allsums = [torch.zeros_like(mine) for _ in range(world)]
        td.all_gather(allsums, mine)
        table_sums = [[int(v) for v in t.tolist()] for t in allsums]
        assert all(t == table_sums[0] for t in table_sums), f"rank {rank}: the gathered tables differ between ranks: {table_sums}"

    result = None
    if rank == 0:
        fps = n_total * args.steps / elapsed
        fmt = "gray uint8" if args.channels == 1 else "BGR uint8 (3-channel)"
        result = {
            "metric": f"frames/sec (track->3D) at {W}x{H}, {M} markers", "value": round(fps, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8+f64", "data": "synthetic",
            "config": {"workload": (f"BASELINE config 3/4: {args.frames} synthetic 1280x1024 {fmt} frames per GPU "
                                    f"(13x13 dots, seeded jitter+noise)" if args.workload == "c3" else
                                    f"BASELINE config 5: {args.frames} synthetic 1920x1200 {fmt} frames per GPU "
                                    f"(21x21 dots), plus plane-fit pose" if args.workload == "c5" else
                                    f"the reference's real configuration (BASELINE config 1's frames in bulk): {args.frames} "
                                    f"synthetic 640x480 {fmt} frames per GPU (7x7 dots) through the default crop "
                                    f"(1/8, 1/8, 1/16, 0) = {W}x{H} strided views, small branch" if args.workload == "c1" else
                                    f"the reference's real layout and texture: {args.frames} frames per GPU made on the device from "
                                    f"its one published frame (img/raw_markers.png, 467x437 BGR, 65 dots of ~27 px at a pitch of "
                                    f"35-42 px, edge-padded to the {W}x{H} crop frame it was cut from) by seeded shifts of +-3 px and "
                                    f"noise sigma 2; small branch") +
                                   f", resident in HBM; fused track->3D table + "
                                   f"{'RCCL all-gather + ' if world > 1 else ''}last-seen displacement",
                       "frames_per_gpu": args.frames, "internal_batch": args.batch, "pass_streams": args.pass_streams, "markers": M, "channels": args.channels,
                       "world_size": td.get_world_size() if world > 1 else 1,
                       "backend": td.get_backend() if world > 1 else "none (single process)",
                       "pipelined_gather": bool(args.pipelined) if world > 1 else None,
                       "device_uuids": dev_ids, "distinct_devices": len(set(d.split("|")[0] for d in dev_ids)) if dev_ids else 1,
                       "rccl_version": rccl, "gathered_table_checksum": table_sums[0] if table_sums else None,
                       "gathered_table_checksum_equal_on_all_ranks": bool(table_sums) if world > 1 else None,
                       "frame0_ids": id_check,
                       "tracked_observations": tracked, "xyz_solved": solved,
                       "us_per_frame_per_gpu": round(1e6 * elapsed / args.steps / args.frames, 2),
                       "whole_path_hbm_frac": round(fps / world * (H * W * args.channels + M * 40) / 1e9 / HBM_PEAK_GBS, 6),
                       # every NCC decision equals the float64 one iff ambiguous == 0 (rank 0's frames of the timed steps)
                       "ncc_ambiguous_pixels": ncc_ctr["ambiguous"], "ncc_exact_pixels": ncc_ctr["exact"],
                       "ncc_counter_frames": ncc_ctr["frames"]},
        }

    # ---- the same workload on BGR frames (the reference's input format; gray stays the headline) ----------------------
    side_legs = not crop and args.workload != "real"       # legs that render their own synthetic frames
    if args.channels == 1 and not args.no_extras and side_legs:
        nb_ = min(n_local, 2048)
        fb = as_bgr(gray[:nb_])
        n_keep = n_total
        # (single-rank side measurement on rank 0's device; the other ranks wait at the barrier below)
        if rank == 0:
            for _ in range(1):
                eng.track_to_3d(fb, xy, 20.0, cam, 5.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                tb, _, cb = eng.track_to_3d(fb, xy, 20.0, cam, 5.0)
            torch.cuda.synchronize()
            el_b = time.perf_counter() - t0
            for _ in range(1):
                eng.track_to_3d(gray[:nb_], xy, 20.0, cam, 5.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                tg, _, cg = eng.track_to_3d(gray[:nb_], xy, 20.0, cam, 5.0)
            torch.cuda.synchronize()
            el_g = time.perf_counter() - t0
            assert torch.equal(tb, tg), "BGR (B=G=R) and gray frames must give the same table"
            result["config"]["bgr_fps"] = round(nb_ * reps / el_b, 1)
            result["config"]["bgr_vs_gray_same_frames"] = round(el_g / el_b, 4)
            result["config"]["bgr_frames"] = nb_
        del fb, n_keep

    # ---- live per-kernel timing + the threshold+CCL roofline (rank 0; other ranks idle at the barrier) ----
    if rank == 0:
        nk = min(args.roofline_frames or 2 * args.batch, n_local)
        eng.profile(True)                                                    # (passes on ONE stream from here on)
        eng.track_to_3d(frames[:nk], xy, 20.0, cam, 5.0)                     # warm-up in the profiled configuration,
        torch.cuda.synchronize()
        eng.profile(True)                                                    # its records dropped (vbs_profile clears them)
        passes = max(1, -(-nk // args.batch))
        reps = max(2, -(-8 // passes))                                       # >= 8 launches of every kernel
        for _ in range(reps):
            eng.track_to_3d(frames[:nk], xy, 20.0, cam, 5.0)
        prof = eng.profile_read()
        kernels = {k: {"launches": c, "avg_ms": round(ms / c, 4), "us_per_frame": round(1e3 * ms / (nk * reps), 3)}
                   for k, (c, ms) in prof.items()}
        n_passes = prof["k_finalize"][0] / reps                              # launches of each kernel per call
        fpl = nk / n_passes                                                  # frames per launch
        stage_ms_launch = sum(prof[k][1] / prof[k][0] for k in STAGE if k in prof)      # sum of average launch durations
        alg_bytes_frame = H * W + 24 * M                                     # image read + moment sums
        achieved = alg_bytes_frame * fpl / (stage_ms_launch * 1e-3) / 1e9
        traffic = traffic_src = None
        pm = _newest(f"*_pmc_traffic_{args.workload}.json")
        if pm:
            pj = json.load(open(pm))
            traffic = round(pj["traffic_bytes_per_frame"] * fpl)
            traffic_src = os.path.basename(pm)
        result["roofline"] = {
            "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
            "kernel": "+".join(k for k in STAGE if k in prof),
            "stage": "threshold+CCL as it runs on the timed path (bit-packed masks -> detections)",
            "algorithmic_bytes_per_frame": alg_bytes_frame, "frames_per_launch": round(fpl, 1),
            "stage_ms_per_launch": round(stage_ms_launch, 4),
            "per_kernel_avg_ms": {k: round(prof[k][1] / prof[k][0], 4) for k in STAGE if k in prof},
            "us_per_frame": round(1e3 * stage_ms_launch / fpl, 3)}
        # (a) SURVEY 8(d) defines the stage as band + label + three sums (24 B per label); the kernel on the timed path also
        #     does the opened-mask half (a12-a13: opening, external contours, vertex moments).  `band_stage` is the 8(d) part
        #     alone, from the newest committed phase log of the debug library (tools/gpu_stage_phase.py: the kernel stopped
        #     after the band half's sums are out, `stop 10`); `opened_half_us_per_frame` is the rest of the live time.
        # (b) the stage reads 0.3 x its algorithmic bytes and is bound by vector-instruction issue: `valu` = the share of the
        #     SIMDs' cycles in which k_stage issues a vector instruction, from the newest committed SQ counters.
        result["roofline"]["measured_in_this_run"] = True
        if world == 1 and args.workload in ("c3", "c5") and not args.no_extras:
            live, why = _band_stage_live(args.workload, frames=min(args.batch, 2048))
            if live:
                band_us = float(live[10])
                result["roofline"]["band_stage"] = {
                    "what": "band + 4-connected labels + count / sum x / sum y per label: the SURVEY 8(d) stage proper",
                    "us_per_frame": band_us, "frac": round(alg_bytes_frame / (band_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                    "achieved": round(alg_bytes_frame / (band_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "whole_kernel_us_per_frame_same_child": float(live[0]),
                    "opened_half_us_per_frame": round(max(float(live[0]) - band_us, 0.0), 4),
                    "measured_in_this_run": True,
                    "how": "child process (fresh interpreter) of this run: tools/gpu_stage_phase.py, libvbs_dbg.so, VBS_STAGE_STOP=10 "
                           "(the kernel returns once the band half's sums are out) and 0 (whole kernel), min(batch, 2048) frames per launch, "
                           "HIP events"}
            else:
                result["roofline"]["band_stage"] = {"measured_in_this_run": False, "error": why}
        sqf = _newest("*_sq_counters_stage.json")
        if sqf and args.workload == "c3":
            sj = json.load(open(sqf))
            ks = sj.get("kernels", {}).get("k_stage", {})
            if ks.get("SQ_INSTS_VALU") and ks.get("GRBM_GUI_ACTIVE"):
                cyc = 4.0 * ks["SQ_ACTIVE_INST_VALU"]                       # the counter is in quad-cycles
                avail = 1024.0 * ks["GRBM_GUI_ACTIVE"] / 8.0                 # 1024 SIMDs x the launch's cycles (counter summed over 8 XCDs)
                result["roofline"]["valu"] = {
                    "bound": "valu issue", "kernel": "k_stage",
                    "instructions_per_frame": round(ks["SQ_INSTS_VALU"] / sj["frames_per_launch"]),
                    "cycles_per_instruction": round(cyc / ks["SQ_INSTS_VALU"], 2),
                    "simd_cycles_per_frame": round(avail / sj["frames_per_launch"]),
                    "frac": round(cyc / avail, 4), "waves_per_simd": 3, "measured_in_this_run": False,
                    "same_sources": (sj.get("src_sha16") == _sources_sha16()) if sj.get("src_sha16") else None,
                    "source": f"{os.path.basename(sqf)} (rocprofv3 --pmc passes over a {sj['frames_per_launch']}-frame launch, not this run; "
                              f"`same_sources`: the file's fingerprint of k_stage's sources equals today's - null = the file carries none)"}
        # the staged entry on uint8 images (the reference's `_marker_center(mask, area_mask)` interface)
        if args.channels == 1 and side_legs:
            mask, area = eng.find_markers(frames[:nk])
            torch.cuda.synchronize()
            eng.marker_center(mask, area)                       # warm
            eng.profile(True)
            for _ in range(3):
                eng.marker_center(mask, area)
            sp = eng.profile_read()
            st_ms = sum(sp[k][1] / sp[k][0] for k in ("k_threshold",) + STAGE if k in sp)
            result["roofline"]["staged_u8"] = {
                "entry": "vbs_marker_center (uint8 mask + area_mask)", "stage_ms_per_launch": round(st_ms, 4),
                "k_threshold_avg_ms": round(sp["k_threshold"][1] / sp["k_threshold"][0], 4),
                "k_threshold_GBps_of_2HW": round(2 * H * W * fpl / (sp["k_threshold"][1] / sp["k_threshold"][0] * 1e-3) / 1e9, 1),
                "frac_8d_one_image": round(alg_bytes_frame * fpl / (st_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "frac_two_images": round((2 * H * W + 48 * M) * fpl / (st_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
            del mask, area
        eng.profile(False)
        result["kernels"] = kernels
        # which labelling path did the frames take?  vbs_stage_tables describes the last internal pass: pass by pass over the
        # first `nk` frames (every frame of the `real` workload).  0 = the fused kernel kept the frame; anything else = it
        # was handed to the general kernel (k_morph + k_label), the value says why (include/vbs.h).
        import collections
        hist = collections.Counter()
        n_chk = n_local if args.workload == "real" else nk
        for s0 in range(0, n_chk, args.batch):
            e0 = min(n_chk, s0 + args.batch)
            eng.track_to_3d(frames[s0:e0], xy, 20.0, cam, 5.0)
            hist.update(int(v) for v in eng.stage_tables(e0 - s0)["slow"])
        result["config"]["slow_path_frames"] = int(sum(v for k, v in hist.items() if k))
        result["config"]["slow_path_reasons"] = {str(k): int(v) for k, v in sorted(hist.items()) if k}
        result["config"]["slow_path_frames_checked"] = int(n_chk)
        if args.workload == "real":
            # the general labelling kernel on the same frames (VBS_OPT_STAGE_IMPL = 2: k_morph + k_label label EVERY frame):
            # what a frame costs that the fast path hands on
            e2 = Engine(H, W, max_markers=256, max_batch=args.batch, device=local_rank)
            e2.set_option(L.OPT_STAGE_IMPL, 2)
            e2.set_option(L.OPT_PASS_STREAMS, 1)
            n2 = min(n_local, 2 * args.batch)
            t2, _, c2 = e2.track_to_3d(frames[:n2], xy, 20.0, cam, 5.0)
            tf, _, cf = eng.track_to_3d(frames[:n2], xy, 20.0, cam, 5.0)
            assert torch.equal(t2, tf) and torch.equal(c2, cf), "the general labelling kernel must give the fused kernel's table"
            e2.profile(True)
            for _ in range(3):
                e2.track_to_3d(frames[:n2], xy, 20.0, cam, 5.0)
            p2 = e2.profile_read()
            e2.profile(False)
            result["config"]["general_labelling_path"] = {
                "what": "VBS_OPT_STAGE_IMPL = 2: k_morph + k_label label every frame (the path of a frame with holes or beyond the "
                        "fast path's tables); same table as the fused kernel, asserted",
                "us_per_frame": {k: round(1e3 * v[1] / (3 * n2), 4) for k, v in p2.items() if k in ("k_morph", "k_label", "k_finalize")},
                "frames": n2}
            e2.close()
        # the two matrix-core kernels of the front end against the dense MFMA peaks (MI355X_MICROARCH.md: bf16/f16
        # ~2.5 PFLOP/s, int8 2x that); algorithmic operations = the separable filters as written in the reference
        # (Toeplitz padding, hi/lo splits and the count product are overhead, not counted)
        small = H <= 480
        ta, tb_, ln = (21, 35, 33) if small else (39, 101, 80)
        mf = []
        for name, ops_px, peak, dt in (("k_blur16", 2 * 2 * (ta + tb_), 5000.0, "i8"),     # (whichever blur kernel ran)
                                       ("k_blur_mfma", 2 * 2 * (ta + tb_), 5000.0, "i8"),
                                       ("k_ncc_mfma", 2 * 2 * ln, 2500.0, "f16")):
            if name in prof:
                c, ms = prof[name]
                ach = ops_px * H * W * fpl / (ms / c * 1e-3) / 1e12
                mf.append({"kernel": name, "bound": "mfma", "dtype": dt, "achieved": round(ach, 2), "peak": peak,
                           "unit": "TOP/s" if dt == "i8" else "TFLOP/s", "frac": round(ach / peak, 5),
                           "algorithmic_ops_per_pixel": ops_px, "avg_ms": round(ms / c, 4),
                           "frames_per_launch": round(fpl, 1)})
        result["roofline_mfma"] = mf
    if world > 1:
        td.barrier()

    if rank == 0 and world == 1 and not args.no_extras and side_legs:
        try:
            del frames
            torch.cuda.empty_cache()
            result["config"]["host_path_fps"] = host_path_fps(spec, args.host_frames, args.seed, 128)
        except Exception as e:
            result["config"]["host_path_fps"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            result["config"]["avi_path_fps"] = avi_path_fps(2048, args.seed)
        except Exception as e:
            result["config"]["avi_path_fps"] = {"error": f"{type(e).__name__}: {e}"}
        try:
            result["config"]["single_frame_us"] = single_frame_us(spec, args.seed, (K, dist, R, T))
        except Exception as e:
            result["config"]["single_frame_us"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and side_legs:
        try:
            result["cpu_baseline"] = cpu_baseline(spec, args.seed, (K, dist, R, T), args.cpu_single_frames,
                                                  args.cpu_workers, args.cpu_frames_per_worker, probe=cpu_probe)
        except Exception as e:                                  # the baseline must not sink the GPU number
            result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port",
                                      "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
