"""Where the host path's time goes (NumPy frames in host memory -> rows): upload rates from page-locked and pageable memory,
then MarkerTracker.process_frames at several batch sizes.  usage: gpu_host_path.py [frames]"""
import os, sys, time, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vbs_amd.synth as S
from vbs_amd.marker_detection import MarkerTracker, pinned_frames
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spec = S.config2()
fr = S.make_frames_torch(spec, range(n), seed=0, device="cuda").cpu().numpy()
pin = pinned_frames(fr.shape); pin[:] = fr
dev = torch.empty(fr.shape, dtype=torch.uint8, device="cuda")
for name, src in (("pageable", fr), ("pinned", pin)):
    t = torch.from_numpy(src)
    print(name, "is_pinned", t.is_pinned())
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dev.copy_(t, non_blocking=True); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"  upload {src.nbytes / 1e9:.2f} GB in {dt * 1e3:.1f} ms = {src.nbytes / dt / 1e9:.1f} GB/s = {n / dt:.0f} frames/s")
import tempfile
with tempfile.TemporaryDirectory() as td, contextlib.redirect_stdout(sys.stderr):
    clip = os.path.join(td, "c.npy"); open(clip, "wb").close()
    for batch, div in ((64, 4), (128, 1), (128, 2), (128, 4), (128, 8), (256, 4), (512, 4)):
        for name, src in (("pinned", pin), ("pageable", fr)):
            best = 0
            for rep in range(5):
                trk = MarkerTracker({"video_path": clip, "output_dir": os.path.join(td, f"o{batch}{div}{name}{rep}"), "crop_ratios": (0, 0, 0, 0),
                                     "id_mode": "full", "batch": batch, "first_batch_div": div})
                t0 = time.perf_counter(); rows = trk.process_frames(src); t1 = time.perf_counter()
                trk._save_results(rows); t2 = time.perf_counter()
                best = max(best, n / (t2 - t0))
                last = (n / (t1 - t0), n / (t2 - t0))
            print(f"batch {batch} (first batch 1/{div}) {name}: {last[0]:.0f} frames/s to rows, {last[1]:.0f} to CSV (best of 5: {best:.0f})", file=sys.__stdout__, flush=True)
