"""How many markers a 1280x1024 frame may hold before the 256-thread labelling instance (768 records, 1536 queued pairs per frame)
hands it on to the general kernel where the 768-thread one (2048 / 4096) would not: grids of n x n dots through a pass of 768
frames under VBS_OPT_STAGE_IMPL 0 (256 threads at this pass size) and 3 (768).  usage: gpu_dense_markers.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd import _lib as L
from vbs_amd.engine import Engine

N = 768
for n, pitch, dia in ((13, 72, 36), (17, 56, 30), (21, 46, 28), (22, 44, 28)):
    spec = S.grid_spec(1280, 1024, n, pitch, dia, name="dense", noise_sigma=2.0)
    ft = S.make_frames_torch(spec, range(16), seed=1, device="cuda").repeat(N // 16, 1, 1)
    eng = Engine(1024, 1280, max_markers=512, max_batch=N)
    line = f"{n}x{n} dots of {dia} px at pitch {pitch}:"
    for impl in (0, 3):
        eng.set_option(L.OPT_STAGE_IMPL, impl)
        eng.track_to_3d(ft, None); torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, c = eng.track_to_3d(ft, None); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        slow = eng.stage_tables(N)["slow"]
        line += f"  impl {impl}: {int(c.min())}-{int(c.max())} detections, {int((slow != 0).sum())} of {N} frames handed on {sorted(set(slow[slow != 0].tolist()))}, {1e6 * dt / N:.2f} us per frame;"
    print(line, flush=True)
    eng.close()
