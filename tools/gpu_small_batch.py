"""Passes of a few frames: the call's latency with the several-workgroups labelling kernel (VBS_OPT_LATENCY_FRAMES = 32) against
the one-workgroup-per-frame kernel (0), per pass size.  usage: gpu_small_batch.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd import _lib as L
from vbs_amd.engine import Engine
from vbs_amd.pipeline import reference_from_frame0

spec = S.config2()
ft = S.make_frames_torch(spec, range(32), seed=0, device="cuda")
cam = L.make_camera(*S.default_camera(spec), 2.0)
for nb in (1, 4, 8, 12, 16, 24, 32):
    eng = Engine(spec.height, spec.width, max_markers=512, max_batch=nb)
    ids, xy = reference_from_frame0(eng, ft[:1], 5, "full", "optimal")
    xy_d = torch.as_tensor(xy, dtype=torch.float64, device="cuda")
    row = []
    for lat in (32, 0):
        eng.set_option(L.OPT_LATENCY_FRAMES, lat)
        for _ in range(5):
            eng.track_to_3d(ft[:nb], xy_d, 20.0, cam, 5.0)
        torch.cuda.synchronize()
        ts = []
        for _ in range(100):
            t0 = time.perf_counter()
            eng.track_to_3d(ft[:nb], xy_d, 20.0, cam, 5.0)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        row.append(1e6 * sorted(ts)[50])
    print(f"{nb:3d} frames per call: {row[0]:7.1f} us with k_stage_lat, {row[1]:7.1f} us with k_stage", flush=True)
    eng.close()
