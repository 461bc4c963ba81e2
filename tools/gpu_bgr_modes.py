"""BGR input: conversion on the side stream (VBS_OPT_GRAY_SIDE_STREAM = 1) against in line (0), and gray frames.
usage: gpu_bgr_modes.py [frames=2048]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vbs_amd.synth as S
from vbs_amd import _lib as L
from vbs_amd.engine import Engine
from vbs_amd.pipeline import reference_from_frame0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
spec = S.config2()
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=512)
gray = S.make_frames_torch(spec, range(n), seed=0, device="cuda", chunk=16)
bgr = gray.unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()
cam = L.make_camera(*S.default_camera(spec), 2.0)
ids, xy = reference_from_frame0(eng, gray[:1], 5, "full", "optimal")


def timed(fr, reps=3):
    eng.track_to_3d(fr, xy, 20.0, cam, 5.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = eng.track_to_3d(fr, xy, 20.0, cam, 5.0)
    torch.cuda.synchronize()
    return n * reps / (time.perf_counter() - t0), out[0]


fg, tg = timed(gray)
print("gray", round(fg))
for mode in (1, 0):
    eng.set_option(L.OPT_GRAY_SIDE_STREAM, mode)
    fb, tb = timed(bgr)
    assert torch.equal(tb, tg)
    print("bgr side_stream", mode, round(fb), "ratio", round(fb / fg, 3), flush=True)
