"""The few-frames labelling kernel under repetition: its workgroups talk through global memory (arrival counters, one bounded
wait, lists written on one XCD and read on another), so a missing fence would show as a RARE wrong table, not in one parity
run.  64 distinct frames, their tables by the batch kernel once (VBS_OPT_LATENCY_FRAMES = 0), then `reps` rounds of calls with
1, 3, 8 and 19 frames per call in shuffled order through k_stage_lat: every table, detection row and count must be identical.
usage: gpu_lat_stress.py [reps = 40]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd import _lib as L
from vbs_amd.engine import Engine
from vbs_amd.pipeline import reference_from_frame0

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for spec, nm in ((S.config2(), 512), (S.config1(), 512), (S.config5(), 1024)):
    n = 64 if spec.width <= 1280 else 24
    ft = S.make_frames_torch(spec, range(n), seed=9, device="cuda")
    cam = L.make_camera(*S.default_camera(spec), 2.0)
    eng = Engine(spec.height, spec.width, max_markers=nm, max_batch=n)
    ids, xy = reference_from_frame0(eng, ft[:1], 5, "full", "optimal")
    xy_d = torch.as_tensor(xy, dtype=torch.float64, device="cuda")
    eng.set_option(L.OPT_LATENCY_FRAMES, 0)
    t0, d0, c0 = eng.track_to_3d(ft, xy_d, 20.0, cam, 5.0, want_det=True)
    t0, d0, c0 = t0.clone(), d0.clone(), c0.clone()
    eng.set_option(L.OPT_LATENCY_FRAMES, 32)
    rng = np.random.default_rng(1)
    calls = 0
    for r in range(reps):
        for nb in (1, 3, 8, 19):
            order = rng.permutation(n)
            for a in range(0, n - nb + 1, nb):
                idx = torch.as_tensor(order[a:a + nb], device="cuda")
                t, d, c = eng.track_to_3d(ft[idx].contiguous(), xy_d, 20.0, cam, 5.0, want_det=True)
                calls += 1
                if not (torch.equal(t, t0[idx]) and torch.equal(c, c0[idx]) and torch.equal(d, d0[idx])):
                    bad += 1
                    print(f"{spec.width}x{spec.height}: round {r}, {nb} frames per call, frames {order[a:a + nb].tolist()}: DIFFERENT "
                          f"(counts {c.tolist()} against {c0[idx].tolist()})", flush=True)
    torch.cuda.synchronize()
    print(f"{spec.width}x{spec.height}: {calls} calls, {int(c0.min())}..{int(c0.max())} markers per frame, different: {bad}", flush=True)
    eng.close()
print("lat stress all OK" if bad == 0 else f"lat stress: {bad} DIFFERENT")
sys.exit(1 if bad else 0)
