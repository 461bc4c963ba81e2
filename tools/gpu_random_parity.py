"""Random end-to-end parity soak (uses the oracle: a checker, like the tests): N random geometries - frame size (any width >=
128, height >= 64), dot grid (count, pitch, diameter), noise, gray or BGR, a random crop - through the HIP path (masks of
_find_markers, detections of the fused track path, a batch of 3 frames and one frame per call) against oracle/stages.py:
masks bit-exact, the same detections in the same order with bit-exact centroids, axes within 1e-3 px.
usage: gpu_random_parity.py [cases=40] [seed=0] [latency_frames=24] [stage_impl=0]   (latency_frames 0: the batch of 3 goes through
the BATCH labelling kernel k_stage - its 256-thread instance on small frames - instead of k_stage_lat; stage_impl 4: 256 threads
also on large frames, as passes of >= 512 frames run them)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vbs_amd.synth as S
from vbs_amd.engine import Engine
from vbs_amd.marker_detection import _det_to_markers
from oracle import stages as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t_start = time.time()
for it in range(cases):
    small = bool(rng.integers(0, 2))
    if small:                                                # the reference's small branch (h <= 480): dots ~20 px
        H, W = int(rng.integers(64, 481)), int(rng.integers(128, 700))
        dia, pitch = int(rng.integers(14, 26)), int(rng.integers(40, 70))
    else:
        H, W = int(rng.integers(481, 900)), int(rng.integers(176, 1000))
        dia, pitch = int(rng.integers(28, 46)), int(rng.integers(60, 90))
    n = max(1, min((min(H, W) - dia) // pitch, int(rng.integers(1, 9))))
    spec = S.grid_spec(W, H, n, pitch, dia, name="rand", noise_sigma=float(rng.uniform(0, 6)))
    ch = 3 if rng.integers(0, 2) else 1
    frames = S.make_frames(spec, [0, 1, 2], seed=int(rng.integers(0, 1000)), channels=ch)
    l, r, t, b = 0, W, 0, H
    if rng.integers(0, 3) == 0 and W >= 260 and H >= 130:    # a strided crop view
        l, t = int(rng.integers(0, 40)), int(rng.integers(0, 30))
        r, b = W - int(rng.integers(0, 40)), H - int(rng.integers(0, 30))
    if (r - l) < 128 or (b - t) < 64:
        l, r, t, b = 0, W, 0, H
    what = f"case {it}: {H}x{W} crop ({l},{t})-({r},{b}) ch {ch} grid {n}x{n} pitch {pitch} dia {dia} noise {spec.noise_sigma:.1f}"
    try:
        eng = Engine(b - t, r - l, max_markers=512, max_batch=3)
        if len(sys.argv) > 3:
            from vbs_amd import _lib as L
            eng.set_option(L.OPT_LATENCY_FRAMES, int(sys.argv[3]))
            if len(sys.argv) > 4:
                eng.set_option(L.OPT_STAGE_IMPL, int(sys.argv[4]))
        ft = torch.from_numpy(frames).cuda()[:, t:b, l:r]
        mask, area = eng.find_markers(ft)
        _, det, counts = eng.track_to_3d(ft, None, want_det=True)
        det1 = [eng.track_to_3d(ft[i:i + 1], None, want_det=True) for i in range(3)]
        mask, area, det, counts = mask.cpu().numpy(), area.cpu().numpy(), det.cpu().numpy(), counts.cpu().numpy()
        ok = True
        for i in range(3):
            fr = frames[i][t:b, l:r]
            om, oa = O.find_markers(fr)
            want = O.marker_center(om, oa)
            got = _det_to_markers(det[i], int(counts[i]))
            one = _det_to_markers(det1[i][1][0].cpu().numpy(), int(det1[i][2][0]))
            if not (np.array_equal(mask[i], om) and np.array_equal(area[i], oa)):
                ok = False; print("   masks differ in frame", i)
            if len(got) != len(want) or len(one) != len(want):
                ok = False; print("   detections:", len(got), len(one), "oracle", len(want))
            else:
                for g, o1, w in zip(got, one, want):
                    if g["center"] != w["center"] or o1["center"] != w["center"] or abs(g["major_axis"] - w["major_axis"]) > 1e-3 \
                            or abs(g["minor_axis"] - w["minor_axis"]) > 1e-3 or abs(o1["major_axis"] - w["major_axis"]) > 1e-3:
                        ok = False; print("   marker differs", g, o1, w); break
        eng.close()
    except Exception as e:                                    # a geometry the library refuses must say so, not crash
        ok = isinstance(e, ValueError)
        print("   ", type(e).__name__, str(e)[:200])
    bad += not ok
    print(what, "OK" if ok else "DIFF", f"[{time.time() - t_start:.0f} s]", flush=True)
print("random parity cases:", cases, "bad:", bad)
sys.exit(1 if bad else 0)
