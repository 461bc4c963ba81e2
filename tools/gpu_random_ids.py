"""Random soak of the first-frame ID assignment (a14 / f4): `vbs_assign_ids` on the device against oracle.process_first_frame
(the reference's `_process_first_frame`, :275-347) and the host restatement, on random point sets - rings with jitter, random
clouds, grids - for 1..16 layers and both ID modes.  Same keys in the same order, same coordinates per key (float64, bit for
bit).  usage: gpu_random_ids.py [cases=500] [seed=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from vbs_amd.engine import Engine
from vbs_amd import ids as I
from oracle import stages as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
eng = Engine(480, 640, max_markers=512, max_batch=1)
bad = skipped = 0
for it in range(cases):
    kind = int(rng.integers(0, 3))
    if kind == 0:                                            # rings around a centre, like the sensor
        pts = [[320.0 + rng.normal(0, 1), 240.0 + rng.normal(0, 1)]]
        for k in range(1, int(rng.integers(2, 7))):
            cnt = 6 * k - int(rng.integers(0, 3))
            a0 = rng.uniform(0, 2 * np.pi)
            for j in range(cnt):
                a = a0 + 2 * np.pi * j / cnt
                pts.append([320 + 38 * k * np.cos(a) + rng.normal(0, 2), 240 + 38 * k * np.sin(a) + rng.normal(0, 2)])
        pts = np.array(pts)
    elif kind == 1:
        pts = rng.uniform([0, 0], [640, 480], (int(rng.integers(1, 300)), 2))
    else:
        n = int(rng.integers(2, 15))
        gx, gy = np.meshgrid(np.arange(n) * 40.0 + 30, np.arange(n) * 30.0 + 20)
        pts = np.stack([gx.ravel(), gy.ravel()], axis=1) + rng.normal(0, 0.5, (n * n, 2))
    m = len(pts)
    layers = int(rng.integers(1, 17))
    mode = ["as_written", "full"][int(rng.integers(0, 2))]
    markers = [{"center": (float(x), float(y)), "major_axis": 20.0, "minor_axis": 19.0, "angle": 0.0} for x, y in pts]
    det = torch.zeros((1, 512, 6), dtype=torch.float64, device="cuda")
    det[0, :m, 0:2] = torch.from_numpy(pts)
    det[0, :m, 2] = 20.0; det[0, :m, 3] = 19.0
    counts = torch.tensor([m], dtype=torch.int32, device="cuda")
    try:
        oref = O.process_first_frame(markers, layers, mode, "optimal")
    except Exception as e:                                   # (the reference raises too: fewer markers than layers ...)
        try:
            eng.assign_ids(det, counts, layers, mode)
            print(f"case {it}: the oracle raised {type(e).__name__} but the device did not"); bad += 1
        except Exception:
            skipped += 1
        continue
    ids, xy = eng.assign_ids(det, counts, layers, mode)
    ids, xy = ids.cpu().numpy().astype(np.int64), xy.cpu().numpy()
    okeys = list(oref.keys())
    oxy = np.array([[oref[k]["Ox"], oref[k]["Oy"]] for k in okeys]).reshape(-1, 2)
    hi, hxy = I.reference_arrays(I.assign_ids(markers, layers, mode, "optimal"))
    ok = [tuple(int(v) for v in k) for k in ids.tolist()] == okeys and np.array_equal(xy, oxy) and np.array_equal(ids, hi) and np.array_equal(xy, hxy)
    if not ok:
        bad += 1
        nk = [tuple(int(v) for v in k) for k in ids.tolist()] == okeys
        print(f"case {it}: kind {kind} m {m} layers {layers} mode {mode}: keys equal {nk}, slots with other coordinates "
              f"{int((xy != oxy).any(axis=1).sum()) if xy.shape == oxy.shape else 'shape'}", flush=True)
print("random ID cases:", cases, "compared:", cases - skipped, "both raised:", skipped, "bad:", bad)
sys.exit(1 if bad else 0)
