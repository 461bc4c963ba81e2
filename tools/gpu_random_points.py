"""Random soak of the float64 point interfaces (a19 undistortPoints, a20 _calculate_3d_position) against the oracle: random
cameras (focal lengths, principal point, Brown-Conrady coefficients, rotations, translations, marker diameters) and points,
also the rejected ones (on the principal point, zero diameter).  usage: gpu_random_points.py [cases=300] [seed=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from vbs_amd import _lib as L
from vbs_amd.engine import undistort_points, calculate_3d
from oracle import stages as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
worst_u = worst_x = 0.0
for it in range(cases):
    W, H = rng.uniform(300, 2000), rng.uniform(200, 1500)
    K = np.array([[rng.uniform(300, 3000), 0, W / 2 + rng.normal(0, 20)], [0, rng.uniform(300, 3000), H / 2 + rng.normal(0, 20)], [0, 0, 1]], dtype=np.float32)
    dist = (np.array([rng.normal(0, 0.15), rng.normal(0, 0.05), rng.normal(0, 0.002), rng.normal(0, 0.002), rng.normal(0, 0.02)])
            * (rng.integers(0, 4) > 0)).astype(np.float32)
    q = rng.normal(0, 1, (3, 3)); Q, _ = np.linalg.qr(q)
    R = (Q * np.sign(np.linalg.det(Q))).astype(np.float32)
    T = rng.normal(0, 30, 3).astype(np.float32)
    dmm = float(np.float32(rng.uniform(0.5, 5.0)))
    cam = L.make_camera(K, dist, R, T, dmm)
    n = 400
    pts = rng.uniform([0, 0], [W, H], (n, 2))
    got_u = undistort_points(pts, cam).cpu().numpy()
    want_u = O.undistort_points(pts, K, dist)
    eu = float(np.abs(got_u - want_u).max())
    worst_u = max(worst_u, eu)
    d = rng.uniform(3, 80, n)
    d[:3] = [0.0, 1e-300, 1e300]
    uvd = np.column_stack([got_u, d])
    uvd[5, 0:2] = [float(K[0, 2]), float(K[1, 2])]           # on the principal point: the reference raises
    xyz, ok = calculate_3d(uvd, cam)
    xyz, ok = xyz.cpu().numpy(), ok.cpu().numpy().astype(bool)
    case_ok = eu <= 1e-9
    for i in range(n):
        try:
            with np.errstate(all="ignore"):
                w = O.calculate_3d_position(uvd[i, 0], uvd[i, 1], uvd[i, 2], K, R, T, np.float32(dmm))
        except (ValueError, ZeroDivisionError, FloatingPointError):
            w = None
        if (w is None) != (not ok[i]):
            case_ok = False; print(f"case {it} point {i}: accepted {bool(ok[i])}, oracle {'raised' if w is None else 'ok'} {uvd[i]}"); break
        if w is not None:
            ex = float(np.max(np.abs(xyz[i] - w) / np.maximum(np.abs(w), 1e-6)))
            worst_x = max(worst_x, ex)
            if ex > 1e-11:
                case_ok = False; print(f"case {it} point {i}: rel err {ex:.2e} {xyz[i]} {w}"); break
    bad += not case_ok
print(f"random point cases: {cases} bad: {bad}; worst undistort |err| {worst_u:.2e} px, worst 3-D rel err {worst_x:.2e}")
sys.exit(1 if bad else 0)
