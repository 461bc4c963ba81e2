"""Random soak of the last-seen displacement (a21, 3d_reconstruction.py:263-314): tables with random presence, invalid 3-D rows,
rows under the size filter, jumps beyond the limit, warm-ups, and frame ranges (a rank's shard) - the chunk-parallel kernel
against the plain sequential loop of the reference.  usage: gpu_random_displacement.py [cases=200] [seed=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from vbs_amd.engine import Engine

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
eng = Engine(480, 640, max_markers=64, max_batch=1)
bad = 0
for it in range(cases):
    n, m = int(rng.integers(1, 1500)), int(rng.integers(1, 65))
    p_present, p_ok, p_small = rng.uniform(0.05, 1.0), rng.uniform(0.5, 1.0), rng.uniform(0, 0.2)
    tab = np.zeros((n, m, 10), dtype=np.float32)
    present = rng.random((n, m)) < p_present
    if rng.integers(0, 2):
        present[:int(rng.integers(0, min(n, 20)))] = False
    for _ in range(int(rng.integers(0, 4))):                 # long gaps
        a = int(rng.integers(0, n)); present[a:a + int(rng.integers(1, 400)), int(rng.integers(0, m))] = False
    ok = rng.random((n, m)) < p_ok
    tab[..., 0] = present * (1 + 2 * ok)
    tab[..., 3] = np.where(rng.random((n, m)) < p_small, 4.0, 20.0)
    xyz = np.cumsum(rng.normal(0, rng.uniform(0.05, 3.0), (n, m, 3)), axis=0) + 30
    for _ in range(int(rng.integers(0, 6))):
        xyz[int(rng.integers(0, n)):, int(rng.integers(0, m))] += rng.uniform(30, 120)
    tab[..., 6:9] = xyz
    warm, minsz, lim = int(rng.integers(0, 150)), 5.0, float(rng.uniform(5, 80))
    want = np.zeros((n, m, 5), dtype=np.float64)
    seen = (tab[..., 0].astype(int) & 1 > 0) & (tab[..., 3] >= minsz)
    if seen.any():
        fmin = int(np.nonzero(seen.any(1))[0][0])
        for r in range(m):
            last = None
            for f in np.nonzero(seen[:, r])[0]:
                if f < fmin + warm:
                    continue
                good = int(tab[f, r, 0]) & 2
                cur = tab[f, r, 6:9].astype(np.float64)
                if last is not None and last[0] and good:
                    d = cur - last[1]
                    mm = np.sqrt((d * d).sum())
                    if not mm > lim:
                        want[f, r] = [1, d[0], d[1], d[2], mm]
                last = (good, cur)
    tt = torch.from_numpy(tab).cuda()
    full = eng.displacement(tt, warm, minsz, lim).cpu().numpy()
    ok_case = np.array_equal(full[..., 0], want[..., 0]) and np.allclose(full[..., 1:], want[..., 1:], rtol=1e-6, atol=1e-6)
    for _ in range(3):
        a = int(rng.integers(0, n)); b = int(rng.integers(a, n + 1))
        part = eng.displacement(tt, warm, minsz, lim, frame_range=(a, b)).cpu().numpy()
        ok_case = ok_case and np.array_equal(part, full[a:b])
    if not ok_case:
        bad += 1
        print(f"case {it}: n {n} m {m} warm {warm} lim {lim:.1f}: DIFF ({int((full[..., 0] != want[..., 0]).sum())} flags)", flush=True)
print("random displacement cases:", cases, "bad:", bad)
sys.exit(1 if bad else 0)
