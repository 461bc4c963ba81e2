"""A/B of the labelling stage on the GPU: the fused kernel (k_stage, VBS_OPT_STAGE_IMPL = 0) against the separate kernels
(k_morph + k_ccl, = 1) on the same frames, table by table (vbs_stage_tables) and detection row by row.
usage: gpu_stage_ab.py [c1|c2|c5|crop|blobs] [frames]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd import _lib as L
from vbs_amd.engine import Engine


def tables(eng, impl, run):
    eng.set_option(L.OPT_STAGE_IMPL, impl)
    out = run()
    torch.cuda.synchronize()
    return out, eng.stage_tables(run.n)


def compare(t0, t1, tag):
    bad = 0
    n = t0["ncomp"].shape[0]
    for i in range(n):
        if t0["slow"][i] or t1["slow"][i]:
            print(tag, "frame", i, "slow flags fused / separate:", t0["slow"][i], t1["slow"][i])
        nb0, na0 = t0["ncomp"][i]; nb1, na1 = t1["ncomp"][i]
        if (nb0, na0) != (nb1, na1):
            f0, f1 = t0["area_first"][i][:na0].tolist(), t1["area_first"][i][:na1].tolist()
            W = getattr(compare, "W", 1)
            print(tag, "frame", i, "ncomp", (nb0, na0), "!=", (nb1, na1), "only fused (y,x):", [(p // W, p % W) for p in f0 if p not in f1][:8],
                  "only separate:", [(p // W, p % W) for p in f1 if p not in f0][:8]); bad += 1; continue
        for k, cnt, cols in (("band_sums", nb0, 3), ("area_first", na0, None), ("area_sums", na0, 15), ("probe", nb0, 4)):
            a, b = t0[k][i][:cnt], t1[k][i][:cnt]
            if cols: a, b = a[:, :cols], b[:, :cols]
            if not np.array_equal(a, b):
                w = np.argwhere(a != b)
                print(tag, "frame", i, k, "differs at", w[:6].tolist(), "fused", a[tuple(w[0])], "separate", b[tuple(w[0])], "of", len(w)); bad += 1
    return bad


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    bad = 0
    if wl == "blobs":
        rng = np.random.default_rng(7)
        for (h, w) in ((480, 640), (450, 480), (1024, 1280), (700, 900), (333, 517), (1000, 1200)):
            eng = Engine(h, w, max_markers=512, max_batch=n)
            mask = np.zeros((n, h, w), np.uint8); area = np.zeros((n, h, w), np.uint8)
            yy, xx = np.mgrid[0:h, 0:w]
            for f in range(n):
                for _ in range(rng.integers(5, 60)):
                    cx, cy = rng.uniform(0, w), rng.uniform(0, h)
                    a, b, th = rng.uniform(4, 40), rng.uniform(4, 40), rng.uniform(0, np.pi)
                    u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
                    e = (u / a) ** 2 + (v / b) ** 2 <= 1
                    area[f][e] = 255
                    mask[f][(u / (0.7 * a)) ** 2 + (v / (0.7 * b)) ** 2 <= 1] = 1
                if f % 2:
                    area[f][rng.random((h, w)) < 0.002] = 0          # pin holes (mostly removed by the opening)
            mt, at = torch.from_numpy(mask).cuda(), torch.from_numpy(area).cuda()
            run = lambda: eng.marker_center(mt, at); run.n = n
            compare.W = w
            (d0, c0), t0 = tables(eng, 0, run)
            (d1, c1), t1 = tables(eng, 1, run)
            bad += compare(t0, t1, f"blobs {h}x{w}")
            if not (torch.equal(c0, c1) and torch.equal(d0, d1)):
                print("blobs", h, w, "detections differ", c0.tolist(), c1.tolist()); bad += 1
            print("blobs", h, w, "counts", c0.tolist(), "slow fused", t0["slow"].tolist(), flush=True)
            eng.close()
    else:
        spec = {"c1": S.config1, "c2": S.config2, "c5": S.config5, "crop": S.config2}[wl]()
        frames = S.make_frames_torch(spec, range(n), seed=3, device="cuda")
        if wl == "crop":
            frames = frames[:, 64:1024, 160:1120]
        eng = Engine(frames.shape[1], frames.shape[2], max_markers=512 if wl != "c5" else 1024, max_batch=n)
        run = lambda: eng.track_to_3d(frames, want_det=True); run.n = n
        compare.W = frames.shape[2]
        (_, d0, c0), t0 = tables(eng, 0, run)
        (_, d1, c1), t1 = tables(eng, 1, run)
        bad += compare(t0, t1, wl)
        if not (torch.equal(c0, c1) and torch.equal(d0, d1)):
            print(wl, "detections differ", c0.tolist(), c1.tolist()); bad += 1
        print(wl, "counts", c0.tolist(), "ncomp", t0["ncomp"].tolist()[:2], "slow fused", t0["slow"].tolist())
    print("A/B", wl, "OK" if bad == 0 else f"{bad} DIFFERENCES")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
