#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE'S OWN FUNCTION BODIES.

Run only in the build container (needs /root/reference; the GPU box never sees it):
    python tests/golden/make_golden.py

The reference modules cannot be imported (`import cv2` fails: OpenCV is not installed and cannot be),
so each cv2-free method is pulled out of the reference file by AST, compiled on its own and executed
with the real NumPy / SciPy / scikit-learn / pandas of this container (versions recorded in
`meta.json`).  Nothing of OpenCV is imitated: methods that call cv2 are not run, except that
 * `_marker_center` is cut after its SciPy-only prefix (`marker_detection.py:170-185`), and
 * `_track_markers` of both classes run on a bare `self` whose drawing hook (`_draw_tracking`) is a
   no-op and whose `_undistort_points` is the identity (valid for the zero-distortion camera used).
Only inputs and outputs are stored — no reference source text.
"""
import ast
import json
import logging
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/code"
MD = os.path.join(REF, "Marker_Tracking", "marker_detection.py")
R3 = os.path.join(REF, "Marker_Calibration", "3d_reconstruction.py")


def extract(path, cls, name, env, cut_before=None, ret=None):
    """Compile one method of `cls` from `path`; optionally keep only the statements before the
    first one whose source contains `cut_before`, and append `return <ret>`."""
    src = open(path).read()
    tree = ast.parse(src)
    cdef = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    fdef = next(n for n in cdef.body if isinstance(n, ast.FunctionDef) and n.name == name)
    fdef.decorator_list = []
    if cut_before is not None:
        keep = []
        for st in fdef.body:
            if cut_before in ast.get_source_segment(src, st):
                break
            keep.append(st)
        keep.append(ast.parse(f"return {ret}").body[0])
        fdef.body = keep
    mod = ast.Module(body=[fdef], type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(env)
    exec(compile(mod, f"<{cls}.{name} from reference>", "exec"), ns)
    return ns[name]


def md_env():
    import math
    import pandas as pd
    from scipy import ndimage
    from scipy.ndimage import maximum_filter, minimum_filter
    from scipy.signal import fftconvolve
    from scipy.spatial.distance import cdist
    from sklearn.cluster import KMeans
    return dict(np=np, pd=pd, ndimage=ndimage, maximum_filter=maximum_filter,
                minimum_filter=minimum_filter, fftconvolve=fftconvolve, cdist=cdist, KMeans=KMeans,
                math=math)


def rle_encode(bits):
    """Row-major run lengths of a 0/1 image, starting with a run of zeros."""
    flat = np.asarray(bits, dtype=np.uint8).ravel()
    change = np.flatnonzero(np.diff(flat)) + 1
    edges = np.concatenate([[0], change, [flat.size]])
    runs = np.diff(edges)
    if flat[0] == 1:
        runs = np.concatenate([[0], runs])
    return runs.astype(np.int32)


def main():
    import scipy, sklearn, pandas
    import vbs_amd.synth as S
    from oracle import stages as O

    env = md_env()
    meta = dict(numpy=np.__version__, scipy=scipy.__version__, sklearn=sklearn.__version__,
                pandas=pandas.__version__, python=sys.version.split()[0],
                note="outputs of reference function bodies executed in the build container")
    out = {}

    # ---- (1) _gkern ---------------------------------------------------------------------------
    gk = extract(MD, "MarkerTracker", "_gkern", env)
    for l, sig in ((33, 7.4), (80, 13.0), (5, 1.0)):
        k = gk(l=l, sig=sig)
        out[f"gkern_{l}_row"] = k[l // 2].copy()
        out[f"gkern_{l}_diag"] = np.diag(k).copy()
        out[f"gkern_{l}_sum"] = np.array([k.sum(), k[0, 0], k.max()])

    # ---- (2) _normxcorr2 ----------------------------------------------------------------------
    nx = extract(MD, "MarkerTracker", "_normxcorr2", env)
    rng = np.random.default_rng(7)
    img = np.zeros((64, 72), dtype=np.uint8)
    yy, xx = np.mgrid[0:64, 0:72]
    for cy, cx, r in ((16, 18, 7), (40, 50, 9), (50, 12, 5), (5, 66, 6)):
        img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 255
    img[rng.random(img.shape) < 0.01] = 255
    out["ncc_image"] = img
    for l, sig in ((8, 2.0), (9, 2.0), (14, 3.0)):
        res = nx(gk(l=l, sig=sig), img)
        out[f"ncc_out_l{l}"] = res
    big = np.zeros((120, 130), dtype=np.uint8)
    yy, xx = np.mgrid[0:120, 0:130]
    for cy, cx, r in ((30, 30, 10), (30, 90, 11), (85, 60, 9), (100, 120, 8)):
        big[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 255
    out["ncc_big_image"] = big
    out["ncc_big_out_l33"] = nx(gk(l=33, sig=7.4), big)

    # ---- (3) band + label + centroids (SciPy prefix of _marker_center) --------------------------
    mc_prefix = extract(MD, "MarkerTracker", "_marker_center", env,
                        cut_before="np.max(area_mask)", ret="(centers, labeled, num_objects)")
    for tag, spec, crop in (("c1", S.config1(), (1 / 8, 1 / 8, 1 / 16, 0)),
                            ("c2", S.config2(), (1 / 8, 1 / 8, 1 / 16, 0))):
        fr = S.make_frames(spec, [1], seed=11, channels=3)[0]
        l_, r_, t_, b_ = O.crop_box(spec.width, spec.height, crop)
        mask, area = O.find_markers(fr[t_:b_, l_:r_])
        centers, labeled, n = mc_prefix(mask, area)
        out[f"band_{tag}_shape"] = np.array(mask.shape, dtype=np.int32)
        out[f"band_{tag}_mask_rle"] = rle_encode(mask)
        out[f"band_{tag}_centers"] = np.asarray(centers, dtype=np.float64)
        out[f"band_{tag}_npix"] = np.bincount(labeled.ravel())[1:].astype(np.int32)
        out[f"band_{tag}_first"] = np.array(
            [np.flatnonzero(labeled.ravel() == i + 1)[0] for i in range(n)], dtype=np.int64)
    # a ragged case: random blobs, some touching the border, 1-px bridges
    rng = np.random.default_rng(3)
    m = (ndimage_blobs(rng, (200, 260))).astype(np.uint8)
    centers, labeled, n = mc_prefix(m, m)
    out["band_rand_mask_rle"] = rle_encode(m)
    out["band_rand_shape"] = np.array(m.shape, dtype=np.int32)
    out["band_rand_centers"] = np.asarray(centers, dtype=np.float64)
    out["band_rand_npix"] = np.bincount(labeled.ravel())[1:].astype(np.int32)

    np.savez_compressed(os.path.join(HERE, "stages.npz"), **out)

    # ---- (4) first-frame IDs and tracking rows --------------------------------------------------
    pff = extract(MD, "MarkerTracker", "_process_first_frame", env)
    trk = extract(MD, "MarkerTracker", "_track_markers", env)
    ids = {}
    layouts = {
        "ring65": S.ring65_spec(),
        "grid7": S.config1(),
    }
    for name, spec in layouts.items():
        truth = S.dot_truth(spec, 5, [0, 1, 2])
        rng = np.random.default_rng(42)
        order = rng.permutation(spec.n_markers)
        def mk(fi, drop=()):
            ms = []
            for k in order:
                if k in drop:
                    continue
                x, y, d = truth[fi, k]
                ms.append({"center": (float(x) - 0.25, float(y) - 0.25), "major_axis": float(d),
                           "minor_axis": float(d) - 0.5, "angle": 90.0})
            return ms
        results = []
        for rep in range(5):
            me = types.SimpleNamespace(config={"num_layers": 5, "min_marker_distance": 20},
                                       first_frame_markers={}, frame_count=0,
                                       _draw_tracking=lambda *a, **k: None)
            pff(me, mk(0))
            ref = [[int(k[0]), int(k[1]), float(v["Ox"]), float(v["Oy"])]
                   for k, v in me.first_frame_markers.items()]
            rows = []
            for fi, drop in ((0, ()), (1, ()), (2, (int(order[0]), int(order[3])))):
                me.frame_count = fi
                for r in trk(me, None, mk(fi, drop)):
                    rows.append([r[c] if not isinstance(r[c], (np.floating, np.integer))
                                 else r[c].item() for c in O.CSV_COLUMNS])
            results.append((ref, rows))
        assert all(r == results[0] for r in results), f"KMeans unstable on {name}"
        ids[name] = dict(frames=[mk(0), mk(1), mk(2, (int(order[0]), int(order[3])))],
                         ref=results[0][0], rows=results[0][1], num_layers=5, min_dist=20)
    json.dump(ids, open(os.path.join(HERE, "ids_as_written.json"), "w"))

    # ---- (5) _calculate_3d_position ----------------------------------------------------------------
    env3 = dict(np=np, logger=logging.getLogger("golden"))
    c3d = extract(R3, "MarkerAnalysis", "_calculate_3d_position", env3)
    cams = {
        "cam_a": dict(K=[[1400.0, 0, 640.0], [0, 1400.0, 512.0], [0, 0, 1]], R=np.eye(3).tolist(),
                      T=[0.0, 0.0, 30.0]),
        "cam_b": dict(K=[[912.25, 0, 331.5], [0, 915.75, 236.125], [0, 0, 1]],
                      R=rot(0.2, -0.1, 0.3).tolist(), T=[1.5, -2.25, 41.0]),
    }
    g3 = {}
    for cname, cam in cams.items():
        me = types.SimpleNamespace(
            camera=types.SimpleNamespace(matrix=np.array(cam["K"], dtype=np.float32),
                                         R_world_to_cam=np.array(cam["R"], dtype=np.float32),
                                         T_world_to_cam=np.array(cam["T"], dtype=np.float32).reshape(3, 1)),
            config=types.SimpleNamespace(marker_diameter_mm=2.0))
        pts = []
        for u in (12.5, 331.5, 640.0, 700.25, 1279.0):
            for v in (3.0, 236.125, 512.0, 900.75):
                for d in (5.0, 18.6, 40.25):
                    try:
                        p = c3d(me, np.float64(u), np.float64(v), np.float64(d))
                        pts.append([u, v, d] + [float(x) for x in p])
                    except ValueError:
                        pts.append([u, v, d, None, None, None])
        g3[cname] = dict(cam=cam, pts=pts)

    # ---- (6) 3-D displacement rows (last-seen semantics, warm-up, > limit rejection) ---------------
    import pandas as pd
    trk3 = extract(R3, "MarkerAnalysis", "_track_markers",
                   dict(np=np, pd=pd, logger=logging.getLogger("golden")))
    cam = cams["cam_a"]
    me = types.SimpleNamespace(
        camera=types.SimpleNamespace(matrix=np.array(cam["K"], dtype=np.float32),
                                     R_world_to_cam=np.array(cam["R"], dtype=np.float32),
                                     T_world_to_cam=np.array(cam["T"], dtype=np.float32).reshape(3, 1)),
        config=types.SimpleNamespace(marker_diameter_mm=2.0, warmup_frames=2,
                                     max_displacement_px=50.0),
        _undistort_points=lambda pts: pts)
    me._calculate_3d_position = lambda u, v, d: c3d(me, u, v, d)
    rng = np.random.default_rng(5)
    rows = []
    idsl = [(0, 0), (1, 0), (1, 1), (2, 3)]
    base = {k: (300.0 + 90 * i, 200.0 + 70 * i, 30.0 + i) for i, k in enumerate(idsl)}
    for f in range(8):
        for k in idsl:
            if f == 4 and k == (1, 1):
                continue                      # gap frame: (1,1) not seen in frame 4
            u, v, d = base[k]
            u += f * 1.5 + rng.normal(0, 0.2)
            v += -f * 0.75 + rng.normal(0, 0.2)
            d += rng.normal(0, 0.05)
            if f == 6 and k == (2, 3):
                d = 4.0                       # apparent diameter collapse -> > 50 mm jump
            rows.append(dict(frameno=f + 10, row=k[0], col=k[1], u=u, v=v, major_axis=d))
    df = pd.DataFrame(rows)
    res = trk3(me, df)
    g3["disp"] = dict(cam=cam, warmup=2, limit=50.0, rows_in=rows,
                      rows_out=[[float(r[c]) for c in O.XYZ_COLUMNS] for _, r in res.iterrows()])
    json.dump(g3, open(os.path.join(HERE, "solve3d.json"), "w"))
    json.dump(meta, open(os.path.join(HERE, "meta.json"), "w"), indent=1)
    print("golden vectors written to", HERE)


def real_frame():
    """The one real sensor frame the reference holds (`img/raw_markers.png`, README.md; 65 printed dots in the layout
    `code/ForceDistribution/ForceDistribution.py:29-95`): decoded pixels only, stored BGR like cv2.imread would."""
    from PIL import Image
    rgb = np.array(Image.open("/root/reference/img/raw_markers.png").convert("RGB"))
    np.savez_compressed(os.path.join(HERE, "raw_markers_bgr.npz"), bgr=np.ascontiguousarray(rgb[..., ::-1]))
    print("raw_markers_bgr.npz", rgb.shape)


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def ndimage_blobs(rng, shape):
    from scipy import ndimage
    a = rng.random(shape)
    a = ndimage.gaussian_filter(a, 6.0)
    m = a > np.quantile(a, 0.62)
    m[100, 20:200] = True      # a long 1-px bridge
    return m


if __name__ == "__main__":
    real_frame()
    main()
