"""Where does the axis offset between the oracle (on the README's still) and the reference's figure come from?

Runs the CPU oracle on `raw_markers_bgr.npz` with ONE restated stage changed at a time and prints, for each variant, the
mean (and sd) over the 65 markers of `minor_axis - figure` and `major_axis - figure` (figure = figure_2d.json).  Test
infrastructure (drives `oracle/`); its table is quoted in DESIGN.md section 6 and held by
tests/test_oracle_golden.py::test_real_sensor_frame_against_the_published_figure.

    python tests/golden/figure_axis_experiments.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import stages as O          # noqa: E402
from figure_check import axis_offsets   # noqa: E402


def pipeline(gray, blur=O.gaussian_blur_u8, thresh=None, approx_simple=True, opening=True, edge=0.0):
    """`_find_markers` + `_marker_center` of the oracle with the named stage exchanged."""
    p = O.branch_params(gray.shape[0])
    b3, b8 = blur(gray, p["k3"], p["s3"]), blur(gray, p["k8"], p["s8"])
    dog = ((b8.astype(int) - b3.astype(int) + 15) & 255).astype(np.uint8)
    area = O.in_range(dog, p["thresh"] if thresh is None else thresh, p["hi"])
    with np.errstate(all="ignore"):
        mask = (O.normxcorr2(O.gkern(p["tl"], p["tsig"]), area) > 0.1).astype("uint8")
    centers, _, _ = O.band_centroids(mask)
    fg = O.morph_open5(area != 0) if opening else (area != 0)
    cxy = [(c[1], c[0]) for c in centers]
    out = []
    for c in O.find_contours_external(fg, approx_simple=approx_simple):
        if len(c) < 5:
            continue
        (cx, cy), (w, h), ang = O.fit_ellipse(c)
        major, minor = max(w, h) + edge, min(w, h) + edge
        d = [(x - cx) ** 2 + (y - cy) ** 2 for x, y in cxy]
        i = int(np.argmin(d))
        if minor >= 5 and d[i] < (minor / 10) ** 2:
            out.append({"center": cxy[i], "major_axis": major, "minor_axis": minor})
    return out


def contrast(gray, gain):
    g = gray.astype(float)
    return np.clip((g - g.mean()) * gain + g.mean(), 0, 255).round().astype(np.uint8)


def variants(bgr):
    gray = O.bgr2gray(bgr, 15)
    yield "oracle as restated", pipeline(gray)
    yield "BGR2GRAY 14-bit coefficient set", pipeline(O.bgr2gray(bgr, 14))
    yield "GaussianBlur: float kernel, round once", pipeline(gray, blur=O.gaussian_blur_u8_float)
    yield "no 5x5 opening", pipeline(gray, opening=False)
    yield "fitEllipse on every boundary pixel (CHAIN_APPROX_NONE)", pipeline(gray, approx_simple=False)
    yield "contour on the outer pixel EDGES (+1 px)", pipeline(gray, edge=1.0)
    yield "PNG's R and B exchanged", pipeline(O.bgr2gray(bgr[..., ::-1].copy(), 15))
    for th in (36, 37, 38, 39, 40):
        yield f"DoG threshold {th} (reference: 35)", pipeline(gray, thresh=th)
    for gain in (0.9, 0.85, 0.8):
        yield f"image contrast x {gain:.2f} about its mean", pipeline(contrast(gray, gain))


def main():
    bgr = np.load(os.path.join(HERE, "raw_markers_bgr.npz"))["bgr"]
    g = O.bgr2gray(bgr)
    h = np.bincount(g.ravel(), minlength=256)
    print(f"still: {h[0]} pixels at exactly 0 ({100 * h[0] / g.size:.1f} %), {h[1]} at 1, {h[2]} at 2 -> blacks are clipped")
    print(f"{'variant':58s} {'n':>3s} {'minor - figure':>18s} {'major - figure':>18s}")
    for name, mk in variants(bgr):
        dmin, dmaj, _ = axis_offsets(HERE, mk)
        print(f"{name:58s} {len(mk):3d} {dmin.mean():+8.3f} (sd {dmin.std():.3f}) {dmaj.mean():+8.3f} (sd {dmaj.std():.3f})")


if __name__ == "__main__":
    main()
