"""Read the reference's own published output for its real frame into a data fixture  (BUILD CONTAINER ONLY).

`/root/reference/img/2d_visualization.png` (README.md:41, "Figure 3 (b): 2D Recognition (Static Frame)") is a matplotlib
plot of what the reference's `MarkerTracker` produced for the scene of `img/raw_markers.png`: per marker a dot at
(Cx, Cy) in the 480x450 crop frame of `marker_detection.py:481`, its ID as a number 1..65 (the `full` numbering of
`marker_detection.py:337-347` / `tracking.py:13-16`), a green ellipse drawn in DATA coordinates with the time-averaged
(major_axis, minor_axis, angle) of `_marker_center` (`marker_detection.py:203-243`), a red line along the major axis,
and the dot coloured by the average minor axis through the colourbar at the right.  It is the only artefact the
reference holds that carries numbers from the cv2 stages (`GaussianBlur`, `findContours`, `fitEllipse`).

This script measures the figure; it stores numbers, no pixels and no reference source:
  * axes calibration from the tick marks (pixel columns / rows of the ticks 50..400), colourbar calibration from its
    four ticks (20..23 px);
  * per marker a geometric ellipse fit to the pixels of the green stroke, weighted by the stroke's coverage of each
    pixel (the stroke is symmetric about the drawn ellipse, so the fit follows its centre line; pixels the red line,
    a grid line or the label text touch are left out by their colour and by a residual cut);
  * the dot's colour read through the figure's own colourbar where at least one pixel of the 5-px dot is neither under
    the red line nor on the dot's dark edge - a cross-check that the ellipse's short axis IS `minor_axis`
    (asserted: |ellipse - colour| <= 0.2 px on every dot with a clean pixel);
  * the labels, transcribed by hand from the figure (LABELS below: number, approximate position as read), attached
    to the nearest fitted ellipse; asserted to be a bijection onto 1..65.

Output: tests/golden/figure_2d.json.  Used by tests/test_oracle_golden.py::test_real_sensor_frame_against_the_published_figure
(CPU, oracle) and tests/test_gpu_parity.py::test_real_sensor_frame (GPU, HIP path).
"""
import json
import os

import numpy as np
from PIL import Image
from scipy import ndimage
from scipy.optimize import least_squares

SRC = "/root/reference/img/2d_visualization.png"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "figure_2d.json")

# hand transcription: label -> (u, v) as read off the figure (only used to attach a label to the nearest ellipse)
LABELS = {
    56: (241, 56), 57: (285, 63), 55: (195, 64), 65: (394, 76), 58: (329, 80), 64: (88, 82), 54: (155, 83),
    33: (230, 102), 34: (276, 104), 59: (367, 105), 53: (117, 110), 32: (188, 111), 35: (317, 122), 60: (395, 141),
    31: (150, 141), 52: (91, 146), 16: (219, 148), 17: (265, 148), 36: (350, 153), 18: (305, 169), 15: (182, 171),
    30: (126, 178), 61: (413, 183), 7: (254, 187), 51: (76, 189), 37: (370, 194), 6: (213, 200), 19: (326, 207),
    14: (161, 210), 2: (286, 218), 29: (114, 221), 38: (421, 227), 1: (243, 231), 50: (70, 234), 20: (372, 237),
    5: (202, 243), 8: (328, 250), 13: (162, 254), 3: (274, 260), 28: (118, 268), 4: (234, 272), 39: (415, 272),
    49: (77, 279), 21: (362, 282), 9: (307, 291), 12: (184, 292), 27: (138, 307), 11: (224, 314), 10: (269, 312),
    40: (397, 317), 48: (94, 321), 22: (337, 323), 26: (171, 338), 23: (300, 348), 41: (370, 351), 25: (212, 356),
    47: (125, 358), 24: (260, 362), 42: (333, 383), 46: (159, 383), 62: (401, 384), 63: (93, 391), 45: (203, 403),
    43: (295, 400), 44: (248, 408),
}

# tick marks found as dark runs just outside the axes box (columns at rows 563..565, rows at columns 37..39, colourbar
# ticks at columns 617..619); the values are the printed tick labels
X_TICK_COLS = [52, 115, 178, 241, 304, 367, 429, 492]
Y_TICK_ROWS = [73, 136, 199, 262, 325, 388, 451, 514]
TICK_VALUES = [50, 100, 150, 200, 250, 300, 350, 400]
CBAR_TICK_ROWS = {23: 126, 22: 246, 21: 366, 20: 486}
CBAR_ROWS = (16, 591)                # first / last coloured row of the bar
AXES_BOX = (46, 562, 42, 555)        # rows, columns inside the frame lines


def find_ticks(rgb):
    dark = rgb.sum(2) < 200
    xs = [c for c in range(AXES_BOX[2], AXES_BOX[3]) if dark[563:566, c].all()]
    ys = [r for r in range(AXES_BOX[0], AXES_BOX[1]) if dark[r, 37:40].all()]
    cb = [r for r in range(rgb.shape[0]) if dark[r, 617:620].all()]
    return xs, ys, cb


def sampson(p, x, y, w):
    cx, cy, a, b, t = p
    c, s = np.cos(t), np.sin(t)
    X = (x - cx) * c + (y - cy) * s
    Y = -(x - cx) * s + (y - cy) * c
    F = (X / a) ** 2 + (Y / b) ** 2 - 1
    g = 2 * np.sqrt((X / a ** 2) ** 2 + (Y / b ** 2) ** 2)
    return np.sqrt(w) * F / g            # ~ signed distance to the ellipse, in pixels


def fit_stroke(xs, ys, w):
    cx0, cy0 = np.average(xs, weights=w), np.average(ys, weights=w)
    r0 = np.average(np.hypot(xs - cx0, ys - cy0), weights=w)
    keep = np.ones(len(xs), bool)
    for _ in range(3):                   # refit without pixels farther than 1.6 px from the curve (contamination)
        best = None
        for t0 in (0.0, 0.8, 1.6, 2.4):
            r = least_squares(sampson, [cx0, cy0, r0 * 1.05, r0 * 0.95, t0], args=(xs[keep], ys[keep], w[keep]))
            if best is None or r.cost < best.cost:
                best = r
        d = np.abs(sampson(best.x, xs, ys, np.ones_like(w)))
        keep = d < 1.6
    cx, cy, a, b, t = best.x
    a, b = abs(a), abs(b)
    if b > a:
        a, b, t = b, a, t + np.pi / 2
    rms = float(np.sqrt(np.average(sampson(best.x, xs[keep], ys[keep], np.ones(keep.sum())) ** 2, weights=w[keep])))
    return cx, cy, a, b, np.degrees(t) % 180.0, rms, int(keep.sum())


def main():
    rgb = np.array(Image.open(SRC).convert("RGB")).astype(float)
    assert rgb.shape == (621, 692, 3), rgb.shape
    xs, ys, cb = find_ticks(rgb)
    assert xs == X_TICK_COLS and ys == Y_TICK_ROWS and cb == sorted(CBAR_TICK_ROWS.values()), (xs, ys, cb)
    bx, ax = np.polyfit(TICK_VALUES, X_TICK_COLS, 1)          # column = ax + bx * u
    by, ay = np.polyfit(TICK_VALUES, Y_TICK_ROWS, 1)          # row    = ay + by * v
    scale = 0.5 * (bx + by)                                   # figure px per data px (1.2571 / 1.2600: ticks snap to pixels)
    cb_rows = np.arange(CBAR_ROWS[0], CBAR_ROWS[1] + 1)
    cb_val = 23.0 + (CBAR_TICK_ROWS[23] - cb_rows) / 120.0    # 120 rows per px of minor axis
    cb_rgb = rgb[cb_rows, 595:610].mean(1)

    green, white = np.array([76.5, 166.4, 76.5]), np.array([255.0, 255.0, 255.0])   # 'green', alpha 0.7, over white
    d = green - white
    cov = ((rgb - white) @ d) / (d @ d)                                             # the stroke's coverage of a pixel
    res = np.linalg.norm(rgb - (white + cov[..., None] * d), axis=2)                # distance from the white-green line
    inside = np.zeros(rgb.shape[:2], bool)
    inside[AXES_BOX[0]:AXES_BOX[1], AXES_BOX[2]:AXES_BOX[3]] = True
    stroke = (cov > 0.08) & (cov < 1.15) & (res < 10) & inside
    red = (rgb[..., 0] > 200) & (rgb[..., 1] < 140) & (rgb[..., 2] < 140) & inside
    lab, n = ndimage.label(ndimage.binary_dilation((stroke & (cov > 0.3)) | red), structure=np.ones((3, 3)))
    sizes = ndimage.sum((stroke & (cov > 0.3)) | red, lab, range(1, n + 1))
    comps = [i + 1 for i, s in enumerate(sizes) if s > 100]
    assert len(comps) == 65, len(comps)

    markers = []
    for k in comps:
        yy, xx = np.nonzero((lab == k) & stroke)
        cx, cy, a, b, ang, rms, npx = fit_stroke(xx.astype(float), yy.astype(float), np.clip(cov[yy, xx], 0, 1))
        # the dot's colour through the colourbar: the best-matching pixel of the dot that is not under the red line
        best = (1e9, None)
        for y in range(int(cy) - 3, int(cy) + 5):
            for x in range(int(cx) - 3, int(cx) + 5):
                p = rgb[y, x]
                if np.hypot(x - cx, y - cy) > 2.2 or (p[0] > 200 and p[1] < 120):
                    continue
                dist = np.linalg.norm(cb_rgb - p, axis=1)
                j = int(dist.argmin())
                if dist[j] < best[0]:
                    best = (float(dist[j]), float(cb_val[j]))
        markers.append(dict(u=(cx - ax) / bx, v=(cy - ay) / by, major_axis=2 * a / scale, minor_axis=2 * b / scale,
                            angle=ang, fit_rms_figure_px=rms, fit_pixels=npx,
                            minor_axis_from_colour=best[1] if best[0] < 12.0 else None, colour_residual=best[0]))
    P = np.array([[m["u"], m["v"]] for m in markers])
    used = {}
    for label, (u, v) in LABELS.items():
        dd = np.hypot(P[:, 0] - u, P[:, 1] - v)
        j = int(dd.argmin())
        assert dd[j] < 6.0 and j not in used, (label, dd[j])
        used[j] = label
        markers[j]["label"] = label
    assert sorted(used.values()) == list(range(1, 66))
    markers.sort(key=lambda m: m["label"])
    clean = [m for m in markers if m["minor_axis_from_colour"] is not None]
    gap = np.array([m["minor_axis"] - m["minor_axis_from_colour"] for m in clean])
    assert len(clean) >= 15 and np.abs(gap).max() <= 0.2, (len(clean), gap)
    mn = np.array([m["minor_axis"] for m in markers])
    cb_range = (float(cb_val[-1]), float(cb_val[0]))          # the bar auto-scales to the data's min / max
    out = dict(
        source="img/2d_visualization.png (README.md:41), measured by tests/golden/make_figure_fixture.py",
        frame="480x450 crop of marker_detection.py:481; (u, v) = (Cx, Cy); axes = averaged (major_axis, minor_axis) of "
              "marker_detection.py:212-217, angle = direction of the major axis in image coordinates (y down), mod 180",
        calibration=dict(col_of_u=[float(ax), float(bx)], row_of_v=[float(ay), float(by)], figure_px_per_px=float(scale),
                         colourbar_min_max=cb_range, ellipse_minor_min_max=[float(mn.min()), float(mn.max())],
                         colour_vs_ellipse_minor=dict(n=len(clean), mean=float(gap.mean()), max_abs=float(np.abs(gap).max()))),
        markers=[{k: (round(v, 4) if isinstance(v, float) else v) for k, v in m.items()} for m in markers])
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print(f"{OUT}: 65 markers; minor axis {mn.min():.2f}..{mn.max():.2f} (colourbar {cb_range[0]:.2f}..{cb_range[1]:.2f}); "
          f"colour cross-check on {len(clean)} dots: mean {gap.mean():+.3f}, max |.| {np.abs(gap).max():.3f}; "
          f"stroke fit rms {np.mean([m['fit_rms_figure_px'] for m in markers]):.2f} figure px")


if __name__ == "__main__":
    main()
