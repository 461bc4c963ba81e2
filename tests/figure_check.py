"""Shared checker: a pipeline's output for the reference's real frame (`tests/golden/raw_markers_bgr.npz`) against
the reference's own published result for that scene (`tests/golden/figure_2d.json`, measured from
`img/2d_visualization.png` by `tests/golden/make_figure_fixture.py`).  Used by the CPU oracle test and the GPU test.

What the figure pins, and how tightly (the figure's own reading precision is ~0.1 px: its ellipses' short axes
span 19.04..23.84 against the colourbar's 19.12..23.92, and 15 dot colours agree with their ellipse within 0.13 px):
  * centres: a similarity fit (one scale, one translation) of our (Cx, Cy) onto the figure's - `raw_markers.png` is a
    467x437 window of the 480x450 crop frame - must give scale 1 +- 0.003 and residuals <= 0.75 px (measured: scale
    1.0005, rms 0.25 px, max 0.47 px);
  * IDs: under `id_mode="full"` our (layer, idx) of every marker must be the figure's printed number:
    label = 1 + [0, 1, 7, 19, 37, 61][layer] + idx  (65 of 65) - this is the only reference-held statement of the `full`
    order of `marker_detection.py:337-347`;
  * axes: the figure's are averages over a video that is not in the repository, ours come from the README's still, so
    they are compared as a documented offset, not as equality: measured minor +0.84 px (sd 0.22), major +0.69 (sd 0.26),
    correlation 0.975 - see DESIGN section 6 for the experiment table (which restated stage moves the axes by how much;
    the offset equals 3 grey levels of the DoG threshold = 15 % of image contrast, and the README still has clipped
    blacks).  Bounds asserted: mean offset of either axis in [0.3, 1.2] px, per-marker deviation from that mean <= 0.8 px,
    correlation >= 0.95, direction of the major axis within 20 degrees wherever both ellipses have major / minor > 1.1.
"""
import json
import os

import numpy as np

LAYER_OFFSET = [0, 1, 7, 19, 37, 61]


def load_figure(golden_dir):
    with open(os.path.join(golden_dir, "figure_2d.json")) as f:
        fig = json.load(f)
    ms = sorted(fig["markers"], key=lambda m: m["label"])
    assert [m["label"] for m in ms] == list(range(1, 66))
    return fig, ms


def similarity_fit(P, Q):
    """Q ~ s * P + t (least squares, one isotropic scale): returns s, t, residual vectors."""
    Pm, Qm = P.mean(0), Q.mean(0)
    s = ((P - Pm) * (Q - Qm)).sum() / ((P - Pm) ** 2).sum()
    t = Qm - s * Pm
    return s, t, Q - (s * P + t)


def axis_offsets(golden_dir, markers):
    """Per-marker (minor - figure, major - figure) after attaching every marker to the nearest figure marker."""
    _, ms = load_figure(golden_dir)
    F = np.array([[m["u"], m["v"]] for m in ms])
    P = np.array([m["center"] for m in markers], dtype=np.float64)
    # a first translation from the means, then nearest neighbours (markers are >= 35 px apart)
    j = np.argmin(np.linalg.norm((P + (F.mean(0) - P.mean(0)))[:, None] - F[None], axis=2), axis=1)
    assert len(set(j.tolist())) == len(markers)
    dmin = np.array([m["minor_axis"] for m in markers]) - np.array([ms[k]["minor_axis"] for k in j])
    dmaj = np.array([m["major_axis"] for m in markers]) - np.array([ms[k]["major_axis"] for k in j])
    return dmin, dmaj, j


def check_against_figure(golden_dir, markers, ref_full, report=None):
    """`markers`: list of {'center', 'major_axis', 'minor_axis', 'angle'} (`_marker_center`'s output for the frame);
    `ref_full`: the ordered {(layer, idx): {.., 'Ox', 'Oy'}} of `_process_first_frame` under id_mode='full'."""
    _, ms = load_figure(golden_dir)
    F = np.array([[m["u"], m["v"]] for m in ms])
    assert len(markers) == 65 and len(ref_full) == 65
    # IDs: the figure's number of the marker at every (layer, idx)
    keys = list(ref_full.keys())
    O = np.array([[ref_full[k]["Ox"], ref_full[k]["Oy"]] for k in keys], dtype=np.float64)
    ours = np.array([1 + LAYER_OFFSET[k[0]] + k[1] for k in keys])
    assert sorted(ours.tolist()) == list(range(1, 66))
    Q = F[ours - 1]                                   # where the figure puts the marker with OUR number
    s, t, r = similarity_fit(O, Q)
    rn = np.hypot(r[:, 0], r[:, 1])
    assert abs(s - 1) <= 0.003, s
    assert rn.max() <= 0.75 and np.sqrt((rn ** 2).mean()) <= 0.35, (rn.max(), np.sqrt((rn ** 2).mean()))
    # (a wrong label would put a marker >= 35 px from its figure position, so the residual bound IS the 65 / 65 ID check;
    #  said explicitly as well:)
    nearest = np.argmin(np.linalg.norm((s * O + t)[:, None] - F[None], axis=2), axis=1) + 1
    assert np.array_equal(nearest, ours), "full IDs differ from the numbers printed in the reference's figure"
    # axes
    dmin, dmaj, j = axis_offsets(golden_dir, markers)
    mn = np.array([m["minor_axis"] for m in markers])
    mj = np.array([m["major_axis"] for m in markers])
    fmn = np.array([ms[k]["minor_axis"] for k in j])
    fmj = np.array([ms[k]["major_axis"] for k in j])
    for d in (dmin, dmaj):
        assert 0.3 <= d.mean() <= 1.2 and np.abs(d - d.mean()).max() <= 0.8, (d.mean(), np.abs(d - d.mean()).max())
    assert np.corrcoef(mn, fmn)[0, 1] >= 0.95 and np.corrcoef(mj, fmj)[0, 1] >= 0.95
    ang = np.array([m["angle"] for m in markers])
    fang = np.array([ms[k]["angle"] for k in j])
    both = (mj / mn > 1.1) & (fmj / fmn > 1.1)
    dang = (ang - fang + 90.0) % 180.0 - 90.0
    assert both.sum() >= 10 and np.abs(dang[both]).max() <= 20.0, (both.sum(), np.abs(dang[both]).max())
    out = dict(scale=float(s), translation=[float(t[0]), float(t[1])], centre_rms_px=float(np.sqrt((rn ** 2).mean())),
               centre_max_px=float(rn.max()), ids_equal=65, minor_offset_mean=float(dmin.mean()), minor_offset_sd=float(dmin.std()),
               major_offset_mean=float(dmaj.mean()), major_offset_sd=float(dmaj.std()),
               minor_corr=float(np.corrcoef(mn, fmn)[0, 1]), major_corr=float(np.corrcoef(mj, fmj)[0, 1]),
               angle_max_deg=float(np.abs(dang[both]).max()), angle_markers=int(both.sum()))
    if report is not None:
        report.update(out)
    return out
