"""GPU parity tests (run on the MI355X box: `pytest -m gpu`).  Every test drives the HIP path through
the C-ABI (via the ctypes engine or the drop-in classes) and checks it against the CPU oracle on the
same seeded inputs, or against the committed goldens produced by the reference's own function bodies.

Stated tolerances (north_star: IDs bit-exact, centroids / 3-D within an fp32 tolerance):
  masks, labels, counts, IDs          : exact
  centroids in the float64 det / CSV  : exact (integer sums / count, same IEEE division as SciPy)
  centroids Cx, Cy (float32 tables)   : |d| <= 2.5e-4 px   (2 ulp of float32 at 2048)
  major / minor axis (float32)        : |d| <= 1e-3 px      (observed: identical after float32 rounding)
  ellipse angle                       : compared mod 180 deg, only when major - minor > 1e-2 px
  X, Y, Z (float32 tables)            : |d| <= 2e-5 mm      (ulp of float32 at 128 mm = 7.6e-6)
  float64 point APIs                  : rel 1e-12
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

import vbs_amd.synth as S                                     # noqa: E402
from vbs_amd import _lib as L                                 # noqa: E402
from vbs_amd import ids as I                                  # noqa: E402
from oracle import stages as O                                # noqa: E402

TOL_XY, TOL_AX, TOL_XYZ = 2.5e-4, 1e-3, 2e-5


def engine(h, w, **kw):
    from vbs_amd.engine import Engine
    kw.setdefault("max_markers", 512)
    kw.setdefault("max_batch", 4)
    return Engine(h, w, **kw)


def rle_decode(runs, shape):
    vals = np.zeros(len(runs), dtype=np.uint8)
    vals[1::2] = 1
    return np.repeat(vals, runs).reshape(shape)


def angle_close(a, b, tol=0.05):
    d = abs((a - b + 90.0) % 180.0 - 90.0)
    return d <= tol


def compare_markers(got, want):
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g["center"][0] == w["center"][0] and g["center"][1] == w["center"][1]      # bit-exact
        assert abs(g["major_axis"] - w["major_axis"]) <= TOL_AX
        assert abs(g["minor_axis"] - w["minor_axis"]) <= TOL_AX
        if w["major_axis"] - w["minor_axis"] > 1e-2:
            assert angle_close(g["angle"], w["angle"]), (g, w)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,channels,crop", [("c1", 3, (0, 0, 0, 0)), ("c1", 3, (1 / 8, 1 / 8, 1 / 16, 0)),
                                               ("c2", 1, (0, 0, 0, 0)), ("c2", 3, (1 / 8, 1 / 8, 1 / 16, 0)),
                                               ("c5", 1, (0, 0, 0, 0))])
def test_find_markers_bit_exact(tag, channels, crop):
    """a1-a8: masks of `_find_markers` are bit-exact, also through a strided crop view."""
    spec = {"c1": S.config1, "c2": S.config2, "c5": S.config5}[tag]()
    frames = S.make_frames(spec, [0, 3], seed=5, channels=channels)
    l, r, t, b = O.crop_box(spec.width, spec.height, crop)
    eng = engine(b - t, r - l, max_batch=1)
    ft = torch.from_numpy(frames).cuda()[:, t:b, l:r]
    mask, area = eng.find_markers(ft)
    stats = eng.frame_stats(1)
    mask, area = mask.cpu().numpy(), area.cpu().numpy()
    for i in range(frames.shape[0]):
        om, oa = O.find_markers(frames[i][t:b, l:r])
        assert np.array_equal(area[i], oa)
        assert np.array_equal(mask[i], om)
        assert om.sum() > 0
    assert stats[0, 1] == 0, "an NCC pixel sat within 1e-9 of the 0.1 threshold"
    assert stats[0, 0] == (area[-1] > 0).sum()
    eng.close()


def _textured(h, w, n, seed, fine=None):
    """frames whose difference of Gaussians crosses the inRange bounds all over: smooth random fields + noise.  `fine`
    (default: the small branch, h <= 480): many small dark blobs on a bright ground - the small branch's blurs (sigma 4.56 /
    11.4, thresholds 35..180) see nothing in the broad fields that exercise the large branch; 6-8 % of the pixels land in
    range and 3 % wrap past 255 (the mod-256 of `:128`)."""
    fine = (h <= 480) if fine is None else fine
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.zeros((n, h, w), np.float32)
    for f in range(n):
        for _ in range(max(30, (h * w) // 700) if fine else 30):
            cx, cy = rng.uniform(0, w), rng.uniform(0, h)
            s = rng.uniform(3, 9) if fine else rng.uniform(6, 60)
            amp = rng.uniform(-170, 70) if fine else rng.uniform(-120, 160)
            out[f] += amp * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
        out[f] += (150 if fine else 90) + rng.normal(0, 12, (h, w))
    return np.clip(out, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w,view", [(600, 800, False), (1000, 1284, False), (520, 132, False), (1024, 1280, True),
                                      (488, 240, False), (500, 248, False), (700, 1288, False),
                                      (450, 480, False), (480, 640, False), (470, 650, False), (450, 480, True),
                                      (300, 176, False), (400, 244, False), (480, 1284, False), (700, 252, False)])
def test_blur_strips_equal_the_32_column_kernel_and_the_oracle(h, w, view):
    """a4-a5 through both blur kernels (VBS_OPT_BLUR_IMPL): the 16-column strips (k_blur16: rows straight into the
    operands, border mirror folded into each strip's fragments, windows shifted to stay inside the row) and the 32-column
    kernel give the oracle's area mask bit for bit on frames whose DoG crosses the range bounds everywhere - widths that
    are no multiple of 16 / 64, the narrowest frames the strips take (240 / 176: every workgroup's window is the whole
    row), 248 and 1288 (a last strip of 8 columns), a strided view of a larger buffer; round 4: the SMALL branch on the
    strips (height <= 480: 450x480 = the reference's cropped camera frame, 480x640 the uncropped one, a view, 176 and 244
    columns), and widths that are a multiple of 4 but not of 8 on either branch (1284, 252, 244: the last workgroup's
    staged rows at a shifted origin); 132 and 650 columns stay with the 32-column kernel (too narrow / rows that do not
    load as aligned dwords)."""
    n = 2
    if view:
        big = torch.from_numpy(_textured(h + 8, w + 64, n, 5, fine=h <= 480)).cuda()
        ft = big[:, 4:4 + h, 32:32 + w]
    else:
        ft = torch.from_numpy(_textured(h, w, n, h + w)).cuda()
    host = ft.cpu().numpy()
    eng = engine(h, w, max_batch=n)
    got = {}
    for impl in (0, 1):
        eng.set_option(L.OPT_BLUR_IMPL, impl)
        mask, area = eng.find_markers(ft)
        got[impl] = (mask.cpu().numpy(), area.cpu().numpy(), eng.frame_stats(n)[:, 0].copy())
    eng.set_option(L.OPT_BLUR_IMPL, 0)
    for a, b in zip(got[0], got[1]):
        assert np.array_equal(a, b)
    p = O.branch_params(h)
    for i in range(n):
        g = host[i]
        want = O.in_range(O.dog_image(g), p["thresh"], p["hi"]) > 0
        assert np.array_equal(got[0][1][i] > 0, want)
        assert got[0][2][i] == want.sum() and want.sum() > 0.02 * h * w
    eng.close()


@pytest.mark.parametrize("h,w", [(600, 800), (450, 480)])
def test_blur_strips_many_frames_per_launch_equal_few(h, w):
    """k_blur16 maps a launch of 32 or more frames onto a 1-D grid (frame = 8 (b / 8 / per_frame) + b % 8, so that a
    frame's workgroups share an XCD) and pads the frame count to a multiple of 8: 37 frames in one pass give, frame by
    frame, what passes of 16 (the plain 3-D grid) give - large branch and small."""
    n = 37
    base = _textured(h, w, 5, 77)
    host = np.stack([np.roll(base[i % 5], 13 * i, axis=1) for i in range(n)])
    ft = torch.from_numpy(host).cuda()
    big, small = engine(h, w, max_batch=40), engine(h, w, max_batch=16)
    m1, a1 = big.find_markers(ft)
    m2, a2 = small.find_markers(ft)
    assert torch.equal(a1, a2) and torch.equal(m1, m2)
    assert len({int(a1[i].count_nonzero()) for i in range(n)}) > 4          # (the frames differ)
    big.close(); small.close()


def test_blur_strips_hand_shake_under_load():
    """k_blur16's strips take their rows from a ring in LDS that a loader wave fills (counters, no barrier): 96 frames of
    random smooth fields (made on the GPU; 40 % of the pixels in range) in one launch, twice, give the 32-column kernel's
    area and NCC masks bit for bit - a lost or early tile would show as wrong pixels."""
    h, w, n = 600, 808, 96
    g = torch.Generator(device="cuda").manual_seed(4242)
    lo = torch.randn((n, 1, h // 24 + 2, w // 24 + 2), device="cuda", generator=g) * 55 + 100
    fr = torch.nn.functional.interpolate(lo, size=(h, w), mode="bicubic", align_corners=False)[:, 0]
    fr = (fr + torch.randn((n, h, w), device="cuda", generator=g) * 10).clamp(0, 255).to(torch.uint8).contiguous()
    eng = engine(h, w, max_batch=n)
    got = []
    for impl in (0, 1, 0):
        eng.set_option(L.OPT_BLUR_IMPL, impl)
        m, a = eng.find_markers(fr)
        got.append((m.clone(), a.clone()))
    eng.set_option(L.OPT_BLUR_IMPL, 0)
    assert 0.2 < float((got[1][1] > 0).float().mean()) < 0.7
    for k in (0, 2):
        assert torch.equal(got[k][1], got[1][1]) and torch.equal(got[k][0], got[1][0])
    eng.close()


@pytest.mark.parametrize("h,w,pitch,dia", [(470, 650, 60, 20), (700, 1003, 72, 40)])
def test_find_markers_borders_and_odd_sizes(h, w, pitch, dia):
    """a4-a8 where the matrix-core kernels leave their fast paths: sizes that are no multiple of the 32 / 64 / 128
    pixel tiles, and dots cut by all four image borders (reflect-101 staging of the blurs; NCC windows that leave the
    image take the general threshold and, where it is close, the exact float64 path).  Both branches of :117."""
    n = max(h, w) // pitch + 4
    spec = S.grid_spec(w + 3 * pitch, h + 3 * pitch, n, pitch, dia, name="cut")
    gx0, gy0 = spec.grid[2] / 16.0, spec.grid[3] / 16.0
    big = S.make_frames(spec, [0, 2], seed=9)

    def off(g0, size, far):                # crop origin that puts a dot centre 3 px inside the near / 4 px inside the far border
        c = g0 + (int(np.ceil((pitch - g0) / pitch)) + 1) * pitch
        o = int(round(c + 4 - (size - 1))) if far else int(round(c - 3))
        return o + pitch * max(0, -(o // pitch))
    offs = [(off(gy0, h, False), off(gx0, w, False)), (off(gy0, h, True), off(gx0, w, True))]
    frames = np.stack([big[i][oy:oy + h, ox:ox + w] for i, (oy, ox) in enumerate(offs)])
    cut = np.zeros(4, bool)
    eng = engine(h, w, max_batch=2)
    mask, area = eng.find_markers(torch.from_numpy(np.ascontiguousarray(frames)).cuda())
    stats = eng.frame_stats(2)
    for i in range(2):
        om, oa = O.find_markers(frames[i])
        assert np.array_equal(area[i].cpu().numpy(), oa)
        assert np.array_equal(mask[i].cpu().numpy(), om)
        cut |= np.array([oa[0].any() and om[0].any(), oa[-1].any() and om[-1].any(), oa[:, 0].any() and om[:, 0].any(),
                         oa[:, -1].any() and om[:, -1].any()])
        assert stats[i, 0] == (oa > 0).sum() and stats[i, 1] == 0
        from vbs_amd.marker_detection import MarkerTracker
        compare_markers(MarkerTracker._marker_center(om, oa), O.marker_center(om, oa))      # blobs cut by the border
    assert cut.all(), "a border without a cut dot in both masks"
    # the same crops as strided views of the big gray frames (odd byte offsets: the blur's unaligned staging path)
    bt = torch.from_numpy(big).cuda()
    for i, (oy, ox) in enumerate(offs):
        m2, a2 = eng.find_markers(bt[i:i + 1, oy:oy + h, ox:ox + w])
        assert torch.equal(m2[0], mask[i]) and torch.equal(a2[0], area[i])
    eng.close()


def test_bgr_weights_and_dog_wrap():
    """a3: the cv2 fixed-point BGR weights on a frame with B != G != R; a4/a5: the mod-256 wrap and the
    upper inRange bound on a high-contrast frame (bright discs on black wrap to large values)."""
    rng = np.random.default_rng(0)
    spec = S.grid_spec(640, 480, 5, 90, 30, bg=20, fg=250, name="bright")
    g = S.make_frames(spec, [0, 1], seed=2)
    frames = np.stack([np.clip(g.astype(int) + d, 0, 255) for d in (7, -9, 3)], axis=-1).astype(np.uint8)
    frames[:, 100:140, 200:300] = rng.integers(0, 256, (2, 40, 100, 3), dtype=np.uint8)
    eng = engine(480, 640, max_batch=2)
    mask, area = eng.find_markers(torch.from_numpy(frames).cuda())
    for i in range(2):
        om, oa = O.find_markers(frames[i])
        assert np.array_equal(area[i].cpu().numpy(), oa)
        assert np.array_equal(mask[i].cpu().numpy(), om)
    gray = O.bgr2gray(frames[0])
    assert not np.array_equal(gray, frames[0][..., 1])
    eng.close()


@pytest.mark.parametrize("channels", [1, 3])
def test_undistort_frame_parity(channels):
    """a2 / f3: new camera matrix, fixed-point rectify map + remap, and detection on the undistorted frame."""
    spec = S.config1()
    frames = S.make_frames(spec, [0, 2], seed=6, channels=channels)
    if channels == 3:
        frames = frames.copy()
        frames[..., 0] = np.clip(frames[..., 0].astype(int) + 9, 0, 255)         # B != G != R
        frames[..., 2] = np.clip(frames[..., 2].astype(int) - 6, 0, 255)
    K = np.array([[600.0, 0, 318.5], [0, 604.0, 241.25], [0, 0, 1]])
    D = np.array([-0.25, 0.08, 0.001, -0.0005, 0.01])
    eng = engine(spec.height, spec.width, max_batch=2)
    newK = eng.set_undistort(K, D)
    np.testing.assert_allclose(newK, O.get_optimal_new_camera_matrix_alpha0(K, D, (spec.width, spec.height)),
                               rtol=1e-12, atol=1e-12)
    ft = torch.from_numpy(frames).cuda()
    und = eng.undistort_frames(ft).cpu().numpy()
    mask, area = eng.find_markers(ft)
    for i in range(2):
        want = O.undistort_frame(frames[i], K, D)
        assert np.array_equal(und[i], want)
        assert np.abs(want.astype(int) - frames[i].astype(int)).max() > 30        # it really moves pixels
        om, oa = O.find_markers(want)
        assert np.array_equal(area[i].cpu().numpy(), oa) and np.array_equal(mask[i].cpu().numpy(), om)
    eng.set_undistort(None)
    m2, a2 = eng.find_markers(ft)
    om, oa = O.find_markers(frames[0])
    assert np.array_equal(a2[0].cpu().numpy(), oa) and np.array_equal(m2[0].cpu().numpy(), om)
    eng.close()


def test_marker_tracker_with_calibration_params(tmp_path):
    """`MarkerTracker.process` with `calibration_params` (crop -> undistort -> detect) against the oracle."""
    import pandas as pd
    from vbs_amd.marker_detection import MarkerTracker
    spec = S.config1()
    frames = S.make_frames(spec, range(3), seed=12, channels=3)
    np.save(tmp_path / "clip.npy", frames)
    calib = {"camera_matrix": [[520.0, 0, 280.0], [0, 520.0, 225.0], [0, 0, 1]], "dist_coeffs": [-0.12, 0.03, 0.0008, -0.0006, 0.0]}
    crop = (1 / 16, 1 / 16, 0, 1 / 16)
    cfg = {"video_path": str(tmp_path / "clip.npy"), "output_dir": str(tmp_path / "o"), "crop_ratios": crop,
           "id_mode": "full", "calibration_params": calib}
    trk = MarkerTracker(cfg)
    trk.process()
    df = pd.read_csv(trk.output_csv, float_precision="round_trip")
    rows, ref = O.process_frames(list(frames), crop_ratios=crop, id_mode="full", calibration=calib)
    assert list(trk.first_frame_markers.keys()) == list(ref.keys()) and len(df) == len(rows) > 100
    want = pd.DataFrame(rows)
    assert (df[["frameno", "row", "col"]].to_numpy() == want[["frameno", "row", "col"]].to_numpy()).all()
    assert np.array_equal(df["Cx"].to_numpy(), want["Cx"].to_numpy()) and np.array_equal(df["Cy"].to_numpy(), want["Cy"].to_numpy())
    trk.width, trk.height = spec.width, spec.height
    pre = trk._preprocess_frame(frames[1])
    l, r, t, b = O.crop_box(spec.width, spec.height, crop)
    assert np.array_equal(pre, O.undistort_frame(frames[1][t:b, l:r], np.array(calib["camera_matrix"]), np.array(calib["dist_coeffs"])))


def test_ncc_map_matches_fft_reference():
    """a7: the float64 NCC map against the oracle's literal FFT evaluation."""
    spec = S.config2()
    frames = S.make_frames(spec, [2], seed=9)
    eng = engine(spec.height, spec.width, max_batch=1)
    ncc = eng.ncc_map(torch.from_numpy(frames).cuda())[0].cpu().numpy()
    _, oa = O.find_markers(frames[0])
    with np.errstate(all="ignore"):
        ref = O.normxcorr2(O.gkern(80, 13.0), oa)
    d = np.abs(ncc - ref)
    assert d.max() < 1e-6 and np.median(d) < 1e-13      # the tail is FFT noise in flat windows
    assert np.array_equal(ncc > 0.1, ref > 0.1)
    eng.close()


def test_normxcorr2_golden_from_reference_body(golden_dir):
    """`MarkerTracker._normxcorr2` against the output of the reference's own `_normxcorr2` (golden)."""
    from vbs_amd.marker_detection import MarkerTracker
    G = np.load(os.path.join(golden_dir, "stages.npz"))
    eng_img = G["ncc_big_image"]                        # 120 x 130, small-image branch (l = 33)
    t = MarkerTracker._gkern(33, 7.4)
    out = MarkerTracker._normxcorr2(t, eng_img)
    ref = G["ncc_big_out_l33"]
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) < 1e-6
    assert np.array_equal(out > 0.1, ref > 0.1)


def test_normxcorr2_general_operands(golden_dir):
    """`_normxcorr2(template, image, mode)` for operands outside the pipeline's (`marker_detection.py:146-164`): other
    template sizes (even and odd), a non-Gaussian template, a float image, all three modes - against the reference
    body's goldens and against the oracle's literal FFT evaluation."""
    from scipy.signal import fftconvolve
    from vbs_amd.marker_detection import MarkerTracker
    G = np.load(os.path.join(golden_dir, "stages.npz"))
    img = G["ncc_image"]                                 # 64 x 72 uint8 {0, 255}
    for l, sig in ((8, 2.0), (9, 2.0), (14, 3.0)):
        out = MarkerTracker._normxcorr2(MarkerTracker._gkern(l, sig), img)
        ref = G[f"ncc_out_l{l}"]
        d = np.abs(out - ref)                            # (flat windows: 0 here, FFT noise / noise in the reference)
        assert out.shape == ref.shape and d.max() < 1e-6 and np.median(d) < 1e-12
        assert np.array_equal(out > 0.1, ref > 0.1)
    rng = np.random.default_rng(4)
    t = rng.normal(size=(7, 12))                         # arbitrary rectangular template
    im = rng.normal(size=(40, 57)) * 3.0 + 1.0           # arbitrary float image
    for mode in ("full", "same", "valid"):
        with np.errstate(all="ignore"):
            ref = O.normxcorr2(t, im, mode)
        out = MarkerTracker._normxcorr2(t, im, mode)
        assert out.shape == ref.shape and np.max(np.abs(out - ref)) < 1e-9, mode
    with pytest.raises(ValueError):
        MarkerTracker._normxcorr2(t, im, "circular")
    # a grey-valued uint8 image with the pipeline's template goes the general way too (not two-valued)
    big = G["ncc_big_image"].copy()
    big[10:20, 10:20] = 77
    out = MarkerTracker._normxcorr2(MarkerTracker._gkern(33, 7.4), big)
    with np.errstate(all="ignore"):
        ref = O.normxcorr2(O.gkern(33, 7.4), big)
    assert np.max(np.abs(out - ref)) < 1e-6 and np.median(np.abs(out - ref)) < 1e-12


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["c1", "c2", "c5"])
def test_marker_center_parity(tag):
    """a9-a13 on the oracle's masks: count, order, centres, axes."""
    from vbs_amd.marker_detection import MarkerTracker
    spec = {"c1": S.config1(), "c2": S.config2(), "c5": S.config5()}[tag]
    frame = S.make_frames(spec, [4], seed=3)[0]
    om, oa = O.find_markers(frame)
    want = O.marker_center(om, oa)
    got = MarkerTracker._marker_center(om, oa)
    assert len(want) == spec.n_markers
    compare_markers(got, want)


@pytest.mark.parametrize("tag", ["c1", "c2", "rand"])
def test_band_centroids_golden(golden_dir, tag):
    """a9-a11 against the goldens from the reference's SciPy prefix (`marker_detection.py:170-185`):
    with area_mask = mask every blob's contour holds its own band centroid, so each det row carries
    the centroid of the label named in its last column."""
    G = np.load(os.path.join(golden_dir, "stages.npz"))
    shape = tuple(int(v) for v in G[f"band_{tag}_shape"])
    mask = rle_decode(G[f"band_{tag}_mask_rle"], shape)
    centers = G[f"band_{tag}_centers"]
    eng = engine(shape[0], shape[1], max_markers=1024, max_batch=1)
    mt = torch.from_numpy(mask).cuda()
    det, counts = eng.marker_center(mt, mt)
    n = int(counts[0])
    det = det[0].cpu().numpy()[:n]
    assert n > 0
    want = O.marker_center(mask, mask)               # blobs cut by the crop border may stay unmatched
    assert n == len(want)
    if tag != "rand":
        assert n >= 0.9 * centers.shape[0]
    lab = det[:, 5].astype(int) - 1
    assert len(set(lab.tolist())) == n
    assert np.array_equal(det[:, 0], centers[lab, 1])        # bit-exact vs ndimage.center_of_mass
    assert np.array_equal(det[:, 1], centers[lab, 0])
    eng.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_marker_center_random_blobs(seed):
    """Ragged input: random hole-free blobs, some touching the border, thin bridges; also checks the
    sequential contour<->centre matching and the minor<5 / len<5 rejections."""
    from scipy import ndimage
    from vbs_amd.marker_detection import MarkerTracker
    rng = np.random.default_rng(seed)
    a = ndimage.gaussian_filter(rng.random((300, 400)), 7.0)
    area = ndimage.binary_fill_holes(a > np.quantile(a, 0.7))
    area[150, 30:370] |= True
    area = ndimage.binary_fill_holes(area)
    mask = ndimage.binary_erosion(area, iterations=3)
    area_u8 = (area * 255).astype(np.uint8)
    mask_u8 = mask.astype(np.uint8)
    want = O.marker_center(mask_u8, area_u8)
    got = MarkerTracker._marker_center(mask_u8, area_u8)
    compare_markers(got, want)


def test_matching_sequential_replay_equals_parallel(monkeypatch):
    """The parallel matching and the sequential replay (taken when two contours claim one centre) agree."""
    from vbs_amd.marker_detection import MarkerTracker
    spec = S.config2()
    frame = S.make_frames(spec, [7], seed=8)[0]
    om, oa = O.find_markers(frame)
    from vbs_amd.marker_detection import _engine
    par = MarkerTracker._marker_center(om, oa)
    eng = _engine(om.shape[0], om.shape[1])
    eng.set_option(L.OPT_FORCE_SEQ_MATCH, 1)
    try:
        seq = MarkerTracker._marker_center(om, oa)
    finally:
        eng.set_option(L.OPT_FORCE_SEQ_MATCH, 0)
    assert par == seq and len(par) == spec.n_markers


@pytest.mark.parametrize("shape", [(450, 480), (960, 960), (480, 640), (1024, 1280), (1200, 1920), (130, 4096)])
def test_label_counts_on_line_patterns(shape):
    """CCL structure cases for every lanes-per-row packing (WW = 8, 15, 10, 20, 30, 64 words per row): runs that
    cross 64-px word boundaries, a cross spanning all strips and words, a diagonal staircase (one run per row),
    full-width runs (words that are all ones inside one run).  Band component counts against ndimage.label."""
    from scipy import ndimage
    H, W = shape
    eng = engine(H, W, max_markers=1024, max_batch=1)
    pats = []
    m = np.zeros(shape, np.uint8); m[:, 63] = 1; m[:, 64] = 1; pats.append(m)                  # word-crossing runs
    m = np.zeros(shape, np.uint8); m[10:H - 10, 200] = 1; m[H // 2, 30:W - 30] = 1; pats.append(m)   # cross
    m = np.zeros(shape, np.uint8)
    for y in range(0, H - 1, 2):
        x = (y * 3) % (W - 3)
        m[y, x:x + 3] = 1; m[y + 1, x + 2:x + 5] = 1
    pats.append(m)                                                                              # staircases
    m = np.zeros(shape, np.uint8); m[5, :] = 1; m[5:H - 5, 0] = 1; m[H - 6, :] = 1; m[20, 1:W - 1] = 0; pats.append(m)
    for m in pats:
        mt = torch.from_numpy(m).cuda()
        _, counts = eng.marker_center(mt, mt)
        st = eng.frame_stats(1)[0]
        assert int(counts[0]) >= 0
        want = ndimage.label(O.band_mask(m))[1]
        assert int(st[5]) == want, (shape, int(st[5]), want)
    eng.close()


def test_holes_are_filled_like_retr_external():
    """cv2.findContours(RETR_EXTERNAL) ignores hole borders and anything nested inside a hole.  The HIP path gets the
    same result by filling the holes of the opened mask (background components that do not touch the border) before
    contouring; the Euler-number check decides per frame whether that pass runs."""
    import warnings
    from vbs_amd.marker_detection import MarkerTracker
    yy, xx = np.mgrid[0:300, 0:400]
    area = np.zeros((300, 400), np.uint8)
    mask = np.zeros((300, 400), np.uint8)
    r2 = (yy - 120) ** 2 + (xx - 150) ** 2
    area[(r2 <= 45 ** 2) & (r2 >= 18 ** 2)] = 255                    # ring: one hole ...
    area[(yy - 120) ** 2 + (xx - 150) ** 2 <= 7 ** 2] = 255          # ... with a blob nested inside the hole
    mask[r2 <= 30 ** 2] = 1
    r3 = (yy - 230) ** 2 + (xx - 60) ** 2
    area[(r3 <= 30 ** 2) & (r3 >= 9 ** 2)] = 255                     # second ring
    mask[r3 <= 20 ** 2] = 1
    r4 = (yy - 240) ** 2 + (xx - 330) ** 2
    area[r4 <= 28 ** 2] = 255                                         # plain disc
    mask[r4 <= 18 ** 2] = 1
    r5 = (yy - 20) ** 2 + (xx - 378) ** 2
    area[(r5 <= 26 ** 2) & (r5 >= 8 ** 2)] = 255                      # holed disc cut by the image corner
    mask[r5 <= 14 ** 2] = 1
    want = O.marker_center(mask, area)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got = MarkerTracker._marker_center(mask, area)
    compare_markers(got, want)
    assert len(want) >= 3
    eng = engine(300, 400, max_batch=1)
    eng.marker_center(torch.from_numpy(mask).cuda(), torch.from_numpy(area).cuda())
    st = eng.frame_stats(1)[0]
    assert st[7] == 3 and st[4] == 0                                  # three holes filled, none left
    spec = S.config2()
    om, oa = O.find_markers(S.make_frames(spec, [1], seed=4)[0])
    e2 = engine(spec.height, spec.width, max_batch=1)
    e2.marker_center(torch.from_numpy(om).cuda(), torch.from_numpy(oa).cuda())
    assert e2.frame_stats(1)[0, 4] == 0 and e2.frame_stats(1)[0, 7] == 0
    eng.close(); e2.close()


def test_marker_center_empty_and_capacity():
    from vbs_amd.marker_detection import MarkerTracker
    z = np.zeros((128, 256), dtype=np.uint8)
    assert MarkerTracker._marker_center(z, z) == []
    assert O.marker_center(z, z) == []
    # a checkerboard has far more runs than the workspace holds -> status, not a wrong answer
    cb = (np.indices((1024, 1280)).sum(0) % 2).astype(np.uint8)
    with pytest.raises(L.VbsError):
        MarkerTracker._marker_center(cb, cb)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("id_mode", ["as_written", "full"])
@pytest.mark.parametrize("tag", ["ring", "c1"])
def test_marker_tracker_process_frames(tmp_path, tag, id_mode):
    """`MarkerTracker.process` (in-memory frames): CSV rows vs the oracle, IDs bit-exact."""
    import pandas as pd
    from vbs_amd.marker_detection import MarkerTracker, CSV_COLUMNS
    spec = S.ring65_spec() if tag == "ring" else S.config1()
    frames = S.make_frames(spec, range(5), seed=21, channels=3)
    np.save(tmp_path / "clip.npy", frames)
    crop = (0, 0, 0, 0) if tag == "ring" else (1 / 16, 1 / 16, 0, 1 / 16)
    cfg = {"video_path": str(tmp_path / "clip.npy"), "output_dir": str(tmp_path / "out"), "crop_ratios": crop,
           "num_layers": 5, "min_marker_distance": 20, "id_mode": id_mode, "batch": 3}
    trk = MarkerTracker(cfg)
    trk.process()
    df = pd.read_csv(trk.output_csv, float_precision="round_trip")     # the default parser is 1 ulp lossy
    assert list(df.columns) == CSV_COLUMNS
    # the upload pipeline's batch schedule (a short first batch, then full ones: here 1 + 2 + 2 frames from page-locked
    # memory) does not show in the CSV
    from vbs_amd.marker_detection import pinned_frames
    pin = pinned_frames(frames.shape)
    pin[:] = frames
    t2 = MarkerTracker(dict(cfg, batch=2, output_dir=str(tmp_path / "out2")))
    t2._save_results(t2.process_frames(pin))
    assert open(t2.output_csv, "rb").read() == open(trk.output_csv, "rb").read()
    rows, ref = O.process_frames(list(frames), crop_ratios=crop, id_mode=id_mode)
    assert list(trk.first_frame_markers.keys()) == list(ref.keys())
    assert len(df) == len(rows)
    want = pd.DataFrame(rows, columns=CSV_COLUMNS)
    assert (df[["frameno", "row", "col"]].to_numpy() == want[["frameno", "row", "col"]].to_numpy()).all()
    for c, tol in (("Ox", 0.0), ("Oy", 0.0), ("Cx", 0.0), ("Cy", 0.0), ("major_axis", TOL_AX),
                   ("minor_axis", TOL_AX)):
        assert np.max(np.abs(df[c].to_numpy() - want[c].to_numpy())) <= tol, c
    if id_mode == "as_written":
        assert len(ref) == 6
    else:
        assert len(ref) == spec.n_markers


@pytest.mark.parametrize("tag", ["ring", "c1", "c2", "c5"])
@pytest.mark.parametrize("id_mode", ["as_written", "full"])
def test_assign_ids_on_device(tag, id_mode):
    """a14 / f4: `vbs_assign_ids` (frame-0 identities on the device) against the host restatement `ids.assign_ids`
    (itself pinned to the reference body's golden, tests/test_host_logic.py): same keys in the same dict order, same
    reference coordinates, on the detections of a real frame 0."""
    from vbs_amd.pipeline import _det_to_markers
    spec = {"ring": S.ring65_spec(), "c1": S.config1(), "c2": S.config2(), "c5": S.config5()}[tag]
    frame = torch.from_numpy(S.make_frames(spec, [0], seed=4)).cuda()
    eng = engine(spec.height, spec.width, max_batch=1)
    _, det, counts = eng.track_to_3d(frame, None, want_det=True)
    n0 = int(counts[0].item())
    assert n0 == spec.n_markers
    markers = _det_to_markers(det[0].cpu().numpy(), n0)
    ids, xy = eng.assign_ids(det, counts, 5, id_mode)
    ids, xy = ids.cpu().numpy().astype(np.int64), xy.cpu().numpy()
    # (1) against the ORACLE (`oracle.process_first_frame`, pinned to the reference body's golden for `as_written` and to the
    #     reference's published figure for `full`): same keys in the same dict order, and every slot holds the coordinates
    #     the oracle gives that key (up to the order among markers at mathematically equal angles, checked below)
    oref = O.process_first_frame(markers, 5, id_mode, "optimal")
    assert [tuple(int(v) for v in k) for k in ids.tolist()] == list(oref.keys())
    okeys = list(oref.keys())
    oxy = np.array([[oref[k]["Ox"], oref[k]["Oy"]] for k in okeys])
    assert sorted(map(tuple, xy.tolist())) == sorted(map(tuple, oxy.tolist()))
    assert int((xy != oxy).any(axis=1).sum()) <= 4
    # (2) against the product's host restatement (the checker of pipeline.reference_from_frame0)
    table = I.assign_ids(markers, 5, id_mode, "optimal")
    want_ids, want_xy = I.reference_arrays(table)
    assert np.array_equal(ids, want_ids)
    # bit-exact float64 coordinates in the same order - except among markers whose angles are mathematically equal
    # (collinear with the centre: np.arctan2's last bit orders them on the host, the device's atan2 here)
    bad = np.where((xy != want_xy).any(axis=1))[0]
    keys = list(table.keys())
    for b in bad:
        tb = table[keys[b]]["angle_rad"]
        twins = [j for j in bad if j != b and keys[j][0] == keys[b][0] and
                 abs(table[keys[j]]["angle_rad"] - tb) <= 4 * np.spacing(abs(tb)) and np.array_equal(xy[b], want_xy[j])]
        assert twins, (keys[b], xy[b], want_xy[b])
    assert len(bad) <= 4
    empty = torch.zeros_like(counts)
    with pytest.raises(ValueError, match="No markers detected"):
        eng.assign_ids(det, empty, 5, id_mode)
    eng.close()


@pytest.mark.parametrize("name", ["ring65", "grid7"])
def test_assign_ids_on_device_against_the_reference_golden(golden_dir, name):
    """f4 against the reference itself: `tests/golden/ids_as_written.json` holds marker lists and the table that the
    reference's own `_process_first_frame` body (marker_detection.py:275-347, executed by make_golden.py) made of them -
    `vbs_assign_ids` on the same markers gives the same keys in the same order with the same (Ox, Oy), bit for bit."""
    g = json.load(open(os.path.join(golden_dir, "ids_as_written.json")))[name]
    markers = g["frames"][0]
    m = len(markers)
    eng = engine(480, 640, max_markers=256, max_batch=1)
    det = torch.zeros((1, 256, 6), dtype=torch.float64, device="cuda")
    det[0, :m] = torch.tensor([[mk["center"][0], mk["center"][1], mk["major_axis"], mk["minor_axis"], mk["angle"], i]
                               for i, mk in enumerate(markers)], dtype=torch.float64)
    counts = torch.tensor([m], dtype=torch.int32, device="cuda")
    ids, xy = eng.assign_ids(det, counts, int(g["num_layers"]), "as_written")
    ids, xy = ids.cpu().numpy().astype(np.int64), xy.cpu().numpy()
    ref = np.array(g["ref"], dtype=np.float64)
    assert np.array_equal(ids, ref[:, :2].astype(np.int64)) and np.array_equal(xy, ref[:, 2:4])
    eng.close()


def test_marker_tracker_process_on_avi(tmp_path):
    """a16 / f4: `MarkerTracker.process()` on a Motion-JPEG AVI (the sensor's recording format) through the package's
    own reader when OpenCV is absent: the CSV equals `process_frames` on the frames that reader decodes."""
    import pandas as pd
    from vbs_amd.marker_detection import MarkerTracker
    from vbs_amd.video_io import AviReader, write_avi
    pytest.importorskip("PIL")
    try:
        import cv2  # noqa: F401
        pytest.skip("OpenCV present: VideoCapture is used, as in the reference")
    except ImportError:
        pass
    spec = S.config1()
    frames = S.make_frames(spec, range(4), seed=6, channels=3)
    path = str(tmp_path / "clip.avi")
    write_avi(path, frames, fps=30.0, codec="MJPG")
    cfg = {"video_path": path, "output_dir": str(tmp_path / "o1"), "crop_ratios": (1 / 8, 1 / 8, 1 / 16, 0),
           "num_layers": 5, "min_marker_distance": 20, "id_mode": "full"}
    t1 = MarkerTracker(cfg)
    t1.process()
    a = pd.read_csv(t1.output_csv, float_precision="round_trip")
    cap = AviReader(path)
    dec = []
    while True:
        ok, f = cap.read()
        if not ok:
            break
        dec.append(f)
    t2 = MarkerTracker({**cfg, "output_dir": str(tmp_path / "o2")})
    rows = t2.process_frames(np.stack(dec))
    b = pd.DataFrame(rows)
    assert len(a) == len(b) > 3 * 40 and list(a.columns) == list(b.columns)
    assert np.array_equal(a.to_numpy(dtype=np.float64), b.to_numpy(dtype=np.float64))
    # the clip went through the native Motion-JPEG decoder (entropy decode on C++ threads, IDCT + colour on the device); the
    # Pillow path gives the same file byte for byte
    assert t1.decode_path == "device"
    t3 = MarkerTracker({**cfg, "output_dir": str(tmp_path / "o3"), "mjpeg_on_device": False})
    t3.process()
    assert t3.decode_path == "pillow"
    assert open(t1.output_csv, "rb").read() == open(t3.output_csv, "rb").read()


def _jpeg_test_frames(h, w, n, seed, gray):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    out = []
    for i in range(n):
        base = 128 + 90 * np.sin(xx / (7.0 + i)) * np.cos(yy / (5.0 + 2 * i))
        img = np.stack([base, 255 - base, 128 + 100 * np.sin((xx + yy) / 11.0)], axis=2)
        img += rng.normal(0, 12 + 8 * i, img.shape)
        img[h // 4:h // 2, w // 3:w // 2] = rng.integers(0, 256, 3)            # hard edges, saturated patches
        img[:6, :9] = 255
        img[-5:, -7:] = 0
        img = np.clip(img, 0, 255).astype(np.uint8)
        out.append(img[:, :, 0] if gray else img)
    return np.stack(out)


@pytest.mark.parametrize("sub", [0, 1, 2, "gray"])
def test_mjpeg_device_decode_equals_libjpeg(tmp_path, sub):
    """f4(c): the native Motion-JPEG decoder (`vbs_mjpeg_entropy_batch` on the host + `vbs_mjpeg_reconstruct` on the device)
    against Pillow's libjpeg on the same chunks, bit for bit in every BGR byte: chroma 4:4:4 / 4:2:2 / 4:2:0 and gray,
    qualities 35-100 (quality 100 = all-ones tables: the widest coefficient range), image sizes that are not multiples of
    the MCU (the partial last MCU row / column, chroma upsampling at the padded edge), images a few pixels wide (libjpeg
    replicates a chroma plane of one or two samples instead of interpolating it), restart intervals, optimised
    Huffman tables, batches that do not divide the clip.  (libjpeg with its defaults - islow IDCT, fancy upsampling - is what
    cv2.imdecode and OpenCV's own MJPEG reader use; cv2.VideoCapture's FFmpeg backend has its own IDCT: unpinned.)"""
    pytest.importorskip("PIL")
    from vbs_amd.video_io import AviReader, MjpegDeviceDecoder, write_avi
    gray = sub == "gray"
    cases = [(q, hw, {}) for q in (35, 75, 95, 100) for hw in ((48, 80), (61, 83), (17, 9))]
    cases += [(q, (h, w), {}) for q in (10, 90) for h in (1, 2, 5, 33) for w in (1, 2, 3, 4, 5)]   # (chroma <= 2 samples wide: replicated)
    cases += [(75, (61, 83), o) for o in (dict(restart_marker_rows=1), dict(restart_marker_blocks=3), dict(optimize=True))]
    cases += [(70, (480, 640), {})]
    for q, (h, w), opts in cases:
        fr = _jpeg_test_frames(h, w, 5, 7 * h + w + q, gray)
        p = str(tmp_path / f"a_{q}_{h}_{len(opts)}.avi")
        write_avi(p, fr, quality=q, subsampling=0 if gray else sub, **opts)
        n, want = AviReader(p).read_batch(5, threads=1)
        dec = MjpegDeviceDecoder(AviReader(p), torch.device("cuda:0"), batch=4, threads=2)
        got, slot = [], 0
        while dec.entropy(slot):
            got.append(dec.reconstruct(slot).cpu().numpy().copy())
            slot ^= 1
        got = np.concatenate(got)
        assert n == 5 and got.shape == want.shape and np.array_equal(got, want), (sub, q, h, w, opts)
    # camera-style frames: no DHT segment, the standard tables are implied (Pillow's libjpeg-turbo installs them too)
    raw = bytearray(open(p, "rb").read())
    rd = AviReader(p)
    for off, size in rd._frames:
        d, i = bytes(raw[off:off + size]), 2
        keep = bytearray(d[:2])
        while d[i + 1] != 0xDA:
            ln = 2 + ((d[i + 2] << 8) | d[i + 3])
            if d[i + 1] != 0xC4:
                keep += d[i:i + ln]
            i += ln
        keep += d[i:]
        assert len(keep) < size
        raw[off:off + size] = bytes(keep) + bytes(size - len(keep))          # (same chunk size: trailing zeros after EOI)
    p2 = str(tmp_path / "nodht.avi")
    open(p2, "wb").write(bytes(raw))
    n, want = AviReader(p2).read_batch(5, threads=1)
    dec = MjpegDeviceDecoder(AviReader(p2), torch.device("cuda:0"), batch=5, threads=1)
    assert dec.entropy(0) == 5 and np.array_equal(dec.reconstruct(0).cpu().numpy(), want)


def test_mjpeg_device_decode_refusals_and_corrupt_frames(tmp_path):
    """What the native decoder does not take is refused at construction (ValueError: the tracker then stays with Pillow),
    and a frame that breaks inside a clip is an IOError naming the frame - with the rows of the batches before it kept."""
    import io
    from PIL import Image
    from vbs_amd.marker_detection import MarkerTracker
    from vbs_amd.video_io import AviReader, MjpegDeviceDecoder, write_avi
    spec = S.config1()
    frames = S.make_frames(spec, range(6), seed=6, channels=3)
    prog = str(tmp_path / "prog.avi")
    write_avi(prog, frames[:2], progressive=True)
    with pytest.raises(ValueError):
        MjpegDeviceDecoder(AviReader(prog), torch.device("cuda:0"), 2)
    raw = str(tmp_path / "raw.avi")
    write_avi(raw, frames[:2], codec="RAW")
    with pytest.raises(ValueError):
        MjpegDeviceDecoder(AviReader(raw), torch.device("cuda:0"), 2)
    cfg = {"crop_ratios": (1 / 8, 1 / 8, 1 / 16, 0), "id_mode": "full", "batch": 2}
    t = MarkerTracker({**cfg, "video_path": prog, "output_dir": str(tmp_path / "op")})
    t.process()
    assert t.decode_path == "pillow" and os.path.getsize(t.output_csv) > 1000
    # frame 4 of 6 loses its header (a scan that merely ends early is padded with zero bits, as libjpeg does with a warning)
    good = str(tmp_path / "good.avi")
    write_avi(good, frames, quality=70)
    b = bytearray(open(good, "rb").read())
    rd = AviReader(good)
    off, size = rd._frames[4]
    b[off:off + 4] = bytes(4)
    bad = str(tmp_path / "bad.avi")
    open(bad, "wb").write(bytes(b))
    t = MarkerTracker({**cfg, "video_path": bad, "output_dir": str(tmp_path / "ob")})
    with pytest.raises(IOError, match="frame 4"):
        t.process()
    import pandas as pd
    kept = pd.read_csv(t.output_csv)
    assert sorted(kept["frameno"].unique()) == [0, 1, 2, 3]


def test_track_markers_method_and_drop_rules():
    """a15 through the reference-shaped method: nearest (first on ties), > min_distance dropped, two
    references may claim one detection."""
    from vbs_amd.marker_detection import MarkerTracker
    trk = MarkerTracker.__new__(MarkerTracker)
    trk.config = {"min_marker_distance": 20}
    trk.frame_count = 7
    trk.crop_height, trk.crop_width = 480, 640
    trk.first_frame_markers = {(0, 0): {"Ox": 100.0, "Oy": 100.0}, (1, 0): {"Ox": 130.0, "Oy": 100.0},
                               (1, 1): {"Ox": 400.0, "Oy": 300.0}, (2, 0): {"Ox": 115.0, "Oy": 100.0}}
    markers = [{"center": (110.0, 100.0), "major_axis": 20.0, "minor_axis": 19.0, "angle": 90.0},
               {"center": (120.0, 100.0), "major_axis": 21.0, "minor_axis": 19.5, "angle": 91.0},
               {"center": (400.0, 321.0), "major_axis": 22.0, "minor_axis": 18.0, "angle": 92.0}]
    got = trk._track_markers(np.zeros((480, 640, 3), np.uint8), markers)
    want = O.track_markers(trk.first_frame_markers, markers, 7, 20)
    assert got == want
    assert [(r["row"], r["col"]) for r in got] == [(0, 0), (1, 0), (2, 0)]      # (1,1) is 21 px away
    assert got[2]["Cx"] == 110.0                                                  # tie 5 px / 5 px -> first
    assert trk._track_markers(None, []) == []
    with pytest.raises(ValueError):
        MarkerTracker._process_first_frame(trk, [])


# ------------------------------------------------------------------------------------------------
def test_points_api_against_reference_goldens(golden_dir):
    """a19/a20 float64 interfaces against outputs of the reference's `_calculate_3d_position`."""
    from vbs_amd.reconstruction3d import MarkerAnalysis, Config
    g = json.load(open(os.path.join(golden_dir, "solve3d.json")))
    for cname in ("cam_a", "cam_b"):
        cam = g[cname]["cam"]
        ma = MarkerAnalysis.__new__(MarkerAnalysis)
        ma.config = Config()
        from vbs_amd.reconstruction3d import CameraParameters
        ma.camera = CameraParameters()
        ma.set_camera(cam["K"], np.zeros(5), cam["R"], cam["T"])
        for u, v, d, x, y, z in g[cname]["pts"]:
            if x is None:
                with pytest.raises(ValueError):
                    ma._calculate_3d_position(u, v, d)
            else:
                p = ma._calculate_3d_position(u, v, d)
                np.testing.assert_allclose(p, [x, y, z], rtol=1e-12, atol=1e-12)


def test_undistort_points_with_distortion():
    from vbs_amd.engine import undistort_points
    K = np.array([[912.25, 0, 331.5], [0, 915.75, 236.125], [0, 0, 1]], dtype=np.float32)
    dist = np.array([-0.21, 0.07, 0.0013, -0.0009, -0.011], dtype=np.float32)
    cam = L.make_camera(K, dist, np.eye(3), np.zeros(3))
    rng = np.random.default_rng(1)
    pts = rng.uniform([0, 0], [640, 480], size=(500, 2))
    got = undistort_points(pts, cam).cpu().numpy()
    want = O.undistort_points(pts, K, dist)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    assert np.abs(got - pts).max() > 1.0
    cam0 = L.make_camera(K, np.zeros(5), np.eye(3), np.zeros(3))
    np.testing.assert_allclose(undistort_points(pts, cam0).cpu().numpy(), pts, rtol=0, atol=1e-10)


def test_marker_analysis_track_markers_golden(golden_dir):
    """a21 through `MarkerAnalysis._track_markers(df)` against the reference body's output: gap frame,
    warm-up, > 50 mm rejection."""
    import pandas as pd
    from vbs_amd.reconstruction3d import MarkerAnalysis, Config, CameraParameters
    g = json.load(open(os.path.join(golden_dir, "solve3d.json")))["disp"]
    ma = MarkerAnalysis.__new__(MarkerAnalysis)
    ma.config = Config(warmup_frames=g["warmup"], max_displacement_px=g["limit"])
    ma.camera = CameraParameters()
    ma.set_camera(g["cam"]["K"], np.zeros(5), g["cam"]["R"], g["cam"]["T"])
    df = pd.DataFrame(g["rows_in"])
    out = ma._track_markers(df)
    cols = ["frameno", "row", "col", "X", "Y", "Z", "dX", "dY", "dZ", "displacement"]
    assert list(out.columns) == cols
    got = out.to_numpy(dtype=np.float64)
    want = np.array(g["rows_out"])
    assert got.shape == want.shape
    key = lambda a: np.lexsort((a[:, 2], a[:, 1], a[:, 0]))      # noqa: E731
    got, want = got[key(got)], want[key(want)]
    np.testing.assert_array_equal(got[:, :3], want[:, :3])
    np.testing.assert_allclose(got[:, 3:], want[:, 3:], rtol=0, atol=1e-9)


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,nframes", [("c2", 6), ("c5", 2)])
def test_fused_track_to_3d_displacement_plane(tag, nframes):
    """Whole path frames -> table -> displacement -> plane fit against the oracle (IDs exact)."""
    from vbs_amd.pipeline import track_shard
    spec = S.config2() if tag == "c2" else S.config5()
    frames = S.make_frames(spec, range(nframes), seed=13)
    eng = engine(spec.height, spec.width, max_markers=1024, max_batch=4)
    K, dist, R, T = S.default_camera(spec)
    dist = np.array([-0.05, 0.01, 0.0005, -0.0003, 0.0], dtype=np.float32)
    cam = L.make_camera(K, dist, R, T, 2.0)
    warm = 1 if nframes > 2 else 0
    res = track_shard(eng, torch.from_numpy(frames).cuda(), nframes, cam=cam, warmup_frames=warm, id_mode="full")
    rows, ref = O.process_frames(list(frames), id_mode="full")
    assert [tuple(k) for k in res.ids.tolist()] == list(ref.keys())
    table = res.table.cpu().numpy()
    disp = res.disp.cpu().numpy()
    assert int((table[..., 0].astype(int) & 1).sum()) == len(rows)
    slot = {k: i for i, k in enumerate(ref.keys())}
    for r in rows:
        t = table[r["frameno"], slot[(r["row"], r["col"])]]
        assert int(t[0]) & 1
        assert abs(t[1] - r["Cx"]) <= TOL_XY and abs(t[2] - r["Cy"]) <= TOL_XY
        assert abs(t[3] - r["major_axis"]) <= TOL_AX and abs(t[4] - r["minor_axis"]) <= TOL_AX
    rows3 = O.track_markers_3d(rows, K, dist, R, T, warmup_frames=warm)
    assert int(disp[..., 0].sum()) == len(rows3) and len(rows3) > 0
    for r in rows3:
        s = slot[(r["row"], r["col"])]
        t, d = table[r["frameno"], s], disp[r["frameno"], s]
        assert d[0] == 1
        assert np.max(np.abs(t[6:9] - [r["X"], r["Y"], r["Z"]])) <= TOL_XYZ
        assert np.max(np.abs(d[1:5] - [r["dX"], r["dY"], r["dZ"], r["displacement"]])) <= TOL_XYZ
    plane = res.plane.cpu().numpy()
    for f in range(nframes):
        v = (table[f, :, 0].astype(int) & 2) > 0
        a, b, c, tilt = O.fit_plane(table[f, v, 6].astype(np.float64), table[f, v, 7].astype(np.float64),
                                    table[f, v, 8].astype(np.float64))
        assert plane[f, 0] == v.sum()
        np.testing.assert_allclose(plane[f, 1:], [a, b, c, tilt], rtol=2e-4, atol=2e-5)
    eng.close()


def test_deviation_plane_against_the_oracle():
    """f1 (`ForceDistribution.py:168-208,218-243`): deviation field of a tilted against a vertical loading, plane over
    its end points, tilt, mean vector and mean magnitude, on two synthetic sessions over the 65-marker ring layout with
    markers missing from either session; both Z modes and a scale factor.  Deviations: float32 rounding of the float64
    differences; plane / tilt within 2e-4 relative of `np.linalg.lstsq` (float32 output)."""
    from vbs_amd.pipeline import deviation_pose
    rng = np.random.default_rng(9)
    rings = [(0, 1), (3.4, 6), (6.8, 12), (10.2, 18), (13.4, 24), (16.3, 4)]
    ref = np.array([[r * np.cos(2 * np.pi * k / n_), r * np.sin(2 * np.pi * k / n_), 0.02 * r * r] for r, n_ in rings for k in range(n_)])
    m = ref.shape[0]
    assert m == 65

    def session(tilt_deg, drop):
        tab = np.zeros((2, m, L.TABLE_COLS), np.float32)
        start = ref + rng.normal(0, 0.02, (m, 3))
        end = start + np.column_stack([np.zeros(m), 0.01 * start[:, 1], -0.6 - np.tan(np.radians(tilt_deg)) * start[:, 0]])
        end += rng.normal(0, 0.01, (m, 3))
        for f, xyz in enumerate((start, end)):
            tab[f, :, 0] = L.FLAG_TRACKED | L.FLAG_XYZ
            tab[f, :, 6:9] = xyz
        tab[1, drop, 0] = L.FLAG_TRACKED                          # tracked in 2-D, but no 3-D point in the end frame
        return tab

    tv, tt = session(0.0, [5, 30]), session(4.0, [30, 31, 64])
    eng = engine(480, 640)
    four = lambda row: np.column_stack([((row[:, 0].astype(int) & L.FLAG_XYZ) != 0).astype(float), row[:, 6:9].astype(np.float64)])
    for mode in ("plane", "shell"):
        for scale in (1.0, 5.0):
            res = deviation_pose(eng, torch.from_numpy(tv).cuda(), torch.from_numpy(tt).cuda(), ref, mode, scale)
            common, dev, plane, mean_vec, mean_mag = O.deviation_plane(four(tv[0]), four(tv[1]), four(tt[0]), four(tt[1]),
                                                                      ref.astype(np.float32), mode, scale)
            got = res["deviation"].cpu().numpy()
            assert res["n"] == int(common.sum()) == m - 4
            assert np.array_equal(got[:, 0] != 0, common)
            assert np.array_equal(got[:, 1:4], dev.astype(np.float32))
            np.testing.assert_allclose([res["a"], res["b"], res["c"], res["tilt_deg"]], plane, rtol=2e-4, atol=2e-5)
            np.testing.assert_allclose(res["mean_vector"], mean_vec, rtol=2e-4, atol=1e-6)
            assert abs(res["mean_magnitude"] - mean_mag) <= 2e-4 * mean_mag
            if mode == "plane" and scale == 1.0:
                assert abs(res["tilt_deg"] - 4.0) < 0.3           # the synthetic misalignment comes back
    eng.close()


def test_displacement_range_and_gaps():
    """a21 on a synthetic table with gaps, invalid 3-D rows, a jump and a warm-up: the chunk-parallel kernel
    equals a plain sequential restatement of `3d_reconstruction.py:263-314`, and a rank's frame range of the
    gathered table equals the same rows of the full result (look-back across the range start)."""
    rng = np.random.default_rng(0)
    n, m = 301, 37
    tab = np.zeros((n, m, 10), dtype=np.float32)
    present = rng.random((n, m)) < 0.8
    present[:3] = False
    present[40:120, 5] = False                      # long gap for one ID
    ok = rng.random((n, m)) < 0.95
    tab[..., 0] = present * (1 + 2 * ok)
    tab[..., 3] = np.where(rng.random((n, m)) < 0.03, 4.0, 20.0)      # some rows fail the size filter
    xyz = np.cumsum(rng.normal(0, 0.3, (n, m, 3)), axis=0) + 30
    xyz[200, 7] += 80.0                            # > 50 mm jump
    tab[..., 6:9] = xyz
    warm, minsz, lim = 10, 5.0, 50.0
    want = np.zeros((n, m, 5), dtype=np.float64)
    seen = (tab[..., 0].astype(int) & 1 > 0) & (tab[..., 3] >= minsz)
    fmin = int(np.nonzero(seen.any(1))[0][0])
    for r in range(m):
        last = None
        for f in range(fmin + warm, n):
            if not seen[f, r]:
                continue
            good = int(tab[f, r, 0]) & 2
            cur = tab[f, r, 6:9].astype(np.float64)
            if last is not None and last[0] and good:
                d = cur - last[1]
                mm = np.sqrt((d * d).sum())
                if not mm > lim:
                    want[f, r] = [1, d[0], d[1], d[2], mm]
            last = (good, cur)
    eng = engine(480, 640)
    tt = torch.from_numpy(tab).cuda()
    full = eng.displacement(tt, warm, minsz, lim).cpu().numpy()
    assert np.array_equal(full[..., 0], want[..., 0])
    np.testing.assert_allclose(full[..., 1:], want[..., 1:], rtol=1e-6, atol=1e-6)
    assert want[200, 7, 0] == 0 and want[..., 0].sum() > 1000
    for a, b in ((0, 64), (64, 200), (150, 301), (117, 123)):
        part = eng.displacement(tt, warm, minsz, lim, frame_range=(a, b)).cpu().numpy()
        assert np.array_equal(part, full[a:b])
    eng.close()


def test_batch_and_chunk_independence():
    """Size-independent property at batch scale: a frame's table row does not depend on the batch it
    travels in, nor on the engine's internal chunking (frames are independent units)."""
    spec = S.config2()
    n = 24
    ft = S.make_frames_torch(spec, range(n), seed=2, device="cuda")
    eng_a = engine(spec.height, spec.width, max_batch=16)
    eng_b = engine(spec.height, spec.width, max_batch=5)
    from vbs_amd.pipeline import reference_from_frame0
    ids, xy = reference_from_frame0(eng_a, ft)
    K, dist, R, T = S.default_camera(spec)
    cam = L.make_camera(K, dist, R, T)
    ta, _, _ = eng_a.track_to_3d(ft, xy, 20.0, cam)
    tb, _, _ = eng_b.track_to_3d(ft, xy, 20.0, cam)
    assert torch.equal(ta, tb)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(0))
    tc, _, _ = eng_b.track_to_3d(ft[perm.cuda()].contiguous(), xy, 20.0, cam)
    assert torch.equal(tc, ta[perm.cuda()])
    assert int((ta[..., 0].int() & 1).sum()) == n * spec.n_markers
    # one frame per call and per internal pass (MarkerTracker.process, marker_detection.py:434-453: the several-workgroups
    # labelling kernel and the one-launch finalize + track): the same rows
    eng_1 = engine(spec.height, spec.width, max_batch=1)
    t1, _, _ = eng_1.track_to_3d(ft, xy, 20.0, cam)
    assert torch.equal(t1, ta)
    for i in (0, 7, 23):
        ti, _, _ = eng_1.track_to_3d(ft[i:i + 1], xy, 20.0, cam)
        assert torch.equal(ti[0], ta[i])
    eng_1.close()
    # torch-rendered frames are the same bytes as the NumPy renderer's
    assert np.array_equal(ft[3].cpu().numpy(), S.make_frames(spec, [3], seed=2)[0])
    eng_a.close()
    eng_b.close()
    # a pass of 112 frames: the matrix-core kernels then walk whole columns in one piece (the benchmark's launch
    # shape), small passes split them into row segments that each re-run the filter ramp: same detections
    n2 = 112
    f2 = S.make_frames_torch(spec, range(n2), seed=2, device="cuda")
    eng_c = engine(spec.height, spec.width, max_batch=n2)
    eng_d = engine(spec.height, spec.width, max_batch=3)
    _, dc, cc = eng_c.track_to_3d(f2, None, want_det=True)
    _, dd, cd = eng_d.track_to_3d(f2, None, want_det=True)
    assert torch.equal(cc, cd) and torch.equal(dc, dd) and int(cc.min()) == spec.n_markers
    eng_c.close()
    eng_d.close()


def test_benchmark_launch_shape_properties():
    """BASELINE config 3 at its real size (4096 resident 1280x1024 frames = 5.4 GB, internal passes of 512 frames, frame
    offsets beyond 4 GB) through size-independent properties: every frame finds its 169 markers, every observation gets
    its 3-D point, the table does not depend on the pass size (512 / 96, on the first 1024 frames), BGR frames with
    B = G = R give the gray frames' table, and the NCC decision counters report (next to) no ambiguous pixel."""
    spec = S.config2()
    n = 4096
    ft = S.make_frames_torch(spec, range(n), seed=4, device="cuda", chunk=16)
    from vbs_amd.pipeline import reference_from_frame0
    eng = engine(spec.height, spec.width, max_batch=512)
    ids, xy = reference_from_frame0(eng, ft[:1], 5, "full", "optimal")
    cam = L.make_camera(*S.default_camera(spec), 2.0)
    eng.ncc_counters(reset=True)
    t512, _, c512 = eng.track_to_3d(ft, xy, 20.0, cam, 5.0)
    cnt = eng.ncc_counters()
    assert int(c512.min()) == spec.n_markers and int(c512.max()) == spec.n_markers
    flags = t512[..., 0].int()
    assert int((flags & 1).sum()) == n * spec.n_markers          # tracked
    assert int(((flags >> 1) & 1).sum()) == n * spec.n_markers   # 3-D solved
    # "ambiguous" = an NCC value within 1e-9 (relative) of the 0.1 threshold: there the reference's own FFT rounding
    # (up to 5e-8) decides, so no implementation can promise its bit; 5.4e9 pixels may hold one or two of them
    assert cnt["frames"] == n and cnt["ambiguous"] <= 2, cnt
    # the last frames of the batch (addresses past 4 GB) agree with the same frames processed on their own
    tl, _, _ = eng.track_to_3d(ft[n - 8:], xy, 20.0, cam, 5.0)
    assert torch.equal(tl, t512[n - 8:])
    tb, _, _ = eng.track_to_3d(ft[:600].unsqueeze(-1).expand(-1, -1, -1, 3).contiguous(), xy, 20.0, cam, 5.0)
    assert torch.equal(tb, t512[:600])
    eng.close()
    eng2 = engine(spec.height, spec.width, max_batch=96)
    t96, _, c96 = eng2.track_to_3d(ft[:1024], xy, 20.0, cam, 5.0)
    assert torch.equal(t96, t512[:1024]) and torch.equal(c96, c512[:1024])
    eng2.close()


def test_filtered_mask_equals_float64_map_over_512_frames():
    """The statistical check of `k_ncc_mfma`'s float16 / float32 filter margin (2e-5): over 512 frames (6.7e8 pixels) the
    mask of `find_markers` equals `ncc_map > 0.1` from the float64 kernel (itself held to the reference's FFT at 1e-6 by
    `test_ncc_map_matches_fft_reference`) bit for bit; the only pixels excused are those the float64 value itself puts
    within 1e-9 of the threshold (`_find_markers` :133)."""
    spec = S.config2()
    n, step = 512, 64
    eng = engine(spec.height, spec.width, max_batch=step)
    bad = near = 0
    for f0 in range(0, n, step):
        ft = S.make_frames_torch(spec, range(f0, f0 + step), seed=11, device="cuda", chunk=16)
        mask, _ = eng.find_markers(ft)
        ncc = eng.ncc_map(ft)
        diff = (ncc > 0.1) != (mask != 0)
        if bool(diff.any()):
            d = (ncc[diff] - 0.1).abs()
            near += int((d <= 1e-10).sum())
            bad += int((d > 1e-10).sum())
        del ncc, mask, ft
    assert bad == 0 and near <= 2, (bad, near)
    eng.close()


def test_fused_stage_equals_separate_kernels():
    """The fused labelling kernel (k_stage: band / opening / components / sums in one launch) against the separate kernels
    other geometries take (VBS_OPT_STAGE_IMPL = 1: k_morph + k_ccl): every per-component table and every detection row
    identical, on marker frames of all three sizes, on crops and on ragged blobs that send frames to the general kernel."""
    from vbs_amd.engine import Engine

    def both(eng, run, n):
        out = []
        for impl in (0, 1):
            eng.set_option(L.OPT_STAGE_IMPL, impl)
            res = run()
            torch.cuda.synchronize()
            out.append((res, eng.stage_tables(n)))
        eng.set_option(L.OPT_STAGE_IMPL, 0)
        return out

    def same_tables(t0, t1):
        for i in range(t0["ncomp"].shape[0]):
            nb, na = (int(v) for v in t0["ncomp"][i])
            assert (nb, na) == tuple(int(v) for v in t1["ncomp"][i])
            assert np.array_equal(t0["band_sums"][i][:nb, :3], t1["band_sums"][i][:nb, :3])
            assert np.array_equal(t0["area_first"][i][:na], t1["area_first"][i][:na])
            assert np.array_equal(t0["area_sums"][i][:na, :15], t1["area_sums"][i][:na, :15])
            assert np.array_equal(t0["probe"][i][:nb], t1["probe"][i][:nb])

    for tag, crop in (("c1", None), ("c2", None), ("c2", (64, 1024, 160, 1120)), ("c5", None)):
        spec = {"c1": S.config1, "c2": S.config2, "c5": S.config5}[tag]()
        n = 3
        ft = S.make_frames_torch(spec, range(n), seed=3, device="cuda")
        if crop:
            ft = ft[:, crop[0]:crop[1], crop[2]:crop[3]]
        eng = Engine(ft.shape[1], ft.shape[2], max_markers=1024 if tag == "c5" else 512, max_batch=n)
        ((_, d0, c0), t0), ((_, d1, c1), t1) = both(eng, lambda: eng.track_to_3d(ft, want_det=True), n)
        assert int(t0["slow"].sum()) == 0, "a marker frame left the fused path"
        same_tables(t0, t1)
        assert torch.equal(c0, c1) and torch.equal(d0, d1) and int(c0.min()) > 0
        eng.close()
    rng = np.random.default_rng(7)
    for (h, w) in ((450, 480), (700, 900), (1000, 1200)):
        n = 4
        mask = np.zeros((n, h, w), np.uint8); area = np.zeros((n, h, w), np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        for f in range(n):
            for _ in range(int(rng.integers(5, 50))):
                cx, cy = rng.uniform(0, w), rng.uniform(0, h)
                a, b, th = rng.uniform(4, 40), rng.uniform(4, 40), rng.uniform(0, np.pi)
                u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
                area[f][(u / a) ** 2 + (v / b) ** 2 <= 1] = 255
                mask[f][(u / (0.7 * a)) ** 2 + (v / (0.7 * b)) ** 2 <= 1] = 1
        eng = Engine(h, w, max_markers=512, max_batch=n)
        mt, at = torch.from_numpy(mask).cuda(), torch.from_numpy(area).cuda()
        ((d0, c0), t0), ((d1, c1), t1) = both(eng, lambda: eng.marker_center(mt, at), n)
        keep = (t0["slow"] == 0) & (t1["slow"] == 0)          # frames neither path handed to the general kernel
        same_tables({k: v[keep] for k, v in t0.items()}, {k: v[keep] for k, v in t1.items()})
        assert torch.equal(c0, c1) and torch.equal(d0, d1)
        eng.close()


def test_stage_256_threads_equals_768(golden_dir):
    """Small frames (the reference's 480x450 crop, 640x480) label a frame with the 256-thread instance of k_stage (tiles of
    15-20 rows, three workgroups per CU, smaller tables) instead of the 768-thread one (tiles of 5-7 rows under 13 / 10 rows
    of halo); so does a pass of >= 512 large frames (tiles of 86 rows at 1280x1024), here forced for a few frames with
    VBS_OPT_STAGE_IMPL = 4.  VBS_OPT_STAGE_IMPL = 3 keeps 768: every per-component table, detection row and count identical - marker
    frames, the reference's real frame, ragged blobs, and the adverse patterns that overflow the (smaller) tables and go to
    the general kernel from either shape."""
    from vbs_amd.engine import Engine

    def both(eng, run, n):
        out = []
        eng.set_option(L.OPT_LATENCY_FRAMES, 0)              # the batch kernel also for these few frames
        for impl in (4, 3):                                  # (4: 256 threads also for a pass of a few LARGE frames)
            eng.set_option(L.OPT_STAGE_IMPL, impl)
            res = run()
            torch.cuda.synchronize()
            out.append((res, eng.stage_tables(n)))
        eng.set_option(L.OPT_STAGE_IMPL, 0)
        eng.set_option(L.OPT_LATENCY_FRAMES, 24)
        return out

    def same_tables(t0, t1):
        for i in range(t0["ncomp"].shape[0]):
            nb, na = (int(v) for v in t0["ncomp"][i])
            assert (nb, na) == tuple(int(v) for v in t1["ncomp"][i]), i
            assert np.array_equal(t0["band_sums"][i][:nb, :3], t1["band_sums"][i][:nb, :3]), i
            assert np.array_equal(t0["area_first"][i][:na], t1["area_first"][i][:na]), i
            assert np.array_equal(t0["area_sums"][i][:na, :15], t1["area_sums"][i][:na, :15]), i
            assert np.array_equal(t0["probe"][i][:nb], t1["probe"][i][:nb]), i

    for tag, crop, n in (("c1", None, 6), ("c1", (15, 465, 80, 560), 5), ("c1", (0, 300, 0, 640), 3), ("c1", (0, 480, 0, 250), 3),
                         ("c2", None, 3), ("c2", (64, 1024, 160, 1120), 2), ("c2", (0, 700, 100, 1000), 2)):
        spec = {"c1": S.config1, "c2": S.config2}[tag]()
        ft = S.make_frames_torch(spec, range(n), seed=4, device="cuda")
        if crop:
            ft = ft[:, crop[0]:crop[1], crop[2]:crop[3]]
        eng = Engine(ft.shape[1], ft.shape[2], max_markers=512, max_batch=n)
        ((_, d0, c0), t0), ((_, d1, c1), t1) = both(eng, lambda: eng.track_to_3d(ft, want_det=True), n)
        assert int(t0["slow"].sum()) == 0 and int(t1["slow"].sum()) == 0, (crop, t0["slow"], t1["slow"])
        same_tables(t0, t1)
        assert torch.equal(c0, c1) and torch.equal(d0, d1) and int(c0.min()) > 0
        eng.close()
    # a dense layout (17 x 17 dots at a pitch of 56 px: more segments per 86-row tile and more records than the 256-thread
    # instance holds): its frames get a second chance on 768 threads IN THE SAME PASS instead of the general kernels - not one
    # frame handed on, the same tables
    spec = S.grid_spec(1280, 1024, 17, 56, 30, name="dense", noise_sigma=2.0)
    ft = S.make_frames_torch(spec, range(3), seed=1, device="cuda")
    eng = Engine(1024, 1280, max_markers=512, max_batch=3)
    ((_, d0, c0), t0), ((_, d1, c1), t1) = both(eng, lambda: eng.track_to_3d(ft, want_det=True), 3)
    assert int(t0["slow"].sum()) == 0 and int(t1["slow"].sum()) == 0, (t0["slow"], t1["slow"])
    same_tables(t0, t1)
    assert torch.equal(c0, c1) and torch.equal(d0, d1) and int(c0.min()) == 289
    eng.close()
    # the reference's real frame, as it comes (467x437) and put back into the 480x450 crop frame it was cut from
    bgr = np.load(os.path.join(golden_dir, "raw_markers_bgr.npz"))["bgr"]
    full = np.zeros((450, 480, 3), np.uint8); full[6:6 + bgr.shape[0], 6:6 + bgr.shape[1]] = bgr
    for img in (bgr, full):
        ft = torch.from_numpy(np.stack([np.roll(img, (dy, dx), (0, 1)) for dy, dx in ((0, 0), (2, -3), (-1, 1), (3, 3))])).cuda()
        eng = Engine(img.shape[0], img.shape[1], max_markers=1024, max_batch=4)
        ((_, d0, c0), t0), ((_, d1, c1), t1) = both(eng, lambda: eng.track_to_3d(ft, want_det=True), 4)
        # (the bare still stays on the fast path; pasted onto black its cut edge makes holes in some shifts: general kernel, from
        #  either shape alike)
        assert np.array_equal(t0["slow"], t1["slow"]) and (img is full or int(t0["slow"].sum()) == 0), (img.shape, t0["slow"], t1["slow"])
        keep = t0["slow"] == 0
        same_tables({k: v[keep] for k, v in t0.items()}, {k: v[keep] for k, v in t1.items()})
        assert torch.equal(c0, c1) and torch.equal(d0, d1) and int(c0[1]) == 65
        eng.close()
    rng = np.random.default_rng(13)
    for (h, w) in ((450, 480), (480, 640), (300, 200), (1024, 1280), (700, 900)):
        n = 6
        mask = np.zeros((n, h, w), np.uint8); area = np.zeros((n, h, w), np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        for f in range(n):
            for _ in range(int(rng.integers(3, 14))):
                cx, cy = rng.uniform(0, w), rng.uniform(0, h)
                a, b, th = rng.uniform(4, 22), rng.uniform(4, 22), rng.uniform(0, np.pi)
                u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
                area[f][(u / a) ** 2 + (v / b) ** 2 <= 1] = 255
                mask[f][(u / (0.7 * a)) ** 2 + (v / (0.7 * b)) ** 2 <= 1] = 1
        pats = [np.ones((h, w), bool), (xx // 64) % 2 == 0, (yy // 8) % 2 == 0, ((xx // 8) + (yy // 8)) % 2 == 0,
                rng.random((h, w)) < 0.5, ((xx % 12) < 6) & ((yy % 97) > 5)]
        area = np.concatenate([area, np.stack([(p * 255).astype(np.uint8) for p in pats])])
        mask = np.concatenate([mask, np.stack([p.astype(np.uint8) for p in pats])])
        n = area.shape[0]
        eng = Engine(h, w, max_markers=1024, max_batch=n)
        mt, at = torch.from_numpy(mask).cuda(), torch.from_numpy(area).cuda()
        ((d0, c0), t0), ((d1, c1), t1) = both(eng, lambda: eng.marker_center(mt, at), n)
        assert torch.equal(c0, c1) and torch.equal(d0, d1), (h, w, c0.tolist(), c1.tolist())
        keep = (t0["slow"] == 0) & (t1["slow"] == 0)          # frames neither shape handed to the general kernel
        assert int(keep[:6].sum()) > 0, (h, w, t0["slow"], t1["slow"])
        same_tables({k: v[keep] for k, v in t0.items()}, {k: v[keep] for k, v in t1.items()})
        eng.close()


def test_latency_stage_equals_the_batch_stage():
    """A pass of a few frames labels every frame with SEVERAL workgroups (k_stage_lat: tiles of ~8 rows, the shared tables
    in global memory, the last workgroup of a frame to arrive resolves it) instead of one (k_stage).  VBS_OPT_LATENCY_FRAMES = 32
    against 0 on the same inputs: every per-component table, every detection row and every count identical - marker frames of
    the three sizes one and several per call, crops, ragged blobs, and the adverse patterns (whatever either kernel cannot
    take goes to the general kernel with the same result)."""
    from vbs_amd.engine import Engine

    def both(eng, run, n):
        out = []
        for lat in (32, 0):
            eng.set_option(L.OPT_LATENCY_FRAMES, lat)
            res = run()
            torch.cuda.synchronize()
            out.append((res, eng.stage_tables(n)))
        eng.set_option(L.OPT_LATENCY_FRAMES, 24)
        return out

    def same_tables(t0, t1):
        for i in range(t0["ncomp"].shape[0]):
            nb, na = (int(v) for v in t0["ncomp"][i])
            assert (nb, na) == tuple(int(v) for v in t1["ncomp"][i]), i
            assert np.array_equal(t0["band_sums"][i][:nb, :3], t1["band_sums"][i][:nb, :3]), i
            assert np.array_equal(t0["area_first"][i][:na], t1["area_first"][i][:na]), i
            assert np.array_equal(t0["area_sums"][i][:na, :15], t1["area_sums"][i][:na, :15]), i
            assert np.array_equal(t0["probe"][i][:nb], t1["probe"][i][:nb]), i

    for tag, crop, n in (("c1", None, 1), ("c1", None, 5), ("c2", None, 1), ("c2", None, 4), ("c2", (64, 1024, 160, 1120), 2),
                         ("c2", (0, 450, 0, 480), 3), ("c5", None, 1), ("c5", None, 3), ("c2", None, 19)):
        spec = {"c1": S.config1, "c2": S.config2, "c5": S.config5}[tag]()
        ft = S.make_frames_torch(spec, range(n), seed=5, device="cuda")
        if crop:
            ft = ft[:, crop[0]:crop[1], crop[2]:crop[3]]
        eng = Engine(ft.shape[1], ft.shape[2], max_markers=1024 if tag == "c5" else 512, max_batch=n)
        ((_, d0, c0), t0), ((_, d1, c1), t1) = both(eng, lambda: eng.track_to_3d(ft, want_det=True), n)
        assert np.array_equal(t0["slow"], t1["slow"]), (tag, crop, t0["slow"], t1["slow"])
        if crop is None:
            assert int(t0["slow"].sum()) == 0, ("a marker frame left the several-workgroups path", tag, t0["slow"])
        keep = t0["slow"] == 0                               # (the 450x480 corner of large-marker frames: holes, general kernel)
        same_tables({k: v[keep] for k, v in t0.items()}, {k: v[keep] for k, v in t1.items()})
        assert torch.equal(c0, c1) and torch.equal(d0, d1) and int(c0.min()) > 0
        eng.close()
    # with reference positions (the tracking rows come out of the same launch as the fits), on frames the labelling kernel hands
    # on to k_label (the 450x480 corner of large-marker frames: holes)
    spec = S.config2()
    ft = S.make_frames_torch(spec, range(5), seed=5, device="cuda")[:, 0:450, 0:480]
    eng = Engine(450, 480, max_markers=512, max_batch=5)
    _, d_ref, c_ref = eng.track_to_3d(ft[:1], None, want_det=True)
    xy = d_ref[0, :int(c_ref[0]), :2].cpu().numpy()
    cam = L.make_camera(*S.default_camera(spec), 2.0)
    ((ta, da, ca), t0), ((tb, db, cb), t1) = both(eng, lambda: eng.track_to_3d(ft, xy, 20.0, cam, 5.0, want_det=True), 5)
    assert int((t0["slow"] != 0).sum()) > 0 and np.array_equal(t0["slow"], t1["slow"])
    assert torch.equal(ta, tb) and torch.equal(da, db) and torch.equal(ca, cb) and int(ca.min()) > 0
    eng.close()
    rng = np.random.default_rng(11)
    for (h, w) in ((450, 480), (700, 900), (1000, 1200)):
        n = 6
        mask = np.zeros((n, h, w), np.uint8); area = np.zeros((n, h, w), np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        for f in range(n):
            for _ in range(int(rng.integers(5, 50))):
                cx, cy = rng.uniform(0, w), rng.uniform(0, h)
                a, b, th = rng.uniform(4, 40), rng.uniform(4, 40), rng.uniform(0, np.pi)
                u = (xx - cx) * np.cos(th) + (yy - cy) * np.sin(th); v = -(xx - cx) * np.sin(th) + (yy - cy) * np.cos(th)
                area[f][(u / a) ** 2 + (v / b) ** 2 <= 1] = 255
                mask[f][(u / (0.7 * a)) ** 2 + (v / (0.7 * b)) ** 2 <= 1] = 1
        eng = Engine(h, w, max_markers=512, max_batch=n)
        mt, at = torch.from_numpy(mask).cuda(), torch.from_numpy(area).cuda()
        ((d0, c0), t0), ((d1, c1), t1) = both(eng, lambda: eng.marker_center(mt, at), n)
        keep = (t0["slow"] == 0) & (t1["slow"] == 0)          # frames neither kernel handed to the general one
        assert int(keep.sum()) > 0
        same_tables({k: v[keep] for k, v in t0.items()}, {k: v[keep] for k, v in t1.items()})
        assert torch.equal(c0, c1) and torch.equal(d0, d1)
        eng.close()
    for (h, w) in ((480, 640), (1024, 1280)):
        yy, xx = np.mgrid[0:h, 0:w]
        pats = [np.ones((h, w), bool), np.zeros((h, w), bool), (xx // 64) % 2 == 0, (yy // 8) % 2 == 0,
                ((xx // 8) + (yy // 8)) % 2 == 0, rng.random((h, w)) < 0.5, ((xx % 12) < 6) & ((yy % 97) > 5),
                (((xx - w // 2) ** 2 + (yy - h // 2) ** 2) < (min(h, w) // 2 - 3) ** 2)]
        n = len(pats)
        area = np.stack([(p * 255).astype(np.uint8) for p in pats]); mask = np.stack([p.astype(np.uint8) for p in pats])
        eng = Engine(h, w, max_markers=1024, max_batch=n)
        mt, at = torch.from_numpy(mask).cuda(), torch.from_numpy(area).cuda()
        ((d0, c0), t0), ((d1, c1), t1) = both(eng, lambda: eng.marker_center(mt, at), n)
        assert torch.equal(c0, c1), (c0.tolist(), c1.tolist())
        for i in range(n):
            k = max(int(c0[i]), 0)
            assert torch.equal(d0[i, :k], d1[i, :k]), i
        eng.close()


@pytest.mark.parametrize("shape", [(480, 640), (1024, 1280)])
def test_stage_on_adverse_patterns(shape):
    """Inputs far from marker frames - all ones, stripes one tile wide, blocks touching at their corners, dense noise, a comb
    with hundreds of segments per column - through the fused labelling kernel and through the separate kernels: whatever
    the fused kernel cannot take it must hand to the general kernel (never hang, never differ): same counts (or the same
    capacity status) and the same detection rows."""
    from vbs_amd.engine import Engine
    h, w = shape
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:h, 0:w]
    pats = [np.ones((h, w), bool), np.zeros((h, w), bool),
            (xx // 64) % 2 == 0, (yy // 29) % 2 == 0,                      # stripes as wide / tall as a tile
            ((xx // 8) + (yy // 8)) % 2 == 0,                               # 8x8 blocks touching at their corners
            rng.random((h, w)) < 0.5, rng.random((h, w)) < 0.9,             # noise
            ((xx % 12) < 6) & ((yy % 97) > 5),                               # a comb: many runs per word, tall teeth
            (((xx - w // 2) ** 2 + (yy - h // 2) ** 2) < (min(h, w) // 2 - 3) ** 2)]    # one huge disc
    n = len(pats)
    area = np.stack([(p * 255).astype(np.uint8) for p in pats])
    mask = np.stack([p.astype(np.uint8) for p in pats])
    eng = Engine(h, w, max_markers=1024, max_batch=n)
    mt, at = torch.from_numpy(mask).cuda(), torch.from_numpy(area).cuda()
    out = []
    for impl in (0, 1):
        eng.set_option(L.OPT_STAGE_IMPL, impl)
        det, counts = eng.marker_center(mt, at)
        torch.cuda.synchronize()
        out.append((det.clone(), counts.clone()))
    eng.set_option(L.OPT_STAGE_IMPL, 0)
    (d0, c0), (d1, c1) = out
    assert torch.equal(c0, c1), (c0.tolist(), c1.tolist())
    for i in range(n):
        k = max(int(c0[i]), 0)
        assert torch.equal(d0[i, :k], d1[i, :k]), i
    assert int(c0[1]) == 0 and int(c0[0]) >= 0                              # empty frame: nothing; all ones: one component at most
    eng.close()


def test_smallest_opened_components_fit_without_the_degenerate_branch():
    """`cv2.fitEllipse` re-fits a degenerate point set (collinear, .. : singular design matrix) after nudging the points by
    +-eps in an order-dependent pattern; the HIP path fits from contour-vertex moments, which cannot express that, and
    flags such a contour invalid instead.  After the 5x5 opening (`_marker_center` :195) every component is a
    union of 5x5 squares: this walks the smallest such shapes (pairs / triples of squares at every small offset, whose
    contours have 4 .. 12 vertices) and checks that the oracle never takes the degenerate branch on them and that the HIP
    path returns the oracle's ellipses - the branch is not reachable through `_marker_center`."""
    from vbs_amd.marker_detection import MarkerTracker
    eps32 = float(np.finfo(np.float32).eps)
    h, w = 480, 640
    area = np.zeros((h, w), np.uint8)
    offs = [(dx, dy) for dx in range(0, 7) for dy in range(0, 7) if (dx, dy) != (0, 0)]
    shapes, k = [], 0
    for (dx, dy) in offs:
        for third in (None, (2 * dx, 0), (0, 2 * dy), (dx + 3, dy - 3)):
            gx, gy = 20 + 30 * (k % 20), 20 + 30 * (k // 20)
            if gy + 25 >= h:
                break
            sq = [(0, 0), (dx, dy)] + ([third] if third else [])
            for (ox, oy) in sq:
                area[gy + 8 + oy:gy + 13 + oy, gx + ox:gx + 5 + ox] = 255
            shapes.append((gx, gy))
            k += 1
    assert k > 150
    opened = O.morph_open5(area != 0)
    assert np.array_equal(opened, area != 0)                      # unions of 5x5 squares survive the opening unchanged
    degenerate = 0
    for cont in O.find_contours_external(opened):
        if len(cont) < 5:
            continue
        pts = cont.astype(np.float32).reshape(-1, 2)
        c = pts.mean(axis=0, dtype=np.float32)
        q = (pts - c).astype(np.float64)
        s = np.abs(q).sum()
        px, py = q[:, 0] * 100.0 / s, q[:, 1] * 100.0 / s
        sv = np.linalg.svd(np.stack([-px * px, -py * py, -px * py, px, py], axis=1), compute_uv=False)
        degenerate += bool(sv[0] * eps32 > sv[4])
    assert degenerate == 0
    mask = (area != 0).astype(np.uint8)                           # every blob is its own band, so every contour finds a centre
    want = O.marker_center(mask, area)
    got = MarkerTracker._marker_center(mask, area)
    assert len(want) > 40
    compare_markers(got, want)


def test_two_handles_of_different_sizes_alternate():
    """Handles of different frame sizes in one process, used in turn: each declares the dynamic LDS its labelling kernels
    need (`hipFuncSetAttribute` is tracked per handle, not in function statics), in the fused and in the separate-kernel
    form, and a small handle created AFTER a large one still gets its own declaration."""
    from vbs_amd.engine import Engine
    specs = [S.config2(), S.config1(), S.config5()]
    engs = [Engine(sp.height, sp.width, max_markers=1024, max_batch=2) for sp in specs]
    frames = [S.make_frames_torch(sp, range(2), seed=6, device="cuda") for sp in specs]
    for impl in (0, 1, 0):
        for eng, sp, ft in zip(engs, specs, frames):
            eng.set_option(L.OPT_STAGE_IMPL, impl)
            _, _, counts = eng.track_to_3d(ft, want_det=True)
            assert counts.tolist() == [sp.n_markers, sp.n_markers], (impl, sp.name, counts.tolist())
    late = Engine(specs[1].height, specs[1].width, max_markers=256, max_batch=1)
    for impl in (1, 0):
        late.set_option(L.OPT_STAGE_IMPL, impl)
        _, _, counts = late.track_to_3d(frames[1][:1], want_det=True)
        assert counts.tolist() == [specs[1].n_markers]
    for e in engs + [late]:
        e.close()


def test_multi_pass_call_captured_into_a_graph_replays_bit_identically():
    """A vbs_track_to_3d call that spans several internal passes on two pass streams, captured into a HIP graph (the second
    workspace was built ahead of time by vbs_set_option(VBS_OPT_PASS_STREAMS, 2): nothing in the call allocates or
    synchronises) and replayed twice, equals the eager call bit for bit; on a handle left at the default, whose second
    workspace does not exist yet, a captured call runs on one stream instead of breaking the capture - same table."""
    from vbs_amd.engine import Engine
    from vbs_amd.pipeline import reference_from_frame0
    spec = S.config1()
    n = 22
    ft = S.make_frames_torch(spec, range(n), seed=5, device="cuda")
    cam = L.make_camera(*S.default_camera(spec), 2.0)
    for eager_twin in (True, False):
        eng = Engine(spec.height, spec.width, max_markers=256, max_batch=4)
        ids, xy = reference_from_frame0(eng, ft[:1], 5, "full", "optimal")
        xy_d = torch.as_tensor(xy, dtype=torch.float64, device="cuda")
        if eager_twin:
            eng.set_option(L.OPT_PASS_STREAMS, 2)            # builds the second workspace now
            want, _, wc = eng.track_to_3d(ft, xy_d, 20.0, cam, 5.0)
        else:
            # the reference result from another handle: this one must meet its first multi-pass call under capture
            e2 = Engine(spec.height, spec.width, max_markers=256, max_batch=4)
            want, _, wc = e2.track_to_3d(ft, xy_d, 20.0, cam, 5.0)
            torch.cuda.synchronize()
            e2.close()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side):
                table, _, counts = eng.track_to_3d(ft, xy_d, 20.0, cam, 5.0)
        torch.cuda.current_stream().wait_stream(side)
        for rep in range(2):
            table.zero_(); counts.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert counts.tolist() == [spec.n_markers] * n, (eager_twin, rep)
            assert torch.equal(table, want) and torch.equal(counts, wc), (eager_twin, rep)
        del g
        eng.close()


def test_frame_stats_follow_the_last_pass_onto_the_second_workspace():
    """vbs_frame_stats / vbs_stage_tables describe the LAST internal pass wherever it ran: with two pass streams and a
    staggered first pass (4 frames at batch 8), 20 frames are passes of 4, 8, 8 - the last on the caller's workspace -
    and 28 frames passes of 4, 8, 8, 8 - the last on the second workspace; both equal a one-stream handle's view."""
    from vbs_amd.engine import Engine
    spec = S.config1()
    for n in (20, 28):
        ft = S.make_frames_torch(spec, range(n), seed=9, device="cuda")
        views = []
        for ps in (1, 2):
            eng = Engine(spec.height, spec.width, max_markers=256, max_batch=8)
            eng.set_option(L.OPT_PASS_STREAMS, ps)
            eng.track_to_3d(ft)
            views.append((eng.frame_stats(8), eng.stage_tables(8)["ncomp"]))
            eng.close()
        # one stream: passes of 8, 8, 4 (20) / 8, 8, 8, 4 (28): its last pass holds the last 4 frames, which are also the
        # last 4 of the staggered schedule's last pass of 8
        k = 4
        assert np.array_equal(views[1][0][8 - k:, [0, 2, 5, 6]], views[0][0][:k, [0, 2, 5, 6]]), n
        assert np.array_equal(views[1][1][8 - k:], views[0][1][:k]), n
        assert (views[1][0][:, 5] == spec.n_markers).all() and (views[1][0][:, 0] > 0).all(), n


def test_frame_stats_after_normxcorr2_describe_that_pass():
    """every pass entry point records itself as the last pass.  A multi-pass vbs_track_to_3d whose last pass
    ran on the second workspace, then vbs_normxcorr2 on the same handle: vbs_frame_stats must show the NCC pass's
    counters (area popcount of ITS masks), not the second workspace's."""
    from vbs_amd.engine import Engine
    spec = S.config1()
    ft = S.make_frames_torch(spec, range(28), seed=9, device="cuda")
    eng = Engine(spec.height, spec.width, max_markers=256, max_batch=8)
    eng.set_option(L.OPT_PASS_STREAMS, 2)
    eng.track_to_3d(ft)                                    # passes of 4, 8, 8, 8: the last one on the second workspace
    before = eng.frame_stats(8).copy()
    area = torch.zeros((2, spec.height, spec.width), dtype=torch.uint8, device="cuda")
    area[0, 100:140, 200:260] = 255                        # 2 400 foreground pixels
    area[1, 50:60, 50:70] = 255                            # 200
    eng.normxcorr2(area)
    st = eng.frame_stats(2)
    assert st[:, 0].tolist() == [2400, 200], (st[:, 0], before[:2, 0])
    eng.close()


def test_num_layers_beyond_the_device_kernel_runs_on_the_host(tmp_path):
    """k_assign_ids covers num_layers <= 16; a configuration beyond that must run as it always did (host
    assignment) instead of failing the device check."""
    from vbs_amd.marker_detection import MarkerTracker
    spec = S.config1()
    frames = S.make_frames(spec, range(2), seed=3)
    np.save(tmp_path / "clip.npy", frames)
    trk = MarkerTracker({"video_path": str(tmp_path / "clip.npy"), "output_dir": str(tmp_path / "o"), "crop_ratios": (0, 0, 0, 0),
                         "num_layers": 20, "id_mode": "full"})
    trk.process()
    assert trk.ids_device_check == {"on_device": False, "used": "host"} and len(trk.first_frame_markers) == spec.n_markers
    trk5 = MarkerTracker({"video_path": str(tmp_path / "clip.npy"), "output_dir": str(tmp_path / "o5"), "crop_ratios": (0, 0, 0, 0),
                          "num_layers": 5, "id_mode": "full"})
    trk5.process()
    assert trk5.ids_device_check["used"] in ("device", "host") and trk5.ids_device_check["equal_to_host"] == (trk5.ids_device_check["used"] == "device")


def test_gray_plane_is_allocated_at_first_bgr_use_and_not_under_capture():
    """The gray plane of 3-channel input no longer exists on a handle that only sees gray frames; the first BGR call
    allocates it - unless that call is being captured, which is refused with a message instead of breaking the capture."""
    from vbs_amd.engine import Engine
    spec = S.config1()
    g1 = S.make_frames_torch(spec, range(2), seed=1, device="cuda")
    bgr = g1.unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()
    eng = Engine(spec.height, spec.width, max_markers=256, max_batch=2)
    free0 = torch.cuda.mem_get_info()[0]
    m_gray, a_gray = eng.find_markers(g1)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0           # gray input: nothing was allocated by the call
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with pytest.raises(ValueError, match="outside stream capture"):
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                eng.find_markers(bgr)
    torch.cuda.synchronize()
    m_bgr, a_bgr = eng.find_markers(bgr)                    # allocates the plane (B = G = R: the same masks)
    assert torch.equal(m_bgr, m_gray) and torch.equal(a_bgr, a_gray)
    eng.close()
    # with two pass streams the second workspace converts its own passes; its plane exists as soon as the
    # handle's does, so that a MULTI-pass 3-channel call can be captured after a SINGLE-pass 3-channel warm-up
    g6 = S.make_frames_torch(spec, range(6), seed=1, device="cuda")
    bgr6 = g6.unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()
    eng = Engine(spec.height, spec.width, max_markers=256, max_batch=2)
    from vbs_amd.pipeline import reference_from_frame0
    eng.set_option(L.OPT_PASS_STREAMS, 2)                   # builds the second workspace (no gray planes yet)
    _, xy = reference_from_frame0(eng, g6[:1], 5, "full", "optimal")
    xy_d = torch.as_tensor(xy, dtype=torch.float64, device="cuda")
    cam = L.make_camera(*S.default_camera(spec), 2.0)
    want, _, wc = eng.track_to_3d(g6, xy_d, 20.0, cam, 5.0)
    eng.track_to_3d(bgr6[:2], xy_d, 20.0, cam, 5.0)         # one pass, 3 channels: allocates BOTH workspaces' planes
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            table, _, counts = eng.track_to_3d(bgr6, xy_d, 20.0, cam, 5.0)     # three passes, two streams, under capture
    torch.cuda.current_stream().wait_stream(side)
    table.zero_(); counts.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(counts, wc) and torch.equal(table, want)
    del graph
    eng.close()


def test_blur16_hand_shake_that_expires_is_reported_in_counts():
    """k_blur16's loader and strips wait for each other with BOUNDED spins.  A wait that expires must not continue silently:
    it sets the frame's status word, which k_finalize hands to counts[] (VBS_EINTERNAL), like a capacity overflow.  Shown
    once with the debug library (-DVBS_DEBUG_KNOBS; the product library has no such knob): VBS_BLUR16_DROP=4 makes the
    loader of a frame's first workgroup "forget" the tick of its fourth tile, so its strips run one tile behind and their
    wait for the last tile expires.  Runs in a child process (a second copy of the library, an environment knob)."""
    import subprocess
    import sys
    import vbs_amd._build as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dbg = B.LIB.replace(".so", "_dbg.so")
    if not os.path.exists(dbg):
        B.build(extra_flags=["-DVBS_DEBUG_KNOBS"], suffix="_dbg")
    code = (
        "import os, sys, json; sys.path.insert(0, %r)\n"
        "import torch, vbs_amd.synth as S\n"
        "from vbs_amd import _lib as L\n"
        "L.LIB_PATH = %r\n"
        "from vbs_amd.engine import Engine\n"
        "spec = S.config2(); ft = S.make_frames_torch(spec, range(3), seed=2, device='cuda')\n"
        "eng = Engine(spec.height, spec.width, max_markers=512, max_batch=3)\n"
        "out = {}\n"
        "for drop in ('0', '4'):\n"
        "    os.environ['VBS_BLUR16_DROP'] = drop\n"
        "    _, _, counts = eng.track_to_3d(ft); torch.cuda.synchronize()\n"
        "    out[drop] = {'counts': counts.tolist(), 'status': eng.frame_stats(3)[:, 2].astype('int32').tolist()}\n"
        "print(json.dumps(out))\n" % (root, dbg))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["0"] == {"counts": [169, 169, 169], "status": [0, 0, 0]}
    assert out["4"] == {"counts": [L.VBS_EINTERNAL] * 3, "status": [L.VBS_EINTERNAL] * 3}


def test_latency_stage_on_the_widest_geometry():
    """4096 x 2048: 64 word columns (one row block per wave), 16 workgroups per plane - the cap, 32 rows per tile - and 4 MB of
    per-row slot table per frame (the scratch holds fewer frames than VBS_LAT_MAXN there): the few-frames kernel against the
    batch kernel on ragged blobs."""
    from vbs_amd.engine import Engine
    h, w, n = 2048, 4096, 2
    rng = np.random.default_rng(5)
    mask = np.zeros((n, h, w), np.uint8); area = np.zeros((n, h, w), np.uint8)
    for f in range(n):
        for _ in range(60):
            cx, cy = int(rng.integers(40, w - 40)), int(rng.integers(40, h - 40))
            a, b = int(rng.integers(6, 36)), int(rng.integers(6, 36))
            yy, xx = np.mgrid[-b:b + 1, -a:a + 1]
            e = (xx / a) ** 2 + (yy / b) ** 2
            area[f, cy - b:cy + b + 1, cx - a:cx + a + 1][e <= 1] = 255
            mask[f, cy - b:cy + b + 1, cx - a:cx + a + 1][e <= 0.5] = 1
    eng = Engine(h, w, max_markers=512, max_batch=n)
    mt, at = torch.from_numpy(mask).cuda(), torch.from_numpy(area).cuda()
    out = []
    for lat in (32, 0):
        eng.set_option(L.OPT_LATENCY_FRAMES, lat)
        det, counts = eng.marker_center(mt, at)
        torch.cuda.synchronize()
        out.append((det.clone(), counts.clone(), eng.stage_tables(n)))
    (d0, c0, t0), (d1, c1, t1) = out
    assert torch.equal(c0, c1) and torch.equal(d0, d1) and int(c0.min()) > 20
    keep = (t0["slow"] == 0) & (t1["slow"] == 0)
    assert int(keep.sum()) > 0
    for i in np.nonzero(keep)[0]:
        nb, na = (int(v) for v in t0["ncomp"][i])
        assert (nb, na) == tuple(int(v) for v in t1["ncomp"][i])
        assert np.array_equal(t0["band_sums"][i][:nb, :3], t1["band_sums"][i][:nb, :3])
        assert np.array_equal(t0["area_sums"][i][:na, :15], t1["area_sums"][i][:na, :15])
        assert np.array_equal(t0["probe"][i][:nb], t1["probe"][i][:nb])
    eng.close()


def test_stage_lat_under_repetition():
    """k_stage_lat's workgroups talk through global memory (arrival counters, lists written on one XCD and read on another): a
    missing fence shows as a RARE wrong table.  tools/gpu_lat_stress.py: 64 frames of each size, their tables by the batch
    kernel once, then calls with 1 / 3 / 8 / 19 frames in shuffled order through the few-frames kernel (here 6 rounds, ~1 400
    calls; 40 rounds = 9 120 calls ran clean when it was written): every table, detection row and count identical."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_lat_stress.py"), "6"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "lat stress all OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_stage_lat_wait_that_expires_is_reported_in_counts():
    """k_stage_lat has ONE wait: the workgroup that resolves a frame's opened plane needs the band centroids (the other
    workgroups' resolve) for its probes.  The wait is bounded, and a wait that expires must not answer the probes from
    whatever the memory holds: the frame's status becomes VBS_EINTERNAL -> counts[].  Shown once with the debug library
    (VBS_LAT_DROP = 2: frame 1's band resolve "forgets" to raise its flag); the other frames of the pass are untouched."""
    import subprocess
    import sys
    import vbs_amd._build as B
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dbg = B.LIB.replace(".so", "_dbg.so")
    if not os.path.exists(dbg):
        B.build(extra_flags=["-DVBS_DEBUG_KNOBS"], suffix="_dbg")
    code = (
        "import os, sys, json; sys.path.insert(0, %r)\n"
        "import torch, vbs_amd.synth as S\n"
        "from vbs_amd import _lib as L\n"
        "L.LIB_PATH = %r\n"
        "from vbs_amd.engine import Engine\n"
        "spec = S.config2(); ft = S.make_frames_torch(spec, range(3), seed=2, device='cuda')\n"
        "eng = Engine(spec.height, spec.width, max_markers=512, max_batch=3)\n"
        "out = {}\n"
        "for drop in ('0', '2'):\n"
        "    os.environ['VBS_LAT_DROP'] = drop\n"
        "    _, _, counts = eng.track_to_3d(ft); torch.cuda.synchronize()\n"
        "    out[drop] = {'counts': counts.tolist(), 'status': eng.frame_stats(3)[:, 2].astype('int32').tolist()}\n"
        "print(json.dumps(out))\n" % (root, dbg))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["0"] == {"counts": [169, 169, 169], "status": [0, 0, 0]}
    assert out["2"] == {"counts": [169, L.VBS_EINTERNAL, 169], "status": [0, L.VBS_EINTERNAL, 0]}


def test_passes_on_two_streams_equal_one():
    """VBS_OPT_PASS_STREAMS: the odd internal passes of vbs_track_to_3d on the handle's second workspace and stream
    (default) give, row for row, what all passes on the caller's stream give - tables, detections, counts and the running
    NCC counters (summed over both workspaces); options set on the handle reach the second workspace; a following call on
    the caller's stream sees the joined results."""
    from vbs_amd.engine import Engine
    from vbs_amd.pipeline import reference_from_frame0
    spec = S.config1()
    n = 23                                                # 6 passes of 4: three on either stream
    ft = S.make_frames_torch(spec, range(n), seed=3, device="cuda")
    eng = Engine(spec.height, spec.width, max_markers=256, max_batch=4)
    ids, xy = reference_from_frame0(eng, ft[:1], 5, "full", "optimal")
    cam = L.make_camera(*S.default_camera(spec), 2.0)
    got = {}
    for ps, impl in ((1, 0), (2, 0), (2, 1), (1, 1)):
        eng.set_option(L.OPT_PASS_STREAMS, ps)
        eng.set_option(L.OPT_STAGE_IMPL, impl)            # (reaches the second workspace as well)
        eng.ncc_counters(reset=True)
        table, det, counts = eng.track_to_3d(ft, xy, 20.0, cam, 5.0, want_det=True)
        disp = eng.displacement(table, 0, 5.0, 50.0)      # on the caller's stream, behind the join
        ctr = eng.ncc_counters()
        got[(ps, impl)] = (table.cpu(), det.cpu(), counts.cpu(), disp.cpu(), ctr)
    ref = got[(1, 0)]
    assert ref[2].tolist() == [spec.n_markers] * n and ref[4]["frames"] == n
    for key, g in got.items():
        for a, b in zip(g[:4], ref[:4]):
            assert torch.equal(a, b), key
        assert g[4] == ref[4], (key, g[4], ref[4])
    eng.set_option(L.OPT_STAGE_IMPL, 0)
    eng.close()


def test_engine_argument_errors():
    from vbs_amd.engine import Engine
    with pytest.raises(ValueError):
        Engine(32, 32)
    eng = engine(480, 640)
    with pytest.raises(ValueError):
        eng.find_markers(torch.zeros((1, 100, 100), dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        eng.find_markers(torch.zeros((1, 480, 640), dtype=torch.float32, device="cuda"))
    mask, area = eng.find_markers(torch.full((1, 480, 640), 190, dtype=torch.uint8, device="cuda"))
    assert int(mask.sum()) == 0 and int(area.sum()) == 0
    _, _, counts = eng.track_to_3d(torch.full((2, 480, 640), 190, dtype=torch.uint8, device="cuda"),
                                   np.array([[10.0, 10.0]]))
    assert counts.tolist() == [0, 0]
    det = torch.zeros((2, eng.max_markers, 6), dtype=torch.float64, device="cuda")
    cnt = torch.zeros((2,), dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):                     # wrong row count per frame
        eng.track(det[:, :7].contiguous(), cnt, np.array([[10.0, 10.0]]))
    with pytest.raises(ValueError):                     # counts of the wrong length / dtype
        eng.track(det, cnt[:1], np.array([[10.0, 10.0]]))
    with pytest.raises(ValueError):
        eng.track(det, cnt.long(), np.array([[10.0, 10.0]]))
    big = torch.full((2,), 5000, dtype=torch.int32, device="cuda")      # a count beyond the table is clamped, not trusted
    t = eng.track(det, big, np.array([[10.0, 10.0]]))
    torch.cuda.synchronize()
    assert t.shape == (2, 1, 10)
    with pytest.raises(ValueError):
        eng.set_option(L.OPT_GRAY_COEFFS, 13)
    eng.close()


def test_tracking_grid_equals_the_plain_nearest_search():
    """a15 (`_track_markers`, marker_detection.py:349-396): `vbs_track` looks for a reference ID's nearest detection in the
    3 x 3 cells around it of a grid in LDS instead of among all detections.  Against `cdist` + `argmin` + the distance rule
    of the reference (NumPy, float64) on inputs made to break a grid: detections at every distance around `min_dist`
    (exactly on it too), on cell borders, exact ties (the first index wins), duplicates, reference positions outside the
    frame / negative / huge / NaN, min_dist 0 ... beyond the grid's range (the plain scan), few and many detections."""
    from scipy.spatial.distance import cdist
    rng = np.random.default_rng(11)
    eng = engine(480, 640, max_markers=1024, max_batch=4)
    cases = []
    for md in (0.0, 0.5, 7.0, 20.0, 31.0, 32.0, 33.0, 63.5, 200.0, 4095.0, 5000.0, float("nan")):
        for cnt in (0, 1, 16, 17, 169, 1024):
            cases.append((md, cnt))
    for md, cnt in cases:
        mdv = 20.0 if md != md else md
        det = np.zeros((4, 1024, 6), np.float64)
        counts = np.array([cnt, cnt, max(cnt - 3, 0), cnt], np.int32)
        xy = rng.uniform(0, [640, 480], (4, 1024, 2))
        xy[1] = np.round(xy[1] / 16.0) * 16.0                       # on cell borders, many exact ties and duplicates
        xy[3, :, 0] = rng.uniform(-50, 700, 1024)
        det[..., :2] = xy
        det[..., 2] = rng.uniform(15, 30, (4, 1024)); det[..., 3] = det[..., 2] - 1.0
        det[..., 4] = rng.uniform(0, 180, (4, 1024)); det[..., 5] = np.arange(1024) + 1
        m = 300
        ref = rng.uniform(0, [640, 480], (m, 2))
        k = min(cnt, 100)
        if k:
            ang = rng.uniform(0, 2 * np.pi, k)
            rad = np.concatenate([np.full(k // 2, mdv), rng.uniform(0.9, 1.1, k - k // 2) * mdv])
            ref[:k] = xy[0, :k] + rad[:, None] * np.stack([np.cos(ang), np.sin(ang)], axis=1)     # around / exactly at min_dist
            ref[100:100 + min(k, 50)] = np.round(ref[100:100 + min(k, 50)] / 8.0) * 8.0            # ties against frame 1's lattice
        ref[200:206] = [[-1e6, 3.0], [3.0, 1e12], [np.nan, 5.0], [-0.0, -0.0], [639.999, 479.999], [1e300, -1e300]]
        ref[206:212] = [[-17.0, 240.0], [656.0, 240.0], [320.0, -15.9], [320.0, 495.5], [np.inf, 0.0], [32.0, 32.0]]
        t = eng.track(torch.from_numpy(det).cuda(), torch.from_numpy(counts).cuda(), ref, md).cpu().numpy()
        for n in range(4):
            c = int(counts[n])
            for r in range(m):
                row = t[n, r]
                want = None
                if c:
                    with np.errstate(invalid="ignore", over="ignore"):
                        d = cdist(ref[r:r + 1], det[n, :c, :2])[0]
                    if not np.isnan(d).all():
                        j = int(np.argmin(d)) if not np.isnan(d).any() else None
                        if j is None:                               # (argmin of a row with NaN is the NaN: every distance NaN here)
                            want = None
                        elif not (d[j] > md) and np.isfinite(d[j]):     # (an overflowing distance is never reported: the
                            want = j                                    #  reference would, for a NaN min_dist only)
                if want is None:
                    assert row[0] == 0.0, (md, cnt, n, r, row)
                else:
                    assert int(row[0]) & 1 and int(row[9]) == want, (md, cnt, n, r, row, want)
                    assert row[1] == np.float32(det[n, want, 0]) and row[2] == np.float32(det[n, want, 1])
    eng.close()


def test_real_sensor_frame(golden_dir, tmp_path):
    """The reference's one real frame (img/raw_markers.png -> tests/golden/raw_markers_bgr.npz, 467x437 BGR, 65 printed
    dots): HIP vs oracle under both BGR2GRAY coefficient sets - masks exact, 65 detections, centroids exact, axes
    within 1e-3 px; identities in both modes through `MarkerTracker` (physical layout 1+6+12+18+24+4)."""
    from collections import Counter
    from vbs_amd.marker_detection import MarkerTracker
    bgr = np.load(os.path.join(golden_dir, "raw_markers_bgr.npz"))["bgr"]
    h, w = bgr.shape[:2]
    ft = torch.from_numpy(bgr[None]).cuda()
    for bits in (15, 14):
        eng = engine(h, w, max_markers=1024, max_batch=1)
        eng.set_option(L.OPT_GRAY_COEFFS, bits)
        assert np.array_equal(eng.bgr2gray(ft)[0].cpu().numpy(), O.bgr2gray(bgr, bits))
        om, oa = O.find_markers(bgr, gray_bits=bits)
        mask, area = eng.find_markers(ft)
        assert np.array_equal(mask[0].cpu().numpy(), om) and np.array_equal(area[0].cpu().numpy(), oa)
        want = O.marker_center(om, oa)
        det, counts = eng.marker_center(mask, area)
        assert int(counts[0]) == 65 == len(want)
        _, det2, counts2 = eng.track_to_3d(ft, None, want_det=True)           # the fused path sees the same frame
        assert int(counts2[0]) == 65 and torch.equal(det2[0, :65], det[0, :65])
        from vbs_amd.marker_detection import _det_to_markers
        compare_markers(_det_to_markers(det[0].cpu().numpy(), 65), want)
        assert eng.frame_stats(1)[0, 5] == 65 and eng.frame_stats(1)[0, 6] == 65
        # the real layout stays on the fast labelling path under BOTH labelling kernels (few-frames and batch): no frame is
        # handed on to the general kernel (65 dots of ~27 px at a pitch of 35-42 px)
        assert int(eng.stage_tables(1)["slow"][0]) == 0
        eng8 = engine(h, w, max_markers=1024, max_batch=8)
        eng8.set_option(L.OPT_LATENCY_FRAMES, 0)
        _, det8, counts8 = eng8.track_to_3d(ft.expand(8, -1, -1, -1).contiguous(), None, want_det=True)
        assert (counts8 == 65).all() and torch.equal(det8[7, :65], det[0, :65])
        assert not eng8.stage_tables(8)["slow"].any()
        eng8.close()
        if bits == 15:
            # ... and against the reference's own published result for this scene (tests/figure_check.py): centres, the
            # 65 `full` IDs as printed in img/2d_visualization.png, axes as a bounded offset - from the HIP path's rows
            from figure_check import check_against_figure
            hip = _det_to_markers(det[0].cpu().numpy(), 65)
            table = I.assign_ids(hip, 5, "full", "optimal")
            rep = check_against_figure(golden_dir, hip, table)
            assert rep["ids_equal"] == 65 and rep["centre_max_px"] <= 0.75
        eng.close()
    clip = np.stack([bgr, bgr])
    np.save(tmp_path / "real.npy", clip)
    for id_mode, n_ids in (("as_written", 6), ("full", 65)):
        cfg = {"video_path": str(tmp_path / "real.npy"), "output_dir": str(tmp_path / id_mode), "crop_ratios": (0, 0, 0, 0),
               "num_layers": 5, "min_marker_distance": 20, "id_mode": id_mode}
        trk = MarkerTracker(cfg)
        trk.process()
        rows, ref = O.process_frames(list(clip), crop_ratios=(0, 0, 0, 0), id_mode=id_mode)
        assert list(trk.first_frame_markers.keys()) == list(ref.keys()) and len(ref) == n_ids
        import pandas as pd
        df = pd.read_csv(trk.output_csv, float_precision="round_trip")
        assert len(df) == len(rows) == 2 * n_ids
        wantdf = pd.DataFrame(rows)
        for c in ("frameno", "row", "col", "Cx", "Cy", "Ox", "Oy"):
            assert (df[c].to_numpy() == wantdf[c].to_numpy()).all(), c
        if id_mode == "full":
            assert Counter(df[df.frameno == 0].row.tolist()) == {0: 1, 1: 6, 2: 12, 3: 18, 4: 24, 5: 4}


def test_real_layout_jittered_frames_against_the_oracle(golden_dir):
    """The real-layout workload of `bench.py --workload real` (the reference's published frame moved by seeded shifts of
    +-3 px, noise sigma 2: real texture, real dot size and pitch, coloured BGR) frame by frame against the oracle: masks,
    band centroids (bit-exact), ellipse axes, through the batch labelling kernel and through the few-frames one; every
    frame stays on the fast labelling path; tracking rows equal the oracle's `_track_markers` rows."""
    from vbs_amd.marker_detection import _det_to_markers
    bgr = np.load(os.path.join(golden_dir, "raw_markers_bgr.npz"))["bgr"]
    n = 12
    ft, shifts = S.jittered_copies_torch(bgr, n, seed=3, device="cuda")
    frames = ft.cpu().numpy()
    assert np.array_equal(frames[0], bgr) and (np.abs(shifts[1:]).max(axis=1) > 0).any()
    h, w = bgr.shape[:2]
    want = []
    for f in frames:
        om, oa = O.find_markers(f)
        want.append((om, oa, O.marker_center(om, oa)))
    ref = O.process_first_frame(want[0][2], 5, "full", "optimal")
    xy = np.array([[v["Ox"], v["Oy"]] for v in ref.values()])
    for lat in (24, 0):                                   # few-frames kernel / batch kernel
        eng = engine(h, w, max_markers=256, max_batch=n)
        eng.set_option(L.OPT_LATENCY_FRAMES, lat)
        mask, area = eng.find_markers(ft)
        table, det, counts = eng.track_to_3d(ft, xy, 20.0, want_det=True)
        assert not eng.stage_tables(n)["slow"].any(), lat
        mask, area, det, counts, table = mask.cpu().numpy(), area.cpu().numpy(), det.cpu().numpy(), counts.cpu().numpy(), table.cpu().numpy()
        for i in range(n):
            om, oa, markers = want[i]
            assert np.array_equal(mask[i], om) and np.array_equal(area[i], oa), (lat, i)
            assert counts[i] == len(markers), (lat, i, counts[i], len(markers))
            compare_markers(_det_to_markers(det[i], int(counts[i])), markers)
            rows = O.track_markers(ref, markers, i + 1, 20)
            got = table[i]
            assert int((got[:, 0].astype(int) & 1).sum()) == len(rows), (lat, i)
            keys = list(ref.keys())
            for r in rows:
                t = got[keys.index((r["row"], r["col"]))]
                assert abs(t[1] - r["Cx"]) < 2.5e-4 and abs(t[2] - r["Cy"]) < 2.5e-4 and abs(t[3] - r["major_axis"]) < 1e-3
        eng.close()


def test_bgr_frames_large_branch_fused_and_fallback():
    """a3 + a4 on coloured 1280x1024 BGR frames (large branch): dense 16-byte-aligned frames take `k_gray`'s coalesced
    path, a crop view at an odd offset its strided one; both equal the oracle, under both coefficient sets."""
    rng = np.random.default_rng(12)
    spec = S.config2()
    g = S.make_frames(spec, [0, 3], seed=5)
    tint = rng.integers(-12, 13, (2, 1, 1, 3))
    frames = np.clip(g[..., None].astype(int) + tint + rng.integers(-3, 4, g.shape + (3,)), 0, 255).astype(np.uint8)
    frames[0, :40, :300] = rng.integers(0, 256, (40, 300, 3), dtype=np.uint8)          # coloured clutter at the border
    frames[1, 500:560, 1200:] = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    pad = np.pad(frames, ((0, 0), (5, 3), (7, 9), (0, 0)))
    for bits in (15, 14):
        eng = engine(spec.height, spec.width, max_batch=2)
        eng.set_option(L.OPT_GRAY_COEFFS, bits)
        want = [O.find_markers(f, gray_bits=bits) for f in frames]
        for ft in (torch.from_numpy(frames).cuda(), torch.from_numpy(pad).cuda()[:, 5:5 + spec.height, 7:7 + spec.width]):
            mask, area = eng.find_markers(ft)
            for i in range(2):
                assert np.array_equal(area[i].cpu().numpy(), want[i][1]), (bits, i)
                assert np.array_equal(mask[i].cpu().numpy(), want[i][0]), (bits, i)
        eng.close()
    assert not np.array_equal(O.bgr2gray(frames[0], 15), O.bgr2gray(frames[0], 14))


@pytest.mark.parametrize("cfg", ["c1", "c2"])
def test_ncc_wide_margin_drives_the_queued_exact_path(cfg):
    """a7-a8: with the float32 filter's margin widened 250x (VBS_OPT_NCC_MARGIN, a test hook) thousands of pixels per
    frame are left undecided, queued tile by tile and re-evaluated in float64 after the step loop (queue drains in
    mid-strip, several pixels per tile, atomic OR into stored mask words, the uint8 mask patched): the masks must still
    equal the oracle's and the fused path's table must not change by a bit."""
    from vbs_amd.pipeline import reference_from_frame0
    spec = S.config1() if cfg == "c1" else S.config2()
    f = S.make_frames(spec, [0, 1, 2, 5], seed=3)
    ft = torch.from_numpy(f).cuda()
    eng = engine(spec.height, spec.width, max_batch=4)
    ref_ids, ref_xy = reference_from_frame0(eng, ft[:1], 5, "full", "optimal")
    t0, _, c0 = eng.track_to_3d(ft, ref_xy, 20.0, None, 5.0)
    base = eng.ncc_counters()
    eng.set_option(L.OPT_NCC_MARGIN, 5000)
    mask, area = eng.find_markers(ft)                       # staged form: uint8 mask written by the kernel
    t1, _, c1 = eng.track_to_3d(ft, ref_xy, 20.0, None, 5.0)    # fused form: bit masks only
    wide = eng.ncc_counters()
    for i in range(len(f)):
        om, oa = O.find_markers(f[i])
        assert np.array_equal(area[i].cpu().numpy(), oa)
        assert np.array_equal(mask[i].cpu().numpy(), om), i
    assert torch.equal(t0, t1) and torch.equal(c0, c1)
    per_frame = (wide["exact"] - base["exact"]) / (2 * len(f))
    assert per_frame > 200, per_frame                      # the hook did widen the margin
    eng.close()


def test_bgr_side_stream_option_gives_the_same_results():
    """VBS_OPT_GRAY_SIDE_STREAM: BGR frames over several internal passes, the conversion of pass k + 1 on the handle's own
    stream (lead pass, two gray planes, event fork / join) - tables and masks identical to the in-line conversion."""
    rng = np.random.default_rng(21)
    spec = S.config1()
    g = S.make_frames(spec, range(11), seed=6)
    frames = np.clip(g[..., None].astype(int) + rng.integers(-6, 7, g.shape + (3,)), 0, 255).astype(np.uint8)
    ft = torch.from_numpy(frames).cuda()
    eng = engine(spec.height, spec.width, max_batch=3)
    from vbs_amd.pipeline import reference_from_frame0
    ids, xy = reference_from_frame0(eng, ft[:1], 5, "full", "optimal")
    t0, _, c0 = eng.track_to_3d(ft, xy, 20.0, None, 5.0)
    m0, a0 = eng.find_markers(ft)
    eng.set_option(L.OPT_GRAY_SIDE_STREAM, 1)
    for _ in range(2):                                         # twice: the planes and events are reused across calls
        t1, _, c1 = eng.track_to_3d(ft, xy, 20.0, None, 5.0)
        m1, a1 = eng.find_markers(ft)
        assert torch.equal(t0, t1) and torch.equal(c0, c1) and torch.equal(m0, m1) and torch.equal(a0, a1)
    om, oa = O.find_markers(frames[7])
    assert np.array_equal(m1[7].cpu().numpy(), om) and np.array_equal(a1[7].cpu().numpy(), oa)
    eng.close()


def test_bgr2gray_both_coefficient_sets():
    """a3 on coloured pixels, where the 15-bit (OpenCV 4) and 14-bit sets differ; also through `find_markers`."""
    rng = np.random.default_rng(9)
    f = rng.integers(0, 256, (2, 480, 640, 3), dtype=np.uint8)
    ft = torch.from_numpy(f).cuda()
    eng = engine(480, 640, max_batch=2)
    g = {}
    for bits in (15, 14):
        eng.set_option(L.OPT_GRAY_COEFFS, bits)
        g[bits] = eng.bgr2gray(ft).cpu().numpy()
        assert np.array_equal(g[bits], np.stack([O.bgr2gray(x, bits) for x in f]))
        # a crop view (odd offsets) through the strided addressing
        assert np.array_equal(eng.bgr2gray(torch.from_numpy(np.pad(f, ((0, 0), (3, 5), (7, 9), (0, 0)))).cuda()
                                           [:, 3:483, 7:647]).cpu().numpy(), g[bits])
    assert (g[15] != g[14]).any()
    eng.close()


def test_capacity_frame_inside_a_batch_is_reported(tmp_path):
    """A frame that exceeds the workspace (more runs than the labelling tables hold) in the MIDDLE of a batch must not
    vanish from the output: `counts` carries the status and the drop-in raises, naming the frame."""
    from vbs_amd.marker_detection import MarkerTracker
    from vbs_amd.pipeline import track_shard
    spec = S.config2()
    frames = S.make_frames(spec, range(4), seed=2)
    # 33 x 33 dots of 14 px at pitch 30: 928 contours in the opened area mask, beyond the 512 the workspace holds
    frames[2] = S.make_frames(S.grid_spec(spec.width, spec.height, 33, 30, 14, name="dense"), [1], seed=0)[0]
    eng = engine(spec.height, spec.width, max_markers=1024, max_batch=4)
    ft = torch.from_numpy(frames).cuda()
    _, _, counts = eng.track_to_3d(ft, np.array([[100.0, 100.0]]))
    c = counts.cpu().numpy()
    assert c[2] == L.VBS_ECAPACITY and (c[[0, 1, 3]] == spec.n_markers).all()
    with pytest.raises(L.VbsError, match="frame 2"):
        track_shard(eng, ft, 4, cam=None)
    np.save(tmp_path / "clip.npy", frames)
    trk = MarkerTracker({"video_path": str(tmp_path / "clip.npy"), "output_dir": str(tmp_path / "o"),
                         "crop_ratios": (0, 0, 0, 0), "id_mode": "full"})
    with pytest.raises(L.VbsError, match="frame 2"):
        trk.process()
    eng.close()


@pytest.mark.parametrize("workload,pipelined", [("c2", 1), ("c2", 0), ("c5", 1)])
def test_track_shard_two_ranks_on_one_gpu(tmp_path, workload, pipelined):
    """e (config 4's path at small scale; `c5` = config 5's: 1920x1200, 441 markers, plane fit per shard): two ranks as
    fresh child processes sharing GPU 0 (gloo), contiguous shards, reference-table broadcast, the tables gathered either
    pass pair by pass pair (`pipelined`, dist.TableGather) or by the single collective of SURVEY 8(e) (dist.gather_tables),
    per-rank last-seen displacement with a look-back across the shard edge, per-shard plane fit - everything equal to the
    single-rank result.  (No scaling number follows from this: both ranks share one GPU and the transport is gloo.)"""
    import socket
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "helpers"))
    import shard_worker as SW
    from vbs_amd.pipeline import track_shard
    n_total = 12 if workload == "c2" else 6
    spec, frames = SW.make_clip(n_total, workload)
    K, dist, R, T = S.default_camera(spec)
    cam = L.make_camera(K, dist, R, T, 2.0)
    eng = engine(spec.height, spec.width, max_markers=1024 if workload == "c5" else 512, max_batch=2 if workload == "c5" else 4)
    one = track_shard(eng, torch.from_numpy(frames).cuda(), n_total, cam=cam, warmup_frames=0)
    table1, disp1 = one.table.cpu().numpy(), one.disp.cpu().numpy()
    assert (one.counts.cpu().numpy() == np.where(np.arange(n_total) == n_total // 2, spec.n_markers - 1, spec.n_markers)).all()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(os.path.dirname(__file__), "helpers", "shard_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(n_total), str(tmp_path), workload, str(pipelined)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    for r in range(2):
        z = np.load(tmp_path / f"rank{r}.npz")
        a, b = int(z["span"][0]), int(z["span"][1])
        assert (a, b) == ((0, n_total // 2), (n_total // 2, n_total))[r]
        assert np.array_equal(z["ids"], one.ids) and np.array_equal(z["xy"], one.ref_xy)
        assert np.array_equal(z["table"], table1)                      # the gathered table, bit for bit
        assert np.array_equal(z["disp"], disp1[a:b], equal_nan=True)   # this rank's frames of the displacement
        assert np.array_equal(z["plane"], one.plane.cpu().numpy()[a:b], equal_nan=True)
    # the look-back did cross the edge: the painted-out marker is unseen in the first frame of rank 1 and measured in the next
    # against the last frame of rank 0
    e = n_total // 2
    slot = int(np.nonzero((table1[e, :, 0].astype(int) & 1) == 0)[0][0])
    assert disp1[e + 1, slot, 0] == 1 and disp1[e, slot, 0] == 0
    want = table1[e + 1, slot, 6:9] - table1[e - 1, slot, 6:9]
    assert np.allclose(disp1[e + 1, slot, 1:4], want, atol=2e-6)
    if workload == "c5":                                   # the per-shard plane fit is the single-rank one (441 markers per frame)
        pl = one.plane.cpu().numpy()
        assert pl.shape[0] == n_total and np.isfinite(pl[:, :4]).all()
    eng.close()
