"""CPU ORACLE — test infrastructure, NOT product code.

A NumPy/SciPy restatement of the reference hot path (UPM-ROB-Lab/Vision-basedSensor,
`code/Marker_Tracking/marker_detection.py` + `code/Marker_Calibration/3d_reconstruction.py`),
stage by stage with the reference's own stage boundaries.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package; the product
(`vision-basedsensor_amd/`) never does and has no CPU fallback.

Pinning status:
  * every cv2-free function body of the reference (`_gkern`, `_normxcorr2`, `_process_first_frame`,
    `_track_markers`, `_calculate_3d_position`) was executed here, extracted from the reference file by
    AST with the real NumPy/SciPy/scikit-learn, and its outputs are committed under `tests/golden/`
    (`tests/golden/make_golden.py`); `tests/test_oracle_golden.py` holds this file to them.  The
    SciPy calls of `_marker_center:170-185` are made literally below.
  * the OpenCV calls (`cvtColor`, `GaussianBlur`, `inRange`, `morphologyEx`, `findContours`,
    `fitEllipse`, `pointPolygonTest`, `undistortPoints`) are restated from the published OpenCV 4.x
    algorithms; OpenCV is not installable here and the reference pins no version and holds no
    fixture, so those stages are **parity unpinned** against real cv2.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
from scipy import ndimage
from scipy.ndimage import maximum_filter, minimum_filter
from scipy.signal import fftconvolve
from scipy.spatial.distance import cdist

# ------------------------------------------------------------------------------------------------
# a1  crop                                                        marker_detection.py:78-91, 62-67
# ------------------------------------------------------------------------------------------------

def crop_box(width: int, height: int, crop_ratios: Sequence[float]) -> Tuple[int, int, int, int]:
    """(left, right, top, bottom) with the reference's int() truncation (`:81-84`)."""
    left = int(width * crop_ratios[0])
    right = width - int(width * crop_ratios[1])
    top = int(height * crop_ratios[2])
    bottom = height - int(height * crop_ratios[3])
    return left, right, top, bottom


# ------------------------------------------------------------------------------------------------
# a3  BGR -> gray                                                  marker_detection.py:114 (cv2)
# ------------------------------------------------------------------------------------------------

GRAY_COEFFS = {15: (3735, 19235, 9798), 14: (1868, 9617, 4899)}     # (B, G, R) weights, sum 2^bits


def bgr2gray(frame: np.ndarray, bits: int = 15) -> np.ndarray:
    """OpenCV 8-bit `COLOR_BGR2GRAY` in fixed point  [OpenCV-knowledge; the version is not pinned by the reference]:
    bits=15 (default): OpenCV 4.x `RGB2Gray<uchar>`, (3735 B + 19235 G + 9798 R + 2^14) >> 15;
    bits=14: OpenCV <= 3.4.1, (1868 B + 9617 G + 4899 R + 2^13) >> 14.
    Both are the identity where B = G = R; on coloured pixels they differ by at most one grey level."""
    if frame.ndim == 2:
        return frame
    cb, cg, cr = GRAY_COEFFS[bits]
    f = frame.astype(np.int64)
    return ((f[..., 0] * cb + f[..., 1] * cg + f[..., 2] * cr + (1 << (bits - 1))) >> bits).astype(np.uint8)


# ------------------------------------------------------------------------------------------------
# a4  GaussianBlur on uint8                                  marker_detection.py:118-119,123-124 (cv2)
# ------------------------------------------------------------------------------------------------

def gaussian_kernel_f64(ksize: int, sigma: float) -> np.ndarray:
    """`cv::getGaussianKernel` for sigma > 0: exp(-(i-(k-1)/2)^2 / 2 sigma^2), normalised."""
    xi = (1 - ksize + 2 * np.arange(ksize)).astype(np.float64)       # 2 * (i - (k-1)/2)
    v = np.exp(xi * xi * (-0.125 / (sigma * sigma)))
    n2 = (ksize - 1) // 2
    s = 2.0 * float(np.sum(v[:n2])) + 1.0
    return v * (1.0 / s)


def gaussian_kernel_q8(ksize: int, sigma: float) -> np.ndarray:
    """OpenCV 4's bit-exact uint8 path: the kernel in 8 fractional bits with error diffusion from
    the ends towards the centre, centre tap = 256 - sum of the others  [OpenCV-knowledge]."""
    k = gaussian_kernel_f64(ksize, sigma)
    n2 = ksize // 2
    out = np.zeros(ksize, dtype=np.int64)
    err, tot = 0.0, 0
    for i in range(n2):
        adj = k[i] * 256.0 + err
        v0 = int(np.rint(adj))            # cvRound: half to even
        err = adj - v0
        out[i] = out[ksize - 1 - i] = v0
        tot += v0
    out[n2] = 256 - 2 * tot
    return out


def gaussian_blur_u8(gray: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """uint8 GaussianBlur, fixed-point model: (sum_y sum_x ky kx p + 2^15) >> 16, BORDER_REFLECT_101.
    Exact integer arithmetic, so independent of summation order."""
    k = gaussian_kernel_q8(ksize, sigma)
    g = gray.astype(np.int64)
    h = ndimage.correlate1d(g, k, axis=1, mode="mirror")      # 'mirror' == reflect-101
    v = ndimage.correlate1d(h, k, axis=0, mode="mirror")
    return ((v + 32768) >> 16).astype(np.uint8)


def gaussian_blur_u8_float(gray: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """Alternative float64 model (round-half-even, saturate) — used only to report how far the
    two plausible cv2 models are apart; the HIP path follows `gaussian_blur_u8`."""
    k = gaussian_kernel_f64(ksize, sigma)
    g = gray.astype(np.float64)
    h = ndimage.correlate1d(g, k, axis=1, mode="mirror")
    v = ndimage.correlate1d(h, k, axis=0, mode="mirror")
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def branch_params(height: int) -> dict:
    """The two parameter sets of `_find_markers` (`:117-126,129`) and `_marker_center` (`:170`)."""
    if height <= 480:
        return dict(k3=21, s3=4.56, k8=35, s8=11.4, tl=33, tsig=7.4, thresh=35, hi=180, ns=8)
    return dict(k3=39, s3=8.0, k8=101, s8=20.0, tl=80, tsig=13.0, thresh=20, hi=200, ns=14)


# ------------------------------------------------------------------------------------------------
# a6/a7  template + normalised cross-correlation                   marker_detection.py:137-164
# ------------------------------------------------------------------------------------------------

def gkern(l: int = 5, sig: float = 1.0) -> np.ndarray:
    """`_gkern` (`:138-143`): l x l Gaussian on linspace(-(l-1)/2, (l-1)/2, l), sum 1."""
    ax = np.linspace(-(l - 1) / 2.0, (l - 1) / 2.0, l)
    xx, yy = np.meshgrid(ax, ax)
    kernel = np.exp(-0.5 * (np.square(xx) + np.square(yy)) / np.square(sig))
    return kernel / np.sum(kernel)


def normxcorr2(template: np.ndarray, image: np.ndarray, mode: str = "same") -> np.ndarray:
    """`_normxcorr2` (`:146-164`), the same three FFT convolutions in float64."""
    t = template - np.mean(template)
    im = image - np.mean(image)
    flipped = np.flipud(np.fliplr(t))
    num = fftconvolve(im, flipped.conj(), mode=mode)
    ones = np.ones(t.shape)
    var = fftconvolve(np.square(im), ones, mode=mode)
    var -= np.square(fftconvolve(im, ones, mode=mode)) / np.prod(t.shape)
    var[var < 0] = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        out = num / np.sqrt(var * np.sum(np.square(t)))
    out[np.logical_not(np.isfinite(out))] = 0
    return out


def ncc_window(l: int) -> Tuple[int, int]:
    """Offsets (lo, hi) of the `mode='same'` correlation window: rows y+lo .. y+hi.
    fftconvolve crops the full result from (l-1)//2, so l=80 -> (-40, +39), l=33 -> (-16, +16)."""
    return -((l - 1) - (l - 1) // 2), (l - 1) // 2


def normxcorr2_direct(template: np.ndarray, image: np.ndarray) -> np.ndarray:
    """Spatial-domain evaluation of the same quantity (no FFT): what the HIP kernel computes.
    Zero padding applies to the MEAN-SUBTRACTED image, so border windows see n < l*l samples:
      num   = sum_W t I - tbar sum_W I - mu (sum_W t - n tbar)
      var   = sum_W I'^2 - (sum_W I')^2 / l^2
    Requires a separable template (true for `gkern`)."""
    l = template.shape[0]
    lo, hi = ncc_window(l)
    H, W = image.shape
    img = image.astype(np.float64)
    mu = np.mean(image)
    tbar = np.mean(template)
    tp = template - tbar
    T2 = np.sum(np.square(tp))
    # separable factors of the (already normalised) template
    gy = template.sum(axis=1)
    gx = template.sum(axis=0)
    gx = gx / gx.sum()
    # correlation window index u = i - y - lo in [0, l)
    def corr1d(a, k, axis):          # scipy window starts at i - l//2 == i + lo for odd and even l
        assert -(l // 2) == lo
        return ndimage.correlate1d(a, k, axis=axis, mode="constant", cval=0.0)
    sum_tI = corr1d(corr1d(img, gx, 1), gy, 0)
    onesk = np.ones(l)
    sum_I = corr1d(corr1d(img, onesk, 1), onesk, 0)
    sum_I2 = corr1d(corr1d(img * img, onesk, 1), onesk, 0)
    inside = np.ones((H, W))
    n = corr1d(corr1d(inside, onesk, 1), onesk, 0)
    sum_t = corr1d(corr1d(inside, gx, 1), gy, 0)
    num = sum_tI - tbar * sum_I - mu * (sum_t - n * tbar)
    s1 = sum_I - n * mu
    s2 = sum_I2 - 2 * mu * sum_I + n * mu * mu
    var = s2 - s1 * s1 / (l * l)
    var[var < 0] = 0
    with np.errstate(divide="ignore", invalid="ignore"):
        out = num / np.sqrt(var * T2)
    out[np.logical_not(np.isfinite(out))] = 0
    return out


# ------------------------------------------------------------------------------------------------
# a3-a8  _find_markers                                             marker_detection.py:112-135
# ------------------------------------------------------------------------------------------------

def dog_image(gray: np.ndarray) -> np.ndarray:
    """`im_blur_8 - im_blur_3 + 15` in uint8 — wraps mod 256 (`:128`)."""
    p = branch_params(gray.shape[0])
    b3 = gaussian_blur_u8(gray, p["k3"], p["s3"])
    b8 = gaussian_blur_u8(gray, p["k8"], p["s8"])
    return ((b8.astype(np.int64) - b3.astype(np.int64) + 15) & 0xFF).astype(np.uint8)


def in_range(img: np.ndarray, lo: int, hi: int) -> np.ndarray:
    """`cv2.inRange`: 255 where lo <= x <= hi (inclusive), else 0."""
    return np.where((img >= lo) & (img <= hi), 255, 0).astype(np.uint8)


def find_markers(frame: np.ndarray, ncc: str = "fft", gray_bits: int = 15) -> Tuple[np.ndarray, np.ndarray]:
    """`_find_markers(frame)` -> (mask uint8 {0,1}, area_mask uint8 {0,255})."""
    gray = bgr2gray(frame, gray_bits)
    p = branch_params(gray.shape[0])
    area_mask = in_range(dog_image(gray), p["thresh"], p["hi"])
    template = gkern(p["tl"], p["tsig"])
    if ncc == "fft":
        nrm = normxcorr2(template, area_mask)
    else:
        nrm = normxcorr2_direct(template, area_mask)
    mask = (nrm > 0.1).astype("uint8")
    return mask, area_mask


# ------------------------------------------------------------------------------------------------
# a9-a11  band, labels, centroids                                  marker_detection.py:170-185
# ------------------------------------------------------------------------------------------------

def band_mask(mask: np.ndarray) -> np.ndarray:
    ns = 8 if mask.shape[0] <= 480 else 14
    data_max = maximum_filter(mask, ns)
    maxima = (mask == data_max)
    diff = ((data_max - minimum_filter(mask, ns)) > 0)
    maxima[diff == 0] = 0
    return maxima


def band_centroids(mask: np.ndarray) -> Tuple[np.ndarray, np.ndarray, int]:
    """(centers [n,2] as (row, col) float64, label image, n) — `ndimage.label` default (4-conn)."""
    maxima = band_mask(mask)
    labeled, n = ndimage.label(maxima)
    if n == 0:
        return np.zeros((0, 2)), labeled, 0
    centers = np.array(ndimage.center_of_mass(mask, labeled, range(1, n + 1)))
    if centers.ndim == 1 and n == 1:
        centers = centers.reshape(1, -1)
    return centers, labeled, n


# ------------------------------------------------------------------------------------------------
# a12  5x5 open + external contours                                marker_detection.py:188-196 (cv2)
# ------------------------------------------------------------------------------------------------

def morph_open5(fg: np.ndarray) -> np.ndarray:
    """Binary 5x5 opening; outside the image never erodes / never dilates (cv2 default border)."""
    st = np.ones((5, 5), dtype=bool)
    er = ndimage.binary_erosion(fg.astype(bool), structure=st, border_value=1)
    return ndimage.binary_dilation(er, structure=st, border_value=0)


# 8-neighbour offsets in OpenCV chain-code order: 0=E, 1=NE, 2=N, 3=NW, 4=W, 5=SW, 6=S, 7=SE (y down)
_DX = (1, 1, 0, -1, -1, -1, 0, 1)
_DY = (0, -1, -1, -1, 0, 1, 1, 1)


def find_contours_external(fg: np.ndarray, approx_simple: bool = True) -> List[np.ndarray]:
    """`cv2.findContours(img, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)` restated  [OpenCV-knowledge].

    Raster scan of a zero-padded copy; an outer border starts where a 0 -> 1 transition is met
    while the last border pixel seen on the row is not a positive ("still inside") mark.  The border
    is followed Suzuki-Abe style over the 8-neighbourhood; visited pixels get +nbd, or -nbd when
    their east neighbour was examined as background (right bound).  Hole borders are not followed.
    With CHAIN_APPROX_SIMPLE a point is kept iff the outgoing step direction differs from the
    incoming one.  The returned list is in OpenCV's order: last contour found comes first.
    """
    H, W = fg.shape
    f = np.zeros((H + 2, W + 2), dtype=np.int32)
    f[1:-1, 1:-1] = (fg != 0)
    contours: List[np.ndarray] = []
    nbd = 1
    for y in range(1, H + 1):
        row = f[y]
        lnbd_x = 0
        prev = 0
        x = 1
        while x <= W + 1:
            p = int(row[x])
            if p == prev:
                x += 1
                continue
            is_outer = (prev == 0 and p == 1)
            if is_outer and not (row[lnbd_x] > 0):
                nbd += 1
                contours.append(_follow_border(f, x, y, nbd, approx_simple))
                prev = int(row[x])          # the scanner resumes at x+1 with prev = new mark,
                x += 1                      # without moving lnbd to the start pixel
                continue
            prev = p
            if p != 0 and p != 1:
                lnbd_x = x
            x += 1
    contours.reverse()
    return contours


def _follow_border(f: np.ndarray, x0: int, y0: int, nbd: int, approx_simple: bool) -> np.ndarray:
    """Trace one outer border starting at padded (x0, y0); returns [k, 2] int32 (x, y) unpadded."""
    # 1) search clockwise from W for the pixel "before" the start (the previous border pixel)
    s_end = s = 4
    found = False
    while True:
        s = (s - 1) & 7
        if f[y0 + _DY[s], x0 + _DX[s]] != 0:
            found = True
            break
        if s == s_end:
            break
    pts: List[Tuple[int, int]] = []
    if not found:                                   # isolated pixel
        f[y0, x0] = -nbd
        return np.array([[x0 - 1, y0 - 1]], dtype=np.int32)
    x1, y1 = x0 + _DX[s], y0 + _DY[s]               # i1: where the closing step comes from
    prev_s = s ^ 4
    x3, y3 = x0, y0
    while True:
        s_end = s
        # 2) counter-clockwise from the direction after the previous pixel: first non-zero
        while True:
            s += 1
            if f[y3 + _DY[s & 7], x3 + _DX[s & 7]] != 0:
                break
        s_found = s & 7
        # "right bound": the scan from s_end+1 .. s passed over direction 0 (east) as background
        passed_east = _passed_east(s_end, s)
        if passed_east:
            f[y3, x3] = -nbd
        elif f[y3, x3] == 1:
            f[y3, x3] = nbd
        if (not approx_simple) or s_found != prev_s:
            pts.append((x3 - 1, y3 - 1))
            prev_s = s_found
        x4, y4 = x3 + _DX[s_found], y3 + _DY[s_found]
        if x4 == x0 and y4 == y0 and x3 == x1 and y3 == y1:
            break
        x3, y3 = x4, y4
        s = (s_found + 4) & 7
    return np.array(pts, dtype=np.int32).reshape(-1, 2)


def _passed_east(s_end: int, s: int) -> bool:
    """Directions s_end+1 .. s-1 were examined as zero; was direction 0 (mod 8) among them?
    (OpenCV: `(unsigned)(s - 1) < (unsigned)s_end` on the unwrapped counter.)"""
    for d in range(s_end + 1, s):
        if (d & 7) == 0:
            return True
    return False


# ------------------------------------------------------------------------------------------------
# a13  fitEllipse / pointPolygonTest                               marker_detection.py:208,228 (cv2)
# ------------------------------------------------------------------------------------------------

def fit_ellipse(points: np.ndarray) -> Tuple[Tuple[float, float], Tuple[float, float], float]:
    """`cv2.fitEllipse` (OpenCV 4 `fitEllipseNoDirect`, the LIN algorithm)  [OpenCV-knowledge].

    Two linear least-squares fits on points centred at their float32 mean and scaled so that
    sum(|x|+|y|) = 100: (1) the general conic A..E with rhs 10000 gives the centre, (2) a
    re-fit of A..C about that centre gives axes and angle.  Returns ((cx,cy),(w,h),angle) as
    float32-rounded Python floats with w <= h, like `RotatedRect`."""
    pts = np.asarray(points, dtype=np.float32).reshape(-1, 2)
    n = pts.shape[0]
    if n < 5:
        raise ValueError("There should be at least 5 points to fit the ellipse")
    c = np.zeros(2, dtype=np.float32)
    for i in range(n):                        # float32 running sum, as Point2f +=
        c = (c + pts[i]).astype(np.float32)
    c = (c / np.float32(n)).astype(np.float32)
    q = (pts - c).astype(np.float32)          # Point2f subtraction
    s = float(np.sum(np.abs(q[:, 0].astype(np.float64)) + np.abs(q[:, 1].astype(np.float64))))
    eps32 = float(np.finfo(np.float32).eps)
    scale = 100.0 / (s if s > eps32 else eps32)

    def design(qq):
        px = qq[:, 0].astype(np.float64) * scale
        py = qq[:, 1].astype(np.float64) * scale
        return px, py, np.stack([-px * px, -py * py, -px * py, px, py], axis=1)

    px, py, A = design(q)
    b = np.full(n, 10000.0)
    u, w, vt = np.linalg.svd(A, full_matrices=False)
    if w[0] * eps32 > w[4]:
        # degenerate (e.g. collinear) input: OpenCV nudges the points by +-eps and refits
        eps = np.float32(s / (n * 2) * 1e-3)
        ofs = np.array([[((i & 1) * 2 - 1) * eps, ((i & 2) - 1) * eps] for i in range(n)],
                       dtype=np.float32)
        q = ((pts + ofs).astype(np.float32) - c).astype(np.float32)
        px, py, A = design(q)
        u, w, vt = np.linalg.svd(A, full_matrices=False)
    gfp = _svd_backsubst(u, w, vt, b)
    # centre: gradient of the conic = 0
    A2 = np.array([[2 * gfp[0], gfp[2]], [gfp[2], 2 * gfp[1]]])
    rp = np.zeros(5)
    rp[:2] = _lstsq_svd(A2, np.array([gfp[3], gfp[4]]))
    # re-fit A..C about that centre
    dx, dy = px - rp[0], py - rp[1]
    A3 = np.stack([dx * dx, dy * dy, dx * dy], axis=1)
    g3 = _lstsq_svd(A3, np.ones(n))
    min_eps = 1e-8
    rp[4] = -0.5 * math.atan2(g3[2], g3[1] - g3[0])
    if abs(g3[2]) > min_eps:
        t = g3[2] / math.sin(-2.0 * rp[4])
    else:
        t = g3[1] - g3[0]
    rp[2] = abs(g3[0] + g3[1] - t)
    if rp[2] > min_eps:
        rp[2] = math.sqrt(2.0 / rp[2])
    rp[3] = abs(g3[0] + g3[1] + t)
    if rp[3] > min_eps:
        rp[3] = math.sqrt(2.0 / rp[3])
    cx = np.float32(np.float32(rp[0] / scale) + c[0])
    cy = np.float32(np.float32(rp[1] / scale) + c[1])
    wd = np.float32(rp[2] * 2 / scale)
    ht = np.float32(rp[3] * 2 / scale)
    ang = np.float32(rp[4] * 180.0 / math.pi)
    if wd > ht:
        wd, ht = ht, wd
        ang = np.float32(90 + rp[4] * 180.0 / math.pi)
    if ang < -180:
        ang = np.float32(ang + 360)
    if ang > 360:
        ang = np.float32(ang - 360)
    return (float(cx), float(cy)), (float(wd), float(ht)), float(ang)


def _svd_backsubst(u, w, vt, b):
    thr = np.finfo(np.float64).eps * 2 * np.sum(w)          # cv::SVBackSubst threshold
    winv = np.where(w > thr, 1.0 / np.where(w > thr, w, 1.0), 0.0)
    return vt.T @ (winv * (u.T @ b))


def _lstsq_svd(A, b):
    u, w, vt = np.linalg.svd(A, full_matrices=False)
    return _svd_backsubst(u, w, vt, b)


def point_polygon_test(contour: np.ndarray, pt: Tuple[float, float]) -> int:
    """`cv2.pointPolygonTest(contour, pt, False)`: +1 inside, 0 on an edge/vertex, -1 outside.
    The query is a Point2f, i.e. rounded to float32 first  [OpenCV-knowledge]."""
    cnt = np.asarray(contour).reshape(-1, 2).astype(np.float64)
    x = float(np.float32(pt[0]))
    y = float(np.float32(pt[1]))
    total = cnt.shape[0]
    counter = 0
    vx, vy = cnt[total - 1]
    for i in range(total):
        v0x, v0y = vx, vy
        vx, vy = cnt[i]
        if (v0y <= y and vy <= y) or (v0y > y and vy > y) or (v0x < x and vx < x):
            if y == vy and (x == vx or (y == v0y and
                                        ((v0x <= x <= vx) or (vx <= x <= v0x)))):
                return 0
            continue
        dist = (y - v0y) * (vx - v0x) - (x - v0x) * (vy - v0y)
        if dist == 0:
            return 0
        if vy < v0y:
            dist = -dist
        counter += dist > 0
    return -1 if counter % 2 == 0 else 1


# ------------------------------------------------------------------------------------------------
# a9-a13  _marker_center                                           marker_detection.py:166-249
# ------------------------------------------------------------------------------------------------

def marker_center(mask: np.ndarray, area_mask: np.ndarray, return_debug: bool = False):
    """`_marker_center(mask, area_mask)` -> list of {'center','major_axis','minor_axis','angle'}.
    The emitted centre is the band centroid (a11), not the ellipse centre."""
    centers, _, n = band_centroids(mask)
    if n == 0 or centers.size == 0:
        return ([], {}) if return_debug else []
    opened = morph_open5(area_mask != 0)
    contours = find_contours_external(opened)
    centers_xy = [(c[1], c[0]) for c in centers]
    unmatched = list(enumerate(centers_xy))
    output = []
    dbg = dict(centers_xy=centers_xy, contours=contours, ellipses=[], matched=[])
    for contour in contours:
        if len(contour) < 5:
            continue
        (cx, cy), (w, h), angle = fit_ellipse(contour)
        if w > h:
            major, minor, ell_angle = w, h, angle
        else:
            major, minor, ell_angle = h, w, angle + 90
        dbg["ellipses"].append(((cx, cy), (w, h), angle))
        if minor < 5:
            continue
        best, min_dist = -1, float("inf")
        threshold = (minor / 10) ** 2
        for i, (_orig, (x, y)) in enumerate(unmatched):
            if point_polygon_test(contour, (x, y)) < 0:
                continue
            dist = (x - cx) ** 2 + (y - cy) ** 2
            if dist < threshold and dist < min_dist:
                min_dist, best = dist, i
        if best != -1:
            orig, (x, y) = unmatched.pop(best)
            dbg["matched"].append(orig)
            output.append({"center": (x, y), "major_axis": float(major),
                           "minor_axis": float(minor), "angle": float(ell_angle)})
    return (output, dbg) if return_debug else output


# ------------------------------------------------------------------------------------------------
# a14  first-frame IDs                                             marker_detection.py:275-347
# ------------------------------------------------------------------------------------------------

def kmeans_1d_optimal(values: np.ndarray, k: int) -> np.ndarray:
    """Deterministic 1-D k-means: the global optimum of the k-means objective by dynamic programming
    over the sorted values (clusters of 1-D data are contiguous).  Returns labels in input order,
    numbered by ascending cluster mean.  Replaces the unseeded `KMeans(n_init=10)` (`:308`)."""
    v = np.asarray(values, dtype=np.float64)
    n = v.size
    k = min(k, n)
    order = np.argsort(v, kind="stable")
    s = v[order]
    p1 = np.concatenate([[0.0], np.cumsum(s)])
    p2 = np.concatenate([[0.0], np.cumsum(s * s)])

    def cost(i, j):        # SSE of s[i:j]
        m = j - i
        sm = p1[j] - p1[i]
        return (p2[j] - p2[i]) - sm * sm / m

    INF = float("inf")
    D = np.full((k + 1, n + 1), INF)
    B = np.zeros((k + 1, n + 1), dtype=np.int64)
    D[0, 0] = 0.0
    for c in range(1, k + 1):
        for j in range(c, n + 1):
            best, arg = INF, c - 1
            for i in range(c - 1, j):
                if D[c - 1, i] == INF:
                    continue
                val = D[c - 1, i] + cost(i, j)
                if val < best:
                    best, arg = val, i
            D[c, j], B[c, j] = best, arg
    labels_sorted = np.zeros(n, dtype=np.int64)
    j = n
    for c in range(k, 0, -1):
        i = B[c, j]
        labels_sorted[i:j] = c - 1
        j = i
    labels = np.zeros(n, dtype=np.int64)
    labels[order] = labels_sorted
    return labels


def process_first_frame(markers: List[dict], num_layers: int = 5, id_mode: str = "as_written",
                        kmeans: str = "optimal") -> Dict[Tuple[int, int], dict]:
    """`_process_first_frame` (`:275-347`).  Returns the ordered dict {(layer, idx): marker+Ox,Oy}.

    id_mode='as_written' reproduces the published behaviour: every non-centre marker is first stored
    under the key (layer, -1) (`:318-321`), so one marker per layer survives (the LAST one in list
    order) and becomes (layer, 0).  id_mode='full' gives every marker its own (layer, angle_idx)
    as the docstring at `tracking.py:13-16` describes; insertion order is layer-major, then by angle.
    kmeans='sklearn' calls `KMeans(n_clusters, n_init=10)` like the reference (unseeded);
    'optimal' uses `kmeans_1d_optimal`."""
    if not markers:
        raise ValueError("No markers detected in first frame!")
    out: Dict[Tuple[int, int], dict] = {}
    centers = np.array([m["center"] for m in markers])
    center_pos = np.mean(centers, axis=0)
    dists = np.linalg.norm(centers - center_pos, axis=1)
    ci = int(np.argmin(dists))
    cm = markers[ci]
    out[(0, 0)] = {**cm, "Ox": cm["center"][0], "Oy": cm["center"][1]}
    remaining = [m for i, m in enumerate(markers) if i != ci]
    if not remaining:
        return out
    rc = np.array([m["center"] for m in remaining])
    vec = rc - cm["center"]
    radii = np.linalg.norm(vec, axis=1)
    angles = np.arctan2(vec[:, 1], vec[:, 0])
    if kmeans == "sklearn":
        from sklearn.cluster import KMeans
        km = KMeans(n_clusters=num_layers, n_init=10)
        km.fit(radii.reshape(-1, 1))
        order = np.argsort(km.cluster_centers_.flatten())
        lmap = {int(o): new + 1 for new, o in enumerate(order)}
        layers = np.array([lmap[int(l)] for l in km.labels_])
    else:
        layers = kmeans_1d_optimal(radii, num_layers) + 1
    if id_mode == "as_written":
        for i, m in enumerate(remaining):
            out[(int(layers[i]), -1)] = {**m, "angle_rad": angles[i], "Ox": m["center"][0],
                                         "Oy": m["center"][1]}
        for layer in range(1, num_layers + 1):
            if (layer, -1) in out:
                m = out.pop((layer, -1))
                out[(layer, 0)] = m
        return out
    for layer in range(1, num_layers + 1):
        idxs = [i for i in range(len(remaining)) if layers[i] == layer]
        if not idxs:
            continue
        idxs.sort(key=lambda i: angles[i])                     # stable, like list.sort
        start = int(np.argmin([abs(angles[i]) for i in idxs]))
        for pos, i in enumerate(idxs):
            m = remaining[i]
            out[(layer, (pos - start) % len(idxs))] = {**m, "angle_rad": angles[i],
                                                       "Ox": m["center"][0], "Oy": m["center"][1]}
    return out


# ------------------------------------------------------------------------------------------------
# a15  per-frame tracking rows                                     marker_detection.py:349-396
# ------------------------------------------------------------------------------------------------

CSV_COLUMNS = ("frameno", "row", "col", "Ox", "Oy", "Cx", "Cy", "major_axis", "minor_axis", "angle")


def track_markers(first_frame_markers: Dict[Tuple[int, int], dict], markers: List[dict],
                  frame_count: int, min_dist: float = 20) -> List[dict]:
    """`_track_markers` (`:349-396`): nearest current marker per reference ID (first on ties),
    dropped when farther than `min_marker_distance`; not exclusive."""
    if not first_frame_markers or not markers:
        return []
    cur = np.array([m["center"] for m in markers])
    lookup = {tuple(m["center"]): m for m in markers}
    rows = []
    for (layer, idx), ref in first_frame_markers.items():
        d = cdist([np.array([ref["Ox"], ref["Oy"]])], cur)[0]
        j = int(np.argmin(d))
        if d[j] > min_dist:
            continue
        c = lookup.get(tuple(cur[j]))
        if c:
            rows.append({"frameno": frame_count, "row": layer, "col": idx, "Ox": ref["Ox"],
                         "Oy": ref["Oy"], "Cx": c["center"][0], "Cy": c["center"][1],
                         "major_axis": c["major_axis"], "minor_axis": c["minor_axis"],
                         "angle": c["angle"]})
    return rows


def process_frames(frames: Sequence[np.ndarray], crop_ratios=(0, 0, 0, 0), num_layers: int = 5,
                   min_dist: float = 20, id_mode: str = "as_written", kmeans: str = "optimal",
                   ncc: str = "fft", calibration: Optional[dict] = None):
    """`MarkerTracker.process` (`:429-462`) over in-memory frames -> (rows, first_frame_markers)."""
    ref: Dict[Tuple[int, int], dict] = {}
    rows: List[dict] = []
    for fc, frame in enumerate(frames):
        H, W = frame.shape[:2]
        l, r, t, b = crop_box(W, H, crop_ratios)
        cropped = frame[t:b, l:r]
        if calibration is not None:                    # `_preprocess_frame` :88-89
            cropped = undistort_frame(cropped, np.asarray(calibration["camera_matrix"], dtype=np.float64),
                                      np.asarray(calibration["dist_coeffs"], dtype=np.float64))
        mask, area = find_markers(cropped, ncc=ncc)
        markers = marker_center(mask, area)
        if fc == 0:
            ref = process_first_frame(markers, num_layers, id_mode, kmeans)
        rows.extend(track_markers(ref, markers, fc, min_dist))
    return rows, ref


# ------------------------------------------------------------------------------------------------
# a19-a21  3-D reconstruction                                      3d_reconstruction.py:185-316
# ------------------------------------------------------------------------------------------------

def undistort_points(points: np.ndarray, K: np.ndarray, dist: np.ndarray) -> np.ndarray:
    """`cv2.undistortPoints(pts, K, dist, None, K)` (`:187-193`): normalise, 5 fixed-point
    iterations of the inverse Brown-Conrady model, re-project with K; float64  [OpenCV-knowledge]."""
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    K = np.asarray(K, dtype=np.float64)
    k = np.zeros(12)
    d = np.asarray(dist, dtype=np.float64).ravel()
    k[:d.size] = d
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x0 = (pts[:, 0] - cx) / fx
    y0 = (pts[:, 1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    if np.any(d != 0):
        for _ in range(5):
            r2 = x * x + y * y
            icd = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
            dxx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
            dyy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
            x = (x0 - dxx) * icd
            y = (y0 - dyy) * icd
    return np.stack([x * fx + cx, y * fy + cy], axis=1)


def calculate_3d_position(u, v, diameter_px, K, R, T, marker_diameter_mm=2.0) -> np.ndarray:
    """`_calculate_3d_position` (`:195-238`).  K, R (world->cam), T are the float32 arrays of
    `load_parameters`; u, v, d arrive as float64 scalars so NumPy computes in float64."""
    fx, fy = K[0, 0], K[1, 1]
    cx, cy = K[0, 2], K[1, 2]
    f_avg = (fx + fy) / 2
    Rr = np.sqrt((u - cx) ** 2 + (v - cy) ** 2)
    if Rr < 1e-6:
        raise ValueError("Marker too close to principal point")
    d_eff = (marker_diameter_mm / f_avg) * np.sqrt(Rr ** 2 + f_avg ** 2)
    h = f_avg * (d_eff / diameter_px)
    P_cam = np.array([h * (u - cx) / fx, h * (v - cy) / fy, h]).reshape(3, 1)
    P_world = (R.T @ (P_cam - np.asarray(T).reshape(3, 1))).flatten()
    if not np.all(np.isfinite(P_world)):
        raise ValueError("Non-finite coordinates calculated")
    return P_world


XYZ_COLUMNS = ("frameno", "row", "col", "X", "Y", "Z", "dX", "dY", "dZ", "displacement")


def track_markers_3d(rows: Sequence[dict], K, dist, R, T, marker_diameter_mm=2.0, warmup_frames=100,
                     min_marker_size_px=5.0, max_displacement=50.0) -> List[dict]:
    """`load_marker_data` filtering (`:172-179`) + `_track_markers` (`:240-316`): drop
    major_axis < min size, drop the first `warmup_frames`, undistort, then per ID the displacement
    against the frame in which that ID was LAST SEEN; rows with |d| > limit are dropped (the
    last-seen table is still updated)."""
    rows = [r for r in rows if r["major_axis"] >= min_marker_size_px]
    if not rows:
        return []
    rows = sorted(rows, key=lambda r: r["frameno"])
    fmin = rows[0]["frameno"]
    if warmup_frames > 0:
        rows = [r for r in rows if r["frameno"] >= fmin + warmup_frames]
    if not rows:
        return []
    uv = undistort_points(np.array([[r["Cx"], r["Cy"]] for r in rows], dtype=np.float64), K, dist)
    last: Dict[Tuple[int, int], Tuple[float, float, float]] = {}
    out: List[dict] = []
    i = 0
    while i < len(rows):
        j = i
        cur = {}
        while j < len(rows) and rows[j]["frameno"] == rows[i]["frameno"]:
            r = rows[j]
            key = (r["row"], r["col"])
            u, v, d = float(uv[j, 0]), float(uv[j, 1]), float(r["major_axis"])
            cur[key] = (u, v, d)
            if key in last:
                try:
                    pu, pv, pd_ = last[key]
                    prev = calculate_3d_position(np.float64(pu), np.float64(pv), np.float64(pd_),
                                                 K, R, T, marker_diameter_mm)
                    curr = calculate_3d_position(np.float64(u), np.float64(v), np.float64(d),
                                                 K, R, T, marker_diameter_mm)
                    disp = curr - prev
                    mm = float(np.linalg.norm(disp))
                    if mm > max_displacement:
                        raise ValueError("Displacement too large")
                    out.append({"frameno": r["frameno"], "row": r["row"], "col": r["col"],
                                "X": curr[0], "Y": curr[1], "Z": curr[2], "dX": disp[0],
                                "dY": disp[1], "dZ": disp[2], "displacement": mm})
                except ValueError:
                    pass
            j += 1
        last.update(cur)
        i = j
    return out


# ------------------------------------------------------------------------------------------------
# f1  plane-fit pose                                               ForceDistribution.py:138-162
# ------------------------------------------------------------------------------------------------

def fit_plane(X, Y, Z) -> Tuple[float, float, float, float]:
    """Least-squares plane Z = aX + bY + c and tilt = atan(sqrt(a^2 + b^2)) in degrees."""
    A = np.vstack([X, Y, np.ones(len(X))]).T
    coeff, *_ = np.linalg.lstsq(A, Z, rcond=None)
    a, b, c = coeff
    return float(a), float(b), float(c), float(np.degrees(np.arctan(np.sqrt(a * a + b * b))))


def deviation_plane(vert_start, vert_end, tilt_start, tilt_end, ref_xyz, mode="plane", scale=1.0):
    """Deviation field and its plane, `ForceDistribution.py`: `process_marker_data` :196-204 (deviation = d_tilt - d_vert
    over the markers common to both loadings and the reference table), `visualize_deviations` :218-243 (end points =
    reference position + scale * deviation, Z from 0 in 'plane' mode :222; plane over the END points), :263 (mean scaled
    deviation), :274 (mean magnitude).  The four inputs are [M,4] arrays (present, X, Y, Z), one row per marker id.
    Returns (common [M] bool, deviation [M,3], (a, b, c, tilt_deg), mean_vec [3], mean_mag)."""
    vs, ve, ts, te = (np.asarray(t, dtype=np.float64) for t in (vert_start, vert_end, tilt_start, tilt_end))
    ref = np.asarray(ref_xyz, dtype=np.float64)
    common = (vs[:, 0] != 0) & (ve[:, 0] != 0) & (ts[:, 0] != 0) & (te[:, 0] != 0)
    d_vert = ve[:, 1:4] - vs[:, 1:4]
    d_tilt = te[:, 1:4] - ts[:, 1:4]
    deviation = np.where(common[:, None], d_tilt - d_vert, 0.0)
    z_start = ref[:, 2] if mode == "shell" else np.zeros_like(ref[:, 2])
    end = np.stack([ref[:, 0], ref[:, 1], z_start], axis=1) + scale * deviation
    plane = fit_plane(end[common, 0], end[common, 1], end[common, 2])
    mean_vec = (scale * deviation[common]).mean(axis=0)
    mean_mag = float(np.linalg.norm(deviation[common], axis=1).mean())
    return common, deviation, plane, mean_vec, mean_mag


# ------------------------------------------------------------------------------------------------
# a2 / f3  frame undistortion                                      marker_detection.py:93-109 (cv2)
# ------------------------------------------------------------------------------------------------
INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
INTER_REMAP_COEF_BITS = 15
INTER_REMAP_COEF_SCALE = 1 << INTER_REMAP_COEF_BITS


def _undistort_points_norm(pts: np.ndarray, K: np.ndarray, dist: np.ndarray) -> np.ndarray:
    """`cvUndistortPoints` to NORMALISED coordinates (no re-projection), 5 iterations."""
    k = np.zeros(12)
    d = np.asarray(dist, dtype=np.float64).ravel()
    k[:d.size] = d
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x0 = (pts[:, 0] - cx) / fx
    y0 = (pts[:, 1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    for _ in range(5):
        r2 = x * x + y * y
        icd = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        dxx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        dyy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        x = (x0 - dxx) * icd
        y = (y0 - dyy) * icd
    return np.stack([x, y], axis=1)


def get_optimal_new_camera_matrix_alpha0(K, dist, size: Tuple[int, int]) -> np.ndarray:
    """`cv2.getOptimalNewCameraMatrix(K, D, (w,h), 0, (w,h))[0]` (`:101-103`)  [OpenCV-knowledge].
    A 9x9 grid over the image is undistorted to normalised coordinates; the INNER rectangle (largest
    axis-aligned rectangle inside the undistorted border) is scaled to fill the new image."""
    w, h = size
    K = np.asarray(K, dtype=np.float64)
    N = 9
    pts = np.array([[x * (w - 1) / (N - 1), y * (h - 1) / (N - 1)] for y in range(N) for x in range(N)], dtype=np.float64)
    # cvUndistortPoints works on float32 (CV_32FC2) points in icvGetRectangles
    p = _undistort_points_norm(pts.astype(np.float32).astype(np.float64), K, dist).astype(np.float32).astype(np.float64)
    p = p.reshape(N, N, 2)
    i_x0 = np.max(p[:, 0, 0]);  i_x1 = np.min(p[:, N - 1, 0])
    i_y0 = np.max(p[0, :, 1]);  i_y1 = np.min(p[N - 1, :, 1])
    # alpha = 0: inner rectangle only
    fx0 = (w - 1) / (i_x1 - i_x0)
    fy0 = (h - 1) / (i_y1 - i_y0)
    cx0 = -fx0 * i_x0
    cy0 = -fy0 * i_y0
    M = np.eye(3)
    M[0, 0], M[1, 1], M[0, 2], M[1, 2] = fx0, fy0, cx0, cy0
    return M


def init_undistort_rectify_map_16sc2(K, dist, newK, size: Tuple[int, int]):
    """`cv2.initUndistortRectifyMap(K, D, None, newK, (w,h), CV_16SC2)` (`:106-108`): for every destination
    pixel the distorted source position in 1/32 px fixed point -> (map1 int16 [h,w,2], map2 uint16 [h,w])."""
    w, h = size
    K = np.asarray(K, dtype=np.float64)
    newK = np.asarray(newK, dtype=np.float64)
    k = np.zeros(12)
    d = np.asarray(dist, dtype=np.float64).ravel()
    k[:d.size] = d
    ir = np.linalg.inv(newK)
    jj, ii = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    X = jj * ir[0, 0] + ii * ir[0, 1] + ir[0, 2]
    Y = jj * ir[1, 0] + ii * ir[1, 1] + ir[1, 2]
    Wc = jj * ir[2, 0] + ii * ir[2, 1] + ir[2, 2]
    x, y = X / Wc, Y / Wc
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    _2xy = 2 * x * y
    kr = (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2) / (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2)
    xd = x * kr + k[2] * _2xy + k[3] * (r2 + 2 * x2)
    yd = y * kr + k[2] * (r2 + 2 * y2) + k[3] * _2xy
    u = K[0, 0] * xd + K[0, 2]
    v = K[1, 1] * yd + K[1, 2]
    iu = np.clip(np.rint(u * INTER_TAB_SIZE), -2**31, 2**31 - 1).astype(np.int64)
    iv = np.clip(np.rint(v * INTER_TAB_SIZE), -2**31, 2**31 - 1).astype(np.int64)
    map1 = np.stack([np.clip(iu >> INTER_BITS, -32768, 32767), np.clip(iv >> INTER_BITS, -32768, 32767)],
                    axis=-1).astype(np.int16)
    map2 = ((iv & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE - 1))).astype(np.uint16)
    return map1, map2


def bilinear_tab_i16() -> np.ndarray:
    """OpenCV's fixed-point bilinear weights: [1024, 4] int (w00, w01, w10, w11), each row sums to 2^15.
    Built like `initInterTab2D(INTER_LINEAR, fixpt=true)`: float32 products, saturate_cast<short>(v * 2^15),
    then the rounding residue is pushed onto the largest (or smallest) weight."""
    tab1 = np.array([[1.0 - i / INTER_TAB_SIZE, i / INTER_TAB_SIZE] for i in range(INTER_TAB_SIZE)], dtype=np.float32)
    out = np.zeros((INTER_TAB_SIZE * INTER_TAB_SIZE, 4), dtype=np.int64)
    for i in range(INTER_TAB_SIZE):            # y fraction
        for j in range(INTER_TAB_SIZE):        # x fraction
            wv = np.array([tab1[i, 0] * tab1[j, 0], tab1[i, 0] * tab1[j, 1], tab1[i, 1] * tab1[j, 0],
                           tab1[i, 1] * tab1[j, 1]], dtype=np.float32)
            iw = np.rint(wv.astype(np.float64) * INTER_REMAP_COEF_SCALE).astype(np.int64)
            isum = int(iw.sum())
            if isum != INTER_REMAP_COEF_SCALE:
                diff = isum - INTER_REMAP_COEF_SCALE
                # ksize = 2: the search window is the whole 2x2 block
                if diff < 0:
                    iw[int(np.argmax(iw))] -= diff
                else:
                    iw[int(np.argmin(iw))] -= diff
            out[i * INTER_TAB_SIZE + j] = iw
    return out


def remap_linear_16sc2(img: np.ndarray, map1: np.ndarray, map2: np.ndarray) -> np.ndarray:
    """`cv2.remap(img, map1, map2, INTER_LINEAR)` on uint8 with BORDER_CONSTANT 0 (`:109`):
    (sum w_i p_i + 2^14) >> 15 with the fixed-point weights above."""
    tab = bilinear_tab_i16()
    h, w = map2.shape
    H, W = img.shape[:2]
    src = img.astype(np.int64)
    if src.ndim == 2:
        src = src[..., None]
    sx = map1[..., 0].astype(np.int64)
    sy = map1[..., 1].astype(np.int64)
    wts = tab[map2.astype(np.int64)]                      # [h, w, 4]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return np.where(ok[..., None], v, 0)

    acc = (tap(sy, sx) * wts[..., 0:1] + tap(sy, sx + 1) * wts[..., 1:2] + tap(sy + 1, sx) * wts[..., 2:3] +
           tap(sy + 1, sx + 1) * wts[..., 3:4])
    out = ((acc + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS).astype(np.uint8)
    return out[..., 0] if img.ndim == 2 else out


def undistort_frame(frame: np.ndarray, K, dist) -> np.ndarray:
    """`MarkerTracker._undistort_frame` (`:93-109`)."""
    h, w = frame.shape[:2]
    newK = get_optimal_new_camera_matrix_alpha0(K, dist, (w, h))
    m1, m2 = init_undistort_rectify_map_16sc2(K, dist, newK, (w, h))
    return remap_linear_16sc2(frame, m1, m2)
