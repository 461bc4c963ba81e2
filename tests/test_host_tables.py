"""CPU tests of the host-side tables inside libvbs.so against the oracle (no GPU, no compute kernels):
the library must load, export every symbol of include/vbs.h, and its constant tables must match."""
import ctypes as C
import os
import re

import numpy as np
import pytest
from scipy import ndimage

import vbs_amd._lib as L
from oracle import stages as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_header_symbol():
    hdr = open(os.path.join(ROOT, "include", "vbs.h")).read()
    declared = set(re.findall(r"\b(vbs_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(L.SYMBOLS)
    lib = L.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.vbs_version() >= 100


def test_option_and_status_constants_match_the_header():
    """`_lib.py` restates the header's #defines (options, status codes, table layout): they must agree."""
    hdr = open(os.path.join(ROOT, "include", "vbs.h")).read()
    defs = {k: int(v) for k, v in re.findall(r"#define\s+(VBS_[A-Z0-9_]+)\s+(-?\d+)", hdr)}
    for name in ("GRAY_COEFFS", "FORCE_SEQ_MATCH", "GRAY_SIDE_STREAM", "NCC_MARGIN", "STAGE_IMPL", "BLUR_IMPL", "PASS_STREAMS"):
        assert getattr(L, "OPT_" + name) == defs["VBS_OPT_" + name], name
    opts = [v for k, v in defs.items() if k.startswith("VBS_OPT_")]
    assert len(opts) == len(set(opts)) == 8
    for k, v in defs.items():
        if hasattr(L, k[4:]) and isinstance(getattr(L, k[4:]), int) and not k.startswith("VBS_OPT_"):
            assert getattr(L, k[4:]) == v, k


@pytest.mark.parametrize("ksize,sigma", [(39, 8.0), (101, 20.0), (21, 4.56), (35, 11.4)])
def test_gaussian_taps_match_oracle(ksize, sigma):
    out = np.zeros(ksize, dtype=np.int32)
    assert L.lib().vbs_gaussian_taps_q8(ksize, sigma, out.ctypes.data_as(C.c_void_p)) == 0
    np.testing.assert_array_equal(out, O.gaussian_kernel_q8(ksize, sigma))
    assert out.sum() == 256 and (out >= 0).all()


@pytest.mark.parametrize("l,sigma", [(80, 13.0), (33, 7.4)])
def test_ncc_template_matches_oracle(l, sigma):
    g = np.zeros(l)
    st = np.zeros(4)
    assert L.lib().vbs_ncc_template(l, sigma, g.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p)) == 0
    t = O.gkern(l, sigma)
    np.testing.assert_allclose(np.outer(g, g), t, rtol=1e-13, atol=0)
    tbar = np.mean(t)
    assert abs(st[0] - tbar) <= 1e-15 * tbar
    T2 = np.sum(np.square(t - tbar))
    assert abs(st[1] - T2) <= 1e-13 * T2
    assert st[2] == l * l and st[3] == 0.1 * 0.1


def _lut():
    lut = np.zeros(256, dtype=np.uint8)
    assert L.lib().vbs_contour_lut(lut.ctypes.data_as(C.c_void_p)) == 0
    return lut


def lut_vertices(fg, lut):
    """Apply the library's table to every border pixel: {(x, y): multiplicity}."""
    H, W = fg.shape
    p = np.zeros((H + 2, W + 2), dtype=np.uint8)
    p[1:-1, 1:-1] = fg
    pat = np.zeros((H, W), dtype=np.int32)
    for d in range(8):
        pat |= p[1 + O._DY[d]:1 + O._DY[d] + H, 1 + O._DX[d]:1 + O._DX[d] + W].astype(np.int32) << d
    four = (pat & 1 > 0) & (pat & 4 > 0) & (pat & 16 > 0) & (pat & 64 > 0)
    border = (fg > 0) & ~four
    out = {}
    for y, x in zip(*np.nonzero(border)):
        m = int(lut[pat[y, x]])
        if m:
            out[(int(x), int(y))] = m
    return out


@pytest.mark.parametrize("seed,opened", [(0, True), (1, True), (2, False), (3, False), (4, False)])
def test_contour_lut_equals_traced_vertices(seed, opened):
    """The claim behind the HIP contour stage: for hole-free foreground, the per-pixel table yields
    exactly the multiset of CHAIN_APPROX_SIMPLE vertices of the traced outer borders."""
    rng = np.random.default_rng(seed)
    a = ndimage.gaussian_filter(rng.random((150, 170)), 3.0 if opened else 1.2)
    fg = a > np.quantile(a, 0.6)
    if opened:
        fg = O.morph_open5(fg)
    else:
        fg[40, 10:120] = True                      # 1-px lines: pixels visited twice
        fg[10:100, 60] = True
    fg = ndimage.binary_fill_holes(fg, structure=np.ones((3, 3)))     # no holes w.r.t. 4-conn background
    fg = ndimage.binary_fill_holes(fg)
    traced = {}
    for cnt in O.find_contours_external(fg):
        for x, y in cnt:
            traced[(int(x), int(y))] = traced.get((int(x), int(y)), 0) + 1
    assert lut_vertices(fg.astype(np.uint8), _lut()) == traced


def test_contour_lut_straight_edges_have_no_vertex():
    """k_label drops the inside of straight horizontal edges by bit operations before the per-pixel table walk:
    that is exact only because both patterns (E, W and the three pixels below / above set) hold no vertex."""
    lut = _lut()
    top = 1 | 16 | 32 | 64 | 128          # E, W, SW, S, SE
    bottom = 1 | 2 | 4 | 8 | 16           # E, NE, N, NW, W
    assert lut[top] == 0 and lut[bottom] == 0


def test_contour_lut_simple_shapes():
    lut = _lut()
    sq = np.zeros((12, 12), np.uint8)
    sq[3:8, 2:9] = 1
    assert lut_vertices(sq, lut) == {(2, 3): 1, (8, 3): 1, (8, 7): 1, (2, 7): 1}
    one = np.zeros((5, 5), np.uint8)
    one[2, 2] = 1
    assert lut_vertices(one, lut) == {(2, 2): 1}
    assert [c.tolist() for c in O.find_contours_external(one)] == [[[2, 2]]]
    line = np.zeros((5, 9), np.uint8)
    line[2, 1:8] = 1
    assert lut_vertices(line, lut) == {(1, 2): 1, (7, 2): 1}


def test_moment_terms_of_the_24_bit_branch_fit_its_operand_range():
    """`k_ccl<1>` / `k_label` form the contour-vertex moment terms with 24-bit multiplies (`v_mul_i32_i24`: operands are
    sign-extended from 24 bits) while |dx|, |dy| <= 150.  With the largest multiplicity the vertex table holds, every
    OPERAND the kernels form must stay below 2^23 and every product below 2^31 (the order of the factors in
    `csrc/k_ccl.hip` is chosen for that: (mult dx) dy first, then x^2)."""
    lut = (C.c_uint8 * 256)()
    assert L.lib().vbs_contour_lut(lut) == 0
    mult, d = max(lut), 150
    assert mult == 4
    operands = [mult * d,            # mx, my
                d * d,               # x2, y2
                mult * d * d,        # mx dx, my dy, mx dy, mult x2
                ]
    products = [mult * d ** 3,       # mx x2 (a result only: never an operand of a further 24-bit multiply)
                mult * d ** 4]       # (mx dy) x2, (mult x2) x2, ...
    assert max(operands) < 2 ** 23 and max(products) < 2 ** 31
    assert mult * d ** 3 >= 2 ** 23  # which is why mx x2 must not be multiplied on: the association matters
