"""Mask of find_markers against the oracle on one synthetic frame: mismatch count, bounding box, exact-path counters.
usage: gpu_mask_diff.py [c1|c2|c5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vbs_amd.synth as S
from vbs_amd import _lib as L
if os.environ.get("VBS_NCC_DBG"):
    L.LIB_PATH = L.LIB_PATH.replace("libvbs.so", "libvbs_dbg.so")
from vbs_amd.engine import Engine
from oracle import stages as O
name = sys.argv[1] if len(sys.argv) > 1 else "c1"
spec = {"c1": S.config1, "c2": S.config2, "c5": S.config5}[name]()
fr = S.make_frames(spec, [3], seed=1)
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=4)
mask, area = eng.find_markers(torch.from_numpy(fr).cuda())
mask = mask.cpu().numpy()[0]; area = area.cpu().numpy()[0]
om, oa = O.find_markers(fr[0], ncc="direct")
if os.environ.get("VBS_NCC_DBG") == "9":
    fgm = (mask & 1).astype(bool); und = (mask & 2).astype(bool)
    print("undecided", int(und.sum()), "filter fg", int(fgm.sum()), "oracle fg", int(om.sum()))
    bad = (~und) & (fgm != om.astype(bool))
    print("decided wrongly", int(bad.sum()))
    ys, xs = np.nonzero(und)
    if und.any():
        print("und bbox y", ys.min(), ys.max(), "x", xs.min(), xs.max())
        print("und x mod 16", np.bincount(xs % 16, minlength=16).tolist())
        print("und y mod 16", np.bincount(ys % 16, minlength=16).tolist())
        H, W = und.shape
        edge = (ys < 16) | (ys >= H - 16 - 16) | (xs < 16) | (xs >= W - 32)
        print("und in border tiles (approx)", int(edge.sum()))
    sys.exit(0)
print("area equal", np.array_equal(area, oa))
if os.environ.get("VBS_NCC_DBG") == "8":
    from scipy import ndimage
    p = O.branch_params(fr[0].shape[0]); l = p["tl"]
    b = (oa > 0).astype(np.int64)
    c = ndimage.correlate1d(ndimage.correlate1d(b, np.ones(l, np.int64), axis=1, mode="constant"), np.ones(l, np.int64), axis=0, mode="constant")
    c8 = (c & 255).astype(np.uint8)
    dd = mask != c8
    print("count mismatches", int(dd.sum()), "of", dd.size)
    ys, xs = np.nonzero(dd)
    if dd.any():
        print("first", [(int(y), int(x), int(mask[y, x]), int(c8[y, x])) for y, x in zip(ys[:12], xs[:12])])
        print("x mod 16 hist", np.bincount(xs % 16, minlength=16).tolist())
        print("y mod 16 hist", np.bincount(ys % 16, minlength=16).tolist())
    sys.exit(0)
d = mask != om
ys, xs = np.nonzero(d)
print("mask mismatches", int(d.sum()), "of fg", int(om.sum()), "gpu fg", int(mask.sum()))
if d.any():
    print("bbox y", ys.min(), ys.max(), "x", xs.min(), xs.max())
    print("first", list(zip(ys[:10].tolist(), xs[:10].tolist())))
    print("x mod 16 hist", np.bincount(xs % 16, minlength=16).tolist())
    print("y mod 16 hist", np.bincount(ys % 16, minlength=16).tolist())
print("counters", eng.ncc_counters())
