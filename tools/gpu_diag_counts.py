import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbs_amd.engine import Engine
from oracle import stages as O
from scipy import ndimage
G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "stages.npz"))
def rle_decode(runs, shape):
    vals = np.zeros(len(runs), dtype=np.uint8); vals[1::2] = 1
    return np.repeat(vals, runs).reshape(shape)
for tag in ("c1", "c2"):
    shape = tuple(int(v) for v in G[f"band_{tag}_shape"])
    mask = rle_decode(G[f"band_{tag}_mask_rle"], shape)
    eng = Engine(shape[0], shape[1], max_markers=1024, max_batch=1)
    mt = torch.from_numpy(mask).cuda()
    det, counts = eng.marker_center(mt, mt)
    st = eng.frame_stats(1)[0]
    band = O.band_mask(mask)
    _, nb = ndimage.label(band)
    opened = O.morph_open5(mask != 0)
    _, na = ndimage.label(opened, structure=np.ones((3, 3)))
    runs_b = int((np.diff(np.pad(band.astype(np.int8), ((0, 0), (1, 0))), axis=1) == 1).sum())
    print(tag, shape, "gpu count", int(counts[0]), "stats", st.tolist(), "oracle band comps", nb, "open comps", na, "band runs", runs_b)
    eng.close()
