import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbs_amd.engine import Engine
from scipy import ndimage
def band_of(mask, ns):
    from scipy.ndimage import maximum_filter, minimum_filter
    dm = maximum_filter(mask, ns); mx = (mask == dm); mx[(dm - minimum_filter(mask, ns)) == 0] = 0
    return mx
for shape in ((450, 480), (960, 960), (480, 640), (1024, 1280)):
    H, W = shape
    eng = Engine(H, W, max_markers=1024, max_batch=1)
    ns = 8 if H <= 480 else 14
    pats = {}
    m = np.zeros(shape, np.uint8); m[:, 100] = 1; pats["vline100"] = m
    m = np.zeros(shape, np.uint8); m[:, 63] = 1; m[:, 64] = 1; pats["vline63_64"] = m
    m = np.zeros(shape, np.uint8); m[10:H-10, 200] = 1; m[H//2, 30:W-30] = 1; pats["cross"] = m
    m = np.zeros(shape, np.uint8)
    for y in range(0, H - 1): m[y, (y * 3) % (W - 2)] = 1; m[y, (y * 3) % (W - 2) + 1] = 1; m[y, ((y*3) % (W-2) + 2) % W] = 1
    pats["diag"] = m
    for name, m in pats.items():
        mt = torch.from_numpy(m).cuda()
        eng.marker_center(mt, mt)
        st = eng.frame_stats(1)[0]
        _, nb = ndimage.label(band_of(m, ns))
        print(shape, name, "gpu band comps", int(st[5]), "oracle", nb, flush=True)
    eng.close()
