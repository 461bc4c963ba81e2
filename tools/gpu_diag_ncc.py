import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd.engine import Engine
from oracle import stages as O
spec = S.config2()
frames = S.make_frames(spec, [2], seed=9)
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=1)
ft = torch.from_numpy(frames).cuda()
for rep in range(3):
    mask, area = eng.find_markers(ft)
    ncc = eng.ncc_map(ft)[0].cpu().numpy()
    om, oa = O.find_markers(frames[0])
    a = area[0].cpu().numpy(); m = mask[0].cpu().numpy()
    print("rep", rep, "area diff", (a != oa).sum(), "mask diff", (m != om).sum(), "stats", eng.frame_stats(1)[0, :3], "oracle cnt", (oa > 0).sum())
    if (a != oa).any():
        yy, xx = np.nonzero(a != oa); print("  area mismatch y", yy.min(), yy.max(), "x", xx.min(), xx.max(), list(zip(yy[:6], xx[:6])))
    with np.errstate(all="ignore"):
        ref = O.normxcorr2(O.gkern(80, 13.0), oa)
    d = np.abs(ncc - ref)
    yy, xx = np.nonzero(d > 1e-6)
    print("  ncc bad px", len(yy), (yy.min(), yy.max(), xx.min(), xx.max()) if len(yy) else None, d.max())
