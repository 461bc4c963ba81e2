"""k_finalize phase timing through VBS_FINAL_STOP (1 = return after the ellipse fits)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd.engine import Engine
n = 512
spec = S.config2()
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=n)
ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
mask, area = eng.find_markers(ft)
torch.cuda.synchronize()
for stop in ("1", "0"):
    os.environ["VBS_FINAL_STOP"] = stop
    eng.marker_center(mask, area)
    eng.profile(True)
    for _ in range(3):
        eng.marker_center(mask, area)
    p = eng.profile_read()
    eng.profile(False)
    print("final_stop", stop, {k: round(v[1] / v[0] / n * 1e3, 3) for k, v in p.items()}, flush=True)
