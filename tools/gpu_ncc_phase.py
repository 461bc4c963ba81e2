import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd.engine import Engine
spec = S.config2(); n = 256
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=n)
ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
for stop in ("1", "0"):
    os.environ["VBS_NCC_STOP"] = stop
    eng.find_markers(ft); eng.profile(True)
    for _ in range(3): eng.find_markers(ft)
    p = eng.profile_read(); eng.profile(False)
    print("ncc_stop", stop, {k: round(v[1] / v[0] * 1e3 / n, 3) for k, v in p.items()})
