"""k_ncc_mfma / k_blur_mfma time per frame for several VBS_NCC_DBG settings (512 frames, HIP events, one process)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd.engine import Engine
n = 512
spec = S.config2()
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=n)
ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
for dbg in (sys.argv[1:] or ["0"]):
    os.environ["VBS_NCC_DBG"] = dbg
    eng.track_to_3d(ft)
    torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(3):
        eng.track_to_3d(ft)
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    print(dbg, {k: round(v[1] / v[0] / n * 1e3, 3) for k, v in prof.items() if "ncc" in k or "blur" in k}, flush=True)
