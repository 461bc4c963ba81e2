"""k_ncc_mfma's time against the strip length: 1280-wide frames of several heights with the same dot pattern, the same number
of pixels per launch.  us per frame and ps per pixel: the fixed cost per strip (workgroup start, prologue, the NT - 1 ramp
steps) shows as the slope against 1 / height.  usage: gpu_ncc_height.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vbs_amd.synth as S
from vbs_amd.engine import Engine

for H, n in ((512, 1024), (1024, 512), (2048, 256), (2048, 512)):
    spec = S.grid_spec(1280, H, 13, 72, 40, name=f"1280x{H}") if H >= 1024 else S.grid_spec(1280, H, 7, 72, 40, name=f"1280x{H}")
    eng = Engine(spec.height, spec.width, max_markers=512, max_batch=n)
    ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
    for _ in range(2):
        eng.track_to_3d(ft)
    torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(4):
        eng.track_to_3d(ft)
    p = eng.profile_read()
    us = {k: round(1e3 * v[1] / v[0] / n, 4) for k, v in p.items() if k in ("k_ncc_mfma", "k_blur16", "k_stage")}
    print(json.dumps({"H": H, "frames": n, "us_per_frame": us, "ps_per_pixel": {k: round(1e6 * v / (1280 * H), 4) for k, v in us.items()},
                      "ncc_steps_per_strip": H // 16 + 5}), flush=True)
    eng.close()
    del ft
