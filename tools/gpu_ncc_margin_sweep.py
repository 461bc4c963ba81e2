"""What the queued float64 path of k_ncc_mfma costs: kernel time and re-evaluated pixels per frame against the filter's relative
margin (VBS_OPT_NCC_MARGIN, units of 1e-6; the product's is 20).  Prices a cheaper filter (fewer matrix products, wider margin).
usage: gpu_ncc_margin_sweep.py [frames]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import vbs_amd.synth as S
from vbs_amd import _lib as L
from vbs_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spec = S.config2()
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=n)
ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
for ppm in (20, 100, 300, 700, 1000, 2000, 4000, 20):
    eng.set_option(L.OPT_NCC_MARGIN, ppm)
    for _ in range(2):
        eng.track_to_3d(ft)
    torch.cuda.synchronize()
    eng.ncc_counters(reset=True) if hasattr(eng, "ncc_counters") else None
    eng.profile(True)
    for _ in range(4):
        eng.track_to_3d(ft)
    torch.cuda.synchronize()
    p = eng.profile_read()
    st = eng.frame_stats(n)
    t = {k: round(1e3 * v[1] / v[0] / n, 4) for k, v in p.items() if "ncc" in k or "blur16" in k}
    print(f"margin {ppm:5d} ppm: {t}  undecided per frame {st[:, 1].mean():.1f}  float64 pixels per frame {st[:, 3].mean():.1f}", flush=True)
