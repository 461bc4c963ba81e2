"""NCC filter statistics: pixels re-evaluated in float64 per frame (frame_stats column 3) and kernel times."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
spec = S.config2()
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=n)
ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
eng.find_markers(ft)
torch.cuda.synchronize()
st = eng.frame_stats(n)
print("exact px per frame: min %d max %d mean %.1f ; ambiguous %d" % (st[:, 3].min(), st[:, 3].max(), st[:, 3].mean(), st[:, 1].sum()))
print(st[:16])
