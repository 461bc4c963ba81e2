"""Run the threshold+CCL stage (vbs_marker_center on uint8 masks) a few times — target of rocprofv3 runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spec = S.config2()
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=n)
ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
mask, area = eng.find_markers(ft)
torch.cuda.synchronize()
for _ in range(4):
    det, cnt = eng.marker_center(mask, area)
torch.cuda.synchronize()
print("frames", n, "counts", int(cnt.min()), int(cnt.max()))
