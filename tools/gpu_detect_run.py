"""Run the fused frames -> detections path (vbs_track_to_3d without a reference table) a few times — the program
handed to rocprofv3 PMC passes that look at k_blur_mfma / k_ncc (filter with --kernel-include-regex)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
spec = S.config2()
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=n)
ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
torch.cuda.synchronize()
for _ in range(reps):
    table, det, counts = eng.track_to_3d(ft)
torch.cuda.synchronize()
print("frames", n, "counts", int(counts.min()), int(counts.max()))
