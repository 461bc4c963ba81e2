"""Run the fused frames -> detections path (vbs_track_to_3d without a reference table) a few times — the program
handed to rocprofv3 PMC passes (filter with --kernel-include-regex).  usage: gpu_detect_run.py [frames] [reps] [c3|c5]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd import _lib as L
if os.environ.get("VBS_LIB") == "dbg":                  # the library with the phase-timing knobs (VBS_STAGE_STOP, ..)
    L.LIB_PATH = L.LIB_PATH.replace("libvbs.so", "libvbs_dbg.so")
from vbs_amd.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = sys.argv[3] if len(sys.argv) > 3 else "c3"
spec = S.config2() if wl == "c3" else S.config5()
eng = Engine(spec.height, spec.width, max_markers=512 if wl == "c3" else 1024, max_batch=n)
ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
if os.environ.get("BLUR_IMPL"):
    eng.set_option(L.OPT_BLUR_IMPL, int(os.environ["BLUR_IMPL"]))
torch.cuda.synchronize()
for _ in range(reps):
    table, det, counts = eng.track_to_3d(ft)
torch.cuda.synchronize()
print("workload", wl, "frames", n, "counts", int(counts.min()), int(counts.max()))
