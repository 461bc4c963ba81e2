"""BASELINE config 2: ONE 1280x1024 frame through track -> 3-D on one GPU: latency of a call (frame resident in HBM,
launches + kernels, stream synchronised), per kernel and in total.  usage: gpu_single_frame.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd import _lib as L
if os.environ.get("VBS_LIB_SUFFIX"):                      # e.g. _dbg: the tools build with its environment knobs
    L.LIB_PATH = L.LIB_PATH.replace("libvbs.so", f"libvbs{os.environ['VBS_LIB_SUFFIX']}.so")
from vbs_amd.engine import Engine
from vbs_amd.pipeline import reference_from_frame0

spec = S.config2()
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=1)
ft = S.make_frames_torch(spec, range(8), seed=0, device="cuda")
cam = L.make_camera(*S.default_camera(spec), 2.0)
ids, xy = reference_from_frame0(eng, ft[:1], 5, "full", "optimal")
for impl in (int(x) for x in os.environ.get('STAGE_IMPLS', '0').split(',')):
  eng.set_option(L.OPT_STAGE_IMPL, impl)
  for i in range(5):
    eng.track_to_3d(ft[i % 8:i % 8 + 1], xy, 20.0, cam, 5.0)
  torch.cuda.synchronize()
  reps = 200
  t0 = time.perf_counter()
  for i in range(reps):
      table, _, counts = eng.track_to_3d(ft[i % 8:i % 8 + 1], xy, 20.0, cam, 5.0)
      torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / reps
  eng.profile(True)
  for i in range(50):
      eng.track_to_3d(ft[i % 8:i % 8 + 1], xy, 20.0, cam, 5.0)
  torch.cuda.synchronize()
  p = eng.profile_read()
  print(f"one frame, call + synchronise: {dt * 1e6:.1f} us ({1 / dt:.0f} frames/s one at a time); tracked {int((table[..., 0].int() & 1).sum())} of {len(ids)}")
  print("stage impl", impl, {k: round(1e3 * v[1] / v[0], 1) for k, v in p.items()}, "us per launch")
  eng.profile(False)

if os.environ.get("VBS_LIB_SUFFIX"):
    sys.exit(0)
# the same call captured into a HIP graph once and replayed per frame (the frame copied into a fixed buffer first): what a
# caller that feeds one frame at a time (marker_detection.py:434-453) can do about launch latency
xy_d = torch.as_tensor(xy, dtype=torch.float64, device="cuda")
buf = ft[:1].clone()
eng.set_option(L.OPT_STAGE_IMPL, 0)
eng.track_to_3d(buf, xy_d, 20.0, cam, 5.0); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        gtable, _, gcounts = eng.track_to_3d(buf, xy_d, 20.0, cam, 5.0)
torch.cuda.current_stream().wait_stream(side)
for i in range(5):
    buf.copy_(ft[i % 8:i % 8 + 1]); g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(reps):
    buf.copy_(ft[i % 8:i % 8 + 1]); g.replay(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
want, _, _ = eng.track_to_3d(ft[(reps - 1) % 8:(reps - 1) % 8 + 1], xy_d, 20.0, cam, 5.0)
torch.cuda.synchronize()
print(f"one frame, copy into the graph's buffer + replay + synchronise: {dt * 1e6:.1f} us ({1 / dt:.0f} frames/s); equal to the eager call: {bool(torch.equal(want, gtable))}")
