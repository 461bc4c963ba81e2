"""The reference's REAL layout through the labelling kernels: which path do the frames take, and at what rate?
  The README's still (tests/golden/raw_markers_bgr.npz, 467x437 BGR, 65 dots) expanded on the device to
N frames by seeded shifts of +-3 px and noise sigma 2 (synth.jittered_copies_torch), and the synthetic ring layout at the real
size (65 dots of 27 px at a pitch of ~38 px in the 480x450 crop).  Prints, per workload and labelling implementation
(VBS_OPT_STAGE_IMPL 0 fused / 1 separate / 2 general kernel on every frame): frames per second, detections, the histogram of
slow-path reasons, us per frame of every labelling kernel."""
import collections
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vbs_amd.synth as S                         # noqa: E402
from vbs_amd import _lib as L                     # noqa: E402
from vbs_amd.engine import Engine                 # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = 512
bgr = np.load(os.path.join(ROOT, "tests", "golden", "raw_markers_bgr.npz"))["bgr"]
real, _ = S.jittered_copies_torch(bgr, N, seed=0, device="cuda")
ring = S.make_frames_torch(S.ring65_spec(480, 450, px_per_mm=11.0, diameter=27), range(N), seed=0, channels=3, device="cuda")
for name, frames in (("real still x shifts + noise", real), ("ring65 d27 pitch38 480x450", ring)):
    n, h, w = frames.shape[:3]
    for impl in (0, 1, 2):
        eng = Engine(h, w, max_markers=256, max_batch=B)
        eng.set_option(L.OPT_STAGE_IMPL, impl)
        eng.set_option(L.OPT_PASS_STREAMS, 1)
        hist, counts = collections.Counter(), []
        for s in range(0, n, B):                       # pass by pass: vbs_stage_tables describes the last pass
            _, _, c = eng.track_to_3d(frames[s:s + B], None, want_det=True)
            hist.update(int(v) for v in eng.stage_tables(min(B, n - s))["slow"])
            counts.append(c.cpu().numpy())
        counts = np.concatenate(counts)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.track_to_3d(frames, None)
        torch.cuda.synchronize()
        fps = 3 * n / (time.perf_counter() - t0)
        eng.profile(True)
        for _ in range(3):
            eng.track_to_3d(frames, None)
        per = {k: round(1e3 * v[1] / (3 * n), 4) for k, v in eng.profile_read().items()}
        eng.profile(False)
        print(json.dumps({"workload": name, "frame": f"{w}x{h}", "frames": n, "stage_impl": impl, "fps": round(fps),
                          "detections": dict(collections.Counter(int(v) for v in counts)), "slow_reasons": dict(hist),
                          "us_per_frame": per}), flush=True)
        eng.close()
