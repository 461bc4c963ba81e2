"""Phase timing of k_label / k_finalize via the debug early-exit knobs (diagnostic, GPU box only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd.engine import Engine

spec = S.config2()
for batch in (512,):
    n = batch
    eng = Engine(spec.height, spec.width, max_markers=512, max_batch=batch)
    ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
    mask, area = eng.find_markers(ft)
    torch.cuda.synchronize()
    for stop in (2, 34, 4, 0):
        os.environ["VBS_LABEL_STOP"] = str(stop)
        os.environ["VBS_FINAL_STOP"] = "1" if stop else "0"
        eng.marker_center(mask, area)
        eng.profile(True)
        for _ in range(3):
            eng.marker_center(mask, area)
        p = eng.profile_read()
        eng.profile(False)
        print(f"batch {batch} label_stop {stop}: " + "  ".join(f"{k} {v[1]/v[0]*1e3/ n:.3f}us/frame" for k, v in p.items()), flush=True)
    os.environ["VBS_LABEL_STOP"] = "0"; os.environ["VBS_FINAL_STOP"] = "0"
    eng.close()
