"""Two handles on two streams against one: the frames -> table path over N frames, the halves of the batch processed
concurrently (each handle its own workspace and stream), so that the tail of one kernel's last round overlaps the other
stream's work.  usage: gpu_two_streams.py [frames] [batch]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 512
spec = S.config2()
ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
ref = torch.from_numpy(S.dot_truth(spec, 0, [0])[0][:, :2].copy()).cuda() if hasattr(S, "dot_truth") else None


def run(engs, streams, reps):
    parts = [ft[i::1][0:0] for i in range(0)] or list(torch.chunk(ft[:(ft.shape[0] // (len(engs) * batch)) * len(engs) * batch], len(engs)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        outs = []
        for e, st, p in zip(engs, streams, parts):
            with torch.cuda.stream(st):
                outs.append(e.track_to_3d(p, ref_xy=ref))
        for st in streams:
            st.synchronize()
    return (time.perf_counter() - t0) / reps, outs


from vbs_amd import _lib as L
for k in (1, 2, 3, 4, 6, 2, 3):
    engs = [Engine(spec.height, spec.width, max_markers=512, max_batch=batch) for _ in range(k)]
    for e in engs:
        e.set_option(L.OPT_PASS_STREAMS, 1)             # (one stream per handle here: k handles = k streams)
    streams = [torch.cuda.Stream() for _ in range(k)]
    run(engs, streams, 1)
    dt, outs = run(engs, streams, 4)
    cnt = torch.cat([o[2] for o in outs])
    nn = int(cnt.numel())
    print(f"{k} stream(s): {nn / dt:10.0f} frames/s  {dt * 1e6 / nn:.3f} us/frame  ({nn} frames) counts {int(cnt.min())}..{int(cnt.max())}", flush=True)
    for e in engs:
        e.close()

# the library's own form: VBS_OPT_PASS_STREAMS = 1 | 2 on one handle (odd internal passes on a second workspace and stream)
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=batch)
for ps in (1, 2, 1, 2):
    eng.set_option(L.OPT_PASS_STREAMS, ps)
    eng.track_to_3d(ft, ref_xy=ref); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        out = eng.track_to_3d(ft, ref_xy=ref)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    print(f"pass_streams {ps}: {n / dt:10.0f} frames/s  {dt * 1e6 / n:.3f} us/frame", flush=True)
