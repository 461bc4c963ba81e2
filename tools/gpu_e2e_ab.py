"""End-to-end rate of the frames -> tracking table path (the bench's timed region: 4096 resident 1280x1024 frames, reference IDs
from frame 0, camera) under several builds of the library, alternating child processes.
usage: gpu_e2e_ab.py <sfx,sfx,...> [frames] [rounds]"""
import os, sys, json, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import vbs_amd.synth as S
    from vbs_amd import _lib as L
    L.LIB_PATH = L.LIB_PATH.replace("libvbs.so", f"libvbs{sys.argv[2]}.so")
    from vbs_amd.engine import Engine
    from vbs_amd.pipeline import reference_from_frame0
    n = int(sys.argv[3]); spec = S.config2()
    eng = Engine(spec.height, spec.width, max_markers=512, max_batch=512)
    ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda", chunk=256)
    cam = L.make_camera(*S.default_camera(spec), 2.0)
    ids, xy = reference_from_frame0(eng, ft[:1], 5, "full", "optimal")
    xy = torch.as_tensor(xy, dtype=torch.float64, device="cuda")
    for _ in range(2):
        out = eng.track_to_3d(ft, xy, 20.0, cam, 5.0)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        out = eng.track_to_3d(ft, xy, 20.0, cam, 5.0)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    tracked = int((out[0][..., 0].int() & 1).sum())
    print(json.dumps({"fps": round(n / best), "us_per_frame": round(best / n * 1e6, 4), "tracked": tracked}))
else:
    sfxs = [x.strip("'\"") for x in sys.argv[1].split(",")]
    n = sys.argv[2] if len(sys.argv) > 2 else "4096"
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    for r in range(rounds):
        for sfx in sfxs:
            out = subprocess.run([sys.executable, __file__, "child", sfx, n], capture_output=True, text=True, timeout=300)
            print(f"lib{sfx or '(product)'}", out.stdout.strip(), out.stderr.strip()[-300:] if out.returncode else "", flush=True)
