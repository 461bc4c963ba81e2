"""k_finalize: centroids + ellipse fits only (VBS_FINAL_STOP=1, debug library) against the whole kernel.  usage: gpu_final_phase.py [frames]"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import torch
    import vbs_amd.synth as S
    from vbs_amd import _lib as L
    L.LIB_PATH = L.LIB_PATH.replace("libvbs.so", "libvbs_dbg.so")
    from vbs_amd.engine import Engine
    n = int(sys.argv[1]); spec = S.config2()
    eng = Engine(spec.height, spec.width, max_markers=512, max_batch=n)
    ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
    eng.track_to_3d(ft); torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(3):
        eng.track_to_3d(ft)
    p = eng.profile_read()
    print(json.dumps({k: round(1e3 * v[1] / v[0] / n, 4) for k, v in p.items() if "final" in k or "track" in k}))
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "512"
    for stop in (1, 0):
        env = dict(os.environ, VBS_FINAL_STOP=str(stop))
        r = subprocess.run([sys.executable, __file__, n, "child"], env=env, capture_output=True, text=True, timeout=300)
        print("stop", stop, r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else "", flush=True)
