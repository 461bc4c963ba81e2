"""gpu_lib_ab.py on the reference's real configuration: 640x480 BGR frames through the default crop (480x450 strided views,
small branch).  usage: gpu_lib_ab_c1.py <sfx,sfx,...> [frames] [rounds]"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import vbs_amd.synth as S
    from vbs_amd import _lib as L
    L.LIB_PATH = L.LIB_PATH.replace("libvbs.so", f"libvbs{sys.argv[2]}.so")
    from vbs_amd.engine import Engine
    from vbs_amd.marker_detection import _crop_box
    n = int(sys.argv[3]); spec = S.config1()
    l, r, t, b = _crop_box(spec.width, spec.height, (1 / 8, 1 / 8, 1 / 16, 0))
    full = S.make_frames_torch(spec, range(n), seed=0, channels=3, device="cuda")
    ft = full[:, t:b, l:r, :]
    eng = Engine(b - t, r - l, max_markers=256, max_batch=n)
    for _ in range(2):
        eng.track_to_3d(ft)
    torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(4):
        eng.track_to_3d(ft)
    torch.cuda.synchronize()
    p = eng.profile_read()
    print(json.dumps({k: round(1e3 * v[1] / v[0] / n, 4) for k, v in p.items() if v[1] / v[0] > 0.01}))
else:
    sfxs = [x.strip("'\"") for x in sys.argv[1].split(",")]
    n = sys.argv[2] if len(sys.argv) > 2 else "512"
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    for r in range(rounds):
        for sfx in sfxs:
            out = subprocess.run([sys.executable, __file__, "child", sfx, n], capture_output=True, text=True, timeout=300)
            print(f"lib{sfx or '(product)'}", out.stdout.strip(), out.stderr.strip()[-200:] if out.returncode else "", flush=True)
