"""Per-kernel times (us per frame, HIP events) of the frames -> table path under two builds of the library, alternating:
gpu_lib_ab.py <suffixA> <suffixB> [frames] [rounds]  |  gpu_lib_ab.py <sfx,sfx,...> [frames] [rounds]   (suffix '' = libvbs.so, '_x' = libvbs_x.so built with
vbs_amd._build.build(extra_flags=[...], suffix='_x'))."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import vbs_amd.synth as S
    from vbs_amd import _lib as L
    L.LIB_PATH = L.LIB_PATH.replace("libvbs.so", f"libvbs{sys.argv[2]}.so")
    from vbs_amd.engine import Engine
    n = int(sys.argv[3]); spec = S.config2()
    eng = Engine(spec.height, spec.width, max_markers=512, max_batch=n)
    ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
    for _ in range(2):
        eng.track_to_3d(ft)
    torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(4):
        eng.track_to_3d(ft)
    torch.cuda.synchronize()
    p = eng.profile_read()
    print(json.dumps({k: round(1e3 * v[1] / v[0] / n, 4) for k, v in p.items() if v[1] / v[0] > 0.02}))
else:
    # suffixes first (a comma-separated list in ONE argument also works: "'',_x,_y"), then frames and rounds
    if "," in sys.argv[1]:
        sfxs = [x.strip("'\"") for x in sys.argv[1].split(",")]
        rest = sys.argv[2:]
    else:
        sfxs = [sys.argv[1], sys.argv[2]]
        rest = sys.argv[3:]
    n = rest[0] if len(rest) > 0 else "512"
    rounds = int(rest[1]) if len(rest) > 1 else 3
    for r in range(rounds):
        for sfx in sfxs:
            out = subprocess.run([sys.executable, __file__, "child", sfx, n], capture_output=True, text=True, timeout=300)
            print(f"lib{sfx or '(product)'}", out.stdout.strip(), out.stderr.strip()[-200:] if out.returncode else "", flush=True)
