"""BGR frames -> table under two builds of the library (suffixes as in gpu_lib_ab.py), alternating; prints frames/s.
usage: gpu_bgr_lib_ab.py <suffixA> <suffixB> [frames]"""
import os, sys, time, subprocess
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
    gray = S.make_frames_torch(spec, range(n), seed=0, device="cuda", chunk=16)
    bgr = gray.unsqueeze(-1).expand(-1, -1, -1, 3).contiguous()
    ids, xy = reference_from_frame0(eng, gray[:1], 5, "full", "optimal")
    for fr, name in ((gray, "gray"), (bgr, "bgr")):
        eng.track_to_3d(fr, xy); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.track_to_3d(fr, xy)
        torch.cuda.synchronize()
        print(name, round(n * 3 / (time.perf_counter() - t0)), end="  ")
    print()
else:
    a, b = sys.argv[1], sys.argv[2]
    n = sys.argv[3] if len(sys.argv) > 3 else "2048"
    for r in range(2):
        for sfx in (a, b):
            out = subprocess.run([sys.executable, __file__, "child", sfx, n], capture_output=True, text=True, timeout=300)
            print(f"lib{sfx or '(product)'}", out.stdout.strip(), out.stderr.strip()[-200:] if out.returncode else "", flush=True)
