"""Phase timing of k_ncc_mfma through the debug library's VBS_NCC_DBG knob (3: loads + horizontal products + ring
only; 2: + vertical products; 0: everything; PHASES=3,2,1,0 selects), or of k_blur_mfma with KNOB=VBS_BLUR_DBG (2: no
vertical products, 1: no range test / store).  usage: gpu_ncc_phase.py [frames]"""
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
    eng.find_markers(ft); torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(3):
        eng.track_to_3d(ft)
    p = eng.profile_read()
    print(json.dumps({k: round(1e3 * v[1] / v[0] / n, 4) for k, v in p.items() if "ncc" in k or "blur" in k}))
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "512"
    for dbg in [int(x) for x in os.environ.get("PHASES", "3,2,1,0").split(",")]:
        env = dict(os.environ, **{os.environ.get("KNOB", "VBS_NCC_DBG"): str(dbg)})
        r = subprocess.run([sys.executable, __file__, n, "child"], env=env, capture_output=True, text=True, timeout=300)
        print("dbg", dbg, r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else "", flush=True)
