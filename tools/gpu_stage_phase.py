"""Phase timing of the fused labelling kernel (k_stage): runs the fused path with the debug library (libvbs_dbg.so, built
with -DVBS_DEBUG_KNOBS) and VBS_STAGE_STOP set, printing the live per-kernel times (us per frame).
stops (STOPS=.. selects): 2 band loop (morph, labels, sums) | 3 + tile links | 4 + components | 10 + sums out, probe requests |
12 + open loop (morph, Euler, labels, vertex moments, probes) | 13 + tile links | 14 + components | 0 + moment shift, probe ids.  usage: gpu_stage_phase.py [frames] [c3|c5]"""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 3 and sys.argv[3] == "child":
    import torch
    import vbs_amd.synth as S
    from vbs_amd import _lib as L
    L.LIB_PATH = L.LIB_PATH.replace("libvbs.so", "libvbs_dbg.so")
    from vbs_amd.engine import Engine
    n = int(sys.argv[1]); spec = S.config2() if sys.argv[2] == "c3" else S.config5()
    eng = Engine(spec.height, spec.width, max_markers=512 if sys.argv[2] == "c3" else 1024, max_batch=n)
    ft = S.make_frames_torch(spec, range(n), seed=0, device="cuda")
    eng.track_to_3d(ft); torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(3):
        _, _, counts = eng.track_to_3d(ft)
    p = eng.profile_read()
    print(json.dumps({k: round(1e3 * v[1] / v[0] / n, 4) for k, v in p.items() if "stage" in k or "label" in k or "morph" in k or "final" in k}))
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "512"
    w = sys.argv[2] if len(sys.argv) > 2 else "c3"
    for stop in [int(x) for x in os.environ.get('STOPS', '2,3,4,10,12,13,14,0').split(',')]:
        env = dict(os.environ, VBS_STAGE_STOP=str(stop))
        r = subprocess.run([sys.executable, __file__, n, w, "child"], env=env, capture_output=True, text=True, timeout=300)
        print("stop", stop, r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else "", flush=True)
