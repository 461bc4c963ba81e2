"""Where k_stage_lat (one frame over several workgroups) spends its time: wall-clock stamps the debug library writes into the
frame's header at the phase boundaries.  usage: gpu_lat_trace.py [frames per call = 1]   (needs libvbs_dbg.so: __graft_entry__.build())"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd import _lib as L
L.LIB_PATH = L.LIB_PATH.replace("libvbs.so", "libvbs_dbg.so")
from vbs_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
spec = S.config2()
eng = Engine(spec.height, spec.width, max_markers=512, max_batch=n)
eng.set_option(L.OPT_LATENCY_FRAMES, 8)
ft = S.make_frames_torch(spec, range(8), seed=0, device="cuda")
names = ["first start", "last start", "band walk done (last)", "band resolve begins", "  components known", "  sums out, flag set",
         "others go on (last)", "opened walk done (last)", "opened resolve begins", "  components known", "  moments shifted", "end (probes answered)",
         "band walk done (first)", "opened walk done (first)", "  parents set, counts read", "  pairs united", "  flattened",
         "  parents set, counts read", "  pairs united", "  flattened", "pairs (band walk)", "pairs (opened walk)", "band records", "moment records"]
order = [0, 1, 12, 2, 3, 14, 15, 16, 4, 5, 13, 7, 8, 17, 18, 19, 9, 10, 11]
NS = 20
acc = np.zeros(NS)
cnts = np.zeros(4)
reps = 20
for i in range(reps + 3):
    eng.track_to_3d(ft[i % 8:i % 8 + n] if n == 1 else ft[:n], want_det=True)
    hdr = np.zeros(n * 128, np.uint32)
    rc = eng.lib.vbs_debug_lat_hdr(eng._h, hdr.ctypes.data_as(C.c_void_p), n)
    assert rc == 0
    st = hdr[96:96 + NS].astype(np.int64)
    st[[0, 12, 13]] = (~hdr[[96, 108, 109]]).astype(np.int64)
    if i >= 3:
        acc += (st - st[0]) / 100.0                       # 100 MHz -> us
        cnts += [hdr[32:48].sum(), hdr[64:80].sum(), hdr[16:32].sum(), hdr[48:64].sum()]
for k in order:
    print(f"{names[k]:28s} {acc[k] / reps:8.1f} us")
for k in range(4):
    print(f"{names[20 + k]:28s} {cnts[k] / reps:8.0f}")
