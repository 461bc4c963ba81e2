"""Stress of k_blur16's loader / strip hand-shake: many textured frames per launch (random smooth fields made on the GPU),
both blur kernels, area mask and NCC mask compared bit for bit, several rounds and sizes.
usage: gpu_blur_stress.py [frames] [rounds]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbs_amd import _lib as L
from vbs_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bad = 0
for (h, w) in ((1024, 1280), (1200, 1920), (600, 808), (450, 480), (480, 644), (1000, 1284)):      # large, small branch, widths 4 (mod 8)
    eng = Engine(h, w, max_markers=512, max_batch=n)
    for r in range(rounds):
        g = torch.Generator(device="cuda").manual_seed(1000 * r + h)
        cell = 24 if h > 480 else 9                     # (the small branch's blurs see nothing in the broad fields)
        lo = torch.randn((n, 1, h // cell + 2, w // cell + 2), device="cuda", generator=g) * 55 + 100
        fr = torch.nn.functional.interpolate(lo, size=(h, w), mode="bicubic", align_corners=False)[:, 0]
        fr = (fr + torch.randn((n, h, w), device="cuda", generator=g) * 10).clamp(0, 255).to(torch.uint8).contiguous()
        out = {}
        for impl in (0, 1, 0):
            eng.set_option(L.OPT_BLUR_IMPL, impl)
            m, a = eng.find_markers(fr)
            torch.cuda.synchronize()
            out.setdefault(impl, []).append((m.clone(), a.clone()))
        same = all(torch.equal(out[0][k][1], out[1][0][1]) and torch.equal(out[0][k][0], out[1][0][0]) for k in range(2))
        frac = float((out[1][0][1] > 0).float().mean())
        print(f"{h}x{w} round {r}: {'OK' if same else 'DIFFERENT'}  area fraction {frac:.3f}", flush=True)
        bad += not same
    eng.close()
print("stress", "all OK" if bad == 0 else f"{bad} rounds differ")
