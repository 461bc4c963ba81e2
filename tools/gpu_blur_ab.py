"""A/B of the two blur kernels on the GPU: 16-column strips (k_blur16, VBS_OPT_BLUR_IMPL = 0) against the 32-column kernel
(k_blur_mfma, = 1) on the same frames: area mask (uint8 and popcount) and NCC mask must be identical.  Then the time of
each over a batch of 1280x1024 frames (HIP events around the launches).
usage: gpu_blur_ab.py [frames_for_timing]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbs_amd.synth as S
from vbs_amd import _lib as L
from vbs_amd.engine import Engine


def textured(h, w, n, seed):
    """frames whose difference of Gaussians crosses the inRange bounds everywhere: smooth random fields + noise"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    out = np.zeros((n, h, w), np.float32)
    for f in range(n):
        for _ in range(40):
            cx, cy, s = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(6, 60)
            out[f] += rng.uniform(-120, 160) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
        out[f] += 90 + rng.normal(0, 12, (h, w))
    return np.clip(out, 0, 255).astype(np.uint8)


def run(eng, frames, impl):
    eng.set_option(L.OPT_BLUR_IMPL, impl)
    mask, area = eng.find_markers(frames)
    torch.cuda.synchronize()
    return mask.cpu().numpy(), area.cpu().numpy(), eng.frame_stats(frames.shape[0])[:, 0].copy()


def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    bad = 0
    cases = [("c3", 1024, 1280), ("c5", 1200, 1920), ("tex", 1024, 1280), ("tex", 600, 800), ("tex", 1000, 1284), ("tex", 520, 132),
             ("tex", 1030, 1300), ("view", 1024, 1280), ("bgr", 1024, 1280)]
    for kind, h, w in cases:
        n = 3
        eng = Engine(h, w, max_markers=512, max_batch=4)
        if kind == "c3":
            fr = S.make_frames_torch(S.config2(), range(n), seed=1)
        elif kind == "c5":
            fr = S.make_frames_torch(S.config5(), range(n), seed=2)
        elif kind == "view":
            big = torch.from_numpy(textured(h + 8, w + 64, n, 5)).cuda()
            fr = big[:, 4:4 + h, 32:32 + w]
        elif kind == "bgr":
            g = textured(h, w, n, 6)
            fr = torch.from_numpy(np.stack([g, np.roll(g, 3, 2), np.roll(g, 5, 1)], -1).copy()).cuda()
        else:
            fr = torch.from_numpy(textured(h, w, n, h + w)).cuda()
        m0, a0, p0 = run(eng, fr, 0)
        m1, a1, p1 = run(eng, fr, 1)
        ok = np.array_equal(a0, a1) and np.array_equal(m0, m1) and np.array_equal(p0, p1)
        print(f"A/B {kind} {h}x{w}: {'OK' if ok else 'DIFFERENT'}  area px {p0.tolist()} / {p1.tolist()}", flush=True)
        if not ok:
            bad += 1
            d = np.argwhere(a0 != a1)
            print("   area differs at", len(d), "pixels; first (f,y,x):", d[:10].tolist(), " rows", np.unique(d[:, 1])[:20].tolist(),
                  " cols", np.unique(d[:, 2])[:40].tolist())
        eng.close()
    # timing
    eng = Engine(1024, 1280, max_markers=512, max_batch=nt)
    fr = S.make_frames_torch(S.config2(), range(nt), seed=1)
    for impl in (1, 0, 1, 0):
        eng.set_option(L.OPT_BLUR_IMPL, impl)
        eng.track_to_3d(fr); torch.cuda.synchronize()
        eng.profile(True)
        for _ in range(3):
            eng.track_to_3d(fr)
        torch.cuda.synchronize()
        pr = eng.profile_read(); eng.profile(False)
        print("impl", impl, {k: round(ms / cnt / nt * 1000, 4) for k, (cnt, ms) in pr.items() if "blur" in k or "ncc" in k}, "us/frame", flush=True)
    print("A/B", "all OK" if bad == 0 else f"{bad} cases differ")


if __name__ == "__main__":
    main()
