"""GPU box: the native Motion-JPEG decoder against Pillow, bit for bit, over sampling / quality / size / restart variants."""
import itertools
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbs_amd import video_io as V  # noqa: E402


def frames_for(h, w, n, seed, gray):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    out = []
    for i in range(n):
        base = 128 + 90 * np.sin(xx / (7.0 + i)) * np.cos(yy / (5.0 + 2 * i))
        img = np.stack([base, 255 - base, 128 + 100 * np.sin((xx + yy) / 11.0)], axis=2)
        img += rng.normal(0, 12 + 8 * i, img.shape)
        img[h // 4:h // 2, w // 3:w // 2] = rng.integers(0, 256, 3)            # hard edges and saturated patches
        img[:6, :9] = 255
        img[-5:, -7:] = 0
        img = np.clip(img, 0, 255).astype(np.uint8)
        out.append(img[:, :, 0] if gray else img)
    return np.stack(out)


def random_cases(n, seed):
    """n random clips: size 1..260 per side, quality 1..100, any sampling, content from flat to noise, random Pillow options"""
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    bad = 0
    for it in range(n):
        h, w = int(rng.integers(1, 261)), int(rng.integers(1, 261))
        sub = [0, 1, 2, "gray"][int(rng.integers(0, 4))]
        q = int(rng.integers(1, 101))
        kind = int(rng.integers(0, 5))
        gray = sub == "gray"
        shape = (3, h, w) if gray else (3, h, w, 3)
        if kind == 0:
            fr = rng.integers(0, 256, shape).astype(np.uint8)
        elif kind == 1:
            fr = np.full(shape, int(rng.integers(0, 256)), np.uint8)
        elif kind == 2:
            fr = (np.indices(shape)[2] * 255 // max(w - 1, 1)).astype(np.uint8)
        elif kind == 3:
            fr = (rng.integers(0, 2, shape) * 255).astype(np.uint8)
        else:
            fr = frames_for(max(h, 8), max(w, 8), 3, it, gray)[:, :h, :w]
        opts = {}
        if rng.integers(0, 4) == 0:
            opts["restart_marker_blocks"] = int(rng.integers(1, 9))
        if rng.integers(0, 4) == 0:
            opts["optimize"] = True
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "a.avi")
            try:
                V.write_avi(p, np.ascontiguousarray(fr), quality=q, subsampling=0 if gray else sub, **opts)
            except OSError:                                 # (Pillow's encoder gives up on some option mixes: not a case)
                continue
            n_, want = V.AviReader(p).read_batch(3, threads=1)
            dec = V.MjpegDeviceDecoder(V.AviReader(p), dev, batch=2, threads=2)
            got, slot = [], 0
            while dec.entropy(slot):
                got.append(dec.reconstruct(slot).cpu().numpy().copy())
                slot ^= 1
            got = np.concatenate(got)
        ok = got.shape == want.shape and np.array_equal(got, want)
        bad += not ok
        if not ok or it % 50 == 0:
            print(f"random case {it}: {h}x{w} sub={sub} q={q} kind={kind} {opts}: {'OK' if ok else 'DIFF'}", flush=True)
    print("random cases:", n, "bad:", bad, flush=True)
    return bad


def main():
    dev = torch.device("cuda:0")
    bad = 0
    if len(sys.argv) > 2 and sys.argv[1] == "--random":
        return 1 if random_cases(int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 0) else 0
    cases = list(itertools.product((0, 1, 2, "gray"), (35, 75, 95, 100), ((48, 80), (61, 83), (480, 640), (17, 9))))
    extra = [dict(restart_marker_rows=1), dict(restart_marker_blocks=3), dict(optimize=True)]
    for sub, q, (h, w) in cases:
        for opts in ([{}] + (extra if (h, w) == (61, 83) else [])):
            gray = sub == "gray"
            fr = frames_for(h, w, 5, 7 * h + w + q, gray)
            with tempfile.TemporaryDirectory() as td:
                p = os.path.join(td, "a.avi")
                V.write_avi(p, fr, quality=q, subsampling=0 if gray else sub, **opts)
                r = V.AviReader(p)
                n, want = r.read_batch(5, threads=1)
                r2 = V.AviReader(p)
                dec = V.MjpegDeviceDecoder(r2, dev, batch=4, threads=2)
                got = []
                slot = 0
                while dec.entropy(slot):
                    got.append(dec.reconstruct(slot).cpu().numpy().copy())
                    slot ^= 1
                got = np.concatenate(got)
            d = np.abs(got.astype(int) - want.astype(int))
            ok = got.shape == want.shape and d.max() == 0
            bad += not ok
            print(f"sub={sub} q={q} {h}x{w} {opts}: {'OK' if ok else 'DIFF'} max|d|={d.max()} differing={int((d > 0).sum())} of {d.size}", flush=True)
            if not ok and d.size < 5000:
                ys, xs, cs = np.nonzero(d[0])
                print("   first diffs (y,x,c,got,want):", [(int(y), int(x), int(c), int(got[0, y, x, c]), int(want[0, y, x, c])) for y, x, c in list(zip(ys, xs, cs))[:8]])
    print("FAILED" if bad else "ALL OK", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
