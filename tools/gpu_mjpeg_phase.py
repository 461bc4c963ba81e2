"""Where the device half of the Motion-JPEG decoder spends its time (640x480, quality 70): uploads and the two kernels by HIP
events, at batch 64 and 256.  usage: gpu_mjpeg_phase.py"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import vbs_amd.synth as S
from vbs_amd.video_io import AviReader, MjpegDeviceDecoder, write_avi
spec = S.config1()
frames = S.make_frames(spec, range(64), seed=0, channels=3)
frames = np.concatenate([frames] * 4)
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "clip.avi")
    write_avi(path, frames, fps=12.0, codec="MJPG", quality=70)
    for batch in (64, 256):
        dec = MjpegDeviceDecoder(AviReader(path), dev, batch, 16)
        m = dec.entropy(0)
        for _ in range(3):
            dec.reconstruct(0)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        acc = np.zeros(3)
        wall = 0.0
        for _ in range(10):
            t0 = time.perf_counter()
            ev[0].record()
            ent, reg = dec._ent[0], dec._regions[0]
            for t in range(dec.threads):
                a, u = int(reg[2 * t]), int(reg[2 * t + 1])
                if u:
                    dec._dent[a:a + u].copy_(ent[a:a + u], non_blocking=True)
            dec._dtab[:m].copy_(dec._tab[0][:m], non_blocking=True)
            dec._dfb[:m].copy_(dec._fb[0][:m], non_blocking=True)
            dec._dqt[:m].copy_(dec._qt[0][:m], non_blocking=True)
            ev[1].record()
            out = dec._out[0]
            rc = dec._lib.vbs_mjpeg_reconstruct(dec._dent.data_ptr(), dec._dtab.data_ptr(), dec._dfb.data_ptr(), dec._dqt.data_ptr(), m,
                                                dec._info, dec._planes.data_ptr(), out.data_ptr(), out.stride(0), out.stride(1),
                                                torch.cuda.current_stream().cuda_stream)
            ev[2].record()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            acc += [ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), (t1 - t0) * 1e3]
            wall += t2 - t0
        acc /= 10
        print(f"batch {batch} ({m} frames): uploads {acc[0] * 1e3 / m:.2f} us per frame on the device timeline, kernels {acc[1] * 1e3 / m:.2f}, "
              f"host time to issue everything {acc[2] * 1e3 / m:.2f}; wall {wall / 10 * 1e6 / m:.2f} us per frame = {m / (wall / 10):.0f} frames/s; "
              f"{sum(int(reg[2 * t + 1]) for t in range(dec.threads)) * 4 / m / 1024:.1f} KiB of coefficients per frame", flush=True)
