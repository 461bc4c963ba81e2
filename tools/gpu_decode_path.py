"""The video front end: Motion-JPEG AVI of the sensor's format (640x480, MJPG) -> frames -> CSV.
Decode rates of the package's own reader (single thread = the reference-like loop, then the thread pool), and
MarkerTracker.process() end to end on the file with the reference's default crop.  usage: gpu_decode_path.py [frames]"""
import os, sys, time, tempfile, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vbs_amd.synth as S
from vbs_amd.video_io import AviReader, write_avi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spec = S.config1()
frames = S.make_frames(spec, range(min(n, 64)), seed=0, channels=3)
frames = np.concatenate([frames] * ((n + len(frames) - 1) // len(frames)))[:n]
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "clip.avi")
    write_avi(path, frames, fps=12.0, codec="MJPG", quality=70)       # collecting.py: MJPG, quality 70, 12 fps
    print(f"{n} frames 640x480 MJPG q70: {os.path.getsize(path) / n / 1024:.1f} KiB per frame", flush=True)
    cap = AviReader(path)
    t0 = time.perf_counter(); k = 0
    while cap.read()[0]:
        k += 1
    dt = time.perf_counter() - t0
    print(f"read() one by one: {k / dt:.0f} frames/s", flush=True)
    buf = np.empty((64, spec.height, spec.width, 3), np.uint8)
    for th in (1, 4, 8, 16):
        cap = AviReader(path)
        t0 = time.perf_counter(); k = 0
        while True:
            m = cap.read_batch(64, buf, threads=th)
            if not m:
                break
            k += m
        dt = time.perf_counter() - t0
        print(f"read_batch, {th:2d} threads: {k / dt:.0f} frames/s", flush=True)
        cap.release()
    try:
        import torch
        if torch.cuda.is_available():
            from vbs_amd.video_io import MjpegDeviceDecoder
            from vbs_amd.marker_detection import MarkerTracker
            dev = torch.device("cuda:0")
            for th in (1, 4, 8, 16):
                cap = AviReader(path)
                dec = MjpegDeviceDecoder(cap, dev, 64, th)
                t0 = time.perf_counter(); k = 0
                while True:
                    m = dec.entropy(0)
                    if not m:
                        break
                    k += m
                dt = time.perf_counter() - t0
                print(f"native entropy decode (host half), {th:2d} threads: {k / dt:.0f} frames/s", flush=True)
            cap = AviReader(path)
            dec = MjpegDeviceDecoder(cap, dev, 64, 16)
            dec.entropy(0)
            for _ in range(3):
                dec.reconstruct(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                dec.reconstruct(0)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"native reconstruct (upload + IDCT + colour, device half): {20 * 64 / dt:.0f} frames/s; "
                  f"{dec.uploaded_bytes / (23 * 64) / 1024:.1f} KiB per frame uploaded ({640 * 480 * 3 / 1024:.0f} KiB of pixels)", flush=True)
            texts = {}
            for on_dev in (False, True):
                for batch in (64, 256):
                    for rep in range(2):
                        out = os.path.join(td, f"o{int(on_dev)}_{batch}_{rep}")
                        with contextlib.redirect_stdout(sys.stderr):
                            trk = MarkerTracker({"video_path": path, "output_dir": out, "mjpeg_on_device": on_dev,
                                                 "crop_ratios": (1 / 8, 1 / 8, 1 / 16, 0), "id_mode": "full", "batch": batch})
                            t0 = time.perf_counter(); trk.process(); dt = time.perf_counter() - t0
                    texts[(on_dev, batch)] = open(trk.output_csv, "rb").read()
                    print(f"MarkerTracker.process() on the AVI, decode path {trk.decode_path}, batch {batch}: {n / dt:.0f} frames/s",
                          flush=True)
            same = len(set(texts.values())) == 1
            print("CSV files of the two decode paths and both batch sizes identical:", same, flush=True)
            if not same:
                sys.exit(1)
    except ImportError:
        pass
