"""Video front end without OpenCV (`MarkerTracker._init_video`, marker_detection.py:50-76).

The sensor's camera is opened with FOURCC MJPG (`Vedio_Capture/collecting.py:100`) and its recordings are AVI files,
which the reference hands to `cv2.VideoCapture`.  This module reads that container itself: a sequential walk of the
RIFF 'movi' list, every 'NNdc' / 'NNdb' chunk one frame.  Motion-JPEG frames are decoded natively (`MjpegDeviceDecoder`:
Huffman on C++ threads, IDCT / upsampling / colour on the device) or with Pillow (libjpeg: what cv2.imdecode and OpenCV's
own MJPEG reader use; cv2's FFmpeg backend has another IDCT); uncompressed DIB frames (24-bit BGR or 8-bit gray, bottom-up)
are reshaped.  The reader mimics the four `cv2.VideoCapture` calls the reference makes (`isOpened`, `get`, `read`,
`release`) and returns BGR frames like cv2 does.  Decoding is outside the benchmarked path (frames resident in HBM); it
bounds the real-world rate from a file: 50 k frames/s natively, 3.3 k through Pillow (640x480).  Other codecs (XVID,
H.264 ...) need a real decoder: IOError, as `cv2` absent did.

`write_avi` is the matching minimal writer (MJPG or uncompressed), used by the tests to make fixtures on the fly.
"""
from __future__ import annotations

import io
import os
import struct

import numpy as np

CAP_PROP_FRAME_WIDTH, CAP_PROP_FRAME_HEIGHT, CAP_PROP_FPS, CAP_PROP_FRAME_COUNT = 3, 4, 5, 7   # cv2's ids


def _chunks(buf: memoryview, start: int, end: int):
    """(fourcc, data_start, size) of the chunks in buf[start:end] (sizes are padded to even)."""
    pos = start
    while pos + 8 <= end:
        cc = bytes(buf[pos:pos + 4])
        size = struct.unpack_from("<I", buf, pos + 4)[0]
        yield cc, pos + 8, size
        pos += 8 + size + (size & 1)


class AviReader:
    """Sequential reader of MJPG / uncompressed AVI files with the `cv2.VideoCapture` calls the reference uses."""

    def __init__(self, path: str):
        self._ok = False
        self._buf = memoryview(b"")
        self._map = None
        self.width = self.height = 0
        self.fps = 0.0
        self._frames = []            # (offset, size) of every video chunk
        self._next = 0
        self._codec = b""
        self._bits = 24
        # The file is mapped, not read: a long recording is gigabytes, and only the chunk headers (here) and one batch of
        # frames at a time (the decoders) are ever touched.  (A private copy-on-write mapping: nothing writes to it, but a
        # read-only one cannot hand its address to the native decoder through ctypes.)
        try:
            import mmap
            with open(path, "rb") as f:
                if os.fstat(f.fileno()).st_size == 0:
                    return
                self._map = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_COPY)
            self._buf = memoryview(self._map)
        except (OSError, ValueError):
            return
        b = self._buf
        if len(b) < 12 or bytes(b[0:4]) != b"RIFF" or bytes(b[8:12]) != b"AVI ":
            return
        # OpenDML: a file beyond 1 GB continues in further RIFF chunks of form 'AVIX', each with its own 'movi' list
        pos = 0
        while pos + 12 <= len(b) and bytes(b[pos:pos + 4]) == b"RIFF" and bytes(b[pos + 8:pos + 12]) in (b"AVI ", b"AVIX"):
            rsize = struct.unpack_from("<I", b, pos + 4)[0]
            riff_end = min(len(b), pos + 8 + rsize)
            for cc, ds, size in _chunks(b, pos + 12, riff_end):
                if cc != b"LIST":
                    continue
                kind = bytes(b[ds:ds + 4])
                if kind == b"hdrl" and pos == 0:
                    self._parse_hdrl(ds + 4, ds + size)
                elif kind == b"movi":
                    self._scan_movi(ds + 4, min(ds + size, len(b)))
            pos = riff_end + (rsize & 1)
        self._ok = bool(self._frames) and self.width > 0 and self.height > 0 and \
            self._codec.upper() in (b"MJPG", b"\x00\x00\x00\x00", b"DIB ", b"RAW ")

    def _parse_hdrl(self, start, end):
        b = self._buf
        for cc, ds, size in _chunks(b, start, end):
            if cc == b"avih" and size >= 40:
                usec = struct.unpack_from("<I", b, ds)[0]
                self.fps = 1e6 / usec if usec else 0.0
                self.width, self.height = struct.unpack_from("<II", b, ds + 32)
            elif cc == b"LIST" and bytes(b[ds:ds + 4]) == b"strl":
                is_video = False
                for c2, d2, s2 in _chunks(b, ds + 4, ds + size):
                    if c2 == b"strh" and s2 >= 28:
                        is_video = bytes(b[d2:d2 + 4]) == b"vids"
                        scale, rate = struct.unpack_from("<II", b, d2 + 20)
                        if is_video and scale:
                            self.fps = rate / scale
                    elif c2 == b"strf" and is_video and s2 >= 40 and not self._codec:
                        w, h = struct.unpack_from("<ii", b, d2 + 4)
                        self._bits = struct.unpack_from("<H", b, d2 + 14)[0]
                        self._codec = bytes(b[d2 + 16:d2 + 20])
                        self.width, self.height = abs(w), abs(h)

    def _scan_movi(self, start, end):
        b = self._buf
        for cc, ds, size in _chunks(b, start, end):
            if cc == b"LIST" and bytes(b[ds:ds + 4]) == b"rec ":
                self._scan_movi(ds + 4, ds + size)
            elif cc[2:4] in (b"dc", b"db") and size > 0:
                self._frames.append((ds, size))

    # ---- the cv2.VideoCapture subset ----
    def isOpened(self) -> bool:
        return self._ok

    def get(self, prop: int) -> float:
        return {CAP_PROP_FRAME_WIDTH: float(self.width), CAP_PROP_FRAME_HEIGHT: float(self.height),
                CAP_PROP_FPS: float(self.fps), CAP_PROP_FRAME_COUNT: float(len(self._frames))}.get(prop, 0.0)

    def read(self):
        if not self._ok or self._next >= len(self._frames):
            return False, None
        off, size = self._frames[self._next]
        self._next += 1
        data = self._buf[off:off + size]
        if self._codec.upper() == b"MJPG":
            from PIL import Image
            out = np.empty((self.height, self.width, 3), dtype=np.uint8)
            self._next -= 1
            self._decode_into(self._next, out)                              # BGR like cv2
            self._next += 1
            return True, out
        stride = (self.width * self._bits // 8 + 3) & ~3                    # DIB rows are dword aligned, bottom-up
        rows = np.frombuffer(data, dtype=np.uint8, count=stride * self.height).reshape(self.height, stride)[::-1]
        if self._bits == 24:
            return True, np.ascontiguousarray(rows[:, :self.width * 3].reshape(self.height, self.width, 3))
        if self._bits == 8:
            g = rows[:, :self.width]
            return True, np.ascontiguousarray(np.repeat(g[:, :, None], 3, axis=2))
        return False, None

    def _decode_into(self, index: int, dst: np.ndarray):
        """frame `index` -> dst [H,W,3] BGR (any memory, e.g. a row of a page-locked batch buffer)"""
        off, size = self._frames[index]
        data = self._buf[off:off + size]
        if self._codec.upper() == b"MJPG":
            from PIL import Image
            im = Image.open(io.BytesIO(data))
            if im.mode != "RGB":
                im = im.convert("RGB")
            # BGR like cv2, packed by PIL's raw encoder in C: the NumPy view `[:, :, ::-1]` copied byte by byte and cost
            # more than the JPEG decode itself (2.2 of 3.8 ms per 640x480 frame; round 5)
            dst[...] = np.frombuffer(im.tobytes("raw", "BGR"), dtype=np.uint8).reshape(self.height, self.width, 3)
            return
        stride = (self.width * self._bits // 8 + 3) & ~3
        rows = np.frombuffer(data, dtype=np.uint8, count=stride * self.height).reshape(self.height, stride)[::-1]
        if self._bits == 24:
            dst[...] = rows[:, :self.width * 3].reshape(self.height, self.width, 3)
        elif self._bits == 8:
            dst[...] = rows[:, :self.width, None]
        else:
            raise IOError(f"unsupported DIB depth {self._bits}")

    def read_batch(self, n: int, out: np.ndarray | None = None, threads: int | None = None):
        """The next up to `n` frames, decoded by a pool of threads (libjpeg runs outside the GIL) straight into `out`
        [>= n, H, W, 3] - e.g. `marker_detection.pinned_frames`, from where the tracker uploads by DMA.  Returns the
        number of frames decoded (0 at the end); without `out` returns (count, array).  Not part of cv2's interface:
        `MarkerTracker.process` uses it when the capture object has it, to decode batch k + 1 while batch k computes."""
        import os
        from concurrent.futures import ThreadPoolExecutor
        m = max(0, min(int(n), len(self._frames) - self._next)) if self._ok else 0
        own = out is None
        if own:
            out = np.empty((m, self.height, self.width, 3), dtype=np.uint8)
        if m:
            first = self._next
            self._next += m
            # (4 by default: with the BGR packing in C a frame's Python-side work under the GIL is what is left, and more
            #  threads only contend for it - 2.0 / 4.1 / 3.5 / 3.3 k frames/s at 1 / 4 / 8 / 16 threads, profiles/r5e_decode_path.log)
            workers = max(1, threads or min(4, os.cpu_count() or 1))
            if workers == 1 or m == 1:
                for i in range(m):
                    self._decode_into(first + i, out[i])
            else:
                # ONE pool per reader, of the size asked for (a short last batch simply leaves workers idle: replacing the
                # pool when the batch length changed leaked the old one's threads)
                pool = getattr(self, "_pool", None)
                if pool is not None and getattr(self, "_pool_workers", 0) != workers:
                    pool.shutdown(wait=True)
                    pool = None
                if pool is None:
                    pool = self._pool = ThreadPoolExecutor(workers)
                    self._pool_workers = workers
                list(pool.map(lambda i: self._decode_into(first + i, out[i]), range(m)))
        return (m, out) if own else m

    def release(self):
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.shutdown(wait=False)
            self._pool = None
        self._buf = memoryview(b"")                       # (the mapping itself goes with its last user: a decoder may hold it)
        self._map = None
        self._frames = []
        self._ok = False


class MjpegDeviceDecoder:
    """Native Motion-JPEG front end for an `AviReader` (csrc/host_mjpeg.hip): the Huffman entropy decode of a batch runs
    on C++ threads (no GIL, no Python per frame) into page-locked buffers, as the non-zero quantised coefficients of every 8x8
    block; de-quantisation, libjpeg's `islow` inverse DCT, its "fancy" chroma upsampling and its fixed-point YCbCr -> RGB
    tables run as HIP kernels, so the pixels are born in HBM and only the non-zero coefficients cross PCIe (about a tenth of
    the pixels' bytes for a quality-70 sensor frame, never more than 1.5 bytes per pixel).  The BGR frames equal Pillow's
    (libjpeg-turbo, default settings: islow IDCT, fancy upsampling) bit for bit - tests/test_gpu_parity.py.

    Baseline sequential 8-bit JPEG with 1 (gray) or 3 components, luma sampling 1x1 / 2x1 / 2x2, chroma 1x1; frames without
    Huffman tables (camera MJPG streams) mean the standard ones of ITU-T T.81 Annex K, as for libjpeg-turbo and FFmpeg: what
    cameras, `cv2.VideoWriter('MJPG')` and Pillow write.  Anything else (progressive, arithmetic coding, 12 bit, CMYK):
    `ValueError` at construction, and the caller stays with Pillow.

    Two slots of host / device buffers, so that `entropy(slot)` of batch k + 1 can run on a helper thread while
    `reconstruct(slot)` and the tracker work on batch k."""

    def __init__(self, reader: "AviReader", device, batch: int, threads: int | None = None):
        import ctypes as C
        import os

        import torch

        from . import _lib as L
        if not reader.isOpened() or reader._codec.upper() != b"MJPG":
            raise ValueError("not a Motion-JPEG clip")
        self._lib = L.lib()
        self._reader = reader
        self._keep = reader._buf.obj                                   # the mapping behind the memoryview, kept alive here
        self._base = C.addressof(C.c_char.from_buffer(self._keep))
        off, size = reader._frames[reader._next if reader._next < len(reader._frames) else 0]
        self._info = (C.c_int32 * 8)()
        first = (C.c_char * size).from_buffer_copy(reader._buf[off:off + size])
        if self._lib.vbs_mjpeg_probe(first, size, self._info) != 0:
            raise ValueError("JPEG variant outside the native decoder (baseline Huffman, 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0)")
        if self._info[0] != reader.width or self._info[1] != reader.height:
            raise ValueError("JPEG geometry differs from the AVI header")
        self.batch = int(batch)
        self.device = torch.device(device)
        self.width, self.height = int(self._info[0]), int(self._info[1])
        self._per, self._pl = int(self._info[6]), int(self._info[7])
        self.threads = max(1, threads or min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
        pin = self.device.type == "cuda"
        self._cap, self._nblk = self._per // 2, self._per // 64
        # host side, two slots: entry words (never more than half the dense coefficients' bytes; typically a few percent of
        # them are used), the per-block table, frames' first words, tables, status, the spans the threads filled
        self._ent = [torch.empty(self.batch * self._cap, dtype=torch.int32, pin_memory=pin) for _ in range(2)]
        self._tab = [torch.empty((self.batch, self._nblk), dtype=torch.int32, pin_memory=pin) for _ in range(2)]
        self._fb = [torch.empty(self.batch, dtype=torch.int64, pin_memory=pin) for _ in range(2)]
        self._qt = [torch.empty((self.batch, 3, 64), dtype=torch.int16, pin_memory=pin) for _ in range(2)]
        self._status = [np.zeros(self.batch, dtype=np.int32) for _ in range(2)]
        self._regions = [np.zeros(2 * self.threads, dtype=np.int64) for _ in range(2)]
        self._count = [0, 0]
        self._dent = torch.empty(self.batch * self._cap, dtype=torch.int32, device=self.device)
        self._dtab = torch.empty((self.batch, self._nblk), dtype=torch.int32, device=self.device)
        self._dfb = torch.empty(self.batch, dtype=torch.int64, device=self.device)
        self._dqt = torch.empty((self.batch, 3, 64), dtype=torch.int16, device=self.device)
        self._planes = torch.empty((self.batch, self._pl), dtype=torch.uint8, device=self.device)
        self._out = [torch.empty((self.batch, self.height, self.width, 3), dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.uploaded_bytes = 0                                        # what crossed to the device so far (all batches)

    def entropy(self, slot: int, n: int | None = None) -> int:
        """Host half for the reader's next up to `n` (default: batch) frames into `slot`.  Returns the count (0 at the end).
        Safe on a helper thread: ctypes drops the GIL for the call and the C side starts its own threads."""
        r = self._reader
        m = max(0, min(self.batch if n is None else min(int(n), self.batch), len(r._frames) - r._next)) if r._ok else 0
        self._count[slot] = m
        if not m:
            return 0
        fr = r._frames[r._next:r._next + m]
        r._next += m
        offs = np.asarray([f[0] for f in fr], dtype=np.int64)
        sizes = np.asarray([f[1] for f in fr], dtype=np.int32)
        st = self._status[slot]
        bad = self._lib.vbs_mjpeg_entropy_batch(self._base, offs.ctypes.data, sizes.ctypes.data, m, self._info,
                                                self._ent[slot].data_ptr(), self._tab[slot].data_ptr(), self._fb[slot].data_ptr(),
                                                self._regions[slot].ctypes.data, self._qt[slot].data_ptr(), st.ctypes.data,
                                                self.threads)
        if bad < 0:
            raise RuntimeError(f"vbs_mjpeg_entropy_batch failed ({bad})")
        if bad:
            i = int(np.flatnonzero(st[:m])[0])
            raise IOError(f"Motion-JPEG frame {r._next - m + i}: corrupt or not of the clip's JPEG variant (status {int(st[i])})")
        return m

    def reconstruct(self, slot: int):
        """Device half of what `entropy(slot)` left: -> uint8 [m, H, W, 3] BGR on the device (a view of the slot's buffer,
        valid until the slot's next `reconstruct`), asynchronous on torch's current stream."""
        import torch
        m = self._count[slot]
        out = self._out[slot]
        if not m:
            return out[:0]
        ent, reg = self._ent[slot], self._regions[slot]
        for t in range(self.threads):                                  # only the words the host threads wrote
            a, u = int(reg[2 * t]), int(reg[2 * t + 1])
            if u:
                self._dent[a:a + u].copy_(ent[a:a + u], non_blocking=True)
                self.uploaded_bytes += 4 * u
        self._dtab[:m].copy_(self._tab[slot][:m], non_blocking=True)
        self._dfb[:m].copy_(self._fb[slot][:m], non_blocking=True)
        self._dqt[:m].copy_(self._qt[slot][:m], non_blocking=True)
        self.uploaded_bytes += m * (4 * self._nblk + 8 + 384)
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
        rc = self._lib.vbs_mjpeg_reconstruct(self._dent.data_ptr(), self._dtab.data_ptr(), self._dfb.data_ptr(), self._dqt.data_ptr(), m,
                                             self._info, self._planes.data_ptr(), out.data_ptr(), out.stride(0), out.stride(1), stream)
        if rc != 0:
            raise RuntimeError(f"vbs_mjpeg_reconstruct failed ({rc})")
        return out[:m]


def write_avi(path: str, frames: np.ndarray, fps: float = 30.0, codec: str = "MJPG", quality: int = 95, subsampling: int = 2,
              riff_frames: int = 0, **jpeg_options):
    """frames uint8 [N,H,W,3] BGR or [N,H,W] gray -> AVI with one 'movi' list and an 'idx1' index.  `subsampling` (0 = 4:4:4,
    1 = 4:2:2, 2 = 4:2:0) and further Pillow JPEG options (`restart_marker_rows`, `optimize` ...) apply to MJPG.  `riff_frames`
    > 0 continues the file after that many frames in 'AVIX' RIFF chunks, the way OpenDML writers split files beyond 1 GB."""
    frames = np.asarray(frames)
    if frames.dtype != np.uint8 or frames.ndim not in (3, 4):
        raise ValueError("frames must be uint8 [N,H,W] or [N,H,W,3]")
    n, h, w = frames.shape[:3]
    gray = frames.ndim == 3
    payloads = []
    if codec.upper() == "MJPG":
        from PIL import Image
        for fr in frames:
            im = Image.fromarray(fr if gray else np.ascontiguousarray(fr[:, :, ::-1]))
            bio = io.BytesIO()
            im.save(bio, format="JPEG", quality=quality, subsampling=0 if gray else subsampling, **jpeg_options)
            payloads.append(bio.getvalue())
        fourcc, bits = b"MJPG", 24
    else:
        bits = 8 if gray else 24
        stride = (w * bits // 8 + 3) & ~3
        for fr in frames:
            rows = np.zeros((h, stride), dtype=np.uint8)
            rows[:, :w * bits // 8] = fr.reshape(h, -1)
            payloads.append(rows[::-1].tobytes())
        fourcc = b"\x00\x00\x00\x00"
    tag = b"00dc" if fourcc == b"MJPG" else b"00db"

    def chunk(cc, data):
        return cc + struct.pack("<I", len(data)) + data + (b"\x00" if len(data) & 1 else b"")

    def lst(kind, data):
        return b"LIST" + struct.pack("<I", len(data) + 4) + kind + data

    maxsz = max(len(p) for p in payloads) if payloads else 0
    avih = struct.pack("<14I", int(round(1e6 / fps)) if fps else 0, 0, 0, 0x10, n, 0, 1, maxsz, w, h, 0, 0, 0, 0)
    rate, scale = int(round(fps * 1000)), 1000
    strh = b"vids" + fourcc + struct.pack("<IHHIIIIIIII4h", 0, 0, 0, 0, scale, rate, 0, n, maxsz, 0xFFFFFFFF, 0,
                                         0, 0, w, h)
    strf = struct.pack("<IiiHH4sIiiII", 40, w, h, 1, bits, fourcc, maxsz, 0, 0, 256 if bits == 8 else 0, 0)
    if bits == 8:
        strf += b"".join(struct.pack("<4B", i, i, i, 0) for i in range(256))
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    first = payloads[:riff_frames] if riff_frames > 0 else payloads
    movi_body, index, off = b"", b"", 4
    for p in first:
        index += tag + struct.pack("<III", 0x10, off, len(p))
        c = chunk(tag, p)
        movi_body += c
        off += len(c)
    body = b"AVI " + hdrl + lst(b"movi", movi_body) + chunk(b"idx1", index)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
        for k in range(len(first), len(payloads), max(riff_frames, 1)):       # OpenDML continuation chunks
            ext = b"AVIX" + lst(b"movi", b"".join(chunk(tag, p) for p in payloads[k:k + riff_frames]))
            f.write(b"RIFF" + struct.pack("<I", len(ext)) + ext)
