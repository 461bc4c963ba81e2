"""Drop-in for the reference's `code/Marker_Tracking/marker_detection.py`: same class, method names,
argument meaning and exceptions; the per-frame work runs in HIP kernels on the MI355X (libvbs.so).

    tracker = MarkerTracker(config)      # config keys as `marker_detection.py:478-489`
    tracker.process()                    # video -> <name>_markers.csv        (`:429-462`)

Differences a maintainer should know (all deliberate):
 * frames are processed in device batches through the fused `vbs_track_to_3d`; the static per-stage
   methods (`_find_markers`, `_marker_center`, `_normxcorr2`) are kept with NumPy in / NumPy out;
 * video decode needs OpenCV, which is outside this path: `video_path` may also be a `.npy` / `.npz`
   array of frames [N,H,W(,3)] uint8, and `process_frames(frames)` takes frames already in memory;
   the annotated AVI (`:69-76,453`) and the drawing calls are not produced (visual only);
 * extra optional config keys: `id_mode` ("as_written" | "full"), `kmeans` ("optimal" | "sklearn"),
   `device` (GPU index), `batch` (frames per device batch), `gray_coeffs` (15 = OpenCV 4's BGR2GRAY fixed-point
   set, the default; 14 = the older set; identical on grey frames).
 * a frame the device workspace cannot hold (more than 30720 runs in a mask, more than 1024 band components or more
   than about 512 contours) RAISES with its frame number instead of being dropped; the rows of the batches before it are
   written to the CSV first (`process`) / ride on the exception as `.rows` (`process_frames`).
There is no CPU fallback: without a GPU or without the built library every compute call raises.
"""
from __future__ import annotations

import os

import numpy as np

from . import ids as _ids

CSV_COLUMNS = ["frameno", "row", "col", "Ox", "Oy", "Cx", "Cy", "major_axis", "minor_axis", "angle"]

_ENGINES = {}


def _engine(height, width, device=None, max_batch=16, calibration=None, gray_coeffs=15):
    """One cached engine per (frame size, device, undistortion setup, BGR2GRAY coefficient set)."""
    import torch
    from .engine import Engine
    dev = torch.cuda.current_device() if (device is None and torch.cuda.is_available()) else (device or 0)
    ukey = None
    if calibration is not None:
        K = np.asarray(calibration["camera_matrix"], dtype=np.float64).reshape(3, 3)
        D = np.asarray(calibration["dist_coeffs"], dtype=np.float64).ravel()
        ukey = (K.tobytes(), D.tobytes())
    key = (int(height), int(width), int(dev), ukey, int(gray_coeffs))
    eng = _ENGINES.get(key)
    if eng is None or eng.max_batch < max_batch:
        eng = Engine(height, width, max_markers=1024, max_batch=max_batch, device=int(dev))
        if calibration is not None:
            eng.set_undistort(K, D)
        if int(gray_coeffs) != 15:
            from . import _lib as L
            eng.set_option(L.OPT_GRAY_COEFFS, int(gray_coeffs))
        _ENGINES[key] = eng
    return eng


def _to_device(arr, eng):
    import torch
    return torch.from_numpy(np.ascontiguousarray(arr)).to(eng.device)


def _crop_box(width, height, crop_ratios):
    """`marker_detection.py:81-84` (int() truncation)."""
    left = int(width * crop_ratios[0])
    right = width - int(width * crop_ratios[1])
    top = int(height * crop_ratios[2])
    bottom = height - int(height * crop_ratios[3])
    return left, right, top, bottom


def _det_to_markers(det, count):
    """det rows [x, y, major, minor, angle, label] -> the reference's marker dicts (`:238-243`)."""
    return [{"center": (float(r[0]), float(r[1])), "major_axis": float(r[2]), "minor_axis": float(r[3]),
             "angle": float(r[4])} for r in det[:count]]


def _format_csv_rows(cols):
    """The CSV text (no header line) of the rows in `cols`, exactly as `DataFrame(cols).to_csv(index=False)` writes them
    (shortest round-trip floats as repr prints them, empty cell for NaN), from the library's host-side formatter
    (`vbs_format_csv`, several threads, no GIL): one repr() per cell in Python was 90 % of the drop-in's wall time."""
    import ctypes as C
    from . import _lib as L
    ints = [np.ascontiguousarray(np.asarray(cols[c]).astype(np.int64, copy=False)) for c in CSV_COLUMNS[:3]]
    flts = [np.ascontiguousarray(np.asarray(cols[c], dtype=np.float64)) for c in CSV_COLUMNS[3:]]
    n = int(ints[0].shape[0])
    if n == 0:
        return b""
    cap = n * (65 + 26 * len(flts))
    buf = np.empty(cap, dtype=np.uint8)
    ptrs = (C.c_void_p * len(flts))(*[a.ctypes.data for a in flts])
    got = L.lib().vbs_format_csv(ints[0].ctypes.data, ints[1].ctypes.data, ints[2].ctypes.data, ptrs, len(flts), n,
                                 buf.ctypes.data, cap, min(8, os.cpu_count() or 1))
    if got < 0:
        raise L.VbsError(f"vbs_format_csv failed ({got})")
    return buf[:got].tobytes()


def _write_csv_columns(path, cols):
    """`DataFrame(cols).to_csv(path, index=False)`, byte for byte."""
    with open(path, "wb") as f:
        f.write((",".join(CSV_COLUMNS) + "\n").encode())
        f.write(_format_csv_rows(cols))


class _RowBlock:
    """The CSV rows of one device batch as columns (NumPy arrays); iterates / indexes as the reference's row dicts."""

    def __init__(self, cols, format_now=False):
        self.cols = cols
        self.n = len(cols["frameno"])
        self._text = self._thread = None
        if format_now:                                  # the batch's CSV text is formatted in the background (the formatter
            import threading                            # holds no GIL) while the device works on the next batch
            self._thread = threading.Thread(target=self._format, daemon=True)
            self._thread.start()

    def _format(self):
        self._text = _format_csv_rows(self.cols)

    def csv_text(self):
        if self._thread is not None:
            self._thread.join()
            self._thread = None
        if self._text is None:
            self._format()
        return self._text

    def __len__(self):
        return self.n

    def __iter__(self):
        names = CSV_COLUMNS
        arrs = [self.cols[c].tolist() for c in names]
        return (dict(zip(names, vals)) for vals in zip(*arrs))


class _Rows:
    """What `process_frames` returns: the batches' blocks; behaves as the flat list of row dicts."""

    def __init__(self, blocks):
        self.blocks = blocks

    def __len__(self):
        return sum(len(b) for b in self.blocks)

    def __iter__(self):
        for b in self.blocks:
            yield from b

    def __getitem__(self, i):
        return list(self)[i]


class MarkerTracker:
    """A comprehensive marker tracking system for video analysis (GPU implementation)."""

    def __init__(self, config):
        self.config = config
        self._validate_config()
        self._setup_paths()
        self.frame_count = 0
        self.first_frame_markers = {}

    # ---- configuration (`:33-48`) ------------------------------------------------------------
    def _validate_config(self):
        for key in ("video_path", "output_dir", "crop_ratios"):
            if key not in self.config:
                raise ValueError(f"Missing required config key: {key}")
        if not os.path.exists(self.config["video_path"]):
            raise FileNotFoundError(f"Video file not found: {self.config['video_path']}")

    def _setup_paths(self):
        os.makedirs(self.config["output_dir"], exist_ok=True)
        video_name = os.path.splitext(os.path.basename(self.config["video_path"]))[0]
        self.output_csv = os.path.join(self.config["output_dir"], f"{video_name}_markers.csv")
        self.output_video = os.path.join(self.config["output_dir"], f"{video_name}_tracked.avi")

    # ---- video source (`:50-76`) ---------------------------------------------------------------
    def _init_video(self):
        path = self.config["video_path"]
        self._frames = None
        self.cap = None
        if path.endswith(".npy") or path.endswith(".npz"):
            data = np.load(path)                     # allow_pickle=False (default)
            if hasattr(data, "files"):
                data = data[data.files[0]]
            if data.dtype != np.uint8 or data.ndim not in (3, 4):
                raise IOError(f"Could not open video: {path}")
            self._frames = data
            self.fps = float(self.config.get("fps", 0.0))
            self.height, self.width = int(data.shape[1]), int(data.shape[2])
        else:
            try:
                # config["video_reader"] = "native": the package's own reader (and with it the device-side Motion-JPEG
                # decode) also where OpenCV is installed; the default keeps the reference's cv2.VideoCapture when there is one
                if self.config.get("video_reader", "auto") == "native":
                    raise ImportError("the package's own reader was asked for")
                import cv2
                self.cap = cv2.VideoCapture(path)
                props = (cv2.CAP_PROP_FPS, cv2.CAP_PROP_FRAME_WIDTH, cv2.CAP_PROP_FRAME_HEIGHT)
            except ImportError:
                # no OpenCV: the package's own reader for the sensor's recordings (Motion-JPEG / uncompressed AVI)
                from . import video_io
                self.cap = video_io.AviReader(path)
                props = (video_io.CAP_PROP_FPS, video_io.CAP_PROP_FRAME_WIDTH, video_io.CAP_PROP_FRAME_HEIGHT)
            if not self.cap.isOpened():
                raise IOError(f"Could not open video: {path}")
            self.fps = self.cap.get(props[0])
            self.width = int(self.cap.get(props[1]))
            self.height = int(self.cap.get(props[2]))
        left, right, top, bottom = _crop_box(self.width, self.height, self.config["crop_ratios"])
        self.crop_width, self.crop_height = right - left, bottom - top

    def _preprocess_frame(self, frame):
        """Crop (a view, `:80-85`) and, when `calibration_params` is configured, undistort (`:88-89`)."""
        left, right, top, bottom = _crop_box(self.width, self.height, self.config["crop_ratios"])
        cropped = frame[top:bottom, left:right]
        if "calibration_params" in self.config:
            cropped = self._undistort_frame(cropped)
        return cropped

    def _undistort_frame(self, frame):
        """`:93-109`: getOptimalNewCameraMatrix(alpha=0) + initUndistortRectifyMap(CV_16SC2) + remap(INTER_LINEAR);
        the maps are built once per (size, K, D), not per frame."""
        frame = np.asarray(frame)
        eng = _engine(frame.shape[0], frame.shape[1], self.config.get("device"), calibration=self.config["calibration_params"])
        return eng.undistort_frames(_to_device(frame[None], eng))[0].cpu().numpy()

    # ---- per-stage static methods, NumPy in / NumPy out ------------------------------------------
    @staticmethod
    def _find_markers(frame):
        """`:112-135` -> (mask uint8 {0,1}, area_mask uint8 {0,255}), both [h, w]."""
        frame = np.asarray(frame)
        if frame.dtype != np.uint8 or frame.ndim not in (2, 3):
            raise ValueError("frame must be a uint8 image [h,w,3] (BGR) or [h,w]")
        eng = _engine(frame.shape[0], frame.shape[1])
        mask, area = eng.find_markers(_to_device(frame[None], eng))
        return mask[0].cpu().numpy(), area[0].cpu().numpy()

    @staticmethod
    def _gkern(l=5, sig=1.0):
        """`:138-143`: l x l Gaussian, sum 1 (host table)."""
        ax = np.linspace(-(l - 1) / 2.0, (l - 1) / 2.0, l)
        xx, yy = np.meshgrid(ax, ax)
        k = np.exp(-0.5 * (np.square(xx) + np.square(yy)) / np.square(sig))
        return k / np.sum(k)

    @staticmethod
    def _normxcorr2(template, image, mode="same"):
        """`:146-164`.  The operands the pipeline uses (`image` a uint8 area_mask with values in {0, 255}, `template` the
        `_gkern` of that image size's branch, mode 'same') run on the hot path's NCC kernel; any other template / image /
        mode is evaluated by the general float64 kernel (`vbs_normxcorr2_general`)."""
        image = np.asarray(image)
        template = np.asarray(template)
        if np.ndim(template) > np.ndim(image) or any(t > i for t, i in zip(template.shape, image.shape)):
            print("Warning: Template larger than image. Arguments may be swapped.")          # `:147-149`
        if image.ndim == 2 and image.dtype == np.uint8 and mode == "same" and image.shape[0] >= 64 \
                and 128 <= image.shape[1] <= 4096:
            l, sig = (33, 7.4) if image.shape[0] <= 480 else (80, 13.0)
            if template.shape == (l, l) and not np.any((image != 0) & (image != 255)) and \
                    np.allclose(template, MarkerTracker._gkern(l, sig), rtol=1e-12, atol=0):
                eng = _engine(image.shape[0], image.shape[1])
                return eng.normxcorr2(_to_device(image[None], eng))[0].cpu().numpy()
        return MarkerTracker._normxcorr2_general(template, image, mode)

    @staticmethod
    def _normxcorr2_general(template, image, mode="same"):
        from .engine import normxcorr2_general
        return normxcorr2_general(template, image, mode).cpu().numpy()

    @staticmethod
    def _marker_center(mask, area_mask, frame=None):
        """`:166-249` -> list of {'center': (x, y), 'major_axis', 'minor_axis', 'angle'} in the
        reference's order.  `frame` (drawing target) is ignored."""
        mask = np.asarray(mask)
        area_mask = np.asarray(area_mask)
        if mask.shape != area_mask.shape or mask.ndim != 2:
            raise ValueError("mask and area_mask must be 2-D arrays of the same shape")
        eng = _engine(mask.shape[0], mask.shape[1])
        det, counts = eng.marker_center(_to_device(mask.astype(np.uint8)[None], eng),
                                        _to_device(area_mask.astype(np.uint8)[None], eng))
        n = int(counts[0].item())
        if n < 0:
            from ._lib import VbsError
            from ._lib import status_text
            raise VbsError(f"_marker_center: device status {n}: {status_text(n, eng.max_markers)}")
        holes = int(eng.frame_stats(1)[0, 4])
        if holes:
            import warnings
            warnings.warn(f"{holes} hole(s) of the opened area mask could not be filled (capacity): contour vertices "
                          "include their borders, cv2.findContours(RETR_EXTERNAL) would ignore them", RuntimeWarning)
        return _det_to_markers(det[0].cpu().numpy(), n)

    # ---- identities and tracking -----------------------------------------------------------------
    def _process_first_frame(self, markers):
        """`:275-347`.  Raises ValueError when no marker was found."""
        table = _ids.assign_ids(markers, self.config.get("num_layers", 5),
                                self.config.get("id_mode", "as_written"), self.config.get("kmeans", "optimal"))
        self.first_frame_markers.update(table)

    def _track_markers(self, frame, markers):
        """`:349-396` -> list of row dicts (CSV_COLUMNS) for this frame."""
        if not self.first_frame_markers or not markers:
            return []
        import torch
        shape = getattr(frame, "shape", None)
        h, w = (shape[0], shape[1]) if shape is not None else (self.crop_height, self.crop_width)
        eng = _engine(h, w, self.config.get("device"))
        det = np.zeros((1, eng.max_markers, 6), dtype=np.float64)
        k = min(len(markers), eng.max_markers)
        for i, m in enumerate(markers[:k]):
            det[0, i, :5] = (m["center"][0], m["center"][1], m["major_axis"], m["minor_axis"], m["angle"])
        _, ref_xy = _ids.reference_arrays(self.first_frame_markers)
        table = eng.track(_to_device(det, eng), torch.tensor([k], dtype=torch.int32, device=eng.device), ref_xy,
                          self.config.get("min_marker_distance", 20)).cpu().numpy()[0]
        rows = []
        for slot, ((layer, idx), ref) in enumerate(self.first_frame_markers.items()):
            if int(table[slot, 0]) & 1:
                cur = markers[int(table[slot, 9])]
                rows.append({"frameno": self.frame_count, "row": layer, "col": idx, "Ox": ref["Ox"],
                             "Oy": ref["Oy"], "Cx": cur["center"][0], "Cy": cur["center"][1],
                             "major_axis": cur["major_axis"], "minor_axis": cur["minor_axis"],
                             "angle": cur["angle"]})
        return rows

    # ---- overlays (`:252-273`, `:398-427`) -------------------------------------------------------------
    # Drawing into the annotated AVI is out of scope (a15 "_draw_tracking is OUT", a16 "AVI writer OUT");
    # the names exist so that a caller written against the reference gets a no-op instead of an AttributeError.
    @staticmethod
    def _draw_marker(frame, center, ellipse, major, minor, angle):
        return None

    def _draw_tracking(self, frame, ref, curr):
        return None

    # ---- main loop (`:429-462`) ---------------------------------------------------------------------
    def process(self):
        self._init_video()
        if self._frames is not None:
            data = self.process_frames(self._frames).blocks
        elif hasattr(self.cap, "read_batch"):
            data = self._process_decoding_ahead(int(self.config.get("batch", 64)))
        else:
            batch = int(self.config.get("batch", 64))
            data, buf = [], []
            while True:
                ret, frame = self.cap.read()
                if ret:
                    buf.append(frame)
                if buf and (len(buf) == batch or not ret):
                    try:
                        data.append(self._process_batch(np.stack(buf)))
                    except Exception:
                        # a frame the workspace cannot hold: the rows of the batches before it are not lost (the reference
                        # would have written a CSV covering them too)
                        if data:
                            self._save_results(data)
                        self._cleanup()
                        raise
                    buf = []
                if not ret:
                    break
        self._save_results(data)
        self._cleanup()

    def _process_decoding_ahead(self, batch):
        """`process` on the package's own AVI reader (no OpenCV): batch k + 1 is decoded while batch k is computed and
        turned into rows.  A Motion-JPEG clip of the variant the native decoder takes (`video_io.MjpegDeviceDecoder`:
        baseline Huffman, 4:4:4 / 4:2:2 / 4:2:0 or gray) is entropy-decoded by C++ threads and reconstructed on the device
        (`config["mjpeg_on_device"]`, default on; `self.decode_path` says which ran); anything else goes through the
        reader's Pillow thread pool into one of two page-locked buffers."""
        from concurrent.futures import ThreadPoolExecutor
        dec = None
        if self.config.get("mjpeg_on_device", True) and getattr(self.cap, "_codec", b"").upper() == b"MJPG":
            try:
                import torch
                from .video_io import MjpegDeviceDecoder
                dev = torch.device(self.config.get("device") or "cuda:0")
                if dev.type == "cuda":
                    dec = MjpegDeviceDecoder(self.cap, dev, batch, self.config.get("decode_threads"))
            except (ValueError, MemoryError):               # a JPEG variant outside the native decoder, or no room for its
                dec = None                                  # page-locked buffers: Pillow, as before
        self.decode_path = "device" if dec is not None else "pillow"
        if dec is not None:
            def ahead_of(slot): return dec.entropy(slot)
            def frames_of(slot, m): return dec.reconstruct(slot)
        else:
            shape = (batch, self.height, self.width, 3)
            try:
                bufs = [pinned_frames(shape), pinned_frames(shape)]
            except Exception:                               # (no GPU runtime to pin with: ordinary memory)
                bufs = [np.empty(shape, np.uint8), np.empty(shape, np.uint8)]
            def ahead_of(slot): return self.cap.read_batch(batch, bufs[slot])
            def frames_of(slot, m): return bufs[slot][:m]
        data, k = [], 0
        with ThreadPoolExecutor(1) as ahead:
            fut = ahead.submit(ahead_of, 0)
            while True:
                try:
                    m = fut.result()                       # (a decode error - a corrupt frame - surfaces here)
                    fut = None
                    if not m:
                        break
                    fut = ahead.submit(ahead_of, (k + 1) & 1)
                    data.append(self._process_batch(frames_of(k & 1, m)))
                except Exception:
                    # the rows of the batches before the failing frame are kept, whichever side failed; the reader is
                    # released only once the decode running ahead has finished with its buffers (a running task cannot be
                    # cancelled)
                    if fut is not None and not fut.cancel():
                        try:
                            fut.result()
                        except Exception:
                            pass
                    if data:
                        self._save_results(data)
                    self._cleanup()
                    raise
                k += 1
        return data

    def process_frames(self, frames):
        """In-memory variant of `process`: frames uint8 [N,H,W(,3)] (NumPy or a torch device tensor)."""
        if not hasattr(self, "width"):
            self.height, self.width = int(frames.shape[1]), int(frames.shape[2])
            left, right, top, bottom = _crop_box(self.width, self.height, self.config["crop_ratios"])
            self.crop_width, self.crop_height = right - left, bottom - top
        batch = int(self.config.get("batch", 256))
        data = []
        self._batch_hint = min(batch, int(frames.shape[0]))      # (the engine's workspace: not the short first batch's size)
        for part in _upload_ahead(frames, batch, self.config.get("device"), self.config.get("first_batch_div", 4)):
            try:
                data.append(self._process_batch(part))
            except Exception as e:
                e.rows = _Rows(data)                    # what the batches before the failing frame produced
                if getattr(self, "_frames", None) is not None and data:
                    self._save_results(data)            # (called from `process`: keep the partial CSV)
                raise
        return _Rows(data)

    def _process_batch(self, frames):
        import torch
        left, right, top, bottom = _crop_box(self.width, self.height, self.config["crop_ratios"])
        eng = _engine(bottom - top, right - left, self.config.get("device"),
                      max(min(int(self.config.get("batch", 64)), max(int(frames.shape[0]), 1)), getattr(self, "_batch_hint", 1)),
                      calibration=self.config.get("calibration_params"),    # undistortion, if any, runs inside the engine
                      gray_coeffs=self.config.get("gray_coeffs", 15))
        ft = frames if isinstance(frames, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(frames))
        ft = ft.to(eng.device)[:, top:bottom, left:right]           # crop = strided view, no copy
        if self.frame_count == 0:
            _, det, counts = eng.track_to_3d(ft[:1], None, want_det=True)
            n0 = int(counts[0].item())
            if n0 < 0:
                from ._lib import VbsError
                raise VbsError(f"device status {n0} in frame 0")
            self._process_first_frame(_det_to_markers(det[0].cpu().numpy(), n0))
            # f4: the same assignment on the device (vbs_assign_ids) with the host table as its checker: where the two agree
            # bit for bit the device's arrays are the ones the tracking calls use, else the host's (the reference's own
            # order: the two may only differ in the order of markers at mathematically equal angles).  `first_frame_markers`
            # - the reference's attribute, marker dicts and all - is the host table either way.  The kernel covers
            # num_layers <= 16 (k_ids.hip): a configuration beyond that runs on the host alone, as it always did.
            self._ref_arrays = None
            if (self.config.get("kmeans", "optimal") == "optimal" and self.config.get("ids_on_device", True)
                    and int(self.config.get("num_layers", 5)) <= 16):
                ids_h, xy_h = _ids.reference_arrays(self.first_frame_markers)
                ids_d, xy_d = eng.assign_ids(det, counts, self.config.get("num_layers", 5),
                                             self.config.get("id_mode", "as_written"))
                ids_d, xy_d = ids_d.cpu().numpy().astype(np.int64), xy_d.cpu().numpy()
                if ids_d.shape != np.asarray(ids_h).shape or not np.array_equal(ids_d, ids_h):
                    from ._lib import VbsError
                    raise VbsError("vbs_assign_ids disagrees with the host assignment on the marker IDs of frame 0")
                same = bool(np.array_equal(xy_d, xy_h))
                self.ids_device_check = {"equal_to_host": same, "used": "device" if same else "host",
                                         "slots_in_another_order": int((xy_d != np.asarray(xy_h)).any(axis=1).sum())}
                if same:
                    self._ref_arrays = (ids_d, xy_d)
            else:
                self.ids_device_check = {"on_device": False, "used": "host"}
        ids, ref_xy = getattr(self, "_ref_arrays", None) or _ids.reference_arrays(self.first_frame_markers)
        table, det, counts = eng.track_to_3d(ft, ref_xy, self.config.get("min_marker_distance", 20),
                                             want_det=True)
        counts = counts.cpu().numpy()
        if (counts < 0).any():                  # the reference would have emitted rows: never drop a frame silently
            from ._lib import VbsError
            bad = int(np.nonzero(counts < 0)[0][0])
            from ._lib import status_text
            raise VbsError(f"device status {int(counts[bad])} in frame {self.frame_count + bad}: "
                           f"{status_text(int(counts[bad]), eng.max_markers)}; rows of the batches before it are kept "
                           f"(`.rows` of this exception / the partial CSV)")
        table = table.cpu().numpy()
        det = det[:, :max(int(counts.max()), 1)].cpu().numpy()      # float64 rows in use: the CSV keeps the reference's precision
        # all rows of the batch at once: (frame, slot) pairs in frame-major, reference-dict order
        fi, si = np.nonzero(table[:, :, 0].astype(np.int64) & 1)
        d = det[fi, table[fi, si, 9].astype(np.int64)]
        refs = np.array([(r["Ox"], r["Oy"]) for r in self.first_frame_markers.values()], dtype=np.float64).reshape(-1, 2)
        block = {"frameno": self.frame_count + fi, "row": np.asarray(ids)[si, 0].astype(np.int64),
                 "col": np.asarray(ids)[si, 1].astype(np.int64), "Ox": refs[si, 0], "Oy": refs[si, 1],
                 "Cx": d[:, 0], "Cy": d[:, 1], "major_axis": d[:, 2], "minor_axis": d[:, 3], "angle": d[:, 4]}
        n_before = self.frame_count
        self.frame_count += table.shape[0]
        for k in range(n_before // 100 + 1, self.frame_count // 100 + 1):
            print(f"Processed frame {100 * k}")
        return _RowBlock(block, format_now=True)

    def _save_results(self, data):
        """`:464-468`.  `data`: row dicts (the reference's form) or the per-batch column blocks of `_process_batch`."""
        import pandas as pd
        if isinstance(data, _Rows):
            data = data.blocks
        if data and all(isinstance(b, _RowBlock) for b in data):
            with open(self.output_csv, "wb") as f:      # header + every batch's text (formatted while the batches ran)
                f.write((",".join(CSV_COLUMNS) + "\n").encode())
                for b in data:
                    f.write(b.csv_text())
        else:
            df = pd.DataFrame(list(data), columns=CSV_COLUMNS if not data else None)
            df.to_csv(self.output_csv, index=False)
        print(f"Saved tracking data to {self.output_csv}")

    def _cleanup(self):
        if getattr(self, "cap", None) is not None:
            self.cap.release()


def pinned_frames(shape):
    """A uint8 NumPy array in page-locked host memory (decode / read frames into it): `process_frames` then uploads batch
    k + 1 by DMA on a copy stream while batch k computes and its rows are built.  Pageable arrays work as before (their
    upload is a staged, synchronous copy)."""
    import torch
    return torch.empty(tuple(shape), dtype=torch.uint8, pin_memory=True).numpy()


def _upload_ahead(frames, batch, device=None, first_div=4):
    """Batches of `frames` as the tracker wants them.  NumPy frames with a GPU present: two device buffers and a copy
    stream; the upload of batch k + 1 is enqueued BEFORE batch k is handed out, so it runs under batch k's kernels, the
    download of its tables and the row building on the host (`_process_batch` blocks on its own results, which also means
    the buffer of batch k - 1 is free again by the time batch k + 1 is written into it).  From page-locked memory
    (`pinned_frames`) that upload is an asynchronous DMA at PCIe speed; from pageable memory the runtime stages it.
    The FIRST batch is a quarter of the others: nothing overlaps its upload, and the path as a whole is bound by the link
    (1 024 frames of 1280x1024 in batches of 128: 3 ms of 27 spent waiting for the first 128 frames)."""
    n = int(frames.shape[0])
    try:
        import torch
        gpu = isinstance(frames, np.ndarray) and frames.dtype == np.uint8 and torch.cuda.is_available() and n > batch
    except ImportError:
        gpu = False
    if not gpu:
        for s in range(0, n, batch):
            yield frames[s:s + batch]
        return
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else
                       (device if isinstance(device, int) else torch.device(device).index or 0))
    host = torch.from_numpy(np.ascontiguousarray(frames))
    bufs = [torch.empty((min(batch, n),) + tuple(host.shape[1:]), dtype=torch.uint8, device=dev) for _ in range(2)]
    copy = torch.cuda.Stream(device=dev)
    done = [None, None]

    first = max(batch // max(int(first_div), 1), 1) if n > 2 * batch else batch
    offs = [0] + list(range(first, n, batch))           # batch k = frames offs[k] .. offs[k + 1]
    offs.append(n)

    def start(k):
        s = offs[k]
        m = offs[k + 1] - s
        with torch.cuda.stream(copy):
            bufs[k & 1][:m].copy_(host[s:s + m], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy)
        done[k & 1] = (ev, m)

    nb = len(offs) - 1
    start(0)
    for k in range(nb):
        ev, m = done[k & 1]
        if k + 1 < nb:
            start(k + 1)
        torch.cuda.current_stream(dev).wait_event(ev)
        yield bufs[k & 1][:m]


# Names `tracking.py:7` imports (absent from the published module)
find_marker = MarkerTracker._find_markers
marker_center = MarkerTracker._marker_center


if __name__ == "__main__":
    config = {
        "video_path": "./video/test2.avi",
        "output_dir": "./results",
        "crop_ratios": (1 / 8, 1 / 8, 1 / 16, 0),
        "num_layers": 5,
        "min_marker_distance": 20,
    }
    MarkerTracker(config).process()
