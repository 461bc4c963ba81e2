"""Thin torch-facing wrapper over the C-ABI: tensors are buffers, all compute is in libvbs.so.

One `Engine` = one `vbs_handle` = one (device, frame size).  Every method enqueues on the current
torch stream of the engine's device and returns device tensors (no synchronisation except where a
host value is needed to raise the reference's exceptions).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    def __init__(self, height: int, width: int, max_markers: int = 512, max_batch: int = 16,
                 device: int | torch.device | None = None):
        if not torch.cuda.is_available():
            raise L.VbsError("no GPU visible: vbs_amd has no CPU path (the oracle lives in oracle/ and "
                             "is test-only)")
        self.lib = L.lib()
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else device.index or 0)
        self.H, self.W = int(height), int(width)
        self.max_markers, self.max_batch = int(max_markers), int(max_batch)
        self.pass_streams = 2                           # the library's default for VBS_OPT_PASS_STREAMS (see set_option)
        h = C.c_void_p()
        rc = self.lib.vbs_create(self.device.index, self.H, self.W, self.max_markers, self.max_batch,
                                 C.byref(h))
        self._h = h
        if rc != L.VBS_OK:
            msg = self.lib.vbs_last_error(h).decode() if h else "vbs_create failed"
            if h:
                self.lib.vbs_destroy(h)
                self._h = None
            raise (ValueError if rc == L.VBS_EINVAL else L.VbsError)(f"vbs_create: {msg} (status {rc})")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vbs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc == L.VBS_OK:
            return
        msg = self.lib.vbs_last_error(self._h).decode()
        raise (ValueError if rc == L.VBS_EINVAL else L.VbsError)(f"{what}: {msg} (status {rc})")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _frames(self, frames: torch.Tensor):
        """uint8 device tensor [N,H,W] or [N,H,W,3] (a crop view is fine) -> (ptr, n, ch, strides)."""
        if frames.dtype != torch.uint8 or frames.device != self.device:
            raise ValueError("frames must be a uint8 tensor on the engine's device")
        if frames.dim() == 2 or (frames.dim() == 3 and frames.shape[-1] == 3 and frames.shape[0] == self.H
                                 and frames.shape[1] == self.W):
            frames = frames.unsqueeze(0)
        ch = 1
        if frames.dim() == 4:
            ch = frames.shape[3]
            if ch not in (1, 3) or frames.stride(3) != 1 or frames.stride(2) != ch:
                raise ValueError("frames must be [N,H,W] or channel-last [N,H,W,3] with unit pixel stride")
        elif frames.dim() != 3 or frames.stride(2) != 1:
            raise ValueError("frames must be [N,H,W] or [N,H,W,3] with unit pixel stride")
        if frames.shape[1] != self.H or frames.shape[2] != self.W:
            raise ValueError(f"engine was built for {self.H}x{self.W}, got {tuple(frames.shape[1:3])}")
        return frames, frames.shape[0], ch, frames.stride(0), frames.stride(1)

    # ---- a2 / f3 ----------------------------------------------------------------------------
    def set_undistort(self, K=None, dist=None):
        """Enable (K, dist given) or disable (K None) frame undistortion; returns the new camera matrix [3,3]."""
        if K is None:
            self._check(self.lib.vbs_set_undistort(self._h, None, None, 0, None, self._stream()), "vbs_set_undistort")
            return None
        Kd = np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(9))
        dd = np.ascontiguousarray(np.asarray([] if dist is None else dist, dtype=np.float64).ravel()[:5])
        newK = np.zeros(9, dtype=np.float64)
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_set_undistort(self._h, Kd.ctypes.data_as(C.c_void_p),
                                                   dd.ctypes.data_as(C.c_void_p) if dd.size else None, int(dd.size),
                                                   newK.ctypes.data_as(C.c_void_p), self._stream()), "vbs_set_undistort")
        return newK.reshape(3, 3)

    def undistort_frames(self, frames):
        frames, n, ch, sn, sr = self._frames(frames)
        shape = (n, self.H, self.W) + ((ch,) if frames.dim() == 4 else ())
        out = torch.empty(shape, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_undistort_frames(self._h, _ptr(frames), n, ch, sn, sr, _ptr(out), self._stream()),
                        "vbs_undistort_frames")
        return out

    # ---- a3-a8 -------------------------------------------------------------------------------
    def bgr2gray(self, frames):
        """cv2.cvtColor(BGR2GRAY) of [N,H,W,3] frames -> uint8 [N,H,W] (coefficient set: OPT_GRAY_COEFFS)."""
        frames, n, ch, sn, sr = self._frames(frames)
        if ch != 3:
            raise ValueError("bgr2gray needs 3-channel frames")
        out = torch.empty((n, self.H, self.W), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_bgr2gray(self._h, _ptr(frames), n, sn, sr, _ptr(out), self._stream()), "vbs_bgr2gray")
        return out

    def find_markers(self, frames):
        frames, n, ch, sn, sr = self._frames(frames)
        mask = torch.empty((n, self.H, self.W), dtype=torch.uint8, device=self.device)
        area = torch.empty_like(mask)
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_find_markers(self._h, _ptr(frames), n, ch, sn, sr, _ptr(mask), _ptr(area),
                                                  self._stream()), "vbs_find_markers")
        return mask, area

    def ncc_map(self, frames):
        frames, n, ch, sn, sr = self._frames(frames)
        out = torch.empty((n, self.H, self.W), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_ncc_map(self._h, _ptr(frames), n, ch, sn, sr, _ptr(out), self._stream()),
                        "vbs_ncc_map")
        return out

    def normxcorr2(self, area_mask, want_mask=False):
        """NCC of a two-valued uint8 area_mask [n,H,W] with the branch template -> float64 map."""
        if area_mask.dim() == 2:
            area_mask = area_mask.unsqueeze(0)
        if area_mask.dtype != torch.uint8 or area_mask.device != self.device or not area_mask.is_contiguous() \
                or tuple(area_mask.shape[1:]) != (self.H, self.W):
            raise ValueError("area_mask must be a contiguous uint8 [n,H,W] tensor on the engine's device")
        n = area_mask.shape[0]
        out = torch.empty((n, self.H, self.W), dtype=torch.float64, device=self.device)
        mask = torch.empty((n, self.H, self.W), dtype=torch.uint8, device=self.device) if want_mask else None
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_normxcorr2(self._h, _ptr(area_mask), n, _ptr(out), _ptr(mask),
                                                self._stream()), "vbs_normxcorr2")
        return (out, mask) if want_mask else out

    def set_option(self, option: int, value: int):
        """`vbs_set_option`: L.OPT_GRAY_COEFFS (15 | 14), L.OPT_GRAY_SIDE_STREAM (0 | 1), test hooks L.OPT_FORCE_SEQ_MATCH,
        L.OPT_NCC_MARGIN (units of 1e-6), L.OPT_STAGE_IMPL / L.OPT_BLUR_IMPL (0 | 1), L.OPT_PASS_STREAMS (1 | 2),
        L.OPT_LATENCY_FRAMES (0 .. 32: passes of at most that many frames take the several-workgroups-per-frame labelling kernel)."""
        self._check(self.lib.vbs_set_option(self._h, int(option), int(value)), "vbs_set_option")
        if int(option) == L.OPT_PASS_STREAMS:
            self.pass_streams = int(value)

    def profile(self, enable: bool):
        self._check(self.lib.vbs_profile(self._h, 1 if enable else 0), "vbs_profile")

    def profile_read(self):
        """{kernel: (launches, total_ms)} from the HIP events recorded since profile(True)."""
        buf = C.create_string_buffer(8192)
        self._check(self.lib.vbs_profile_read(self._h, buf, len(buf)), "vbs_profile_read")
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            out[name] = (int(cnt), float(ms))
        return out

    def frame_stats(self, n):
        out = np.zeros((n, 8), dtype=np.uint32)
        self._check(self.lib.vbs_frame_stats(self._h, out.ctypes.data_as(C.c_void_p), n), "vbs_frame_stats")
        return out

    def stage_tables(self, n):
        """Host copies of the labelling kernels' per-component tables for the first n frames of the last internal pass
        (`vbs_stage_tables`, diagnostic): dict of ncomp [n,2], band_sums [n,M,4], area_first [n,M], area_sums [n,M,16],
        probe [n,M,4], slow [n]."""
        M = self.max_markers
        t = {"ncomp": np.zeros((n, 2), np.uint32), "band_sums": np.zeros((n, M, 4), np.uint64),
             "area_first": np.zeros((n, M), np.uint32), "area_sums": np.zeros((n, M, 16), np.int64),
             "probe": np.zeros((n, M, 4), np.uint16), "slow": np.zeros((n,), np.uint32)}
        self._check(self.lib.vbs_stage_tables(self._h, n, *(t[k].ctypes.data_as(C.c_void_p) for k in
                                                            ("ncomp", "band_sums", "area_first", "area_sums", "probe", "slow"))),
                    "vbs_stage_tables")
        return t

    def ncc_counters(self, reset=False):
        """{ambiguous, exact, frames} over every detection pass since the last reset (`vbs_ncc_counters`)."""
        out = np.zeros(3, dtype=np.uint64)
        self._check(self.lib.vbs_ncc_counters(self._h, out.ctypes.data_as(C.c_void_p), 1 if reset else 0),
                    "vbs_ncc_counters")
        return {"ambiguous": int(out[0]), "exact": int(out[1]), "frames": int(out[2])}

    # ---- a9-a13 ------------------------------------------------------------------------------
    def marker_center(self, mask, area_mask):
        for t in (mask, area_mask):
            if t.dtype != torch.uint8 or t.device != self.device or not t.is_contiguous():
                raise ValueError("mask / area_mask must be contiguous uint8 tensors on the engine's device")
        if mask.dim() == 2:
            mask, area_mask = mask.unsqueeze(0), area_mask.unsqueeze(0)
        n = mask.shape[0]
        if tuple(mask.shape[1:]) != (self.H, self.W) or mask.shape != area_mask.shape:
            raise ValueError("mask shape mismatch")
        det = torch.zeros((n, self.max_markers, L.DET_COLS), dtype=torch.float64, device=self.device)
        counts = torch.zeros((n,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_marker_center(self._h, _ptr(mask), _ptr(area_mask), n, _ptr(det),
                                                   _ptr(counts), self._stream()), "vbs_marker_center")
        return det, counts

    # ---- a15 ---------------------------------------------------------------------------------
    def track(self, det, counts, ref_xy, min_dist=20.0):
        ref = torch.as_tensor(ref_xy, dtype=torch.float64, device=self.device).contiguous().reshape(-1, 2)
        if det.dtype != torch.float64 or not det.is_contiguous() or det.device != self.device or det.dim() != 3 \
                or tuple(det.shape[1:]) != (self.max_markers, L.DET_COLS):
            raise ValueError(f"det must be a contiguous float64 tensor [n, {self.max_markers}, {L.DET_COLS}] on the "
                             "engine's device")
        if counts.dtype != torch.int32 or counts.device != self.device or not counts.is_contiguous() \
                or counts.numel() != det.shape[0]:
            raise ValueError("counts must be a contiguous int32 tensor [n] on the engine's device")
        n, m = det.shape[0], ref.shape[0]
        table = torch.empty((n, m, L.TABLE_COLS), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_track(self._h, _ptr(det), _ptr(counts), n, _ptr(ref), m, float(min_dist),
                                           _ptr(table), self._stream()), "vbs_track")
        return table

    # ---- a19-a20 -----------------------------------------------------------------------------
    def solve3d(self, table, cam: L.Camera, min_marker_size_px=5.0):
        n, m = table.shape[0], table.shape[1]
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_solve3d(self._h, _ptr(table), n, m, C.byref(cam), float(min_marker_size_px),
                                             self._stream()), "vbs_solve3d")
        return table

    # ---- fused ---------------------------------------------------------------------------------
    def track_to_3d(self, frames, ref_xy=None, min_dist=20.0, cam: L.Camera | None = None,
                    min_marker_size_px=5.0, want_det=False):
        frames, n, ch, sn, sr = self._frames(frames)
        table = ref = None
        m = 0
        if ref_xy is not None:
            ref = torch.as_tensor(ref_xy, dtype=torch.float64, device=self.device).contiguous().reshape(-1, 2)
            m = ref.shape[0]
            table = torch.empty((n, m, L.TABLE_COLS), dtype=torch.float32, device=self.device)
        det = torch.zeros((n, self.max_markers, L.DET_COLS), dtype=torch.float64,
                          device=self.device) if want_det else None
        counts = torch.zeros((n,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_track_to_3d(
                self._h, _ptr(frames), n, ch, sn, sr, _ptr(ref), m, float(min_dist),
                C.byref(cam) if cam is not None else None, float(min_marker_size_px), _ptr(table), _ptr(det),
                _ptr(counts), self._stream()), "vbs_track_to_3d")
        return table, det, counts

    # ---- a21, f1 -------------------------------------------------------------------------------
    def displacement(self, table, warmup_frames=100, min_marker_size_px=5.0, max_displacement=50.0,
                     frame_range=None):
        """Last-seen displacement of `table` [n,m,10]; `frame_range=(a, b)` emits only frames [a, b)."""
        table = table.contiguous()
        n, m = table.shape[0], table.shape[1]
        a, b = (0, n) if frame_range is None else (int(frame_range[0]), int(frame_range[1]))
        if not (0 <= a <= b <= n):
            raise ValueError(f"frame_range {frame_range} outside the table's {n} frames")
        disp = torch.empty((b - a, m, L.DISP_COLS), dtype=torch.float32, device=self.device)
        if a == b:                                       # (a rank whose shard is empty: nothing to emit, no null pointer to pass)
            return disp
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_displacement_range(self._h, _ptr(table), n, m, int(warmup_frames),
                                                        float(min_marker_size_px), float(max_displacement), a, b,
                                                        _ptr(disp), self._stream()), "vbs_displacement")
        return disp

    def plane_fit(self, table):
        table = table.contiguous()
        n, m = table.shape[0], table.shape[1]
        plane = torch.empty((n, L.PLANE_COLS), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_plane_fit(self._h, _ptr(table), n, m, _ptr(plane), self._stream()),
                        "vbs_plane_fit")
        return plane


    def deviation_plane(self, vert_start, vert_end, tilt_start, tilt_end, ref_xyz, mode="plane", scale=1.0):
        """Deviation field between a tilted and a vertical loading and the plane through its end points
        (`ForceDistribution.py:168-208,218-243`): four table rows [M,10] (one frame each), reference positions [M,3];
        returns (deviation [M,4] = (common, dX, dY, dZ), out [9] = (n, a, b, c, tilt_deg, mean k dX, k dY, k dZ, mean |d|))."""
        if mode not in ("plane", "shell"):
            raise ValueError("mode must be 'plane' or 'shell'")
        rows = [t.to(device=self.device, dtype=torch.float32).contiguous() for t in (vert_start, vert_end, tilt_start, tilt_end)]
        m = rows[0].shape[0]
        if any(tuple(t.shape) != (m, L.TABLE_COLS) for t in rows):
            raise ValueError(f"table rows must be [M, {L.TABLE_COLS}]")
        ref = torch.as_tensor(np.asarray(ref_xyz, dtype=np.float32).reshape(-1, 3), device=self.device).contiguous()
        if ref.shape[0] != m:
            raise ValueError("ref_xyz must hold one position per table slot")
        dev = torch.empty((m, 4), dtype=torch.float32, device=self.device)
        out = torch.empty((L.DEVPLANE_COLS,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_deviation_plane(self._h, *(_ptr(t) for t in rows), _ptr(ref), m,
                                                     1 if mode == "shell" else 0, float(scale), _ptr(dev), _ptr(out),
                                                     self._stream()), "vbs_deviation_plane")
        return dev, out

    # ---- a14 / f4 ------------------------------------------------------------------------------
    def assign_ids(self, det, counts, num_layers=5, id_mode="as_written"):
        """Frame-0 identities on the device: (ids int32 [M,2], ref_xy float64 [M,2]) as device tensors, in the
        reference dict's order.  `det` / `counts` are frame 0's rows of `marker_center` / `track_to_3d(want_det=True)`."""
        if id_mode not in ("as_written", "full"):
            raise ValueError(f"id_mode must be 'as_written' or 'full', got {id_mode!r}")
        det0 = (det[0] if det.dim() == 3 else det).contiguous()
        cnt0 = counts.reshape(-1)[:1].contiguous()
        cap = det0.shape[0] + 1
        ids = torch.zeros((cap, 2), dtype=torch.int32, device=self.device)
        xy = torch.zeros((cap, 2), dtype=torch.float64, device=self.device)
        m = torch.zeros((1,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.vbs_assign_ids(self._h, _ptr(det0), _ptr(cnt0), int(num_layers),
                                                1 if id_mode == "full" else 0, _ptr(ids), _ptr(xy), cap, _ptr(m),
                                                self._stream()), "vbs_assign_ids")
        mm = int(m.item())
        if mm == -1:
            raise ValueError("No markers detected in first frame!")
        if mm <= -1000:
            raise L.VbsError(f"device status {mm // 1000} in frame 0")
        if mm < 0:
            raise L.VbsError(f"vbs_assign_ids: device status {mm}")
        return ids[:mm], xy[:mm]


def _dev_f64(x, dev, cols):
    if isinstance(x, torch.Tensor):
        return x.to(device=dev, dtype=torch.float64).reshape(-1, cols).contiguous()
    return torch.as_tensor(np.asarray(x, dtype=np.float64).reshape(-1, cols), device=dev).contiguous()


def undistort_points(points, cam: L.Camera, device=None):
    """float64 [n,2] -> [n,2] on the GPU (`MarkerAnalysis._undistort_points`)."""
    if not torch.cuda.is_available():
        raise L.VbsError("no GPU visible: vbs_amd has no CPU path")
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    p = _dev_f64(points, dev, 2)
    out = torch.empty_like(p)
    rc = L.lib().vbs_undistort_points(dev.index, _ptr(p), p.shape[0], C.byref(cam), _ptr(out),
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != L.VBS_OK:
        raise L.VbsError(f"vbs_undistort_points failed ({rc})")
    return out


def normxcorr2_general(template, image, mode="same", device=None):
    """`_normxcorr2` for arbitrary operands: float64 map of the mode's size on the GPU (`vbs_normxcorr2_general`)."""
    if not torch.cuda.is_available():
        raise L.VbsError("no GPU visible: vbs_amd has no CPU path")
    modes = {"full": 0, "same": 1, "valid": 2}
    if mode not in modes:
        raise ValueError("acceptable mode flags are 'valid', 'same', or 'full'")        # scipy's message
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    t = torch.as_tensor(np.asarray(template, dtype=np.float64), device=dev).contiguous()
    im = torch.as_tensor(np.asarray(image, dtype=np.float64), device=dev).contiguous()
    if t.dim() != 2 or im.dim() != 2:
        raise ValueError("template and image must be 2-D")
    (th, tw), (h, w) = t.shape, im.shape
    if tw > 256:
        raise NotImplementedError("vbs_normxcorr2_general holds template rows of at most 256 samples")
    oh, ow = {0: (h + th - 1, w + tw - 1), 1: (h, w), 2: (h - th + 1, w - tw + 1)}[modes[mode]]
    if oh < 1 or ow < 1:
        raise ValueError("For 'valid' mode, one must be at least as large as the other in every dimension")
    out = torch.empty((oh, ow), dtype=torch.float64, device=dev)
    rc = L.lib().vbs_normxcorr2_general(dev.index, _ptr(t), th, tw, _ptr(im), h, w, modes[mode], _ptr(out),
                                        C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != L.VBS_OK:
        raise L.VbsError(f"vbs_normxcorr2_general failed ({rc})")
    return out


def calculate_3d(uvd, cam: L.Camera, device=None):
    """float64 [n,3] (u, v, diameter_px) -> (xyz float64 [n,3], ok int32 [n]) on the GPU."""
    if not torch.cuda.is_available():
        raise L.VbsError("no GPU visible: vbs_amd has no CPU path")
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    p = _dev_f64(uvd, dev, 3)
    xyz = torch.empty_like(p)
    ok = torch.empty((p.shape[0],), dtype=torch.int32, device=dev)
    rc = L.lib().vbs_calculate_3d(dev.index, _ptr(p), p.shape[0], C.byref(cam), _ptr(xyz), _ptr(ok),
                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc == L.VBS_EINVAL:
        raise ValueError("Focal lengths must be positive")
    if rc != L.VBS_OK:
        raise L.VbsError(f"vbs_calculate_3d failed ({rc})")
    return xyz, ok


def displacement_f64(table64, warmup_frames=0, min_marker_size_px=0.0, max_displacement=50.0, device=None):
    """float64 table [n,m,10] (host or device) -> disp float64 [n,m,5] on the GPU (`vbs_displacement_f64`)."""
    if not torch.cuda.is_available():
        raise L.VbsError("no GPU visible: vbs_amd has no CPU path")
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    t = torch.as_tensor(table64, dtype=torch.float64, device=dev).contiguous()
    n, m = t.shape[0], t.shape[1]
    disp = torch.empty((n, m, L.DISP_COLS), dtype=torch.float64, device=dev)
    rc = L.lib().vbs_displacement_f64(dev.index, _ptr(t), n, m, int(warmup_frames), float(min_marker_size_px),
                                      float(max_displacement), _ptr(disp),
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != L.VBS_OK:
        raise L.VbsError(f"vbs_displacement_f64 failed ({rc})")
    return disp
