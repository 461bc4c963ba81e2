"""ctypes binding of csrc/libvbs.so (C-ABI: include/vbs.h).  No fallback: a missing library raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvbs.so")

VBS_OK, VBS_EINVAL, VBS_ECAPACITY, VBS_EHIP, VBS_ENOMEM, VBS_EINTERNAL = 0, -1, -2, -3, -4, -5
DET_COLS, TABLE_COLS, DISP_COLS, PLANE_COLS, DEVPLANE_COLS = 6, 10, 5, 5, 9
FLAG_TRACKED, FLAG_XYZ = 1, 2
OPT_GRAY_COEFFS, OPT_FORCE_SEQ_MATCH, OPT_GRAY_SIDE_STREAM, OPT_NCC_MARGIN, OPT_STAGE_IMPL, OPT_BLUR_IMPL, OPT_PASS_STREAMS, OPT_LATENCY_FRAMES = 1, 2, 3, 4, 5, 6, 7, 8

# every symbol include/vbs.h declares (tests check the export list against the header)
SYMBOLS = ("vbs_create", "vbs_destroy", "vbs_last_error", "vbs_version", "vbs_contour_lut",
           "vbs_gaussian_taps_q8", "vbs_ncc_template", "vbs_set_undistort", "vbs_undistort_frames", "vbs_find_markers", "vbs_ncc_map", "vbs_normxcorr2",
           "vbs_profile", "vbs_profile_read", "vbs_frame_stats", "vbs_undistort_points", "vbs_calculate_3d", "vbs_marker_center",
           "vbs_track", "vbs_solve3d", "vbs_track_to_3d", "vbs_displacement", "vbs_displacement_range", "vbs_displacement_f64",
           "vbs_plane_fit", "vbs_assign_ids", "vbs_set_option", "vbs_bgr2gray", "vbs_ncc_counters", "vbs_normxcorr2_general",
           "vbs_stage_tables", "vbs_deviation_plane", "vbs_format_csv", "vbs_mjpeg_probe", "vbs_mjpeg_entropy_batch",
           "vbs_mjpeg_reconstruct")


class Camera(C.Structure):
    """vbs_camera: float32 fields exactly as `MarkerAnalysis.load_parameters` stores them."""
    _fields_ = [("K", C.c_float * 9), ("dist", C.c_float * 5), ("R", C.c_float * 9),
                ("T", C.c_float * 3), ("marker_diameter_mm", C.c_float)]


class VbsError(RuntimeError):
    pass


def status_text(status: int, max_markers: int = 0) -> str:
    """What a negative per-frame status in `counts[]` means (include/vbs.h)."""
    if status == VBS_ECAPACITY:
        return ("the frame exceeds the device workspace (more than 30720 runs in a mask, more than "
                f"{max_markers or 'max_markers'} band components, or more than about 512 contours)")
    if status == VBS_EINTERNAL:
        return ("a kernel's internal hand-shake timed out: the frame's result is not to be trusted (bounded wait expired; "
                "see VBS_EINTERNAL in include/vbs.h)")
    return "unexpected device status"


_lib = None


def lib():
    """Load libvbs.so once.  Raises (never falls back) when the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VbsError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                       f"g.build()'` (hipcc, gfx950). There is no CPU fallback.")
    # torch first: it brings its own HIP runtime (libamdhip64 of the same soname), and the library must bind to THAT copy.
    # Loaded the other way round - this library before torch - the process holds two runtimes and the second one to
    # initialise fails (hipSetDevice: seen with `python __graft_entry__.py --smoke`, where build() loads the library first).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_double
    cam_p = C.POINTER(Camera)
    sig = {
        "vbs_create": (i32, [i32, i32, i32, i32, i32, C.POINTER(vp)]),
        "vbs_destroy": (i32, [vp]),
        "vbs_last_error": (C.c_char_p, [vp]),
        "vbs_version": (i32, []),
        "vbs_contour_lut": (i32, [vp]),
        "vbs_gaussian_taps_q8": (i32, [i32, f64, vp]),
        "vbs_ncc_template": (i32, [i32, f64, vp, vp]),
        "vbs_set_undistort": (i32, [vp, vp, vp, i32, vp, vp]),
        "vbs_undistort_frames": (i32, [vp, vp, i32, i32, i64, i64, vp, vp]),
        "vbs_find_markers": (i32, [vp, vp, i32, i32, i64, i64, vp, vp, vp]),
        "vbs_ncc_map": (i32, [vp, vp, i32, i32, i64, i64, vp, vp]),
        "vbs_normxcorr2": (i32, [vp, vp, i32, vp, vp, vp]),
        "vbs_profile": (i32, [vp, i32]),
        "vbs_profile_read": (i32, [vp, vp, i32]),
        "vbs_frame_stats": (i32, [vp, vp, i32]),
        "vbs_undistort_points": (i32, [i32, vp, i32, cam_p, vp, vp]),
        "vbs_calculate_3d": (i32, [i32, vp, i32, cam_p, vp, vp, vp]),
        "vbs_marker_center": (i32, [vp, vp, vp, i32, vp, vp, vp]),
        "vbs_track": (i32, [vp, vp, vp, i32, vp, i32, f64, vp, vp]),
        "vbs_solve3d": (i32, [vp, vp, i32, i32, cam_p, f64, vp]),
        "vbs_track_to_3d": (i32, [vp, vp, i32, i32, i64, i64, vp, i32, f64, cam_p, f64, vp, vp, vp, vp]),
        "vbs_displacement": (i32, [vp, vp, i32, i32, i32, f64, f64, vp, vp]),
        "vbs_displacement_range": (i32, [vp, vp, i32, i32, i32, f64, f64, i32, i32, vp, vp]),
        "vbs_displacement_f64": (i32, [i32, vp, i32, i32, i32, f64, f64, vp, vp]),
        "vbs_plane_fit": (i32, [vp, vp, i32, i32, vp, vp]),
        "vbs_assign_ids": (i32, [vp, vp, vp, i32, i32, vp, vp, i32, vp, vp]),
        "vbs_set_option": (i32, [vp, i32, i32]),
        "vbs_bgr2gray": (i32, [vp, vp, i32, i64, i64, vp, vp]),
        "vbs_ncc_counters": (i32, [vp, vp, i32]),
        "vbs_normxcorr2_general": (i32, [i32, vp, i32, i32, vp, i32, i32, i32, vp, vp]),
        "vbs_stage_tables": (i32, [vp, i32, vp, vp, vp, vp, vp, vp]),
        "vbs_deviation_plane": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f64, vp, vp, vp]),
        "vbs_format_csv": (i64, [vp, vp, vp, vp, i32, i64, vp, i64, i32]),
        "vbs_mjpeg_probe": (i32, [vp, i64, vp]),
        "vbs_mjpeg_entropy_batch": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32]),
        "vbs_mjpeg_reconstruct": (i32, [vp, vp, vp, vp, i32, vp, vp, vp, i64, i64, vp]),
    }
    for name in SYMBOLS:
        fn = getattr(L, name)            # AttributeError here = stale library
        fn.restype, fn.argtypes = sig[name]
    _lib = L
    return L


def make_camera(K, dist, R, T, marker_diameter_mm=2.0) -> Camera:
    import numpy as np
    cam = Camera()
    cam.K[:] = np.asarray(K, dtype=np.float32).reshape(9).tolist()
    d = np.zeros(5, dtype=np.float32)
    dd = np.asarray(dist, dtype=np.float32).ravel()[:5]
    d[:dd.size] = dd
    cam.dist[:] = d.tolist()
    cam.R[:] = np.asarray(R, dtype=np.float32).reshape(9).tolist()
    cam.T[:] = np.asarray(T, dtype=np.float32).reshape(3).tolist()
    cam.marker_diameter_mm = float(marker_diameter_mm)
    return cam
