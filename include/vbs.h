/* vbs.h — C-ABI of libvbs.so: MI355X (gfx950) marker tracking -> 3-D displacement.
 *
 * Drop-in boundary for the hot path of UPM-ROB-Lab/Vision-basedSensor.  The reference has no
 * FFI layer; its boundary is the Python call signatures of
 *   code/Marker_Tracking/marker_detection.py   (class MarkerTracker)
 *   code/Marker_Calibration/3d_reconstruction.py (class MarkerAnalysis)
 * and each entry point below names the reference function it replaces (file:line).  The Python
 * shims in `vision-basedsensor_amd/` bind these with ctypes and pass torch device pointers
 * (`tensor.data_ptr()`); nothing here depends on torch.
 *
 * Conventions
 *  - every pointer marked [dev] is device memory of the handle's GPU, owned by the caller;
 *    the library owns only the workspace inside `vbs_handle`.  No allocation on a hot call, with two stated
 *    exceptions that happen ONCE per handle and never under stream capture: the gray plane of 3-channel / undistorted
 *    input (first such call, or vbs_set_undistort), and the second pass workspace of a handle left at the default
 *    VBS_OPT_PASS_STREAMS (first call spanning several internal passes; vbs_set_option(h, VBS_OPT_PASS_STREAMS, 2)
 *    builds it ahead of time - do that before capturing calls into a graph);
 *  - calls are asynchronous on `stream` (a hipStream_t passed as void*, NULL = default stream);
 *    the caller synchronises before reading results;
 *  - return value: VBS_OK or a negative status; `vbs_last_error` gives the text.  Per-frame
 *    conditions found on the device (capacity overflow) are reported in `counts[i]` as a negative
 *    status, because the host cannot see them without a synchronisation;
 *  - one handle per (device, stream); a handle is not thread-safe; one process per GPU.
 *  - images: uint8, pixel (frame i, row y, col x, channel c) at
 *        base + i*stride_n + y*stride_row + x*channels + c        (bytes)
 *    so a crop (marker_detection.py:78-85) is a pointer offset plus the original strides.
 */
#ifndef VBS_H
#define VBS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VBS_OK          0
#define VBS_EINVAL     -1   /* bad argument (ValueError in the shim)                          */
#define VBS_ECAPACITY  -2   /* more runs / components in a frame than the handle was sized for */
#define VBS_EHIP       -3   /* a HIP runtime call failed                                      */
#define VBS_ENOMEM     -4   /* workspace allocation failed                                    */
#define VBS_EINTERNAL  -5   /* per-frame (in counts[i]): a kernel's internal hand-shake timed out - the frame's result is
                               not to be trusted (never seen in practice: the wait is bounded so that a logic error or a
                               lost workgroup cannot hang the GPU, and it reports here instead of continuing silently) */

#define VBS_DET_COLS    6   /* x, y, major_axis, minor_axis, angle, label(band component, 1-based) */
#define VBS_TABLE_COLS 10   /* flags, Cx, Cy, major, minor, angle, X, Y, Z, det_index            */
#define VBS_FLAG_TRACKED 1  /* table col 0 bit: the reference ID matched a detection (2-D row)    */
#define VBS_FLAG_XYZ     2  /* table col 0 bit: the 3-D solve succeeded                           */
#define VBS_DISP_COLS   5   /* flag, dX, dY, dZ, |d|                                              */
#define VBS_PLANE_COLS  5   /* n_used, a, b, c, tilt_deg                                          */
#define VBS_DEVPLANE_COLS 9 /* n_common, a, b, c, tilt_deg, mean k*dX, mean k*dY, mean k*dZ, mean |d| */

typedef struct vbs_handle vbs_handle;

/* Camera of MarkerAnalysis.load_parameters (3d_reconstruction.py:70-130): every field is the
 * float32 value the reference stores; arithmetic on them is float64 as in the reference. */
typedef struct vbs_camera {
    float K[9];      /* row-major camera matrix            (:87-91)   */
    float dist[5];   /* k1 k2 p1 p2 k3                     (:98-102)  */
    float R[9];      /* row-major R_world_to_cam           (:109-112) */
    float T[3];      /* T_world_to_cam                     (:120-124) */
    float marker_diameter_mm;   /* Config.marker_diameter_mm (:21)   */
} vbs_camera;

/* Workspace for frames of `height` x `width` (after cropping).  `max_markers` bounds the connected
 * components per mask per frame, `max_batch` the frames processed per internal pass (larger
 * batches are looped).  height <= 480 selects the reference's small-image parameter set
 * (marker_detection.py:117-126,170).  Throughput: the labelling kernel runs three frames per compute unit at a time, so
 * passes that are a multiple of 768 frames suit an MI355X best (bench.py: 1536; 290-295 k frames/s against 280 k at 512 and
 * 292 k at 1368 or 1640); results do not depend on the pass size. */
int vbs_create(int device, int height, int width, int max_markers, int max_batch, vbs_handle** out);
int vbs_destroy(vbs_handle* h);
const char* vbs_last_error(const vbs_handle* h);
int vbs_version(void);
/* Handle options.  VBS_OPT_GRAY_COEFFS: fixed-point coefficient set of cv2.cvtColor(BGR2GRAY) (marker_detection.py:114)
 * - 15 (default): OpenCV 4.x, (3735 B + 19235 G + 9798 R + 2^14) >> 15;  14: OpenCV <= 3.4.1, (1868 B + 9617 G +
 * 4899 R + 2^13) >> 14.  The two agree wherever B = G = R.  VBS_OPT_FORCE_SEQ_MATCH (test hook): 1 makes
 * vbs_marker_center replay the reference's sequential contour <-> centre matching (:203-243) instead of the parallel
 * form that is proven equal to it.  VBS_OPT_GRAY_SIDE_STREAM (tuning, results identical): 1 converts the
 * BGR frames of internal pass k + 1 on the handle's own stream while pass k computes, 0 (default) converts in line.
 * VBS_OPT_NCC_MARGIN (test hook, results identical): relative margin of the NCC's float32 filter in units of 1e-6
 * (never below the 20 the error bound needs); a wide margin sends thousands of pixels per frame through the queued
 * float64 re-evaluation.  VBS_OPT_STAGE_IMPL (test hook / fallback, results identical): 0 (default) runs band / open /
 * labelling / sums of `_marker_center` (:170-196) in the fused kernel (k_stage.hip) where the frame geometry allows it,
 * 1 always runs the separate kernels (k_morph + k_ccl) that other geometries take, 2 labels EVERY frame with the general
 * kernel (k_morph + k_label: what a frame with holes / beyond the fast path's tables takes), 3 is 0 with the fused kernel
 * at 768 threads per frame always, 4 is 0 with its 256-thread instance wherever the geometry allows (by itself the kernel
 * takes 256 threads for small frames and for passes of >= 512 large ones, 768 otherwise).  VBS_OPT_BLUR_IMPL (test hook /
 * fallback, results identical): 0 (default) runs the two GaussianBlurs (:118-129) on 16-column strips (k_blur16) where the
 * frame allows it (large branch, width >= 240 and a multiple of 8, rows that load as aligned dwords), 1 always runs the 32-column
 * kernel (k_blur_mfma) that every other frame takes.  VBS_OPT_PASS_STREAMS (tuning, results identical): 2 (default) lets
 * vbs_track_to_3d run the odd internal passes of a call (gray or BGR frames converted in line) with a SECOND WORKSPACE on
 * the handle's own stream (forked from and joined to the caller's stream by events), so that the tail of one pass's kernels
 * overlaps the next pass; 1 runs every pass on the caller's stream.  The second workspace is a second copy of every
 * per-pass buffer (twice the device memory of the handle).  Setting the option to 2 EXPLICITLY builds it at once (a set-up
 * call: allocations, copies, one device synchronisation; VBS_ENOMEM / VBS_EHIP if it cannot be built).  A handle left at
 * the default builds it at its first call that spans several passes; if that call's stream is being captured, or the
 * workspace cannot be built, the call runs every pass on the caller's stream instead.  (With vbs_profile on, passes run on
 * one stream: the per-kernel event timings would otherwise overlap.)  VBS_OPT_LATENCY_FRAMES (tuning, results identical):
 * an internal pass of at most this many frames (default 24, at most 32: 1 frame 128 against 264 us, 8: 185 / 324, 16: 269 / 356, 24: 346 / 390, 32: 424 / 406; the reference calls process() with ONE,
 * marker_detection.py:434-453) labels every frame with several workgroups (k_stage_lat) instead of one (k_stage); 0 = never. */
#define VBS_OPT_GRAY_COEFFS      1
#define VBS_OPT_FORCE_SEQ_MATCH  2
#define VBS_OPT_GRAY_SIDE_STREAM 3
#define VBS_OPT_NCC_MARGIN       4
#define VBS_OPT_STAGE_IMPL       5
#define VBS_OPT_BLUR_IMPL        6
#define VBS_OPT_PASS_STREAMS     7
#define VBS_OPT_LATENCY_FRAMES   8
int vbs_set_option(vbs_handle* h, int option, int value);
/* Motion-JPEG front end (the reference reads its AVI through cv2.VideoCapture, marker_detection.py:50-76).
 * vbs_mjpeg_probe: headers of one JPEG frame -> info[8] = {width, height, components, luma h, luma v, restart interval,
 * int16 coefficients per frame, plane bytes per frame}; VBS_EINVAL for a stream this decoder does not take (progressive,
 * arithmetic, 12-bit, sampling other than 4:4:4 / 4:2:2 / 4:2:0 / gray): the caller then decodes with its own reader.  A frame
 * without DHT segments (camera MJPG) is decoded with the standard tables of ITU-T T.81 Annex K.3.
 * vbs_mjpeg_entropy_batch: Huffman-decodes n frames (buf + offs[i], sizes[i]; all of the probed geometry) on `threads` host
 * threads into the compact form the device half reads (HOST memory, e.g. page-locked): tab [n][info[6] / 64] one word per 8x8
 * block (component after component, row-major over the padded block grid) = (first word of the block in ent, relative to the
 * frame's) << 7 | count; count <= 32 entry words (natural-order position << 16 | quantised value as uint16), or 127 = a dense
 * block of 64 int16; ent holds n * info[6] / 2 words, thread t packs its frames from word (its first frame) * info[6] / 2 on
 * and reports (first word, words used) in regions[2 t], regions[2 t + 1] (2 * threads int64): only those spans, tab,
 * frame_base [n] (a frame's first word) and qt [n][3][64] need to reach the device - typically a tenth of the pixels' bytes.
 * status[i] per frame; returns the number of frames that failed.  vbs_mjpeg_reconstruct: DEVICE copies of those arrays ->
 * BGR frames `out` (byte strides out_frame / out_row), planes = device scratch of n * info[7] bytes; dequantisation, the
 * 8x8 "islow" inverse DCT, "fancy" chroma upsampling and the YCbCr -> RGB tables as published in libjpeg, so that the
 * frames equal a libjpeg(-turbo) decode bit for bit; asynchronous on `stream`. */
int vbs_mjpeg_probe(const uint8_t* jpeg, int64_t size, int32_t* info);
int vbs_mjpeg_entropy_batch(const uint8_t* buf, const int64_t* offs, const int32_t* sizes, int n, const int32_t* info,
                            uint32_t* ent, uint32_t* tab, int64_t* frame_base, int64_t* regions, uint16_t* qt, int32_t* status,
                            int threads);
int vbs_mjpeg_reconstruct(const uint32_t* ent, const uint32_t* tab, const int64_t* frame_base, const uint16_t* qt, int n,
                          const int32_t* info, uint8_t* planes, uint8_t* out, int64_t out_frame, int64_t out_row, void* stream);
/* host-only helper: the 256-entry table that classifies a border pixel's 8-neighbourhood into the
 * number of CHAIN_APPROX_SIMPLE vertices it contributes (bit d of the index = neighbour in chain
 * direction d is foreground; 0=E,1=NE,2=N,...,7=SE). */
int vbs_contour_lut(uint8_t out[256]);
/* host-only helper: the body (no header line) of the tracker's CSV - `pandas.DataFrame(rows).to_csv(index=False)`,
 * marker_detection.py:464-468 - for n rows of three int64 columns (frameno, row, col) and nf float64 columns (Ox, Oy, Cx,
 * Cy, major_axis, minor_axis, angle), byte for byte: floats as Python's repr writes them (shortest digits that round-trip,
 * exponent form below 1e-4 and from 1e16), NaN as an empty cell.  Formatted by `threads` host threads into buf.
 * Returns the bytes written; a negative value -c when cap < c = the capacity that is always enough (n * (65 + 26 nf)). */
int64_t vbs_format_csv(const int64_t* frameno, const int64_t* row, const int64_t* col, const double* const* fcols, int nf,
                       int64_t n, char* buf, int64_t cap, int threads);
/* host-only helpers exposing the constant tables the kernels use, so they can be checked without a
 * GPU: the fixed-point GaussianBlur taps (sum 256; marker_detection.py:118-124 via cv2) and the 1-D
 * factor g of the NCC template with stats = {mean(t), sum((t-mean)^2), l*l, 0.1^2}
 * (_gkern :138-143, _normxcorr2 :152,162). */
int vbs_gaussian_taps_q8(int ksize, double sigma, int32_t* out);
int vbs_ncc_template(int l, double sigma, double* g, double* stats);

/* MarkerTracker._undistort_frame (marker_detection.py:93-109), optional (`calibration_params` in the config):
 * getOptimalNewCameraMatrix(K, D, (w,h), 0) + initUndistortRectifyMap(CV_16SC2) are evaluated ONCE here (the
 * reference rebuilds them per frame); afterwards vbs_find_markers / vbs_track_to_3d / vbs_ncc_map first remap every
 * frame (INTER_LINEAR, fixed point, constant border 0) exactly as `_preprocess_frame` does (:88-89).
 * K9 row-major float64 camera matrix, dist = up to 5 coefficients k1 k2 p1 p2 k3; newK9 (may be NULL) receives the
 * new camera matrix.  K9 == NULL switches undistortion off again. */
int vbs_set_undistort(vbs_handle* h, const double* K9, const double* dist, int ndist, double* newK9, void* stream);
/* The remap alone: frames [dev] uint8 (same addressing as below) -> out [dev] uint8 [n,h,w,channels] dense. */
int vbs_undistort_frames(vbs_handle* h, const uint8_t* frames, int n, int channels, int64_t stride_n,
                         int64_t stride_row, uint8_t* out, void* stream);

/* cv2.cvtColor(frame, COLOR_BGR2GRAY) (marker_detection.py:114) on its own: frames [dev] uint8 BGR (3 channels,
 * addressing as above) -> gray [dev] uint8 [n,h,w] dense, with the handle's coefficient set (VBS_OPT_GRAY_COEFFS).
 * (Stage entry for parity tests; on the hot path the same conversion kernel runs in front of the blur.) */
int vbs_bgr2gray(vbs_handle* h, const uint8_t* frames, int n, int64_t stride_n, int64_t stride_row, uint8_t* gray,
                 void* stream);

/* MarkerTracker._find_markers (marker_detection.py:112-135): BGR2GRAY -> 2x GaussianBlur -> uint8
 * difference +15 (mod 256) -> inRange -> area_mask {0,255}; NCC with the Gaussian template
 * (_gkern :138, _normxcorr2 :146) -> mask {0,1} = ncc > 0.1.
 * frames [dev] uint8, channels 1 (gray) or 3 (BGR); mask / area_mask [dev] uint8 [n,h,w] dense
 * (either may be NULL). */
int vbs_find_markers(vbs_handle* h, const uint8_t* frames, int n, int channels, int64_t stride_n,
                     int64_t stride_row, uint8_t* mask, uint8_t* area_mask, void* stream);

/* MarkerTracker._normxcorr2 (marker_detection.py:146-164) for the pipeline's own operands: the
 * float64 correlation map of area_mask with the template, ncc [dev] float64 [n,h,w] dense.
 * (Diagnostic / parity entry: the hot path never materialises this map.) */
int vbs_ncc_map(vbs_handle* h, const uint8_t* frames, int n, int channels, int64_t stride_n,
                int64_t stride_row, double* ncc, void* stream);
/* The same stage from a given two-valued area_mask [dev] uint8 [n,h,w] (the reference call
 * `_normxcorr2(template, area_mask)` :132 with the template of the handle's branch): ncc [dev]
 * float64 [n,h,w] and / or mask [dev] uint8 [n,h,w] = ncc > 0.1 (:133); either may be NULL. */
int vbs_normxcorr2(vbs_handle* h, const uint8_t* area_mask, int n, double* ncc, uint8_t* mask,
                   void* stream);
/* MarkerTracker._normxcorr2(template, image, mode) (marker_detection.py:146-164) for ARBITRARY operands, no handle needed:
 * tmpl [dev] float64 [th,tw] (tw <= 256), image [dev] float64 [h,w], mode 0 'full' | 1 'same' | 2 'valid' (the window
 * scipy.signal.fftconvolve returns), out [dev] float64 of that mode's size ([h+th-1,w+tw-1] | [h,w] | [h-th+1,w-tw+1]).
 * Direct float64 evaluation; agrees with the reference's FFT route to its rounding.  Not on the hot path. */
int vbs_normxcorr2_general(int device, const double* tmpl, int th, int tw, const double* image, int h, int w, int mode,
                           double* out, void* stream);
/* Live kernel timing: while enabled, every kernel launch of this handle is bracketed by a HIP event
 * pair on the launch stream.  vbs_profile(h, on) clears the records; vbs_profile_read synchronises
 * and writes one text line per kernel: "<name> <launches> <total_ms>". */
int vbs_profile(vbs_handle* h, int enable);
int vbs_profile_read(vbs_handle* h, char* buf, int cap);
/* host copy of the per-frame counters of the LAST internal pass (of whichever workspace ran it): out[i*8 + {0: area_mask popcount,
 * 1: NCC pixels within 1e-9 (relative) of the 0.1 threshold, 2: status, 3: NCC pixels re-evaluated in float64,
 * 4: holes LEFT in the opened area mask (components - Euler number; holes are filled before contouring, like
 * cv2.findContours(RETR_EXTERNAL) ignores them, so this is 0 unless the fill pass ran out of capacity), 5 / 6: connected
 * components of the band / opened mask, 7: holes that were filled}] (synchronises). */
int vbs_frame_stats(vbs_handle* h, uint32_t* out, int n);
/* Diagnostic / parity entry: host copies of the per-component tables the labelling kernels left for the first n frames of
 * the LAST internal pass (synchronises; any pointer may be NULL): ncomp [n][2] components of the band / opened mask,
 * band_sums [n][max_markers][4] (count, sum x, sum y, spare), area_first [n][max_markers] first pixel (y * w + x) of every
 * opened component, area_sums [n][max_markers][16] contour-vertex moments about it, probe [n][max_markers][4] component ids
 * of the 2x2 pixel cell around every band centroid (0xFFFF = background), slow [n] non-zero = the general kernel redid the frame (the value says why: +16 = in the opened-mask half; 1 slots, 2 components, 3 mailbox, 4 holes, 5 vertex multiplicity, 6 segments per tile, 7 records, 8 queued unions).
 * Entries past a frame's component counts are unspecified. */
int vbs_stage_tables(vbs_handle* h, int n, uint32_t* ncomp, uint64_t* band_sums, uint32_t* area_first, int64_t* area_sums,
                     uint16_t* probe, uint32_t* slow);
/* Running totals over EVERY internal pass of the detection stage since the last reset (vbs_frame_stats only sees the
 * last pass): out = {NCC pixels inside the ambiguity band of the 0.1 threshold (`:133`; 0 = every decision equals the
 * float64 one), NCC pixels re-evaluated in float64, frames}.  Synchronises. */
int vbs_ncc_counters(vbs_handle* h, uint64_t out[3], int reset);

/* MarkerAnalysis._undistort_points (3d_reconstruction.py:185-193) and _calculate_3d_position
 * (:195-238) on float64 points, no handle needed: pts/out [dev] float64 [n,2]; uvd [dev] float64
 * [n,3] = (u, v, diameter_px); xyz [dev] float64 [n,3]; ok [dev] int32 [n] (0 where the reference
 * raises ValueError). */
int vbs_undistort_points(int device, const double* pts, int n, const vbs_camera* cam, double* out,
                         void* stream);
int vbs_calculate_3d(int device, const double* uvd, int n, const vbs_camera* cam, double* xyz,
                     int32_t* ok, void* stream);

/* MarkerTracker._marker_center (marker_detection.py:166-249): band = mask AND NOT erode(mask)
 * (maximum/minimum_filter :171-174) -> 4-connected labels (:176) -> centroids (:181); 5x5 open
 * (:195) -> external contours (:196) -> fitEllipse (:208) -> contour/centre matching (:222-243).
 * mask, area_mask [dev] uint8 [n,h,w] dense, two-valued (0 / non-zero).
 * det [dev] float64 [n,max_markers,VBS_DET_COLS], rows in the reference's output order (centroids are
 * integer sums / count in float64, bit-identical to ndimage.center_of_mass; axes and angle are the
 * float32 values cv2.fitEllipse would return, widened);
 * counts [dev] int32 [n] = number of rows, or a negative status for that frame. */
int vbs_marker_center(vbs_handle* h, const uint8_t* mask, const uint8_t* area_mask, int n,
                      double* det, int32_t* counts, void* stream);

/* MarkerTracker._track_markers (marker_detection.py:349-396): per reference ID the nearest
 * detection (first on ties), dropped when farther than min_dist.  ref_xy [dev] float64 [m_ref,2]
 * = (Ox, Oy) in reference-dict order; table [dev] float32 [n,m_ref,VBS_TABLE_COLS]; XYZ columns
 * are left 0 and VBS_FLAG_XYZ clear. */
int vbs_track(vbs_handle* h, const double* det, const int32_t* counts, int n, const double* ref_xy,
              int m_ref, double min_dist, float* table, void* stream);

/* MarkerAnalysis._undistort_points + _calculate_3d_position (3d_reconstruction.py:185-238) on the
 * tracked rows of `table` (in place: fills X,Y,Z and VBS_FLAG_XYZ).  Rows with
 * major_axis < min_marker_size_px are skipped (load_marker_data :172-176). */
int vbs_solve3d(vbs_handle* h, float* table, int n, int m_ref, const vbs_camera* cam,
                double min_marker_size_px, void* stream);

/* Fused frames -> table: everything above in one call, intermediates kept on the device in
 * float64 / bit-packed form (no uint8 masks are written).  counts may be NULL. */
int vbs_track_to_3d(vbs_handle* h, const uint8_t* frames, int n, int channels, int64_t stride_n,
                    int64_t stride_row, const double* ref_xy, int m_ref, double min_dist,
                    const vbs_camera* cam, double min_marker_size_px, float* table,
                    double* det, int32_t* counts, void* stream);

/* MarkerAnalysis._track_markers (3d_reconstruction.py:240-316) on a (gathered) table of n
 * consecutive frames: per ID the displacement against the frame where it was LAST SEEN; frames
 * before `first_frame + warmup_frames` are skipped; |d| > max_displacement clears the flag.
 * disp [dev] float32 [n,m_ref,VBS_DISP_COLS]. */
int vbs_displacement(vbs_handle* h, const float* table, int n, int m_ref, int warmup_frames,
                     double min_marker_size_px, double max_displacement, float* disp, void* stream);

/* The same, emitting only frames [frame_begin, frame_end) of a table holding frames [0, n): what a rank of a
 * multi-GPU run calls on the gathered table for its own shard (disp [dev] float32 [frame_end-frame_begin, m_ref, 5]).
 * The look-back for the last-seen frame may reach before frame_begin. */
int vbs_displacement_range(vbs_handle* h, const float* table, int n, int m_ref, int warmup_frames,
                           double min_marker_size_px, double max_displacement, int frame_begin, int frame_end,
                           float* disp, void* stream);

/* The same on float64 tables (same column layout, no handle): what `MarkerAnalysis._track_markers(df)`
 * uses so that the DataFrame interface keeps the reference's float64 results. */
int vbs_displacement_f64(int device, const double* table, int n, int m_ref, int warmup_frames,
                         double min_marker_size_px, double max_displacement, double* disp, void* stream);

/* fit_plane_least_squares (ForceDistribution.py:138-162): per frame Z = aX + bY + c over the rows
 * with VBS_FLAG_XYZ, tilt = atan(sqrt(a^2+b^2)) in degrees.  plane [dev] float32 [n,VBS_PLANE_COLS]. */
int vbs_plane_fit(vbs_handle* h, const float* table, int n, int m_ref, float* plane, void* stream);

/* The deviation field and its plane (ForceDistribution.py: process_marker_data :168-208, visualize_deviations :218-243,
 * :262-268, :274): for every marker with a 3-D point (VBS_FLAG_XYZ) in all four table rows - start / end of the vertical
 * loading, start / end of the tilted one - deviation = (tilt_end - tilt_start) - (vert_end - vert_start) (:196-204); the
 * plane Z = aX + bY + c is fitted over the END POINTS ref + scale * deviation (:229-243; shell_mode 0 = 'plane': Z starts
 * at 0, 1 = 'shell': at the reference Z, :222), tilt = atan(sqrt(a^2 + b^2)); out also carries the mean of the scaled
 * deviation vectors (:263) and the mean magnitude of the deviations (:274).
 *   vert_start .. tilt_end [dev] float32 [m_ref][VBS_TABLE_COLS]: one frame's rows of a table each (same slot order)
 *   ref_xyz [dev] float32 [m_ref][3] reference positions (the embedded MARKER_REF_DATA :29-95 in the reference)
 *   deviation [dev] float32 [m_ref][4] = (1 | 0 common, dX, dY, dZ); out [dev] float32 [VBS_DEVPLANE_COLS] */
int vbs_deviation_plane(vbs_handle* h, const float* vert_start, const float* vert_end, const float* tilt_start,
                        const float* tilt_end, const float* ref_xyz, int m_ref, int shell_mode, double scale,
                        float* deviation, float* out, void* stream);

/* Frame-0 identity assignment on the device — `MarkerTracker._process_first_frame`
 * (marker_detection.py:275-347; inlined again at tracking.py:106-178): the marker nearest the mean is (0,0), the
 * others' radii are clustered into `num_layers` rings (exact 1-D k-means, the deterministic stand-in for the
 * reference's unseeded KMeans) and each ring is ordered by angle from the member nearest angle 0.
 *   det [dev]     float64 [*count][VBS_DET_COLS] rows of frame 0 as vbs_marker_center / vbs_track_to_3d write them
 *   count [dev]   int32[1] number of rows (a device status s < 0 comes back as m_out = 1000 s)
 *   id_mode       0 = "as_written" (the published `(layer,-1)` key collision: 1 + num_layers IDs), 1 = "full"
 *   ids [dev]     int32 [cap][2] (layer, index) in the reference dict's order; ref_xy [dev] float64 [cap][2] (Ox, Oy):
 *                 the table vbs_track / vbs_track_to_3d take as `ref_xy`
 *   m_out [dev]   int32[1] number of IDs; -1 = no markers ("No markers detected in first frame!"), -2 = cap too small */
int vbs_assign_ids(vbs_handle* h, const double* det, const int32_t* count, int num_layers, int id_mode,
                   int32_t* ids, double* ref_xy, int cap, int32_t* m_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VBS_H */
