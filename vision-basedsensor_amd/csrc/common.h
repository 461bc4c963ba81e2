// Internal definitions shared by the gfx950 kernels and the C-ABI (include/vbs.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/vbs.h"

typedef unsigned long long u64;
typedef long long i64;
typedef uint32_t u32;
typedef uint8_t u8;

#define VBS_NCC_MAXL 80
#define VBS_RUN_CAP 30720          // union-find nodes (runs) per mask per frame kept in LDS
#define VBS_AREA_SUMS 16           // n, 14 moments up to order 4, spare
#define VBS_LAT_MAXN 32            // passes of at most this many frames may take the few-frames labelling kernel (k_stage_lat.hip)
#define VBS_LAT_HDR 128            // dwords of counters / flags per frame of that kernel

struct BranchParams {              // marker_detection.py:117-126,129,170
    int taps_a, taps_b;            // GaussianBlur sizes
    int thresh, hi;                // inRange bounds
    int ncc_l;                     // template size
    int ncc_lo, ncc_hi;            // window offsets of mode='same'
    int ns;                        // max/min filter size (14 | 8)
    int small;                     // 1 = small-image branch
};

struct NccConst {
    double g[VBS_NCC_MAXL];        // 1-D normalised Gaussian, template = g (x) g
    double cg[VBS_NCC_MAXL + 1];   // cg[k] = g[0] + ... + g[k-1]
    double tbar, T2, l2, thr2;     // mean(template), sum((t-tbar)^2), l*l, 0.1*0.1
    double inv_l2;
};

// cv2.cvtColor(BGR2GRAY) on uint8 (marker_detection.py:114): (cb B + cg G + cr R + 2^(shift-1)) >> shift
struct GrayCoef { u32 cb, cg, cr, half, shift; };
static inline GrayCoef gray_coef(int bits) {
    return bits == 14 ? GrayCoef{1868u, 9617u, 4899u, 1u << 13, 14u} : GrayCoef{3735u, 19235u, 9798u, 1u << 14, 15u};
}

#ifdef VBS_DEBUG_KNOBS                                  // tools/ builds only: phase timing by early exit
#include <cstdlib>
#define VBS_KNOB(name) (getenv(name) ? atoi(getenv(name)) : 0)
#else
#define VBS_KNOB(name) 0
#endif

struct ProfRec { const char* name; hipEvent_t a, b; };

#define SG_REC 2048                // k_stage / k_stage_lat: segment records per frame, at most (StageGeom::rec_cap)
struct vbs_handle {
    bool prof = false;                     // record a HIP event pair around every kernel launch
    std::vector<ProfRec> recs;
    int device, H, W, P, WW, maxm, maxb;   // P = row pitch (mult. of 64), WW = P/64 words per row
    BranchParams bp;
    NccConst ncc;
    std::string err;
    // ---- device workspace (per internal pass of maxb frames) ----
    u8* gray;          // [maxb][H][P]   gray plane of 3-channel / undistorted input; null until first needed (need_gray)
    u8* gray2;         // second plane: the conversion of pass k + 1 runs on `side` while pass k computes
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_gray[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    uint4* blur_frags; // Toeplitz operand fragments of k_blur_mfma (blur_mfma_fragments)
    uint4* blur16_h = nullptr;   // k_blur16: horizontal fragments per 16-column strip (blur16_fragments); null = not built
    uint4* blur16_v = nullptr;   // k_blur16: the 12 vertical fragment variants
    int blur_impl = 0;           // vbs_set_option(VBS_OPT_BLUR_IMPL): 0 k_blur16 where it applies, 1 always k_blur_mfma
    // vbs_set_option(VBS_OPT_PASS_STREAMS) = 2: the internal passes of vbs_track_to_3d alternate between this handle on the
    // caller's stream and a second workspace (`twin`, created at first use) on `twin_stream`
    int pass_streams = 2;
    vbs_handle* twin = nullptr;
    bool is_twin = false;                  // this handle IS some handle's second workspace (never grows one of its own)
    vbs_handle* last_ws = nullptr;         // workspace (this handle or its twin) and length of the last internal pass
    int last_nb = 0;
    hipStream_t twin_stream = nullptr;
    hipEvent_t ev_tfork = nullptr, ev_tjoin = nullptr;
    u64* area_bits;    // [maxb][H][WW]
    u64* mask_bits;    // [maxb][H][WW]
    u64* band_bits;    // [maxb][H][WW]
    u64* open_bits;    // [maxb][H][WW]
    double* ncc_rx;    // [W]  sum of g over the in-image part of the window (columns)
    double* ncc_ry;    // [H]
    uint4* ncc_frags;  // Toeplitz operand fragments of k_ncc_mfma (ncc_mfma_fragments)
    float2* ncc_rowf;  // [H] {rows of the NCC window inside the image, (float) ncc_ry}: border tiles of k_ncc_mfma
    double* ncc_tab;   // [VBS_NCC_MAXL] g, then [VBS_NCC_MAXL + 1] cg: the exact path of k_ncc_mfma reads them from memory
    u32* fstat;        // [maxb][8]  0: area popcount, 1: ambiguous ncc pixels, 2: status
    u32* wbase;        // [maxb][2][H*WW]   first node index of each word
    u32* stage_mrec = nullptr;   // k_stage's moment records when a frame's slice of wbase would hold fewer than SG_REC of them
                                 // (small frames with many blobs: the reference's real 65-dot layout); null = they live in wbase
    u32* node_pos;     // [maxb][2][RUN_CAP]  y*W + x0 of each run
    u32* node_comp;    // [maxb][2][RUN_CAP]  component id (0-based, raster order) of each run
    u32* ncomp;        // [maxb][2]
    u32* band_first;   // [maxb][maxm]
    u64* band_sums;    // [maxb][maxm][4]   count, sum x, sum y, spare
    u32* area_first;   // [maxb][maxm]
    i64* area_sums;    // [maxb][maxm][VBS_AREA_SUMS]  vertex moments about the component's first pixel
    double* ell;       // [maxb][maxm][8]   cx, cy, w, h, angle, nvert, ok, spare
    double* det64;     // [maxb][maxm][6]
    int32_t* cnt;      // [maxb]
    unsigned short* probe;   // [maxb][maxm][4]  component ids of the 2x2 cell around every band centroid
    u64* ncc_tot;      // [4]  running NCC decision counters (vbs_ncc_counters)
    u32* lat_hdr;      // [VBS_LAT_MAXN][VBS_LAT_HDR] k_stage_lat's per-frame counters; slow_total / slow_flag follow (one fill clears all)
    unsigned char* lat_scratch = nullptr;   // [VBS_LAT_MAXN][stage_lat_scratch()] what the workgroups of a frame share; null = path not available
    int lat_slots = 0;              // frames lat_scratch holds (min(max_batch, VBS_LAT_MAXN), fewer for very large frames)
    int lat_frames = 24;            // vbs_set_option(VBS_OPT_LATENCY_FRAMES): passes of <= this many frames take k_stage_lat (0: never)
    size_t lat_lds_set = 0;
    bool pass_cleared = false;      // detect_pass cleared the labelling headers / flags of this pass together with fstat
    u32* slow_total;   // [1]  frames of this pass the fused kernel handed on (lets the general kernels leave at once)
    u32* slow_flag;    // [maxb]  non-zero = the fast labelling path handed the frame on (the value says why)
    size_t stage_lds_set[2] = {0, 0}, ccl_lds_set[2] = {0, 0};   // dynamic LDS declared for k_stage / k_ccl<0|1> through this handle
    int stage_impl = 0;             // vbs_set_option(VBS_OPT_STAGE_IMPL): 0 fused k_stage, 1 the round-2 kernels (k_morph + k_ccl), 2 k_label for every frame, 3 / 4 fused at 768 / 256 threads
    int gray_bits = 15;             // BGR2GRAY fixed-point coefficient set: 15 (OpenCV 4) | 14 (OpenCV <= 3.4.1)
    bool force_seq_match = false;   // vbs_set_option(VBS_OPT_FORCE_SEQ_MATCH)
    bool gray_side = false;         // vbs_set_option(VBS_OPT_GRAY_SIDE_STREAM)
    int ncc_margin_ppm = 0;         // vbs_set_option(VBS_OPT_NCC_MARGIN): test hook, widens the float32 filter's margin
    u8* lut;           // [256] contour vertex table
    short* umap1;      // [H][W][2] int16 undistortion source pixel (CV_16SC2)
    unsigned short* umap2;   // [H][W] fractional index into the bilinear weight table
    int* uwtab;        // [1024][4] bilinear weights in 1/32768
    bool undist = false;     // frame undistortion enabled (vbs_set_undistort)
    double newK[9];
    std::vector<void*> allocs;
};

#define HIPCHK(h, call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            return VBS_EHIP;                                                              \
        }                                                                                 \
    } while (0)

// kernel launch with optional event bracketing on the launch stream (vbs_profile / vbs_profile_read)
#define VBS_LAUNCH(h, s, name, ...)                                                       \
    do {                                                                                  \
        hipEvent_t a_ = nullptr, b_ = nullptr;                                            \
        if ((h)->prof) {                                                                  \
            (void)hipEventCreate(&a_); (void)hipEventCreate(&b_); (void)hipEventRecord(a_, s); \
        }                                                                                 \
        hipLaunchKernelGGL(__VA_ARGS__);                                                  \
        if ((h)->prof) { (void)hipEventRecord(b_, s); (h)->recs.push_back({name, a_, b_}); } \
    } while (0)

// ---- launchers (each enqueues on `s`; nb = frames in this pass) --------------------------------
void launch_gray(vbs_handle* h, const u8* frames, int nb, int channels, int64_t stride_n,
                 int64_t stride_row, u8* gray, hipStream_t s);
void launch_gray_dense(vbs_handle* h, const u8* frames, int nb, int64_t stride_n, int64_t stride_row, u8* out,
                       hipStream_t s);
void launch_blur(vbs_handle* h, const u8* gray, int64_t gstride_n, int64_t gstride_row, int nb,
                 u8* area_u8, hipStream_t s);
void launch_ncc(vbs_handle* h, int nb, u8* mask_u8, double* ncc_out, hipStream_t s);
void launch_points(int which, const double* in, int n, const vbs_camera& cam, double* out, int32_t* ok,
                   hipStream_t s);
void launch_threshold(vbs_handle* h, const u8* mask, const u8* area, int nb, hipStream_t s);
void launch_labelling(vbs_handle* h, int nb, hipStream_t s);    // band / open planes, their components and sums (a9 - a12)
void launch_finalize(vbs_handle* h, int nb, double* det, int32_t* counts, hipStream_t s);
void launch_track(vbs_handle* h, const double* det, const int32_t* counts32, int nb,
                  const double* ref_xy, int m_ref, double min_dist, float* table, hipStream_t s);
void launch_solve3d(vbs_handle* h, float* table, int n, int m_ref, const vbs_camera& cam,
                    double min_size, hipStream_t s);
void launch_displacement(vbs_handle* h, const float* table, int n, int m_ref, int warmup,
                         double min_size, double max_disp, int f0, int f1, float* disp, hipStream_t s);
void launch_plane_fit(vbs_handle* h, const float* table, int n, int m_ref, float* plane,
                      hipStream_t s);
void launch_deviation_plane(vbs_handle* h, const float* vs, const float* ve, const float* ts, const float* te, const float* ref,
                            int m_ref, int shell, double scale, float* dev, float* out, hipStream_t s);
void launch_assign_ids(vbs_handle* h, const double* det, const int32_t* count, int num_layers, int full_mode,
                       int32_t* ids_out, double* xy_out, int cap, int32_t* m_out, hipStream_t s);
void make_contour_lut(u8 out[256]);
std::vector<u32> ncc_mfma_fragments(const NccConst& nc, int l);
std::vector<u32> blur_mfma_fragments(const std::vector<int>& taps_a, const std::vector<int>& taps_b, int nk,
                                     int sa0, int nka);
void blur16_fragments(const std::vector<int>& taps_s, const std::vector<int>& taps_l, int W, bool small, std::vector<u32>* hfrag,
                      std::vector<u32>* vfrag);
void launch_track_fused(vbs_handle* h, int nb, const double* ref_xy, int m_ref, double min_dist,
                        float* table, const vbs_camera* cam, double min_size, hipStream_t s);
// launch_finalize + launch_track_fused as one launch (k_finalize_track: passes of a few frames)
void launch_finalize_track(vbs_handle* h, int nb, double* det, int32_t* counts, const double* ref_xy, int m_ref, double min_dist,
                           float* table, const vbs_camera* cam, double min_size, hipStream_t s);
void launch_popcount(vbs_handle* h, int nb, hipStream_t s);
size_t stage_lat_scratch(const vbs_handle* h);          // bytes of scratch per frame k_stage_lat needs for this geometry (0: not taken)
// n 32-bit words <- value, as a KERNEL on `s`: the per-pass clears of the hot path.  (Not hipMemsetAsync: captured into a HIP
// graph, the memset nodes of a one-stream multi-pass call left the first pass's status words holding address-like garbage
// from the second replay on - tools/gpu_graph_debug.py, ROCm 7.2 - while kernel nodes replay exactly.)
void launch_fill(u32* p, u32 value, size_t n, hipStream_t s);
int launch_ncc_general(const double* T, int th, int tw, const double* I, int h, int w, int mode, double* out,
                       double* stats, hipStream_t s);
void launch_displacement64(const double* table, int n, int m_ref, int warmup, double min_size, double max_disp,
                           double* disp, int* fmin_scratch, hipStream_t s);
int setup_undistort(vbs_handle* h, const double* K9, const double* dist, int ndist, hipStream_t s);
void launch_remap(vbs_handle* h, const u8* frames, int nb, int channels, int64_t stride_n, int64_t stride_row,
                  u8* out, int to_gray, hipStream_t s);
void bilinear_weights_i16(int32_t* out);
void optimal_new_camera_matrix_alpha0(const double* K, const double* k, int w, int h, double* newK);
