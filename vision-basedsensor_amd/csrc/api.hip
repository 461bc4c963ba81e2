// C-ABI of include/vbs.h: workspace management, host-side constant tables, kernel sequencing.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "common.h"

void launch_track_fused(vbs_handle* h, int nb, const double* ref_xy, int m_ref, double min_dist,
                        float* table, const vbs_camera* cam, double min_size, hipStream_t s);

// ---- host tables -----------------------------------------------------------------------------------
// OpenCV 4 bit-exact uint8 Gaussian kernel in 8 fractional bits (see oracle/stages.py:gaussian_kernel_q8)
static std::vector<int> gaussian_taps_q8(int ksize, double sigma) {
    std::vector<double> v(ksize);
    const double scale2x = -0.125 / (sigma * sigma);
    for (int i = 0; i < ksize; ++i) {
        double xi = (double)(1 - ksize + 2 * i);
        v[i] = std::exp(xi * xi * scale2x);
    }
    const int n2 = (ksize - 1) / 2;
    double s = 0;
    for (int i = 0; i < n2; ++i) s += v[i];
    s = 2.0 * s + 1.0;
    const double mul = 1.0 / s;
    std::vector<int> out(ksize, 0);
    double err = 0;
    int tot = 0;
    for (int i = 0; i < ksize / 2; ++i) {
        double adj = v[i] * mul * 256.0 + err;
        int v0 = (int)std::nearbyint(adj);             // round half to even (cvRound)
        err = adj - v0;
        out[i] = out[ksize - 1 - i] = v0;
        tot += v0;
    }
    out[ksize / 2] = 256 - 2 * tot;
    return out;
}

static void ncc_consts(int l, double sigma, NccConst* nc) {
    std::vector<double> ax(l), e(l);
    // np.linspace(-(l-1)/2, (l-1)/2, l): start + i*step, last point exact
    const double start = -(l - 1) / 2.0, stop = (l - 1) / 2.0, step = (stop - start) / (l - 1);
    for (int i = 0; i < l; ++i) ax[i] = (i == l - 1) ? stop : start + i * step;
    double se = 0;
    for (int i = 0; i < l; ++i) { e[i] = std::exp(-0.5 * ax[i] * ax[i] / (sigma * sigma)); se += e[i]; }
    for (int i = 0; i < VBS_NCC_MAXL; ++i) nc->g[i] = i < l ? e[i] / se : 0.0;
    nc->cg[0] = 0.0;
    for (int i = 0; i < VBS_NCC_MAXL; ++i) nc->cg[i + 1] = nc->cg[i] + nc->g[i];
    // template statistics as `_normxcorr2` forms them: t = K / sum(K), tbar = mean(t), T2 = sum((t - tbar)^2)
    std::vector<double> K((size_t)l * l);
    double sk = 0;
    for (int i = 0; i < l; ++i)
        for (int j = 0; j < l; ++j) {
            K[(size_t)i * l + j] = std::exp(-0.5 * (ax[j] * ax[j] + ax[i] * ax[i]) / (sigma * sigma));
            sk += K[(size_t)i * l + j];
        }
    double st = 0;
    for (auto& x : K) { x /= sk; st += x; }
    nc->tbar = st / ((double)l * l);
    double t2 = 0;
    for (auto& x : K) t2 += (x - nc->tbar) * (x - nc->tbar);
    nc->T2 = t2;
    nc->l2 = (double)l * l;
    nc->inv_l2 = 1.0 / nc->l2;
    nc->thr2 = 0.1 * 0.1;
}

// ---- handle ------------------------------------------------------------------------------------------
template <typename T>
static int dev_alloc(vbs_handle* h, T** p, size_t count) {
    void* q = nullptr;
    if (hipMalloc(&q, count * sizeof(T) + 256) != hipSuccess) {
        h->err = "hipMalloc failed (" + std::to_string(count * sizeof(T)) + " bytes)";
        return VBS_ENOMEM;
    }
    h->allocs.push_back(q);
    *p = (T*)q;
    return VBS_OK;
}

extern "C" int vbs_version(void) { return 100; }

extern "C" int vbs_contour_lut(uint8_t out[256]) {
    if (!out) return VBS_EINVAL;
    make_contour_lut(out);
    return VBS_OK;
}

extern "C" int vbs_gaussian_taps_q8(int ksize, double sigma, int32_t* out) {
    if (!out || ksize < 1 || !(ksize & 1) || !(sigma > 0)) return VBS_EINVAL;
    std::vector<int> k = gaussian_taps_q8(ksize, sigma);
    for (int i = 0; i < ksize; ++i) out[i] = k[i];
    return VBS_OK;
}

extern "C" int vbs_ncc_template(int l, double sigma, double* g, double* stats) {
    if (!g || !stats || l < 2 || l > VBS_NCC_MAXL || !(sigma > 0)) return VBS_EINVAL;
    NccConst nc;
    ncc_consts(l, sigma, &nc);
    for (int i = 0; i < l; ++i) g[i] = nc.g[i];
    stats[0] = nc.tbar; stats[1] = nc.T2; stats[2] = nc.l2; stats[3] = nc.thr2;
    return VBS_OK;
}

extern "C" int vbs_destroy(vbs_handle* h) {
    if (!h) return VBS_EINVAL;
    (void)hipSetDevice(h->device);
    if (h->twin) { (void)vbs_destroy(h->twin); h->twin = nullptr; }
    if (h->twin_stream) (void)hipStreamDestroy(h->twin_stream);
    if (h->ev_tfork) (void)hipEventDestroy(h->ev_tfork);
    if (h->ev_tjoin) (void)hipEventDestroy(h->ev_tjoin);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (int i = 0; i < 2; ++i) { if (h->ev_gray[i]) (void)hipEventDestroy(h->ev_gray[i]); if (h->ev_free[i]) (void)hipEventDestroy(h->ev_free[i]); }
    delete h;
    return VBS_OK;
}

extern "C" const char* vbs_last_error(const vbs_handle* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int vbs_create(int device, int height, int width, int max_markers, int max_batch,
                          vbs_handle** out) {
    if (!out) return VBS_EINVAL;
    *out = nullptr;
    if (height < 64 || width < 128 || width > 4096 || max_markers < 1 || max_markers > 1024 || max_batch < 1)
        return VBS_EINVAL;                                  // k_label keeps one row (<= 64 words) per wave
    vbs_handle* h = new (std::nothrow) vbs_handle();
    if (!h) return VBS_ENOMEM;
    *out = h;                                           // returned even on failure so the text can be read
    h->device = device;
    if (hipSetDevice(device) != hipSuccess) { h->err = "hipSetDevice failed"; return VBS_EHIP; }
    h->H = height; h->W = width; h->maxm = max_markers; h->maxb = max_batch;
    h->P = (width + 63) / 64 * 64;
    h->WW = h->P / 64;
    BranchParams& bp = h->bp;
    bp.small = height <= 480;                           // marker_detection.py:117
    double sa, sb, ts;
    if (bp.small) { bp.taps_a = 21; sa = 4.56; bp.taps_b = 35; sb = 11.4; bp.ncc_l = 33; ts = 7.4; bp.thresh = 35; bp.hi = 180; bp.ns = 8; }
    else { bp.taps_a = 39; sa = 8.0; bp.taps_b = 101; sb = 20.0; bp.ncc_l = 80; ts = 13.0; bp.thresh = 20; bp.hi = 200; bp.ns = 14; }
    bp.ncc_lo = -((bp.ncc_l - 1) - (bp.ncc_l - 1) / 2);
    bp.ncc_hi = (bp.ncc_l - 1) / 2;
    if (height <= bp.taps_b / 2 + 4 || width <= bp.taps_b / 2 + 4) { h->err = "frame smaller than the blur radius"; return VBS_EINVAL; }
    ncc_consts(bp.ncc_l, ts, &h->ncc);
    std::vector<u32> frags, frags16h, frags16v;
    {   // the int8 matrix-core blur needs taps < 128 that sum to 256 (true for every sigma >= 1)
        std::vector<int> ka = gaussian_taps_q8(bp.taps_a, sa), kb = gaussian_taps_q8(bp.taps_b, sb);
        int suma = 0, sumb = 0, mx = 0;
        for (int v : ka) { suma += v; mx = std::max(mx, v); }
        for (int v : kb) { sumb += v; mx = std::max(mx, v); }
        if (suma != 256 || sumb != 256 || mx > 127) { h->err = "internal: blur taps do not fit int8"; return VBS_EINVAL; }
        frags = bp.small ? blur_mfma_fragments(ka, kb, 3, 0, 3) : blur_mfma_fragments(ka, kb, 5, 1, 3);
        if (width >= (bp.small ? 176 : 240) && (width & 3) == 0) blur16_fragments(ka, kb, width, bp.small != 0, &frags16h, &frags16v);
    }

    const size_t B = (size_t)max_batch, HW = (size_t)height * h->WW;
    int rc;
#define ALLOC(field, count) if ((rc = dev_alloc(h, &h->field, (count))) != VBS_OK) return rc
    h->gray = h->gray2 = nullptr;                       // allocated at the first 3-channel / undistorted use (need_gray)
    ALLOC(area_bits, B * HW); ALLOC(mask_bits, B * HW); ALLOC(band_bits, B * HW);
    ALLOC(open_bits, B * HW);
    ALLOC(ncc_rx, (size_t)width); ALLOC(ncc_ry, (size_t)height);
    ALLOC(wbase, B * 2 * HW);
    if (2 * HW < (size_t)16 * SG_REC) ALLOC(stage_mrec, B * 16 * SG_REC);   // (k_stage.hip: stage_geom)
    ALLOC(node_pos, B * 2 * VBS_RUN_CAP); ALLOC(node_comp, B * 2 * VBS_RUN_CAP);
    ALLOC(ncomp, B * 2);
    ALLOC(band_first, B * max_markers); ALLOC(band_sums, B * max_markers * 4);
    ALLOC(area_first, B * max_markers); ALLOC(area_sums, B * max_markers * VBS_AREA_SUMS);
    ALLOC(ell, B * max_markers * 8); ALLOC(det64, B * max_markers * 6);
    ALLOC(cnt, B);
    ALLOC(probe, B * max_markers * 4); ALLOC(ncc_tot, 4);
    // one allocation, one fill per pass: k_stage_lat's headers | slow counter | slow flags | frame statistics
    ALLOC(lat_hdr, (size_t)VBS_LAT_MAXN * VBS_LAT_HDR + 4 + B + B * 8 + 8);     // (+ one spare word: displacement's first-frame cell)
    h->slow_total = h->lat_hdr + (size_t)VBS_LAT_MAXN * VBS_LAT_HDR;
    h->slow_flag = h->slow_total + 4;
    h->fstat = h->slow_flag + B;
    // the few-frames labelling kernel's scratch: optional (without it such passes take the batch kernel), at most 256 MB
    h->lat_slots = 0;
    if (const size_t per = stage_lat_scratch(h)) {
        const size_t slots = std::min<size_t>(std::min<size_t>(B, VBS_LAT_MAXN), ((size_t)256 << 20) / per);
        if (slots >= 1 && dev_alloc(h, &h->lat_scratch, per * slots) == VBS_OK) h->lat_slots = (int)slots;
        else { h->lat_scratch = nullptr; h->err.clear(); (void)hipGetLastError(); }
    }
    ALLOC(lut, 256);
    ALLOC(blur_frags, frags.size() / 4);
    if (!frags16h.empty()) { ALLOC(blur16_h, frags16h.size() / 4); ALLOC(blur16_v, frags16v.size() / 4); }
    std::vector<u32> nfrags = ncc_mfma_fragments(h->ncc, bp.ncc_l);
    ALLOC(ncc_frags, nfrags.size() / 4);
    ALLOC(ncc_tab, (size_t)2 * VBS_NCC_MAXL + 1);
    ALLOC(ncc_rowf, (size_t)height);
    ALLOC(umap1, (size_t)height * width * 2); ALLOC(umap2, (size_t)height * width); ALLOC(uwtab, 4096);
#undef ALLOC
    std::vector<double> rx(width), ry(height);
    for (int x = 0; x < width; ++x) {
        double s = 0;
        for (int j = 0; j < bp.ncc_l; ++j) { int xx = x + bp.ncc_lo + j; if (xx >= 0 && xx < width) s += h->ncc.g[j]; }
        rx[x] = s;
    }
    for (int y = 0; y < height; ++y) {
        double s = 0;
        for (int j = 0; j < bp.ncc_l; ++j) { int yy = y + bp.ncc_lo + j; if (yy >= 0 && yy < height) s += h->ncc.g[j]; }
        ry[y] = s;
    }
    u8 lut[256];
    make_contour_lut(lut);
    HIPCHK(h, hipMemcpy(h->ncc_rx, rx.data(), width * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->ncc_ry, ry.data(), height * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->lut, lut, 256, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->blur_frags, frags.data(), frags.size() * sizeof(u32), hipMemcpyHostToDevice));
    if (h->blur16_h) {
        HIPCHK(h, hipMemcpy(h->blur16_h, frags16h.data(), frags16h.size() * sizeof(u32), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(h->blur16_v, frags16v.data(), frags16v.size() * sizeof(u32), hipMemcpyHostToDevice));
    }
    HIPCHK(h, hipMemcpy(h->ncc_frags, nfrags.data(), nfrags.size() * sizeof(u32), hipMemcpyHostToDevice));
    {
        std::vector<float2> rowf(height);
        for (int y = 0; y < height; ++y) {
            const int ny = std::min(y + bp.ncc_hi, height - 1) - std::max(y + bp.ncc_lo, 0) + 1;
            rowf[y] = make_float2((float)ny, (float)ry[y]);
        }
        HIPCHK(h, hipMemcpy(h->ncc_rowf, rowf.data(), height * sizeof(float2), hipMemcpyHostToDevice));
    }
    HIPCHK(h, hipMemcpy(h->ncc_tab, h->ncc.g, VBS_NCC_MAXL * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->ncc_tab + VBS_NCC_MAXL, h->ncc.cg, (VBS_NCC_MAXL + 1) * sizeof(double), hipMemcpyHostToDevice));
    {
        std::vector<int32_t> wt(4096);
        bilinear_weights_i16(wt.data());
        HIPCHK(h, hipMemcpy(h->uwtab, wt.data(), 4096 * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    HIPCHK(h, hipMemset(h->ncc_tot, 0, 4 * sizeof(u64)));
    HIPCHK(h, hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
    HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) {
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_gray[i], hipEventDisableTiming));
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_free[i], hipEventDisableTiming));
    }
    HIPCHK(h, hipDeviceSynchronize());
    return VBS_OK;
}

static int check_launch(vbs_handle* h) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("kernel launch: ") + hipGetErrorString(e); return VBS_EHIP; }
    return VBS_OK;
}

static bool capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// Gray planes of 3-channel / undistorted input ([maxb][H][P] each; the second one only serves VBS_OPT_GRAY_SIDE_STREAM):
// gray frames never touch them, so they are allocated at the first call that needs them, or by vbs_set_undistort /
// vbs_set_option(VBS_OPT_GRAY_SIDE_STREAM).  Never under stream capture: there the call fails and says why.
static int need_gray(vbs_handle* h, int planes, hipStream_t s) {
    u8** slot[2] = {&h->gray, &h->gray2};
    for (int i = 0; i < planes; ++i) {
        if (*slot[i]) continue;
        if (capturing(s)) {
            h->err = "the gray plane of 3-channel / undistorted input is allocated at first use: run one such call (or "
                     "vbs_set_undistort) outside stream capture first";
            return VBS_EINVAL;
        }
        int rc = dev_alloc(h, slot[i], (size_t)h->maxb * h->H * h->P);
        if (rc != VBS_OK) return rc;
    }
    // the second workspace converts its own passes: its first plane exists whenever this handle's does, so that a multi-pass
    // call on 3-channel frames captured into a graph finds it in place after ANY uncaptured 3-channel call
    if (h->twin && !h->twin->gray && h->gray && !capturing(s)) {
        int rc = need_gray(h->twin, 1, s);
        if (rc != VBS_OK) { h->err = "second pass workspace: " + h->twin->err; return rc; }
    }
    return VBS_OK;
}

// The per-frame statistics of a pass <- 0; for a pass of a few frames also what launch_labelling would clear (k_stage_lat's
// headers, the slow counter and flags lie in front of fstat in one allocation): one launch instead of two
static void clear_pass(vbs_handle* h, int nb, hipStream_t s) {
    h->last_ws = h; h->last_nb = nb;                     // every pass entry point records itself (vbs_track_to_3d names the
                                                         // workspace of ITS last pass afterwards): vbs_frame_stats / vbs_stage_tables
    if (nb <= h->lat_frames && (h->stage_impl == 0 || h->stage_impl >= 3) && nb <= h->lat_slots) {
        launch_fill(h->lat_hdr, 0u, (size_t)VBS_LAT_MAXN * VBS_LAT_HDR + 4 + (size_t)h->maxb + (size_t)nb * 8, s);
        h->pass_cleared = true;
    } else {
        // (the slow counter and the frames' flags lie right in front of the statistics: launch_labelling's fill with this one)
        launch_fill(h->slow_total, 0u, (size_t)4 + (size_t)h->maxb + (size_t)nb * 8, s);
        h->pass_cleared = true;
    }
}

// One internal pass of the detection stage.  `pre` = gray plane already converted for this pass (vbs_detect_loop's
// side-stream pipeline), else the conversion runs here on `s`.
static int detect_pass(vbs_handle* h, const u8* frames, int nb, int channels, int64_t stride_n,
                       int64_t stride_row, u8* mask_u8, u8* area_u8, double* ncc_out, hipStream_t s, const u8* pre = nullptr) {
    if (!pre && (h->undist || channels != 1)) { int rc = need_gray(h, 1, s); if (rc != VBS_OK) return rc; }
    h->last_ws = h; h->last_nb = nb;                     // (vbs_track_to_3d names the workspace of ITS last pass afterwards)
    clear_pass(h, nb, s);
    if (pre) {
        launch_blur(h, pre, (int64_t)h->H * h->P, h->P, nb, area_u8, s);
    } else if (h->undist) {                             // marker_detection.py:88-89: undistort, then cvtColor
        launch_remap(h, frames, nb, channels, stride_n, stride_row, h->gray, 1, s);
        launch_blur(h, h->gray, (int64_t)h->H * h->P, h->P, nb, area_u8, s);
    } else if (channels == 1) {
        launch_blur(h, frames, stride_n, stride_row, nb, area_u8, s);
    } else {
        launch_gray(h, frames, nb, channels, stride_n, stride_row, h->gray, s);
        launch_blur(h, h->gray, (int64_t)h->H * h->P, h->P, nb, area_u8, s);
    }
    launch_ncc(h, nb, mask_u8, ncc_out, s);
    return check_launch(h);
}

// BGR frames over several internal passes with VBS_OPT_GRAY_SIDE_STREAM set: cvtColor of pass k + 1 runs on the handle's
// side stream while pass k's kernels run on `s` (two gray planes; fork / join by events, so a caller may capture the whole
// call in a graph).  Off by default (`on` false: every pass converts in line, see detect_pass).
struct GrayPipe {
    vbs_handle* h; const u8* frames; int n, channels; int64_t stride_n, stride_row; hipStream_t s;
    bool on;
    bool stagger = false;                               // two pass streams: a first pass of half a batch puts them out of phase
    int start() {
        on = h->gray_side && !h->undist && channels == 3 && n > h->maxb;
        if (!on) return VBS_OK;
        { int rc = need_gray(h, 2, s); if (rc != VBS_OK) { on = false; return rc; } }
        HIPCHK(h, hipEventRecord(h->ev_fork, s));
        HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
        return convert(0);
    }
    // pass schedule: with the conversion pipelined, a short first pass keeps the only exposed conversion small
    int lead() const { return on ? std::min(h->maxb, std::max(64, h->maxb / 8)) : (stagger ? std::max(1, h->maxb / 2) : h->maxb); }
    int pass_off(int k) const { return k == 0 ? 0 : lead() + (k - 1) * h->maxb; }
    int pass_len(int k) const { return std::min(k == 0 ? lead() : h->maxb, n - pass_off(k)); }
    int convert(int k) {                                // pass k -> plane k & 1, on the side stream
        const int off = pass_off(k), nb = pass_len(k);
        launch_gray(h, frames + (int64_t)off * stride_n, nb, channels, stride_n, stride_row, k & 1 ? h->gray2 : h->gray, h->side);
        HIPCHK(h, hipEventRecord(h->ev_gray[k & 1], h->side));
        return VBS_OK;
    }
    // before pass k's blur: its plane is ready; the next pass's conversion may start once ITS plane is free again
    int acquire(int k, const u8** plane) {
        *plane = nullptr;
        if (!on) return VBS_OK;
        HIPCHK(h, hipStreamWaitEvent(s, h->ev_gray[k & 1], 0));
        if (pass_off(k + 1) < n) {
            if (k >= 1) HIPCHK(h, hipStreamWaitEvent(h->side, h->ev_free[(k + 1) & 1], 0));
            int rc = convert(k + 1);
            if (rc != VBS_OK) return rc;
        }
        *plane = k & 1 ? h->gray2 : h->gray;
        return VBS_OK;
    }
    int release(int k) {                                // after pass k's blur has been enqueued
        if (!on) return VBS_OK;
        HIPCHK(h, hipEventRecord(h->ev_free[k & 1], s));
        return VBS_OK;
    }
    // an error return in the middle of the passes: the conversion already running on the side stream is joined into `s`
    // (an un-joined fork would break a caller's stream capture and leave a kernel writing the gray planes unordered
    // against the next call); returns rc unchanged
    int fail(int rc) {
        if (on) {
            (void)hipEventRecord(h->ev_gray[0], h->side);
            (void)hipStreamWaitEvent(s, h->ev_gray[0], 0);
        }
        return rc;
    }
};

extern "C" int vbs_find_markers(vbs_handle* h, const uint8_t* frames, int n, int channels, int64_t stride_n,
                                int64_t stride_row, uint8_t* mask, uint8_t* area_mask, void* stream) {
    if (!h) return VBS_EINVAL;
    if (!frames || n < 0 || (channels != 1 && channels != 3) || stride_row < (int64_t)h->W * channels) {
        h->err = "vbs_find_markers: bad argument";
        return VBS_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t hw = (size_t)h->H * h->W;
    GrayPipe gp{h, frames, n, channels, stride_n, stride_row, s, false};
    int rc = gp.start();
    if (rc != VBS_OK) return rc;
    for (int k = 0; gp.pass_off(k) < n; ++k) {
        const int off = gp.pass_off(k), nb = gp.pass_len(k);
        const u8* plane;
        if ((rc = gp.acquire(k, &plane)) != VBS_OK) return gp.fail(rc);
        rc = detect_pass(h, frames + (int64_t)off * stride_n, nb, channels, stride_n, stride_row,
                         mask ? mask + off * hw : nullptr, area_mask ? area_mask + off * hw : nullptr, nullptr, s, plane);
        if (rc != VBS_OK) return gp.fail(rc);
        if ((rc = gp.release(k)) != VBS_OK) return gp.fail(rc);
    }
    return VBS_OK;
}

extern "C" int vbs_bgr2gray(vbs_handle* h, const uint8_t* frames, int n, int64_t stride_n, int64_t stride_row,
                            uint8_t* gray, void* stream) {
    if (!h) return VBS_EINVAL;
    if (!frames || !gray || n < 0 || stride_row < (int64_t)h->W * 3) { h->err = "vbs_bgr2gray: bad argument"; return VBS_EINVAL; }
    HIPCHK(h, hipSetDevice(h->device));
    if (n) launch_gray_dense(h, frames, n, stride_n, stride_row, gray, (hipStream_t)stream);
    return check_launch(h);
}

extern "C" int vbs_ncc_map(vbs_handle* h, const uint8_t* frames, int n, int channels, int64_t stride_n,
                           int64_t stride_row, double* ncc, void* stream) {
    if (!h) return VBS_EINVAL;
    if (!frames || !ncc || n < 0 || (channels != 1 && channels != 3) || stride_row < (int64_t)h->W * channels) {
        h->err = "vbs_ncc_map: bad argument";
        return VBS_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t hw = (size_t)h->H * h->W;
    GrayPipe gp{h, frames, n, channels, stride_n, stride_row, s, false};
    int rc = gp.start();
    if (rc != VBS_OK) return rc;
    for (int k = 0; gp.pass_off(k) < n; ++k) {
        const int off = gp.pass_off(k), nb = gp.pass_len(k);
        const u8* plane;
        if ((rc = gp.acquire(k, &plane)) != VBS_OK) return gp.fail(rc);
        rc = detect_pass(h, frames + (int64_t)off * stride_n, nb, channels, stride_n, stride_row, nullptr, nullptr,
                         ncc + off * hw, s, plane);
        if (rc != VBS_OK) return gp.fail(rc);
        if ((rc = gp.release(k)) != VBS_OK) return gp.fail(rc);
    }
    return VBS_OK;
}

extern "C" int vbs_normxcorr2(vbs_handle* h, const uint8_t* area_mask, int n, double* ncc, uint8_t* mask,
                              void* stream) {
    if (!h) return VBS_EINVAL;
    if (!area_mask || n < 0 || (!ncc && !mask)) { h->err = "vbs_normxcorr2: bad argument"; return VBS_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t hw = (size_t)h->H * h->W;
    for (int off = 0; off < n; off += h->maxb) {
        int nb = std::min(h->maxb, n - off);
        clear_pass(h, nb, s);
        launch_threshold(h, area_mask + off * hw, area_mask + off * hw, nb, s);
        launch_popcount(h, nb, s);
        launch_ncc(h, nb, mask ? mask + off * hw : nullptr, ncc ? ncc + off * hw : nullptr, s);
        int rc = check_launch(h);
        if (rc != VBS_OK) return rc;
    }
    return VBS_OK;
}

static int twin_of(vbs_handle* h);

extern "C" int vbs_set_option(vbs_handle* h, int option, int value) {
    if (!h) return VBS_EINVAL;
    // (same checks, same answer; the twin never converts a pass ahead on a side stream: its second gray plane would be dead memory)
    if (h->twin && option != VBS_OPT_PASS_STREAMS && option != VBS_OPT_GRAY_SIDE_STREAM) (void)vbs_set_option(h->twin, option, value);
    switch (option) {
        case VBS_OPT_PASS_STREAMS:
            if (value != 1 && value != 2) break;
            h->pass_streams = value;
            // an explicit 2 builds the second workspace NOW (a set-up call: allocations, copies, a device synchronisation),
            // so that no later vbs_track_to_3d has to - and a call captured into a graph finds it in place
            if (value == 2 && !h->is_twin) { int rc = twin_of(h); if (rc != VBS_OK) return rc; }
            return VBS_OK;
        case VBS_OPT_FORCE_SEQ_MATCH: h->force_seq_match = value != 0; return VBS_OK;
        case VBS_OPT_GRAY_SIDE_STREAM:
            h->gray_side = value != 0;
            if (h->gray_side) { HIPCHK(h, hipSetDevice(h->device)); int rc = need_gray(h, 2, nullptr); if (rc != VBS_OK) return rc; }
            return VBS_OK;
        case VBS_OPT_NCC_MARGIN:
            if (value < 0 || value > 100000) break;
            h->ncc_margin_ppm = value;
            return VBS_OK;
        case VBS_OPT_STAGE_IMPL:
            if (value < 0 || value > 4) break;
            h->stage_impl = value;
            return VBS_OK;
        case VBS_OPT_LATENCY_FRAMES:
            if (value < 0 || value > VBS_LAT_MAXN) break;
            h->lat_frames = value;
            if (h->twin) h->twin->lat_frames = value;
            return VBS_OK;
        case VBS_OPT_BLUR_IMPL:
            if (value != 0 && value != 1) break;
            h->blur_impl = value;
            return VBS_OK;
        case VBS_OPT_GRAY_COEFFS:
            if (value != 14 && value != 15) break;
            h->gray_bits = value;
            return VBS_OK;
        default: break;
    }
    h->err = "vbs_set_option: unknown option or bad value";
    return VBS_EINVAL;
}

extern "C" int vbs_normxcorr2_general(int device, const double* tmpl, int th, int tw, const double* image, int h, int w,
                                      int mode, double* out, void* stream) {
    if (!tmpl || !image || !out || th < 1 || tw < 1 || h < 1 || w < 1 || mode < 0 || mode > 2) return VBS_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return VBS_EHIP;
    double* stats = nullptr;
    if (hipMallocAsync((void**)&stats, 4 * sizeof(double), (hipStream_t)stream) != hipSuccess) return VBS_ENOMEM;
    const int rc = launch_ncc_general(tmpl, th, tw, image, h, w, mode, out, stats, (hipStream_t)stream);
    (void)hipFreeAsync(stats, (hipStream_t)stream);
    if (rc != VBS_OK) return rc;
    return hipGetLastError() == hipSuccess ? VBS_OK : VBS_EHIP;
}

extern "C" int vbs_profile(vbs_handle* h, int enable) {
    if (!h) return VBS_EINVAL;
    for (auto& r : h->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    h->recs.clear();
    h->prof = enable != 0;
    return VBS_OK;
}

extern "C" int vbs_profile_read(vbs_handle* h, char* buf, int cap) {
    if (!h || !buf || cap < 2) return VBS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    std::vector<std::string> names;
    std::vector<double> tot;
    std::vector<long> cnt;
    for (auto& r : h->recs) {
        float ms = 0;
        HIPCHK(h, hipEventElapsedTime(&ms, r.a, r.b));
        size_t i = 0;
        for (; i < names.size(); ++i) if (names[i] == r.name) break;
        if (i == names.size()) { names.push_back(r.name); tot.push_back(0); cnt.push_back(0); }
        tot[i] += ms; cnt[i] += 1;
    }
    std::string out;
    for (size_t i = 0; i < names.size(); ++i)
        out += names[i] + " " + std::to_string(cnt[i]) + " " + std::to_string(tot[i]) + "\n";
    if ((int)out.size() + 1 > cap) { h->err = "vbs_profile_read: buffer too small"; return VBS_EINVAL; }
    memcpy(buf, out.c_str(), out.size() + 1);
    return VBS_OK;
}

extern "C" int vbs_set_undistort(vbs_handle* h, const double* K9, const double* dist, int ndist, double* newK9,
                                 void* stream) {
    if (!h) return VBS_EINVAL;
    if (!K9) { h->undist = false; return VBS_OK; }      // NULL camera matrix switches undistortion off
    if (ndist < 0 || ndist > 5 || (ndist && !dist) || !(K9[0] > 0) || !(K9[4] > 0)) {
        h->err = "vbs_set_undistort: bad argument";
        return VBS_EINVAL;
    }
    HIPCHK(h, hipSetDevice(h->device));
    int rc = need_gray(h, 1, (hipStream_t)stream);      // (a set-up call: the plane the undistorted frames go to)
    if (rc != VBS_OK) return rc;
    rc = setup_undistort(h, K9, dist, ndist, (hipStream_t)stream);
    if (rc != VBS_OK) return rc;
    h->undist = true;
    if (newK9) for (int i = 0; i < 9; ++i) newK9[i] = h->newK[i];
    return check_launch(h);
}

extern "C" int vbs_undistort_frames(vbs_handle* h, const uint8_t* frames, int n, int channels, int64_t stride_n,
                                    int64_t stride_row, uint8_t* out, void* stream) {
    if (!h) return VBS_EINVAL;
    if (!frames || !out || n < 0 || (channels != 1 && channels != 3) || stride_row < (int64_t)h->W * channels) {
        h->err = "vbs_undistort_frames: bad argument";
        return VBS_EINVAL;
    }
    if (!h->undist) { h->err = "vbs_undistort_frames: call vbs_set_undistort first"; return VBS_EINVAL; }
    HIPCHK(h, hipSetDevice(h->device));
    if (n) launch_remap(h, frames, n, channels, stride_n, stride_row, out, 0, (hipStream_t)stream);
    return check_launch(h);
}

// the workspace the last internal pass ran in: with two pass streams that is the second workspace for an odd last pass
static vbs_handle* last_ws(vbs_handle* h) { return (h->last_ws && (h->last_ws == h || h->last_ws == h->twin)) ? h->last_ws : h; }

extern "C" int vbs_frame_stats(vbs_handle* h, uint32_t* out, int n) {
    if (!h || !out || n < 0 || n > h->maxb) return VBS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());                   // (the last pass may have run on the handle's own stream)
    HIPCHK(h, hipMemcpy(out, last_ws(h)->fstat, (size_t)n * 8 * sizeof(u32), hipMemcpyDeviceToHost));
    return VBS_OK;
}

extern "C" int vbs_stage_tables(vbs_handle* h, int n, uint32_t* ncomp, uint64_t* band_sums, uint32_t* area_first,
                                int64_t* area_sums, uint16_t* probe, uint32_t* slow) {
    if (!h || n < 0 || n > h->maxb) return VBS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const size_t N = (size_t)n, M = (size_t)h->maxm;
    vbs_handle* const caller = h;
    h = last_ws(h);
    (void)caller;
    if (ncomp) HIPCHK(h, hipMemcpy(ncomp, h->ncomp, N * 2 * sizeof(u32), hipMemcpyDeviceToHost));
    if (band_sums) HIPCHK(h, hipMemcpy(band_sums, h->band_sums, N * M * 4 * sizeof(u64), hipMemcpyDeviceToHost));
    if (area_first) HIPCHK(h, hipMemcpy(area_first, h->area_first, N * M * sizeof(u32), hipMemcpyDeviceToHost));
    if (area_sums) HIPCHK(h, hipMemcpy(area_sums, h->area_sums, N * M * VBS_AREA_SUMS * sizeof(i64), hipMemcpyDeviceToHost));
    if (probe) HIPCHK(h, hipMemcpy(probe, h->probe, N * M * 4 * sizeof(unsigned short), hipMemcpyDeviceToHost));
    if (slow) HIPCHK(h, hipMemcpy(slow, h->slow_flag, N * sizeof(u32), hipMemcpyDeviceToHost));
    return VBS_OK;
}

extern "C" int vbs_ncc_counters(vbs_handle* h, uint64_t out[3], int reset) {
    if (!h || !out) return VBS_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out, h->ncc_tot, 3 * sizeof(u64), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(h, hipMemset(h->ncc_tot, 0, 4 * sizeof(u64)));
    if (h->twin) {                                      // the passes the second workspace ran
        uint64_t t[3] = {0, 0, 0};
        HIPCHK(h, hipMemcpy(t, h->twin->ncc_tot, 3 * sizeof(u64), hipMemcpyDeviceToHost));
        if (reset) HIPCHK(h, hipMemset(h->twin->ncc_tot, 0, 4 * sizeof(u64)));
        for (int i = 0; i < 3; ++i) out[i] += t[i];
    }
    return VBS_OK;
}

extern "C" int vbs_undistort_points(int device, const double* pts, int n, const vbs_camera* cam, double* out,
                                    void* stream) {
    if (!pts || !out || !cam || n < 0) return VBS_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return VBS_EHIP;
    if (n) launch_points(0, pts, n, *cam, out, nullptr, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? VBS_OK : VBS_EHIP;
}

extern "C" int vbs_calculate_3d(int device, const double* uvd, int n, const vbs_camera* cam, double* xyz,
                                int32_t* ok, void* stream) {
    if (!uvd || !xyz || !ok || !cam || n < 0) return VBS_EINVAL;
    if (!(cam->K[0] > 0) || !(cam->K[4] > 0)) return VBS_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return VBS_EHIP;
    if (n) launch_points(1, uvd, n, *cam, xyz, ok, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? VBS_OK : VBS_EHIP;
}

extern "C" int vbs_marker_center(vbs_handle* h, const uint8_t* mask, const uint8_t* area_mask, int n,
                                 double* det, int32_t* counts, void* stream) {
    if (!h) return VBS_EINVAL;
    if (!mask || !area_mask || !det || !counts || n < 0) { h->err = "vbs_marker_center: bad argument"; return VBS_EINVAL; }
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(h, hipSetDevice(h->device));
    const size_t hw = (size_t)h->H * h->W;
    for (int off = 0; off < n; off += h->maxb) {
        int nb = std::min(h->maxb, n - off);
        clear_pass(h, nb, s);
        h->last_ws = h; h->last_nb = nb;
        launch_threshold(h, mask + off * hw, area_mask + off * hw, nb, s);
        launch_labelling(h, nb, s);
        launch_finalize(h, nb, det + (size_t)off * h->maxm * VBS_DET_COLS, counts + off, s);
        int rc = check_launch(h);
        if (rc != VBS_OK) return rc;
    }
    return VBS_OK;
}

extern "C" int vbs_track(vbs_handle* h, const double* det, const int32_t* counts, int n, const double* ref_xy,
                         int m_ref, double min_dist, float* table, void* stream) {
    if (!h) return VBS_EINVAL;
    if (!det || !counts || !ref_xy || !table || n < 0 || m_ref < 1) { h->err = "vbs_track: bad argument"; return VBS_EINVAL; }
    HIPCHK(h, hipSetDevice(h->device));
    if (n) launch_track(h, det, counts, n, ref_xy, m_ref, min_dist, table, (hipStream_t)stream);
    return check_launch(h);
}

static int check_cam(vbs_handle* h, const vbs_camera* cam) {
    if (!cam) { h->err = "camera is NULL"; return VBS_EINVAL; }
    if (!(cam->K[0] > 0) || !(cam->K[4] > 0)) { h->err = "Focal lengths must be positive"; return VBS_EINVAL; }
    return VBS_OK;
}

extern "C" int vbs_solve3d(vbs_handle* h, float* table, int n, int m_ref, const vbs_camera* cam,
                           double min_marker_size_px, void* stream) {
    if (!h) return VBS_EINVAL;
    if (!table || n < 0 || m_ref < 1) { h->err = "vbs_solve3d: bad argument"; return VBS_EINVAL; }
    int rc = check_cam(h, cam);
    if (rc != VBS_OK) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    if (n) launch_solve3d(h, table, n, m_ref, *cam, min_marker_size_px, (hipStream_t)stream);
    return check_launch(h);
}

// The second workspace of VBS_OPT_PASS_STREAMS = 2: a whole second handle (same geometry, same options) plus the stream
// its passes run on and the two events that fork it from / join it to the caller's stream.
static int twin_of(vbs_handle* h) {
    if (h->twin) return VBS_OK;
    vbs_handle* t = nullptr;
    int rc = vbs_create(h->device, h->H, h->W, h->maxm, h->maxb, &t);
    if (rc != VBS_OK) {
        h->err = std::string("second pass workspace: ") + (t ? t->err : std::string("allocation failed"));
        if (t) (void)vbs_destroy(t);
        return rc;
    }
    t->pass_streams = 1;
    t->is_twin = true;
    t->gray_bits = h->gray_bits; t->force_seq_match = h->force_seq_match; t->ncc_margin_ppm = h->ncc_margin_ppm;
    t->stage_impl = h->stage_impl; t->blur_impl = h->blur_impl; t->lat_frames = h->lat_frames;
    hipStream_t st = nullptr;
    hipEvent_t ef = nullptr, ej = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ef, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ej, hipEventDisableTiming) != hipSuccess) {
        if (st) (void)hipStreamDestroy(st);             // (whatever of the three exists goes: nothing half-made stays on h)
        if (ef) (void)hipEventDestroy(ef);
        if (ej) (void)hipEventDestroy(ej);
        (void)vbs_destroy(t);
        h->err = "second pass workspace: stream / event creation failed";
        return VBS_EHIP;
    }
    h->twin_stream = st; h->ev_tfork = ef; h->ev_tjoin = ej;
    h->twin = t;
    if (h->gray) { int rg = need_gray(t, 1, nullptr); if (rg != VBS_OK) { h->err = "second pass workspace: " + t->err; return rg; } }
    return VBS_OK;
}

extern "C" int vbs_track_to_3d(vbs_handle* h, const uint8_t* frames, int n, int channels, int64_t stride_n,
                               int64_t stride_row, const double* ref_xy, int m_ref, double min_dist,
                               const vbs_camera* cam, double min_marker_size_px, float* table, double* det,
                               int32_t* counts, void* stream) {
    if (!h) return VBS_EINVAL;
    if (!frames || n < 0 || (channels != 1 && channels != 3) || stride_row < (int64_t)h->W * channels ||
        (table && (!ref_xy || m_ref < 1))) {
        h->err = "vbs_track_to_3d: bad argument";
        return VBS_EINVAL;
    }
    if (cam) { int rc = check_cam(h, cam); if (rc != VBS_OK) return rc; }
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(h, hipSetDevice(h->device));
    GrayPipe gp{h, frames, n, channels, stride_n, stride_row, s, false};
    int rc = gp.start();
    if (rc != VBS_OK) return rc;
    // Several passes: odd passes on the second workspace and stream (VBS_OPT_PASS_STREAMS), so that the partly filled last
    // round of one pass's kernels runs next to the other pass - and, for BGR frames, the bandwidth-bound conversion of one pass
    // next to the matrix-core kernels of the other (each workspace converts into its own gray plane).  Every pass writes its own slice of the
    // caller's tables; the second stream starts behind what the caller's stream holds and is joined before the return.
    bool two = h->pass_streams == 2 && !h->prof && !h->undist && !gp.on && n > h->maxb;     // (gp.on: BGR converted a pass ahead on h->side)
    if (two && !h->twin) {
        // The second workspace normally exists by now (vbs_set_option(VBS_OPT_PASS_STREAMS, 2) builds it).  A handle left at
        // the default builds it at its first call that spans several passes - unless the caller's stream is being captured
        // (allocations and the device synchronisation of a set-up would break the capture), and if it cannot be built
        // (memory) the call simply runs every pass on the caller's stream: results are identical either way.
        if (capturing(s) || twin_of(h) != VBS_OK) { two = false; h->err.clear(); }
    }
    if (two) {
        gp.stagger = true;
        HIPCHK(h, hipEventRecord(h->ev_tfork, s));
        HIPCHK(h, hipStreamWaitEvent(h->twin_stream, h->ev_tfork, 0));
    }
    auto join = [&](int code) {                          // (also on the error returns: an un-joined fork breaks a capture)
        if (two) {
            (void)hipEventRecord(h->ev_tjoin, h->twin_stream);
            (void)hipStreamWaitEvent(s, h->ev_tjoin, 0);
        }
        return code;
    };
    for (int k = 0; gp.pass_off(k) < n; ++k) {
        const int off = gp.pass_off(k), nb = gp.pass_len(k);
        vbs_handle* hh = (two && (k & 1)) ? h->twin : h;
        hipStream_t ss = (two && (k & 1)) ? h->twin_stream : s;
        const u8* plane;
        if ((rc = gp.acquire(k, &plane)) != VBS_OK) return join(gp.fail(rc));
        rc = detect_pass(hh, frames + (int64_t)off * stride_n, nb, channels, stride_n, stride_row, nullptr, nullptr, nullptr,
                         ss, plane);
        if (rc != VBS_OK) { if (hh != h) h->err = hh->err; return join(gp.fail(rc)); }
        h->last_ws = hh; h->last_nb = nb;                // (what vbs_frame_stats / vbs_stage_tables read)
        if ((rc = gp.release(k)) != VBS_OK) return join(gp.fail(rc));
        launch_labelling(hh, nb, ss);
        if (table && nb <= hh->lat_frames)               // a few frames: detections and tracking rows in one launch
                                                         // (for a batch pass the one launch measured nothing: 282.1 k against 283.1 k frames/s)
            launch_finalize_track(hh, nb, det ? det + (size_t)off * h->maxm * VBS_DET_COLS : nullptr, counts ? counts + off : nullptr,
                                  ref_xy, m_ref, min_dist, table + (size_t)off * m_ref * VBS_TABLE_COLS, cam, min_marker_size_px, ss);
        else {
            launch_finalize(hh, nb, det ? det + (size_t)off * h->maxm * VBS_DET_COLS : nullptr,
                            counts ? counts + off : nullptr, ss);
            if (table)
                launch_track_fused(hh, nb, ref_xy, m_ref, min_dist, table + (size_t)off * m_ref * VBS_TABLE_COLS, cam,
                                   min_marker_size_px, ss);
        }
        rc = check_launch(hh);
        if (rc != VBS_OK) { if (hh != h) h->err = hh->err; return join(gp.fail(rc)); }
    }
    return join(VBS_OK);
}

extern "C" int vbs_displacement_range(vbs_handle* h, const float* table, int n, int m_ref, int warmup_frames,
                                      double min_marker_size_px, double max_displacement, int frame_begin,
                                      int frame_end, float* disp, void* stream) {
    if (!h) return VBS_EINVAL;
    if (!table || !disp || n < 0 || m_ref < 1 || frame_begin < 0 || frame_end > n || frame_begin > frame_end) {
        h->err = "vbs_displacement: bad argument";
        return VBS_EINVAL;
    }
    HIPCHK(h, hipSetDevice(h->device));
    if (frame_end > frame_begin)
        launch_displacement(h, table, n, m_ref, warmup_frames, min_marker_size_px, max_displacement, frame_begin,
                            frame_end, disp, (hipStream_t)stream);
    return check_launch(h);
}

extern "C" int vbs_displacement(vbs_handle* h, const float* table, int n, int m_ref, int warmup_frames,
                                double min_marker_size_px, double max_displacement, float* disp, void* stream) {
    return vbs_displacement_range(h, table, n, m_ref, warmup_frames, min_marker_size_px, max_displacement, 0, n, disp,
                                  stream);
}

extern "C" int vbs_displacement_f64(int device, const double* table, int n, int m_ref, int warmup_frames,
                                    double min_marker_size_px, double max_displacement, double* disp,
                                    void* stream) {
    if (!table || !disp || n < 0 || m_ref < 1) return VBS_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return VBS_EHIP;
    if (n) {
        int* cell = nullptr;                            // first-frame cell: a small stream-ordered allocation
        if (hipMallocAsync((void**)&cell, sizeof(int), (hipStream_t)stream) != hipSuccess) return VBS_ENOMEM;
        launch_displacement64(table, n, m_ref, warmup_frames, min_marker_size_px, max_displacement, disp, cell,
                              (hipStream_t)stream);
        (void)hipFreeAsync(cell, (hipStream_t)stream);
    }
    return hipGetLastError() == hipSuccess ? VBS_OK : VBS_EHIP;
}

extern "C" int vbs_plane_fit(vbs_handle* h, const float* table, int n, int m_ref, float* plane, void* stream) {
    if (!h) return VBS_EINVAL;
    if (!table || !plane || n < 0 || m_ref < 1) { h->err = "vbs_plane_fit: bad argument"; return VBS_EINVAL; }
    HIPCHK(h, hipSetDevice(h->device));
    if (n) launch_plane_fit(h, table, n, m_ref, plane, (hipStream_t)stream);
    return check_launch(h);
}

extern "C" int vbs_deviation_plane(vbs_handle* h, const float* vert_start, const float* vert_end, const float* tilt_start,
                                   const float* tilt_end, const float* ref_xyz, int m_ref, int shell_mode, double scale,
                                   float* deviation, float* out, void* stream) {
    if (!h) return VBS_EINVAL;
    if (!vert_start || !vert_end || !tilt_start || !tilt_end || !ref_xyz || !deviation || !out || m_ref < 1 ||
        (shell_mode != 0 && shell_mode != 1)) {
        h->err = "vbs_deviation_plane: bad argument";
        return VBS_EINVAL;
    }
    HIPCHK(h, hipSetDevice(h->device));
    launch_deviation_plane(h, vert_start, vert_end, tilt_start, tilt_end, ref_xyz, m_ref, shell_mode, scale, deviation, out,
                           (hipStream_t)stream);
    return check_launch(h);
}

extern "C" int vbs_assign_ids(vbs_handle* h, const double* det, const int32_t* count, int num_layers, int id_mode,
                              int32_t* ids, double* ref_xy, int cap, int32_t* m_out, void* stream) {
    if (!h) return VBS_EINVAL;
    if (!det || !count || !ids || !ref_xy || !m_out || num_layers < 1 || cap < 1 || (id_mode != 0 && id_mode != 1)) {
        h->err = "vbs_assign_ids: bad argument";
        return VBS_EINVAL;
    }
    HIPCHK(h, hipSetDevice(h->device));
    launch_assign_ids(h, det, count, num_layers, id_mode, ids, ref_xy, cap, m_out, (hipStream_t)stream);
    return check_launch(h);
}

#ifdef VBS_DEBUG_KNOBS
// tools/ builds only: k_stage_lat's per-frame header words of the last pass (phase stamps in words 80..), synchronising
extern "C" int vbs_debug_lat_hdr(vbs_handle* h, uint32_t* out, int frames) {
    if (!h || !out || frames < 1 || frames > VBS_LAT_MAXN) return VBS_EINVAL;
    vbs_handle* w = last_ws(h);
    HIPCHK(h, hipDeviceSynchronize());
    HIPCHK(h, hipMemcpy(out, w->lat_hdr, (size_t)frames * VBS_LAT_HDR * sizeof(u32), hipMemcpyDeviceToHost));
    return VBS_OK;
}
#endif
