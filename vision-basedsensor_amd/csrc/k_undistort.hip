// a2 / f3: frame undistortion — MarkerTracker._undistort_frame (marker_detection.py:93-109):
//   getOptimalNewCameraMatrix(K, D, (w,h), alpha=0) -> initUndistortRectifyMap(..., CV_16SC2) -> remap(INTER_LINEAR).
// The reference rebuilds the maps for every frame; here they depend only on (K, D, size) and are built once per
// handle (vbs_set_undistort).  Arithmetic follows oracle/stages.py (OpenCV's fixed point: source position in 1/32 px,
// bilinear weights in 1/32768 with the rounding residue pushed onto one tap, (sum + 2^14) >> 15, BORDER_CONSTANT 0).
#include <cmath>

#include "common.h"

#pragma clang fp contract(off)

// ---- host: new camera matrix for alpha = 0 (inner rectangle of a 9x9 undistorted grid) ----------------
static void undistort_norm(double u, double v, const double* K, const double* k, double* xo, double* yo) {
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double x0 = (u - cx) / fx, y0 = (v - cy) / fy, x = x0, y = y0;
    for (int it = 0; it < 5; ++it) {
        double r2 = x * x + y * y;
        double icd = 1.0 / (1.0 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        double dxx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
        double dyy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
        x = (x0 - dxx) * icd;
        y = (y0 - dyy) * icd;
    }
    *xo = x; *yo = y;
}

void optimal_new_camera_matrix_alpha0(const double* K, const double* k, int w, int h, double* newK) {
    const int N = 9;
    double ix0 = -1e300, ix1 = 1e300, iy0 = -1e300, iy1 = 1e300;
    for (int y = 0; y < N; ++y)
        for (int x = 0; x < N; ++x) {
            float pu = (float)((double)x * (w - 1) / (N - 1)), pv = (float)((double)y * (h - 1) / (N - 1));
            double xn, yn;
            undistort_norm((double)pu, (double)pv, K, k, &xn, &yn);
            xn = (double)(float)xn; yn = (double)(float)yn;       // the grid lives in CV_32FC2
            if (x == 0) ix0 = std::fmax(ix0, xn);
            if (x == N - 1) ix1 = std::fmin(ix1, xn);
            if (y == 0) iy0 = std::fmax(iy0, yn);
            if (y == N - 1) iy1 = std::fmin(iy1, yn);
        }
    double fx0 = (w - 1) / (ix1 - ix0), fy0 = (h - 1) / (iy1 - iy0);
    for (int i = 0; i < 9; ++i) newK[i] = 0;
    newK[0] = fx0; newK[4] = fy0; newK[2] = -fx0 * ix0; newK[5] = -fy0 * iy0; newK[8] = 1.0;
}

// ---- host: fixed-point bilinear weights, [1024][4], every row sums to 2^15 -----------------------------
void bilinear_weights_i16(int32_t* out) {
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float ty0 = 1.0f - (float)i / 32.0f, ty1 = (float)i / 32.0f, tx0 = 1.0f - (float)j / 32.0f, tx1 = (float)j / 32.0f;
            float wv[4] = {ty0 * tx0, ty0 * tx1, ty1 * tx0, ty1 * tx1};
            int iw[4], sum = 0;
            for (int q = 0; q < 4; ++q) { iw[q] = (int)std::nearbyint((double)wv[q] * 32768.0); sum += iw[q]; }
            if (sum != 32768) {
                int diff = sum - 32768, pick = 0;
                for (int q = 1; q < 4; ++q)
                    if (diff < 0 ? iw[q] > iw[pick] : iw[q] < iw[pick]) pick = q;     // first max / first min
                iw[pick] -= diff;
            }
            for (int q = 0; q < 4; ++q) out[(i * 32 + j) * 4 + q] = iw[q];
        }
}

struct UndistParams {
    double K[9], k[5], ir[9];
};

__global__ __launch_bounds__(256) void k_undist_map(short2* __restrict__ map1, unsigned short* __restrict__ map2,
                                                    int H, int W, UndistParams p) {
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    double jj = (double)x, ii = (double)y;
    double X = jj * p.ir[0] + ii * p.ir[1] + p.ir[2];
    double Y = jj * p.ir[3] + ii * p.ir[4] + p.ir[5];
    double Wc = jj * p.ir[6] + ii * p.ir[7] + p.ir[8];
    double xn = X / Wc, yn = Y / Wc;
    double x2 = xn * xn, y2 = yn * yn, r2 = x2 + y2, _2xy = 2 * xn * yn;
    double kr = (1 + ((p.k[4] * r2 + p.k[1]) * r2 + p.k[0]) * r2);
    double xd = xn * kr + p.k[2] * _2xy + p.k[3] * (r2 + 2 * x2);
    double yd = yn * kr + p.k[2] * (r2 + 2 * y2) + p.k[3] * _2xy;
    double u = p.K[0] * xd + p.K[2], v = p.K[4] * yd + p.K[5];
    double ru = rint(u * 32.0), rv = rint(v * 32.0);
    ru = fmin(fmax(ru, -2147483648.0), 2147483647.0);
    rv = fmin(fmax(rv, -2147483648.0), 2147483647.0);
    long long iu = (long long)ru, iv = (long long)rv;
    long long sx = iu >> 5, sy = iv >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    map1[(int64_t)y * W + x] = make_short2((short)sx, (short)sy);
    map2[(int64_t)y * W + x] = (unsigned short)((iv & 31) * 32 + (iu & 31));
}

// remap (+ BGR->gray when gray_out): out is either uint8 [n,H,W,ch] dense (the drop-in `_undistort_frame`) or the
// handle's gray buffer [n,H,P] that feeds the blur
__global__ __launch_bounds__(256) void k_remap(const u8* __restrict__ frames, int channels, int64_t stride_n,
                                               int64_t stride_row, const short2* __restrict__ map1,
                                               const unsigned short* __restrict__ map2, const int* __restrict__ wtab,
                                               u8* __restrict__ out, int to_gray, int H, int W, int P, GrayCoef gc) {
    int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, n = blockIdx.z;
    if (x >= W) return;
    short2 m = map1[(int64_t)y * W + x];
    const int* wt = wtab + 4 * (int)map2[(int64_t)y * W + x];
    const int w00 = wt[0], w01 = wt[1], w10 = wt[2], w11 = wt[3];
    const int sx = m.x, sy = m.y;
    const u8* src = frames + (int64_t)n * stride_n;
    const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W, y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
    u32 val[3] = {0, 0, 0};
    for (int c = 0; c < channels; ++c) {
        int p00 = (y0 && x0) ? src[(int64_t)sy * stride_row + (int64_t)sx * channels + c] : 0;
        int p01 = (y0 && x1) ? src[(int64_t)sy * stride_row + (int64_t)(sx + 1) * channels + c] : 0;
        int p10 = (y1 && x0) ? src[(int64_t)(sy + 1) * stride_row + (int64_t)sx * channels + c] : 0;
        int p11 = (y1 && x1) ? src[(int64_t)(sy + 1) * stride_row + (int64_t)(sx + 1) * channels + c] : 0;
        val[c] = (u32)((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15);
    }
    if (to_gray) {
        u32 g = channels == 1 ? val[0] : (gc.cb * val[0] + gc.cg * val[1] + gc.cr * val[2] + gc.half) >> gc.shift;
        out[((int64_t)n * H + y) * P + x] = (u8)g;
    } else {
        for (int c = 0; c < channels; ++c) out[(((int64_t)n * H + y) * W + x) * channels + c] = (u8)val[c];
    }
}

int setup_undistort(vbs_handle* h, const double* K9, const double* dist, int ndist, hipStream_t s) {
    UndistParams p;
    for (int i = 0; i < 9; ++i) p.K[i] = K9[i];
    for (int i = 0; i < 5; ++i) p.k[i] = i < ndist ? dist[i] : 0.0;
    double nk[9];
    optimal_new_camera_matrix_alpha0(p.K, p.k, h->W, h->H, nk);
    for (int i = 0; i < 9; ++i) h->newK[i] = nk[i];
    // inverse of the (upper-triangular, zero-skew) new camera matrix
    for (int i = 0; i < 9; ++i) p.ir[i] = 0;
    p.ir[0] = 1.0 / nk[0]; p.ir[4] = 1.0 / nk[4]; p.ir[2] = -nk[2] / nk[0]; p.ir[5] = -nk[5] / nk[4]; p.ir[8] = 1.0;
    if (!(std::isfinite(p.ir[0]) && std::isfinite(p.ir[4]))) { h->err = "degenerate undistortion"; return VBS_EINVAL; }
    dim3 grid((h->W + 255) / 256, h->H);
    VBS_LAUNCH(h, s, "k_undist_map", k_undist_map, grid, dim3(256), 0, s, (short2*)h->umap1, h->umap2, h->H, h->W, p);
    return VBS_OK;
}

void launch_remap(vbs_handle* h, const u8* frames, int nb, int channels, int64_t stride_n, int64_t stride_row,
                  u8* out, int to_gray, hipStream_t s) {
    dim3 grid((h->W + 255) / 256, h->H, nb);
    VBS_LAUNCH(h, s, "k_remap", k_remap, grid, dim3(256), 0, s, frames, channels, stride_n, stride_row,
               (const short2*)h->umap1, (const unsigned short*)h->umap2, (const int*)h->uwtab, out, to_gray, h->H, h->W,
               h->P, gray_coef(h->gray_bits));
}
