// a6-a8: normalised cross-correlation of area_mask with the Gaussian template, thresholded at 0.1.
// Reference: marker_detection.py:132-133 (_normxcorr2 :146-164, _gkern :138-143).
//
// The reference evaluates three FFT convolutions in float64.  Here the same quantity is evaluated
// in the spatial domain, also in float64 (see oracle/stages.py:normxcorr2_direct for the algebra):
// the image is two-valued (I in {0,255} so I^2 = 255 I), the template is separable (t = g (x) g) and
// zero padding applies to the mean-subtracted image, so with c = #foreground and n = #in-image samples
// of the l x l window
//     num = 255 G - tbar 255 c - mu (Rx Ry - n tbar),     G = sum_i g_i sum_j g_j b(y+i, x+j)
//     var = 255^2 c - 2 mu 255 c + n mu^2 - (255 c - n mu)^2 / l^2
//     mask = num / sqrt(var * T2) > 0.1   <=>   var > 0, num > 0, num^2 > 0.01 var T2
// k_ncc_h: horizontal Gaussian pass + horizontal box count straight from the packed bits.
// k_ncc_v: vertical pass, decision, ballot -> packed mask bits (and optional uint8 mask).
#include "common.h"

__device__ __forceinline__ u64 load_bits(const u64* __restrict__ row, int WW, int start) {
    int wi = start >> 6, sh = start & 63;
    u64 a = (wi >= 0 && wi < WW) ? row[wi] : 0ull;
    u64 b = (wi + 1 >= 0 && wi + 1 < WW) ? row[wi + 1] : 0ull;
    return sh ? ((a >> sh) | (b << (64 - sh))) : a;
}

template <int L, int LO>
__global__ __launch_bounds__(256) void k_ncc_h(const u64* __restrict__ bits, double* __restrict__ hx,
                                               u8* __restrict__ cxo, int H, int P, int WW,
                                               NccConst nc) {
    const int groups = P / 8;
    int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= groups * H) return;
    int y = gid / groups, xg = gid - y * groups;
    int n = blockIdx.y;
    int x0 = xg * 8;
    const u64* row = bits + ((int64_t)n * H + y) * WW;
    u64 w0 = load_bits(row, WW, x0 + LO);
    u64 w1 = load_bits(row, WW, x0 + LO + 64);
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (w0 | w1) {
#pragma unroll
        for (int i = 0; i < L + 7; ++i) {
            u32 bit = (i < 64) ? (u32)((w0 >> i) & 1ull) : (u32)((w1 >> (i - 64)) & 1ull);
            double bd = (double)bit;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int j = i - s;
                if (j >= 0 && j < L) acc[s] = __builtin_fma(nc.g[j], bd, acc[s]);
            }
        }
    }
    int64_t o = ((int64_t)n * H + y) * P + x0;
    u64 packed = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        u64 lo = s ? ((w0 >> s) | (w1 << (64 - s))) : w0;
        u32 c;
        if (L >= 64) {
            c = __popcll(lo) + __popcll((w1 >> s) & ((1ull << (L - 64)) - 1ull));
        } else {
            c = __popcll(lo & ((1ull << L) - 1ull));
        }
        packed |= (u64)c << (8 * s);
        hx[o + s] = acc[s];
    }
    *reinterpret_cast<u64*>(cxo + o) = packed;
}

template <int L, int LO>
__global__ __launch_bounds__(256) void k_ncc_v(const double* __restrict__ hx, const u8* __restrict__ cxi,
                                               const double* __restrict__ rx,
                                               const double* __restrict__ ry, u64* __restrict__ mbits,
                                               u8* __restrict__ mask_u8, double* __restrict__ ncc_out,
                                               u32* __restrict__ fstat, int H, int W, int P, int WW,
                                               NccConst nc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int y0 = (blockIdx.y * 4 + wave) * 8;
    const int n = blockIdx.z;
    if (y0 >= H) return;
    constexpr int HI = L - 1 + LO;
    const double* hcol = hx + (int64_t)n * H * P + x;
    const u8* ccol = cxi + (int64_t)n * H * P + x;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u32 cs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < L + 7; ++i) {
        int yy = y0 + LO + i;
        double v = 0.0;
        u32 c = 0;
        if (yy >= 0 && yy < H) {
            v = hcol[(int64_t)yy * P];
            c = ccol[(int64_t)yy * P];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int j = i - s;
            if (j >= 0 && j < L) {
                acc[s] = __builtin_fma(nc.g[j], v, acc[s]);
                cs[s] += c;
            }
        }
    }
    const double mu = (double)(255ull * (u64)fstat[n * 8 + 0]) / (double)((int64_t)H * W);
    u32 amb = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        int y = y0 + s;
        bool pred = false;
        if (y < H && x < W) {
            int ny = min(y + HI, H - 1) - max(y + LO, 0) + 1;
            int nx = min(x + HI, W - 1) - max(x + LO, 0) + 1;
            double nn = (double)(ny * nx);
            double sum_t = ry[y] * rx[x];
            double sum_I = 255.0 * (double)cs[s];
            double num = 255.0 * acc[s] - nc.tbar * sum_I - mu * (sum_t - nn * nc.tbar);
            double s1 = sum_I - nn * mu;
            double s2 = 255.0 * sum_I - 2.0 * mu * sum_I + nn * mu * mu;
            double var = s2 - s1 * s1 / nc.l2;
            double rhs = nc.thr2 * var * nc.T2;
            pred = (var > 0.0) && (num > 0.0) && (num * num > rhs);
            if (var > 1e-6 && num > 0.0 && fabs(num * num - rhs) <= 1e-9 * rhs) amb++;
            if (ncc_out) {                               // the reference's value: non-finite -> 0 (:162-163)
                double q = num / sqrt((var < 0.0 ? 0.0 : var) * nc.T2);
                ncc_out[((int64_t)n * H + y) * W + x] = isfinite(q) ? q : 0.0;
            }
        }
        u64 word = __ballot(pred);
        if (y < H) {
            if (lane == 0) mbits[((int64_t)n * H + y) * WW + blockIdx.x] = word;
            if (mask_u8 && x < W) mask_u8[((int64_t)n * H + y) * W + x] = pred ? 1 : 0;
        }
    }
    if (amb) atomicAdd(&fstat[n * 8 + 1], amb);
}

// area popcount per frame (feeds the global mean of _normxcorr2 :153) when the bits did not come from k_blur_v
__global__ __launch_bounds__(256) void k_popcount(const u64* __restrict__ bits, u32* __restrict__ fstat, int NW) {
    __shared__ u32 part[4];
    const int n = blockIdx.x;
    u32 c = 0;
    for (int i = threadIdx.x; i < NW; i += 256) c += __popcll(bits[(int64_t)n * NW + i]);
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) fstat[n * 8 + 0] = part[0] + part[1] + part[2] + part[3];
}

void launch_popcount(vbs_handle* h, int nb, hipStream_t s) {
    VBS_LAUNCH(h, s, "k_popcount", k_popcount, dim3(nb), dim3(256), 0, s, h->area_bits, h->fstat, h->H * h->WW);
}

void launch_ncc(vbs_handle* h, int nb, u8* mask_u8, double* ncc_out, hipStream_t s) {
    dim3 gh(((h->P / 8) * h->H + 255) / 256, nb);
    dim3 gv(h->WW, (h->H + 31) / 32, nb);
    if (!h->bp.small) {
        VBS_LAUNCH(h, s, "k_ncc_h", (k_ncc_h<80, -40>), gh, dim3(256), 0, s, h->area_bits, h->hx, h->cx, h->H, h->P,
                           h->WW, h->ncc);
        VBS_LAUNCH(h, s, "k_ncc_v", (k_ncc_v<80, -40>), gv, dim3(256), 0, s, h->hx, h->cx, h->ncc_rx, h->ncc_ry,
                           h->mask_bits, mask_u8, ncc_out, h->fstat, h->H, h->W, h->P, h->WW, h->ncc);
    } else {
        VBS_LAUNCH(h, s, "k_ncc_h", (k_ncc_h<33, -16>), gh, dim3(256), 0, s, h->area_bits, h->hx, h->cx, h->H, h->P,
                           h->WW, h->ncc);
        VBS_LAUNCH(h, s, "k_ncc_v", (k_ncc_v<33, -16>), gv, dim3(256), 0, s, h->hx, h->cx, h->ncc_rx, h->ncc_ry,
                           h->mask_bits, mask_u8, ncc_out, h->fstat, h->H, h->W, h->P, h->WW, h->ncc);
    }
}
