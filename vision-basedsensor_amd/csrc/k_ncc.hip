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
// k_ncc: both passes in one kernel, the horizontal one from bit runs into LDS (no float64 image in HBM).
#include <cstdlib>

#include "common.h"

__device__ __forceinline__ u64 load_bits(const u64* __restrict__ row, int WW, int start) {
    int wi = start >> 6, sh = start & 63;
    u64 a = (wi >= 0 && wi < WW) ? row[wi] : 0ull;
    u64 b = (wi + 1 >= 0 && wi + 1 < WW) ? row[wi + 1] : 0ull;
    return sh ? ((a >> sh) | (b << (64 - sh))) : a;
}

// Horizontal Gaussian sum of one pixel from the runs of 1-bits in its window (float64): a run [b, e) in
// window coordinates contributes CG[e] - CG[b], CG the cumulative template factor.
template <int L, int LO>
__device__ __forceinline__ double ncc_row_exact(const u64* __restrict__ row, int WW, int x, const double* cg,
                                                u32* cnt) {
    u64 w0 = load_bits(row, WW, x + LO), w1 = 0;
    if (L > 64) w1 = load_bits(row, WW, x + LO + 64) & ((1ull << (L > 64 ? L - 64 : 1)) - 1ull);
    else w0 &= (1ull << (L < 64 ? L : 0)) - 1ull;
    *cnt += __popcll(w0) + __popcll(w1);
    double h = 0.0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        u64 w = half ? w1 : w0;
        while (w) {
            int b0 = __ffsll((long long)w) - 1;
            u64 t = ~(w >> b0);
            int len = t ? __ffsll((long long)t) - 1 : 64 - b0;
            w &= (len >= 64) ? 0ull : ~(((1ull << len) - 1ull) << b0);
            h += cg[half * 64 + b0 + len] - cg[half * 64 + b0];
        }
    }
    return h;
}

// Exact float64 G(y, x) = sum_i g[i] H(y + LO + i, x), products added in ascending i.  Rare path (pixels the float32
// filter cannot decide, or the diagnostic map): kept out of line so that its loops are not replicated 8x.
template <int L, int LO>
__device__ __attribute__((noinline)) double ncc_exact_G(const u64* fbits, int H, int WW, int y, int x, const double* cg,
                                                        const double* gsh) {
    u32 dummy = 0;
    double Ge = 0.0;
    for (int i = 0; i < L; ++i) {
        int yy = y + LO + i;
        double hrow = (yy >= 0 && yy < H) ? ncc_row_exact<L, LO>(fbits + (int64_t)yy * WW, WW, x, cg, &dummy) : 0.0;
        Ge = __builtin_fma(gsh[i], hrow, Ge);
    }
    return Ge;
}

// One workgroup = 64 columns x 64 output rows.
// Phase 1 fills LDS with the horizontal pass of the 64+L-1 rows the tile needs, 8 px per work item from one
// shared bit window, computed in float64 from runs (2 table lookups per run instead of L multiply-adds) and
// stored as float32.  Phase 2 is the vertical pass out of LDS in float32 (tap-outer, 8 rows per lane).
// float32 is only a filter: with e = 1e-5 bounding the relative error of the float32 sum (80 positive
// products, worst case (L+2) 2^-24 = 4.9e-6 plus the two input roundings), a pixel whose decision is the
// same for G (1 - e) and G (1 + e) is decided; the others (a handful per frame, on the ncc = 0.1 contour)
// recompute G in float64 straight from the bits (ncc_row_exact), so every decision equals the float64 one.
template <int L, int LO>
__global__ __launch_bounds__(256) void k_ncc(const u64* __restrict__ bits, const double* __restrict__ rx,
                                             const double* __restrict__ ry, u64* __restrict__ mbits,
                                             u8* __restrict__ mask_u8, double* __restrict__ ncc_out,
                                             u32* __restrict__ fstat, int H, int W, int WW, int stop, NccConst nc) {
    constexpr int RT = 64, HR = RT + L - 1, HI = L - 1 + LO;
    __shared__ float hxs[HR][64];
    __shared__ __attribute__((aligned(8))) u8 cxs[HR][64];
    __shared__ double cg[L + 1];
    __shared__ double gsh[L];
    __shared__ float g32[L];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * 64, yb = blockIdx.y * RT, n = blockIdx.z;
    const u64* fbits = bits + (int64_t)n * H * WW;
    for (int i = tid; i <= L; i += 256) cg[i] = nc.cg[i];
    for (int i = tid; i < L; i += 256) { gsh[i] = nc.g[i]; g32[i] = (float)nc.g[i]; }
    __syncthreads();
    // phase 1: work item = (row r, 8 consecutive columns); one 64+(L+7-64)-bit window serves all 8
    for (int p = tid; p < HR * 8; p += 256) {
        const int r = p >> 3, c8 = p & 7;
        const int y = yb + LO + r, xs = x0 + 8 * c8;
        double h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        u64 packed = 0;
        if (y >= 0 && y < H) {
            const u64* row = fbits + (int64_t)y * WW;
            const u64 w0 = load_bits(row, WW, xs + LO);
            const u64 w1 = load_bits(row, WW, xs + LO + 64) & ((1ull << (L + 7 - 64 > 0 ? L + 7 - 64 : 1)) - 1ull) &
                           (L + 7 > 64 ? ~0ull : 0ull);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                u64 lo = s ? ((w0 >> s) | (w1 << (64 - s))) : w0;
                u32 c;
                if (L >= 64) c = __popcll(lo) + __popcll((w1 >> s) & ((1ull << (L >= 64 ? L - 64 : 0)) - 1ull));
                else c = __popcll(lo & ((1ull << (L < 64 ? L : 0)) - 1ull));
                packed |= (u64)c << (8 * s);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                u64 w = half ? w1 : ((L + 7 >= 64) ? w0 : (w0 & ((1ull << ((L + 7) & 63)) - 1ull)));
                const int off = half * 64;
                while (w) {
                    int b0 = __ffsll((long long)w) - 1;
                    u64 t = ~(w >> b0);
                    int len = t ? __ffsll((long long)t) - 1 : 64 - b0;
                    w &= (len >= 64) ? 0ull : ~(((1ull << len) - 1ull) << b0);
                    const int rb = off + b0, re = rb + len;     // run [rb, re) in window coordinates
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        // clip the run to window s = [s, s+L); an empty intersection gives cg[k] - cg[k] = 0
                        int lo_ = min(max(rb, s), s + L), hi_ = max(min(re, s + L), lo_);
                        h[s] += cg[hi_ - s] - cg[lo_ - s];
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) hxs[r][8 * c8 + s] = (float)h[s];
        *reinterpret_cast<u64*>(&cxs[r][8 * c8]) = packed;
    }
    __syncthreads();
    if (stop == 1) return;
    const int x = x0 + lane;
    const double mu = (double)(255ull * (u64)fstat[n * 8 + 0]) / (double)((int64_t)H * W);
    u32 amb = 0, nexact = 0;
    for (int oct = 0; oct < RT / 32; ++oct) {
        const int r0 = wave * (RT / 4) + oct * 8;        // first LDS row of this lane's 8 output rows
        const int y0 = yb + r0;
        if (y0 >= H) break;                              // wave-uniform
        const bool interior = (y0 + LO >= 0) && (y0 + 7 + HI <= H - 1) && (x0 + LO >= 0) && (x0 + 63 + HI <= W - 1);
        const double full_t = ry[min(max(-LO, 0), H - 1)] * rx[min(max(-LO, 0), W - 1)];   // rows / columns with a full window
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        float vw[8];
#pragma unroll
        for (int i = 0; i < 7; ++i) vw[i] = hxs[r0 + i][lane];
#pragma unroll 1
        for (int jb = 0; jb < L / 8; ++jb) {             // rolled: keeps the weights' live ranges to one block
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = 8 * jb + jj;
                vw[(jj + 7) & 7] = hxs[r0 + j + 7][lane];
                const float gj = g32[j];                 // LDS broadcast read
#pragma unroll
                for (int s = 0; s < 8; ++s) acc[s] = __builtin_fmaf(gj, vw[(jj + s) & 7], acc[s]);
            }
        }
#pragma unroll
        for (int j = (L / 8) * 8; j < L; ++j) {          // tail taps (L = 33)
            vw[(j + 7) & 7] = hxs[r0 + j + 7][lane];
            const float gj = g32[j];
#pragma unroll
            for (int s = 0; s < 8; ++s) acc[s] = __builtin_fmaf(gj, vw[(j + s) & 7], acc[s]);
        }
        u32 cs0 = 0, pre[8] = {0, 0, 0, 0, 0, 0, 0, 0}, post[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < L + 7; ++i) {
            const u32 c = cxs[r0 + i][lane];
            if (i < L) cs0 += c;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (i < s) pre[s] += c;                  // rows above window s
                if (i >= L && i < L + s) post[s] += c;   // rows that window s gains
            }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int y = y0 + s;
            bool pred = false;
            if (y < H && x < W) {
                double nn, sum_t;
                if (interior) {                          // wave-uniform: every window of these 8 rows x 64 columns is
                    nn = nc.l2; sum_t = full_t;          // inside the image, so n = l*l and sum_W t is the full sum
                } else {
                    int ny = min(y + HI, H - 1) - max(y + LO, 0) + 1;
                    int nx = min(x + HI, W - 1) - max(x + LO, 0) + 1;
                    nn = (double)(ny * nx);
                    sum_t = ry[y] * rx[x];
                }
                double sum_I = 255.0 * (double)(cs0 - pre[s] + post[s]);
                double rest = -nc.tbar * sum_I - mu * (sum_t - nn * nc.tbar);       // num = 255 G + rest
                double s1 = sum_I - nn * mu;
                double s2 = 255.0 * sum_I - 2.0 * mu * sum_I + nn * mu * mu;
                double var = s2 - s1 * s1 * nc.inv_l2;
                double rhs = nc.thr2 * var * nc.T2;
                if (var > 0.0) {
                    double G = (double)acc[s];
                    double nlo = 255.0 * G * (1.0 - 1e-5) + rest, nhi = 255.0 * G * (1.0 + 1e-5) + rest;
                    bool plo = (nlo > 0.0) && (nlo * nlo > rhs), phi = (nhi > 0.0) && (nhi * nhi > rhs);
                    pred = plo;
                    if (plo != phi || ncc_out) {          // undecided by float32 (or a map was asked for): exact
                        const double Ge = ncc_exact_G<L, LO>(fbits, H, WW, y, x, cg, gsh);
                        double num = 255.0 * Ge + rest;
                        pred = (num > 0.0) && (num * num > rhs);
                        if (var > 1e-6 && num > 0.0 && fabs(num * num - rhs) <= 1e-9 * rhs) amb++;
                        nexact++;
                        if (ncc_out) {                   // diagnostic map in the reference's form (:159-163)
                            double v2 = s2 - s1 * s1 / nc.l2;
                            double q = num / sqrt((v2 < 0.0 ? 0.0 : v2) * nc.T2);
                            ncc_out[((int64_t)n * H + y) * W + x] = isfinite(q) ? q : 0.0;
                        }
                    }
                } else if (ncc_out) {
                    ncc_out[((int64_t)n * H + y) * W + x] = 0.0;      // 0/0 or x/0 -> non-finite -> 0 (:163)
                }
            }
            u64 word = __ballot(pred);
            if (y < H) {
                if (lane == 0) mbits[((int64_t)n * H + y) * WW + blockIdx.x] = word;
                if (mask_u8 && x < W) mask_u8[((int64_t)n * H + y) * W + x] = pred ? 1 : 0;
            }
        }
    }
    if (amb) atomicAdd(&fstat[n * 8 + 1], amb);
    if (nexact) atomicAdd(&fstat[n * 8 + 3], nexact);
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
    dim3 grid(h->WW, (h->H + 63) / 64, nb);
    const int stop = getenv("VBS_NCC_STOP") ? atoi(getenv("VBS_NCC_STOP")) : 0;   // debug: phase timing
    if (!h->bp.small) {
        VBS_LAUNCH(h, s, "k_ncc", (k_ncc<80, -40>), grid, dim3(256), 0, s, h->area_bits, h->ncc_rx, h->ncc_ry,
                   h->mask_bits, mask_u8, ncc_out, h->fstat, h->H, h->W, h->WW, stop, h->ncc);
    } else {
        VBS_LAUNCH(h, s, "k_ncc", (k_ncc<33, -16>), grid, dim3(256), 0, s, h->area_bits, h->ncc_rx, h->ncc_ry,
                   h->mask_bits, mask_u8, ncc_out, h->fstat, h->H, h->W, h->WW, stop, h->ncc);
    }
}
