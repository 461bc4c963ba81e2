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
// k_ncc_mfma (the hot path): both passes as Toeplitz products on the float16 matrix cores, a filter in front of the
//             float64 decision.  k_ncc: the float64 MAP of the diagnostic APIs, horizontal pass from bit runs into LDS.
#include <algorithm>
#include <cstdlib>
#include <cstring>

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

// ---- matrix-core path ------------------------------------------------------------------------------
// G and the window count c are banded-Toeplitz products, like the blurs (k_blur.hip), here in float16 operands
// with float32 accumulation on v_mfma_f32_16x16x32_f16:
//   horizontal  h[y][x]  = sum_k b[y][xw+k] * w[k - x]          A = image bits expanded to 0.0 / 1.0 (exact),
//               ch[y][x] = sum_k b[y][xw+k] * 1[k - x]          B = Toeplitz of w = 1024 g, split w = whi + wlo
//   vertical    G[y][x]  = sum_k w[k - y] * h[yw+k][x]           A = the same Toeplitz fragments, B = h split into
//                                                                float16 hi + lo (products hi*hi, hi*lo, lo*hi)
//               c[y][x]  = sum_k 1[k - y] * ch[yw+k][x]          int8 (v_mfma_i32_16x16x64_i8): ch <= l fits a byte
// One wave owns a 16-column strip and slides down it 16 rows per step; horizontal tiles go through a per-wave
// LDS ring stored column-major (h as float16 hi / lo, ch as bytes), so a vertical B operand is one 16-byte read.
// The ring is what bounds the waves per CU, and the kernel's time goes with 1 / waves (measured 4.4 / 3.2 / 3.0 us per
// frame at 2 / 3 / 4 waves per SIMD): the byte count ring and keeping the exact path's tables out of LDS make it four.  The result is a FILTER exactly as
// before: the relative error of G stays below 2^-16 (dropped lo*lo, float16 lo roundings, float32 accumulation of
// <= 300 positive terms), the decision is taken against theta (1 +- 2e-5) and the undecided pixels recompute G in
// float64 from the bits, so every decision equals the float64 one.
// For a window inside the image the mean terms cancel: var = 255^2 c (1 - c / l^2), hence
//   num > 0 and num^2 > rhs  <=>  G > theta(c) = sqrt(thr2 T2 c (l^2 - c)) / l + tbar c + mu (sum_t - l^2 tbar) / 255,
// three float32 operations per pixel (c (l^2 - c) < 2^24 is exact).  An empty window has G = 0 exactly on both
// paths and var = 0 up to rounding: its decision is taken once per wave with the float64 formula.  Border tiles
// (windows that leave the image) evaluate theta per pixel in float64.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define NCC_WSCALE 1024.0                               // weights are scaled so that every wlo is a normal float16
#define NCC_REL 2e-5f
#define NCC_NEVER 3e38                                  // "no G reaches this" (finite, so G - theta stays ordered)
#define NCC_ABS 1e-3f                                   // in units of G * 2^20: far below any theta with c >= 1

// theta on 2^20 G for a window with c foreground and nn in-image samples (+inf where var <= 0)
__device__ __forceinline__ double ncc_theta(double c, double nn, double sum_t, double mu, const NccConst& nc) {
    double sum_I = 255.0 * c;
    double rest = -nc.tbar * sum_I - mu * (sum_t - nn * nc.tbar);
    double s1 = sum_I - nn * mu;
    double s2 = 255.0 * sum_I - 2.0 * mu * sum_I + nn * mu * mu;
    double var = s2 - s1 * s1 * nc.inv_l2;
    double rhs = nc.thr2 * var * nc.T2;
    if (!(var > 0.0)) return NCC_NEVER;
    // an empty window has G = 0 exactly on both paths: decide it here
    if (c == 0.0) return (rest > 0.0 && rest * rest > rhs) ? -NCC_NEVER : NCC_NEVER;
    return (sqrt(rhs) - rest) * (NCC_WSCALE * NCC_WSCALE / 255.0);
}

typedef int i4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u8 gl_u8;                  // global memory, stated (see load_rows)
typedef __attribute__((address_space(1))) unsigned short gl_u16;

// Decision constants of the reference's two templates (marker_detection.py:120-121,125-126: l = 80, sigma = 13 and
// l = 33, sigma = 7.4) on 2^20 G, as instruction literals: kc = tbar 2^20, ks2 = thr2 T2 / l^2 2^40, K1 = ks2 l^2.  The
// kernel checks them against the run-time NccConst and takes the general (border) decision everywhere if they differ.
template <int L> struct NccLit;
template <> struct NccLit<80> { static constexpr float kc = 163.84f, ks2 = 547.2815085234848f, K1 = 3502601.654550303f; };
template <> struct NccLit<33> { static constexpr float kc = 962.8797061524332f, ks2 = 6956.564407359948f, K1 = 7575698.639614983f; };

// k_ncc_mfma's explicit kernel arguments as the kernarg segment lays them out (every argument at its natural alignment, in
// order).  The rare paths (drain, the counters at the end) read what they need from there at the moment they need it: an
// argument the compiler has loaded at the top of the kernel stays in scalar registers across the step loop, and this kernel
// spills scalars into vector lanes that every step then reads back (small branch: ~60 v_readlane per step).
struct NccKArgs {
    const u64* bits; const double* rx; const double* ry; const uint4* wfrag; const double* tab; const float2* rowf;
    u64* mbits; u8* mask_u8; u32* fstat; u64* tot;
    int H, W, WW, tiles_per_seg, dbg_arg; float rel_arg;
    NccConst nc;
};
typedef __attribute__((address_space(4))) const NccKArgs NccKArgsC;
__device__ __forceinline__ NccKArgsC* ncc_kargs() {
    NccKArgsC* p = (NccKArgsC*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));                          // opaque: its loads cannot be hoisted out of the rare path
    return p;
}

template <int L, int LO, bool U8OUT>                    // U8OUT: also the uint8 mask of the staged API
__global__ __launch_bounds__(256, 4) void k_ncc_mfma(const u64* __restrict__ bits, const double* __restrict__ rx,
                                                     const double* __restrict__ ry, const uint4* __restrict__ wfrag,
                                                     const double* __restrict__ tab, const float2* __restrict__ rowf,
                                                     u64* __restrict__ mbits, u8* __restrict__ mask_u8,
                                                     u32* __restrict__ fstat, u64* __restrict__ tot, int H, int W, int WW,
                                                     int tiles_per_seg,
                                                     int dbg_arg, float rel_arg, NccConst nc) {
#ifdef VBS_DEBUG_KNOBS
    const int dbg = dbg_arg;                            // tools/ phase timing and dumps
#else
    constexpr int dbg = 0;
#endif
    constexpr int HI = L - 1 + LO;
    constexpr int NT = (16 + L - 1 + 15) / 16;          // horizontal tiles under one output tile
    constexpr int NKS = (16 * NT + 31) / 32;            // k-steps of 32 (float16 products)
    constexpr int NK8 = (16 * NT + 63) / 64;            // k-steps of 64 (int8 count product)
    constexpr int RING = 16 * NT;
    constexpr int RSTR = RING + 8;                      // halves per ring column (+16 B: conflict-free 16-byte reads)
    constexpr int CSTR = RING + 16;                     // bytes per count-ring column (conflict-free as well)
    constexpr int ECAP = 2 * NT + 2;                    // tiles with undecided pixels a wave may queue before it drains them
    // 39.5 KB for l = 80: four workgroups (16 waves) per CU; the time of this kernel goes with 1 / waves per SIMD.
    // (One struct: the table first, so that a ring address minus one ring length is still a valid LDS address.)
    __shared__ struct __align__(16) {
        uint4 lut[256];                                 // byte -> eight float16 0 / 1
        _Float16 ring[4][2][16 * RSTR];
        u8 ringc[4][16 * CSTR];
        u32 elist[4][ECAP][10];                         // yo, -, uw[0..3] of a queued tile
    } sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform, and known to the compiler to be
    const int g = lane >> 4, q = lane & 15;
    const int n = blockIdx.z;
    const int tilesY = (H + 15) / 16;
    const int tile0 = blockIdx.y * tiles_per_seg;
    const int ntiles = min(tiles_per_seg, tilesY - tile0);
    if (ntiles <= 0) return;
    const int Y0 = tile0 * 16, nsteps = ntiles + NT - 1;
    const int xw = blockIdx.x * 64 + 16 * wave;         // first column of this wave's strip
    const u64* fbits = bits + (int64_t)n * H * WW;
    {
        u32 w[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) w[d] = (((u32)tid >> (2 * d)) & 1u ? 0x3C00u : 0u) | (((u32)tid >> (2 * d + 1)) & 1u ? 0x3C000000u : 0u);
        sm.lut[tid] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    for (int i = lane; i < 2 * 16 * RSTR / 8; i += 64) reinterpret_cast<uint4*>(&sm.ring[wave][0][0])[i] = make_uint4(0, 0, 0, 0);
    for (int i = lane; i < 16 * CSTR / 16; i += 64) reinterpret_cast<uint4*>(&sm.ringc[wave][0])[i] = make_uint4(0, 0, 0, 0);
    h8 whi[NKS], wlo[NKS], one[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
        uint4 a = wfrag[(0 * NKS + s) * 64 + lane], b = wfrag[(1 * NKS + s) * 64 + lane], c = wfrag[(2 * NKS + s) * 64 + lane];
        whi[s] = __builtin_bit_cast(h8, a);
        wlo[s] = __builtin_bit_cast(h8, b);
        one[s] = __builtin_bit_cast(h8, c);
    }
    // int8 Toeplitz of ones (A operand of the vertical count product): lane (g, row q), byte j <-> k = 64 s + 16 g + j
    // (from the fragment table as well: built here it cost 250 vector instructions per wave)
    i4 one8[NK8];
#pragma unroll
    for (int s = 0; s < NK8; ++s) {
        const uint4 a = wfrag[(3 * NKS + s) * 64 + lane];
        one8[s] = i4{(int)a.x, (int)a.y, (int)a.z, (int)a.w};
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // operand fragments landed (see k_blur_mfma)
    __syncthreads();

    const double mu = (double)(255ull * (u64)fstat[n * 8 + 0]) / (double)((int64_t)H * W);
    const double full_t = ry[min(max(-LO, 0), H - 1)] * rx[min(max(-LO, 0), W - 1)];
    // theta(c) = ks sqrt(c (l^2 - c)) + kc c + k0 on 2^20 G; th0 decides the empty window.  Uniform values are made so
    // explicitly and then live in scalar registers.  (As inline assembly: the builtin is folded away for a value the
    // compiler already knows to be uniform, which then stays in the vector register its float64 -> float32 conversion
    // produced.  The s_nop are the hazards the compiler pads for v_readfirstlane after a VALU write, and before a read of
    // the scalar: inside an asm they are ours.)
    auto uni = [](float v) { float o; asm("s_nop 4\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(o) : "v"(v)); return o; };
    const float kc = uni((float)(nc.tbar * (NCC_WSCALE * NCC_WSCALE)));     // (border tiles; the plain decision uses NccLit)
    const float th0 = (float)ncc_theta(0.0, nc.l2, full_t, mu, nc);
    // Loop-invariant conditions as integer bounds on the tile row (a uniform bool costs a 64-bit mask in two scalar
    // registers, and this kernel has none to spare): a tile takes the plain decision for plain_lo <= yo <= plain_hi,
    // and lies wholly inside the image for yo <= valid_hi.
    // The plain decision's constants are the literals of NccLit<L>: they must be this template's (to float32 rounding),
    // and k0 = mu (sum_t - l^2 tbar) 2^20 / 255, which the plain decision leaves out, must vanish (sum_t = l^2 tbar = 1 up
    // to rounding for a window inside the image: k0 ~ 1e-9 against a margin of >= 3e-3); th0 > 0: an empty window
    // (G = 0 and c = 0 exactly, so u = 0 and 0 > 0 fails) is background.
    bool lit_ok;
    {
        const double ksd = sqrt(nc.thr2 * nc.T2) / (double)L * (NCC_WSCALE * NCC_WSCALE);
        const double k0d = mu * (full_t - nc.l2 * nc.tbar) / 255.0 * (NCC_WSCALE * NCC_WSCALE);
        lit_ok = fabs(nc.tbar * (NCC_WSCALE * NCC_WSCALE) - (double)NccLit<L>::kc) <= 2e-7 * (double)NccLit<L>::kc &&
                 fabs(ksd * ksd - (double)NccLit<L>::ks2) <= 2e-7 * (double)NccLit<L>::ks2 && fabs(k0d) < 1e-4;
    }
    const bool th0pos = __builtin_amdgcn_readfirstlane((th0 > 0.0f && lit_ok) ? 1 : 0) != 0;
    const bool xin = (xw + LO >= 0) && (xw + 15 + HI <= W - 1);
#ifdef VBS_DEBUG_KNOBS
    const int plain_lo = dbg == 10 ? -0x40000000 : (th0pos && xin && dbg != 11) ? -LO : 0x7FFFFFFF, plain_hi = dbg == 10 ? 0x40000000 : H - 1 - 15 - HI;
#else
    const int plain_lo = (th0pos && xin) ? -LO : 0x7FFFFFFF, plain_hi = H - 1 - 15 - HI;
#endif
    const int valid_hi = (xw + 15 < W) ? H - 16 : -1;
    // constants of the border tiles' threshold
    const float muf = uni((float)mu), tbarf = uni((float)nc.tbar), ktf = uni((float)(nc.tbar * 255.0)), il2f = uni((float)nc.inv_l2);
    const float krf = uni((float)(nc.thr2 * nc.T2));
    const float mukf = uni((float)(mu / 255.0 * (NCC_WSCALE * NCC_WSCALE)));
    // per-lane (column) factors of a border window, and the window sums of a column whose rows are all inside
    const int xq = min(xw + q, W - 1);
    const float nxf = (float)(min(xq + HI, W - 1) - max(xq + LO, 0) + 1), rxf = (float)rx[xq];
    const float nnc = (float)L * nxf, stc = (float)ry[min(max(-LO, 0), H - 1)] * rxf, dd0c = stc - nnc * tbarf;
    // Ring addresses.  The step loop is unrolled over the NT ring slots, so every slot offset is a constant of its copy
    // of the body and folds into the LDS instruction; what depends on the lane is kept in these pointers.  A k-step of
    // 32 ring rows that starts in the last slot wraps to slot 0 for lane groups 2 and 3: they use rdw.
    _Float16* const ringw = &sm.ring[wave][0][q * RSTR];
    _Float16* const wr = ringw + 4 * g;                 // + 16 slot: this lane's four rows of a new tile
    const _Float16* const rd = ringw + 8 * g;           // + first row of the k-step: eight rows of column q
    const _Float16* const rdw = rd - (g >= 2 ? RING : 0);
    u8* const cw = &sm.ringc[wave][q * CSTR] + 4 * g;
    const u8* const cbase = &sm.ringc[wave][q * CSTR];
    constexpr int LOFS = 16 * RSTR;                     // hi plane -> lo plane, in halves
    u32 amb = 0, nexact = 0;
    unsigned short* mb16 = reinterpret_cast<unsigned short*>(mbits) + ((int64_t)n * H * WW + blockIdx.x) * 4 + wave;   // (uniform)
    // Row bits of a horizontal tile: lane (g, q) needs, of row q, the BYTE (wstart + 32 s) / 8 + g for every k-step s - its
    // eight columns of the 32 - and nothing else (wstart is a multiple of 8: xw of 16, LO of 8).  Strips whose window lies
    // inside the row load the aligned 16 bytes around it with ONE load at lane offset q * rowbytes from a SCALAR base that
    // moves down the strip (no vector instruction per step for the address), issued a step ahead; the byte sits at
    // position c = (b0 & 3) + g + 4 s of the 16, i.e. byte c (< 8, the same in every step) of the dword pair (s + 1 : s):
    // one v_perm per k-step with a per-lane selector puts it into byte 0 over zeros - no alignbit, no bfe.  (Loading the
    // NKS bytes themselves - three byte loads at the lane's own byte offset, nothing to extract - was measured 12 % slower:
    // three times the address work in the texture path)  The first / last strips of a row go through
    // clamped 64-bit loads and build the same 16-byte form.
    const int wstart = xw + LO;                          // wave-uniform
    static_assert((LO & 7) == 0, "the window must start on a byte of the row");
    const int rowbytes = 8 * WW, b0 = wstart >> 3;
    // (uniform conditions as INTEGERS behind an opaque move: as booleans the compiler keeps each of them - and every
    //  bounds test of the clamped loads below, hoisted out of the step loop - as a 64-bit lane mask in two scalar
    //  registers, which this kernel then spills and reloads with v_readlane in every step)
    int wide_i = __builtin_amdgcn_readfirstlane(((b0 >= 0) && ((b0 & ~3) + 16 <= rowbytes)) ? 1 : 0);
    asm volatile("" : "+s"(wide_i));
#define wide (wide_i != 0)
    uint4 nraw = make_uint4(0, 0, 0, 0);
    const u8* fby = reinterpret_cast<const u8*>(fbits) + (b0 & ~3);     // (uniform; only dereferenced by wide strips)
    const u32 lofs = (u32)(q * rowbytes);
    const u32 psel = 0x0C0C0C00u | (u32)((b0 & 3) + g);
    auto load_rows = [&](int t) {
        const int ytile = Y0 + LO + 16 * t;
        if (wide) {
            const gl_u8* pg;
            if (ytile >= 0 && ytile + 15 < H) {          // uniform: the tile's 16 rows are inside the image
                const u8* pb = fby + (int64_t)ytile * rowbytes;
                asm volatile("" : "+s"(pb));             // (a scalar base: else the lane offset is folded into a 64-bit vector multiply-add)
                // (the opaque move hides that this is global memory: say so; and the lane offset must be widened HERE, in
                //  the block of the loads, for them to take the scalar-base + 32-bit-offset form: hoisted out of the loop
                //  as a 64-bit pair it costs a 64-bit vector add per step)
                u32 lo_ = lofs;
                asm volatile("" : "+v"(lo_));
                pg = (const gl_u8*)pb + lo_;
            } else {
                pg = (const gl_u8*)fby + (u32)__mul24(min(max(ytile + q, 0), H - 1), rowbytes);
            }
            typedef u32 u32x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(1))) const u32x4 gl_u4;
            const u32x4 rr = *(gl_u4*)pg;
            nraw = make_uint4(rr.x, rr.y, rr.z, rr.w);
        } else {
            const u64* row = fbits + (int64_t)min(max(ytile + q, 0), H - 1) * WW;
            int ws = __builtin_amdgcn_readfirstlane(wstart);
            asm volatile("" : "+s"(ws));                 // (its bounds tests are redone per step on the scalar unit, not kept)
            const u64 w0 = load_bits(row, WW, ws), w1 = NKS > 2 ? load_bits(row, WW, ws + 64) : 0ull;
            const u32 dw[4] = {(u32)w0, (u32)(w0 >> 32), (u32)w1, (u32)(w1 >> 32)};
            const u32 sh = 8 * (b0 & 3);                 // (the form the wide strips load: the window's first byte at byte b0 & 3)
            nraw = make_uint4(dw[0] << sh, sh ? __builtin_amdgcn_alignbit(dw[1], dw[0], 32 - sh) : dw[1],
                              sh ? __builtin_amdgcn_alignbit(dw[2], dw[1], 32 - sh) : dw[2],
                              sh ? __builtin_amdgcn_alignbit(dw[3], dw[2], 32 - sh) : dw[3]);
        }
    };
    // Undecided pixels (a handful per frame) leave the loop: a tile that has any is queued - its row and the four masks
    // of its undecided pixels - with those pixels stored as background, and the queue is drained every NT steps at the
    // latest: exact float64 G straight from the bits, one pixel at a time by the whole wave (lane i sums rows i and
    // i + 64 of the window from their runs, the L products are added in ascending row order exactly as ncc_exact_G does,
    // row sums broadcast from their lanes, template factors read as scalars), and a pixel found foreground is OR-ed
    // into the stored mask word.
    int ecnt = 0;
    // A finished tile's mask: the decision leaves four 64-bit lane masks pw[r] in scalar registers (bit 16 g + c <-> row
    // 4 g + r, column c), i.e. eight dwords whose halves are the 16-bit row pieces mask_bits wants.  v_writelane puts dword
    // k = 2 r + h into lane k of ONE vector register; lane k then stores its low half to row 8 h + r and its high half to
    // row 8 h + r + 4 (two 2-byte stores of 8 lanes, offsets fixed per lane, base scalar) - no per-lane selection.  The
    // stores are issued at the top of the NEXT step, behind its loads (see there).
    constexpr int NO_TILE = 0x7FFFFFFF;
    u32 pend_x = 0;
    int pend_yo = NO_TILE;
    const u32 off_lo = (u32)((8 * (lane & 1) + ((lane >> 1) & 3)) * rowbytes), off_hi = off_lo + (u32)(4 * rowbytes);   // bytes (lanes 0..7)
    auto flush_pending = [&]() {
        if (pend_yo != NO_TILE) {                        // uniform
            u8* base = reinterpret_cast<u8*>(mb16) + (int64_t)pend_yo * rowbytes;
            asm volatile("" : "+s"(base));               // (scalar base + 32-bit lane offset)
            gl_u8* bg = (gl_u8*)base;
            u32 ol_ = off_lo, oh_ = off_hi;              // (widened here, as in load_rows)
            asm volatile("" : "+v"(ol_), "+v"(oh_));
            if (pend_yo <= H - 16) {                     // uniform: the tile's rows are all inside the image
                if (lane < 8) {
                    *(gl_u16*)(bg + ol_) = (unsigned short)pend_x;
                    *(gl_u16*)(bg + oh_) = (unsigned short)(pend_x >> 16);
                }
            } else {
                const int rl = pend_yo + 8 * (lane & 1) + ((lane >> 1) & 3);
                if (lane < 8 && rl < H) *(gl_u16*)(bg + ol_) = (unsigned short)pend_x;
                if (lane < 8 && rl + 4 < H) *(gl_u16*)(bg + oh_) = (unsigned short)(pend_x >> 16);
            }
        }
        pend_yo = NO_TILE;
    };
    auto drain = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0F70);              // the plain stores of these rows are out before the atomics
        // Small branch: everything this path needs beyond the loop's own values comes from the kernel arguments AGAIN (see
        // NccKArgs): 92 -> 8 spilled scalars, ~60 -> 2 v_readlane per step.  The large branch keeps the values it has: there
        // the same change made the allocator spill MORE (64 -> 115 v_readlane in the loop and a vector register to scratch).
        constexpr bool RELOAD = L < 64;
        const double *d_tab, *d_rx, *d_ry; u64* d_mbits; u8* d_mask_u8; int d_H, d_W, d_WW;
        double d_nc_tbar, d_nc_inv_l2, d_nc_thr2, d_nc_T2, d_mu; const u64* d_fbits;
        if constexpr (RELOAD) {
            NccKArgsC* ka = ncc_kargs();
            d_tab = ka->tab; d_rx = ka->rx; d_ry = ka->ry; d_mbits = ka->mbits; d_mask_u8 = ka->mask_u8;
            d_H = ka->H; d_W = ka->W; d_WW = ka->WW;
            d_nc_tbar = ka->nc.tbar; d_nc_inv_l2 = ka->nc.inv_l2; d_nc_thr2 = ka->nc.thr2; d_nc_T2 = ka->nc.T2;
            d_mu = (double)(255ull * (u64)ka->fstat[n * 8 + 0]) / (double)((int64_t)d_H * d_W);
            d_fbits = ka->bits + (int64_t)n * d_H * d_WW;
        } else {
            d_tab = tab; d_rx = rx; d_ry = ry; d_mbits = mbits; d_mask_u8 = mask_u8; d_H = H; d_W = W; d_WW = WW;
            d_nc_tbar = nc.tbar; d_nc_inv_l2 = nc.inv_l2; d_nc_thr2 = nc.thr2; d_nc_T2 = nc.T2; d_mu = mu; d_fbits = fbits;
        }
        const double* cgd = d_tab + VBS_NCC_MAXL;
        u32* mb32 = reinterpret_cast<u32*>(d_mbits) + (int64_t)n * d_H * d_WW * 2;
        for (int e = 0; e < ecnt; ++e) {
            const int yo = (int)__builtin_amdgcn_readfirstlane((int)sm.elist[wave][e][0]);
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                u64 ur = (u64)(u32)__builtin_amdgcn_readfirstlane((int)sm.elist[wave][e][2 + 2 * r]) |
                         ((u64)(u32)__builtin_amdgcn_readfirstlane((int)sm.elist[wave][e][3 + 2 * r]) << 32);
                while (ur) {
                    const int l = __ffsll((long long)ur) - 1;
                    ur &= ur - 1ull;
                    const int y = yo + 4 * (l >> 4) + r, xe = xw + (l & 15);
                    double hv[(L + 63) / 64];
                    u32 cnt = 0;
#pragma unroll
                    for (int k = 0; k < (L + 63) / 64; ++k) {
                        const int i = lane + 64 * k, yy = y + LO + i;
                        hv[k] = (i < L && yy >= 0 && yy < d_H) ? ncc_row_exact<L, LO>(d_fbits + (int64_t)yy * d_WW, d_WW, xe, cgd, &cnt) : 0.0;
                    }
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) cnt += (u32)__shfl_xor((int)cnt, off);
                    double Ge = 0.0;
#pragma unroll
                    for (int k = 0; k < (L + 63) / 64; ++k) {
                        const u32 hl = (u32)__double_as_longlong(hv[k]), hh = (u32)((u64)__double_as_longlong(hv[k]) >> 32);
#pragma unroll 8
                        for (int i = 64 * k; i < min(L, 64 * k + 64); ++i) {
                            const u64 hb = (u64)(u32)__builtin_amdgcn_readlane((int)hl, i - 64 * k) |
                                           ((u64)(u32)__builtin_amdgcn_readlane((int)hh, i - 64 * k) << 32);
                            Ge = __builtin_fma(d_tab[i], __longlong_as_double((long long)hb), Ge);
                        }
                    }
                    int ny = min(y + HI, d_H - 1) - max(y + LO, 0) + 1;
                    int nx = min(xe + HI, d_W - 1) - max(xe + LO, 0) + 1;
                    double nn = (double)(ny * nx), sum_t = d_ry[y] * d_rx[xe];
                    double sum_I = 255.0 * (double)cnt;
                    double rest = -d_nc_tbar * sum_I - d_mu * (sum_t - nn * d_nc_tbar);
                    double s1 = sum_I - nn * d_mu;
                    double s2 = 255.0 * sum_I - 2.0 * d_mu * sum_I + nn * d_mu * d_mu;
                    double var = s2 - s1 * s1 * d_nc_inv_l2;
                    double rhs = d_nc_thr2 * var * d_nc_T2;
                    double num = 255.0 * Ge + rest;
                    const int flags = __builtin_amdgcn_readfirstlane((var > 0.0 ? 1 : 0) | ((var > 0.0 && num > 0.0 && num * num > rhs) ? 2 : 0) |
                                                                     ((var > 1e-6 && num > 0.0 && fabs(num * num - rhs) <= 1e-9 * rhs) ? 4 : 0));
                    if ((flags & 2) && lane == 0) {
                        atomicOr(&mb32[(u32)__mul24(y, 2 * d_WW) + (u32)(xe >> 5)], 1u << (xe & 31));
                        if (U8OUT) d_mask_u8[((int64_t)n * d_H + y) * d_W + xe] = 1;
                    }
                    if (flags & 1) { nexact++; if (flags & 4) amb++; }
                }
            }
        }
        ecnt = 0;
    };
    // The A operands of tile tt (its rows' bits as float16 0 / 1, one 16-byte table entry per k-step) out of the registers
    // its row load filled; then the load of the tile after it is issued.  The bytes are taken out BEFORE the next load and
    // the mask stores are issued (the empty asm pins that order): loads and stores share one in-order counter that the
    // compiler can only wait on as "all done" once both kinds are in flight, so the wait must sit where everything in
    // flight is a whole step old - behind a new load or store it would expose their full latency in every step.
    h8 a_op[NKS];
    auto fetch_ops = [&](int tt) {
        u32 by[NKS];
        const u32 rw[5] = {nraw.x, nraw.y, nraw.z, nraw.w, 0u};
#pragma unroll
        for (int s = 0; s < NKS; ++s) by[s] = __builtin_amdgcn_perm(rw[s + 1], rw[s], psel);
#pragma unroll
        for (int s = 0; s < NKS; ++s) asm volatile("" : "+v"(by[s]) :: "memory");
        load_rows(tt + 1);                               // (unconditional: past the last step it reads rows nobody uses)
        flush_pending();                                 // the finished tile's mask rows, behind the load
        const int ytile = Y0 + LO + 16 * tt;
        if (ytile < 0 || ytile + 15 >= H) {              // uniform: rows outside the image are empty
            const bool rowin = (ytile + q >= 0) && (ytile + q < H);
#pragma unroll
            for (int s = 0; s < NKS; ++s) by[s] = rowin ? by[s] : 0u;
        }
#pragma unroll
        for (int s = 0; s < NKS; ++s) a_op[s] = __builtin_bit_cast(h8, sm.lut[by[s]]);
    };
    load_rows(0);
    fetch_ops(0);
    for (int t0 = 0; t0 < nsteps; t0 += NT) {
#pragma unroll
        for (int u = 0; u < NT; ++u) {                   // u = ring slot of step t: a constant of this copy of the body
            const int t = t0 + u;
            if (t >= nsteps) break;                      // uniform
            // ---- horizontal tile t, and next to it the part of the vertical product that does not need it ----
            // The output tile of this step sums ring tiles t - NT + 1 .. t, oldest first (slots B, B + 1, ...).  All but the
            // last k-step's rows are in the ring since the previous steps: their operands are read and multiplied HERE, as a
            // second chain of matrix instructions beside the horizontal one, instead of behind tile t's conversion and
            // ring store - the wave is bound by the latency of its own chain (LDS -> products -> convert -> LDS -> products ->
            // decision at four waves per SIMD), not by issue slots, so two independent chains side by side shorten the
            // step.  Same products in the same order: G is bit for bit what the one-chain form gave.
            constexpr int dummy_nt = NT;                 // (u + 1) % NT below is a constant after unrolling
            const int B = (u + 1) % dummy_nt;
            constexpr int NKE = (RING - 16) / 32;          // float16 k-steps clear of the newest 16 rows
            constexpr int NK8E = (RING - 16) / 64;         // int8 k-steps (64 rows) clear of them
            f4 G = {0, 0, 0, 0};
            i4 C = {0, 0, 0, 0};
            auto vert_f16 = [&](int s) {
                const int r0 = (16 * B + 32 * s) % RING;  // first ring row of the k-step
                const _Float16* p = (r0 + 32 <= RING ? rd : rdw) + r0;
                const h8 bh = *reinterpret_cast<const h8*>(p);
                const h8 bl = *reinterpret_cast<const h8*>(p + LOFS);
                G = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[s], bh, G, 0, 0, 0);
                G = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[s], bl, G, 0, 0, 0);
                G = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[s], bh, G, 0, 0, 0);
            };
            auto vert_i8 = [&](int s) {
                // 16 rows per lane group; groups past the ring carry zero weights and may read any valid 16 bytes
                const int gi = min(4 * s + g, NT - 1);
                const int slot = B + gi - (gi >= NT - B ? NT : 0);
                const i4 bc = *reinterpret_cast<const i4*>(cbase + 16 * slot);
                C = __builtin_amdgcn_mfma_i32_16x16x64_i8(one8[s], bc, C, 0, 0, 0);
            };
            const bool vert = !(t < NT - 1 || dbg == 3);     // (uniform; dbg: tools/ phase timing, always 0 in the product library)
            f4 ah = {0, 0, 0, 0}, ac = {0, 0, 0, 0};
            if (vert) {
#pragma unroll
                for (int s = 0; s < NKS; ++s) {
                    ah = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_op[s], whi[s], ah, 0, 0, 0);
                    if (s < NKE) vert_f16(s);
                    ah = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_op[s], wlo[s], ah, 0, 0, 0);
                    if (s < NK8E) vert_i8(s);
                    ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_op[s], one[s], ac, 0, 0, 0);
                }
#pragma unroll
                for (int s = NKS; s < NKE; ++s) vert_f16(s);
            } else {
#pragma unroll
                for (int s = 0; s < NKS; ++s) {
                    ah = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_op[s], whi[s], ah, 0, 0, 0);
                    ah = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_op[s], wlo[s], ah, 0, 0, 0);
                    ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_op[s], one[s], ac, 0, 0, 0);
                }
            }
            // The next tile's operands, into the registers this tile's just left: the wait for its rows (loaded a step
            // ago), the table lookups and their LDS round trip run under this step's conversion, vertical products and
            // decision instead of at the top of the next step, in front of its first matrix instruction.
            fetch_ops(t + 1);
            {
                // hi = the float32 cut to float16's 10 mantissa bits (exact in float16: 0.25 <= h <= 1024), lo = the rest
                // (h - hi is exact in float32), rounded to float16; the count (an integer <= l) goes to the ring as a byte
                // (v_cvt_pkrtz + v_fma_mixlo/hi would be 6 instructions instead of these 12, and was built: no faster - a
                //  v_fma_mix costs 3 x a v_fma on this chip, tools/ubench_valu.hip - and it needed inline assembly)
                h4 vh, vl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hi_f = __uint_as_float(__float_as_uint(ah[r]) & 0xFFFFE000u);
                    vh[r] = (_Float16)hi_f;
                    vl[r] = (_Float16)(ah[r] - hi_f);
                }
                u32 vc = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) vc = __builtin_amdgcn_cvt_pk_u8_f32(ac[r], r, vc);
                *reinterpret_cast<h4*>(wr + 16 * u) = vh;
                *reinterpret_cast<h4*>(wr + LOFS + 16 * u) = vl;
                *reinterpret_cast<u32*>(cw + 16 * u) = vc;
            }
            if (!vert) continue;
            // ---- vertical, the rest: the k-steps that take in tile t.  Output rows yo .. yo + 15 ----
            const int yo = Y0 + 16 * (t - (NT - 1));
#pragma unroll
            for (int s = NKE; s < NKS; ++s) vert_f16(s);
#pragma unroll
            for (int s = NK8E; s < NK8; ++s) vert_i8(s);
#ifdef VBS_DEBUG_KNOBS
            if (dbg == 8 && mask_u8) {                   // tools/gpu_mask_diff.py: the window counts instead of the mask
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int y = yo + 4 * g + r, x = xw + q;
                    if (y < H && x < W) mask_u8[((int64_t)n * H + y) * W + x] = (u8)C[r];
                }
                continue;
            }
#endif
            if (dbg == 2) { asm volatile("" :: "v"(G[0]), "v"(G[1]), "v"(G[2]), "v"(G[3]), "v"(C[0]), "v"(C[1]), "v"(C[2]), "v"(C[3])); continue; }
            // ---- decision: lane (g, q) holds rows yo + 4g + r (r = 0..3) of column xw + q ----
            const int x = xw + q;
            float cf[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) cf[r] = (float)C[r];
            // pw[r] / uw[r]: 64-bit masks (wave-uniform, kept in scalar registers) of the pixels decided foreground / left
            // undecided by the filter, bit = lane
            u64 pw[4] = {0, 0, 0, 0}, uw[4] = {0, 0, 0, 0};
            if (yo >= plain_lo && yo <= plain_hi) {      // wave-uniform: windows inside the image, empty window = background
                // G > theta(c)  <=>  u = G - kc c > 0 and u^2 > a = ks^2 c (l^2 - c), taken with G (1 -+ rel): no square root,
                // and a tile whose every u is negative (the template correlates negatively there) is background after two
                // operations per pixel.  kc, ks^2 and K1 = ks^2 l^2 are instruction literals (NccLit); a = c (K1 - ks^2 c)
                // costs two operations: its absolute error, 0.4 c for l = 80 (half an ulp of K1 and of the fma), moves the
                // threshold on u by 0.2 sqrt(c / (ks^2 (l^2 - c))) < 8e-4 c = the 4.7e-6 G that rel leaves beyond G's own
                // 2^-16 - for every c (l^2 - c) >= 123, i.e. every window that is neither empty nor full; those two have
                // a = 0 and u = 0 -+ rel G and stay undecided or background as before.
                const float rel = rel_arg;               // >= NCC_REL (VBS_OPT_NCC_MARGIN)
                float uu[4], uhi[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    uu[r] = __builtin_fmaf(cf[r], -NccLit<L>::kc, G[r]);
                    uhi[r] = __builtin_fmaf(G[r], rel, uu[r]);
                }
                const float umax = fmaxf(fmaxf(uhi[0], uhi[1]), fmaxf(uhi[2], uhi[3]));
                if (__ballot(umax >= 0.0f) || dbg == 12) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float ulo = __builtin_fmaf(G[r], -rel, uu[r]);
                        const float a = __builtin_fmaf(cf[r], -NccLit<L>::ks2, NccLit<L>::K1) * cf[r];
                        pw[r] = __ballot(ulo * fabsf(ulo) > a);                       // ulo > 0 and ulo^2 > a: foreground
                        uw[r] = ~(pw[r] | __ballot(uhi[r] * fabsf(uhi[r]) <= a));     // uhi < 0 or uhi^2 <= a: background
                    }
                }
            } else {
                // Border tile (or th0 <= 0): theta in float32 from the collected form of the general window,
                //   var = 255^2 c (1 - c / l^2) + (1 - nn / l^2) mu (nn mu - 510 c),   rest = -tbar 255 c - mu (sum_t - nn tbar),
                // whose error stays below 1e-4 theta + 1 in these units (var loses at most 5e-6 to cancellation, the last
                // subtraction 1e-4 absolute on theta 255): pixels within 1e-3 theta + 2 of it go to the exact path.
                // num > 0 <=> G > lin = kc c + (mu 2^20 / 255) d0 with d0 = sum_t - nn tbar: a tile whose every pixel has
                // G - lin + margin < 0 is background (most border tiles: their windows hang over the dark frame).  The
                // margin covers G's 2^-16 and the float32 roundings of lin (1e-6 of its terms, also where they cancel).
                // A strip at the left / right edge of the image is border all the way down, but with whole rows inside
                // for most of it: there nn, sum_t and d0 are the per-lane constants of the prologue.
                float nnv[4], stv[4], dd0[4], uhi[4];
                if (yo + LO >= 0 && yo <= plain_hi) {    // uniform: the windows' rows are all inside
#pragma unroll
                    for (int r = 0; r < 4; ++r) { nnv[r] = nnc; stv[r] = stc; dd0[r] = dd0c; }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float2 rf = rowf[min(yo + 4 * g + r, H - 1)];   // {rows of the window inside the image, (float)ry}
                        nnv[r] = rf.x * nxf;
                        stv[r] = rf.y * rxf;
                        dd0[r] = stv[r] - nnv[r] * tbarf;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float kcc = kc * cf[r];
                    const float mrg = __builtin_fmaf(1e-3f, G[r] + kcc, __builtin_fmaf(1e-5f * mukf, stv[r], 2.0f));
                    uhi[r] = (G[r] - kcc) - mukf * dd0[r] + mrg;
                }
                if (__ballot(fmaxf(fmaxf(uhi[0], uhi[1]), fmaxf(uhi[2], uhi[3])) >= 0.0f)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float cfr = cf[r];
                        const float nn = nnv[r], a0 = 1.0f - nn * il2f;
                        const float rest = -(ktf * cfr) - muf * dd0[r];
                        const float var = 65025.0f * cfr * (1.0f - cfr * il2f) + a0 * muf * (nn * muf - 510.0f * cfr);
                        const float t1 = (__builtin_amdgcn_sqrtf(fmaxf(krf * var, 0.0f)) - rest) * (float)(NCC_WSCALE * NCC_WSCALE / 255.0);
                        // empty window: G = 0, num = rest = -mu d0 and rhs = thr2 T2 a nn mu^2: background unless d0 < 0 and
                        // d0^2 > thr2 T2 a nn (mu cancels); decided here with a factor 2 to spare, else left to the exact path
                        const float t0v = ((dd0[r] > -1e-4f) | (dd0[r] * dd0[r] < 0.5f * krf * a0 * nn)) ? (float)NCC_NEVER : 0.0f;
                        const float th = cfr == 0.0f ? t0v : t1;
                        const float m = __builtin_fmaf(fabsf(th), 1e-3f, 2.0f), d = G[r] - th;
                        pw[r] = __ballot(d > m);
                        uw[r] = ~(pw[r] | __ballot(d < -m));
                    }
                }
            }
            if (yo > valid_hi) {                         // uniform: tiles that stick out of the image
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const u64 v = __ballot((yo + 4 * g + r < H) && (x < W));
                    pw[r] &= v; uw[r] &= v;
                }
            }
#ifdef VBS_DEBUG_KNOBS
            if (dbg == 9 && mask_u8) {                   // tools/gpu_mask_diff.py: the filter's verdict (1 fg, 2 undecided)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int y = yo + 4 * g + r;
                    if (y < H && x < W) mask_u8[((int64_t)n * H + y) * W + x] = (u8)(((pw[r] >> lane) & 1ull) | (((uw[r] >> lane) & 1ull) << 1));
                }
                continue;
            }
#endif
            if (uw[0] | uw[1] | uw[2] | uw[3]) {         // rare: queue the tile (see drain)
                // (one lane writes the ten words: eight `lane == k` selects would each keep a 64-bit lane mask in two scalar
                //  registers for the whole loop, and this kernel spills scalars as it is)
                if (lane == 0) {
                    u32* e = &sm.elist[wave][ecnt][0];
                    e[0] = (u32)yo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { e[2 + 2 * r] = (u32)uw[r]; e[3 + 2 * r] = (u32)(uw[r] >> 32); }
                }
                ecnt++;
            }
            if (U8OUT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int y = yo + 4 * g + r;
                    if (y < H && x < W) mask_u8[((int64_t)n * H + y) * W + x] = (u8)((pw[r] >> lane) & 1ull);
                }
            }
            if (dbg == 1) { asm volatile("" :: "s"(pw[0]), "s"(pw[1]), "s"(pw[2]), "s"(pw[3])); continue; }
            {
                u32 xv = 0;                              // (lanes 8.. are never stored)
#pragma unroll
                for (int r = 0; r < 4; ++r) {            // (no builtin for v_writelane in this compiler; pure register work)
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(xv) : "s"((u32)pw[r]), "n"(2 * r));
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(xv) : "s"((u32)(pw[r] >> 32)), "n"(2 * r + 1));
                }
                pend_x = xv;
                pend_yo = yo;
            }
        }
        if (t0 + NT >= nsteps || ecnt > ECAP - NT) {     // uniform
            flush_pending();
            if (ecnt) drain();
        }
    }
    {
        u32* fstat_ = fstat; u64* tot_ = tot;
        if constexpr (L < 64) { NccKArgsC* ka = ncc_kargs(); fstat_ = ka->fstat; tot_ = ka->tot; }      // (see drain)
        if (lane == 0) {                                 // (wave-uniform counters; per frame and the handle's running totals)
            if (amb) { atomicAdd(&fstat_[n * 8 + 1], amb); atomicAdd(&tot_[0], (u64)amb); }
            if (nexact) { atomicAdd(&fstat_[n * 8 + 3], nexact); atomicAdd(&tot_[1], (u64)nexact); }
        }
        if (tid == 0 && (blockIdx.x | blockIdx.y | blockIdx.z) == 0) atomicAdd(&tot_[2], (u64)gridDim.z);  // frames
    }
}
#undef wide

// Toeplitz fragments of k_ncc_mfma in the lane layout of v_mfma_f32_16x16x32_f16 (lane = 16 g + column, element j
// pairs with the other operand's element j of the same g): weight index (32 s + 8 g + j) - column.  The same
// fragments serve as B operand of the horizontal and as A operand of the vertical product.
std::vector<u32> ncc_mfma_fragments(const NccConst& nc, int l) {
    const int nt = (16 + l - 1 + 15) / 16, nks = (16 * nt + 31) / 32, nk8 = (16 * nt + 63) / 64;
    std::vector<u32> out((size_t)(3 * nks + nk8) * 64 * 4, 0);
    for (int s = 0; s < nk8; ++s)                        // kind 3: int8 ones, byte j of lane (g, row q) <-> k = 64 s + 16 g + j
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 16; ++j) {
                const int idx = 64 * s + 16 * (lane >> 4) + j - (lane & 15);
                if (idx >= 0 && idx < l) out[((size_t)(3 * nks + s) * 64 + lane) * 4 + (j >> 2)] |= 1u << (8 * (j & 3));
            }
    auto bits16 = [](double v) { _Float16 hv = (_Float16)v; unsigned short b; memcpy(&b, &hv, 2); return (u32)b; };
    for (int kind = 0; kind < 3; ++kind)
        for (int s = 0; s < nks; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int g = lane >> 4, col = lane & 15;
                for (int j = 0; j < 8; ++j) {
                    const int idx = 32 * s + 8 * g + j - col;
                    u32 v = 0;
                    if (idx >= 0 && idx < l) {
                        const double w = nc.g[idx] * NCC_WSCALE;
                        const _Float16 hi = (_Float16)w;
                        v = kind == 0 ? bits16((double)hi) : kind == 1 ? bits16(w - (double)hi) : bits16(1.0);
                    }
                    out[((size_t)(kind * nks + s) * 64 + lane) * 4 + (j >> 1)] |= v << (16 * (j & 1));
                }
            }
    return out;
}

// ---- _normxcorr2 for ARBITRARY operands (marker_detection.py:146-164) -------------------------------------------------
// Any float64 template / image, mode 'full' | 'same' | 'valid', evaluated directly in the spatial domain in float64
// (the reference goes through three FFT convolutions; the two agree to the FFT's rounding, ~1e-12 of the map's scale).
// Not on the hot path: the pipeline's own operands (binary area_mask, Gaussian template) take k_ncc_mfma / k_ncc.
//   stats[0] = mean(template), stats[1] = mean(image), stats[2] = sum((template - mean)^2)
__global__ __launch_bounds__(1024) void k_nccg_stats(const double* __restrict__ T, int nt, const double* __restrict__ I, int ni,
                                                     double* __restrict__ stats) {
    __shared__ double part[1024];
    const int tid = threadIdx.x;
    auto block_sum = [&](double v) {                    // fixed order: the result does not depend on scheduling
        part[tid] = v;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) { if (tid < s) part[tid] += part[tid + s]; __syncthreads(); }
        const double r = part[0];
        __syncthreads();
        return r;
    };
    double a = 0;
    for (int i = tid; i < nt; i += 1024) a += T[i];
    const double mt = block_sum(a) / (double)nt;
    a = 0;
    for (int i = tid; i < ni; i += 1024) a += I[i];
    const double mi = block_sum(a) / (double)ni;
    a = 0;
    for (int i = tid; i < nt; i += 1024) { const double d = T[i] - mt; a += d * d; }
    const double t2 = block_sum(a);
    if (tid == 0) { stats[0] = mt; stats[1] = mi; stats[2] = t2; }
}

// block = 8 output rows x 64 output columns; for every template row the image rows under it are staged in LDS
#define NCCG_MAXTW 256
__global__ __launch_bounds__(512) void k_nccg(const double* __restrict__ T, int th, int tw, const double* __restrict__ I, int h,
                                              int w, int oy, int ox, int oh, int ow, const double* __restrict__ stats,
                                              double* __restrict__ out) {
    __shared__ double rows[8][64 + NCCG_MAXTW];
    __shared__ double trow[NCCG_MAXTW];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + tx, y = blockIdx.y * 8 + ty;
    const double mt = stats[0], mi = stats[1], t2 = stats[2];
    double s_it = 0, s_i = 0, s_ii = 0;
    for (int u = 0; u < th; ++u) {
        __syncthreads();
        const int yy = y + oy + u;
        for (int k = tx; k < 64 + tw - 1; k += 64) {
            const int xx = blockIdx.x * 64 + ox + k;
            rows[ty][k] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? I[(int64_t)yy * w + xx] - mi : 0.0;   // zero padding AFTER
        }                                                                                                  // the mean went
        for (int k = threadIdx.x; k < tw; k += 512) trow[k] = T[(int64_t)u * tw + k] - mt;
        __syncthreads();
        for (int v = 0; v < tw; ++v) {
            const double iv = rows[ty][tx + v];
            s_it = __builtin_fma(iv, trow[v], s_it);
            s_i += iv;
            s_ii = __builtin_fma(iv, iv, s_ii);
        }
    }
    if (x >= ow || y >= oh) return;
    double var = s_ii - s_i * s_i / ((double)th * (double)tw);
    if (var < 0.0) var = 0.0;
    double r = s_it / sqrt(var * t2);
    if (!isfinite(r)) r = 0.0;
    out[(int64_t)y * ow + x] = r;
}

int launch_ncc_general(const double* T, int th, int tw, const double* I, int h, int w, int mode, double* out,
                       double* stats, hipStream_t s) {
    if (tw > NCCG_MAXTW) return VBS_EINVAL;
    // scipy.signal.fftconvolve output window: 0 full, 1 same (size of the image, start (t - 1) // 2), 2 valid
    const int sy = mode == 0 ? 0 : mode == 1 ? (th - 1) / 2 : th - 1, sx = mode == 0 ? 0 : mode == 1 ? (tw - 1) / 2 : tw - 1;
    const int oh = mode == 0 ? h + th - 1 : mode == 1 ? h : h - th + 1, ow = mode == 0 ? w + tw - 1 : mode == 1 ? w : w - tw + 1;
    if (oh < 1 || ow < 1) return VBS_EINVAL;
    hipLaunchKernelGGL(k_nccg_stats, dim3(1), dim3(1024), 0, s, T, th * tw, I, h * w, stats);
    hipLaunchKernelGGL(k_nccg, dim3((ow + 63) / 64, (oh + 7) / 8), dim3(512), 0, s, T, th, tw, I, h, w, sy - (th - 1),
                       sx - (tw - 1), oh, ow, stats, out);
    return VBS_OK;
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

// running totals over every internal pass since the last vbs_ncc_counters(reset): {pixels within the ambiguity band of
// the 0.1 threshold, pixels re-evaluated in float64, frames}
__global__ __launch_bounds__(256) void k_stat_accum(const u32* __restrict__ fstat, u64* __restrict__ tot, int nb) {
    u64 a = 0, e = 0;
    for (int n = blockIdx.x * 256 + threadIdx.x; n < nb; n += gridDim.x * 256) { a += fstat[n * 8 + 1]; e += fstat[n * 8 + 3]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); e += __shfl_xor(e, off); }
    if ((threadIdx.x & 63) == 0) {
        if (a) atomicAdd(&tot[0], a);
        if (e) atomicAdd(&tot[1], e);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&tot[2], (u64)nb);
}

static void launch_stat_accum(vbs_handle* h, int nb, hipStream_t s) {
    VBS_LAUNCH(h, s, "k_stat_accum", k_stat_accum, dim3(1), dim3(256), 0, s, h->fstat, h->ncc_tot, nb);
}

void launch_popcount(vbs_handle* h, int nb, hipStream_t s) {
    VBS_LAUNCH(h, s, "k_popcount", k_popcount, dim3(nb), dim3(256), 0, s, h->area_bits, h->fstat, h->H * h->WW);
}

void launch_ncc(vbs_handle* h, int nb, u8* mask_u8, double* ncc_out, hipStream_t s) {
    if (!ncc_out && !VBS_KNOB("VBS_NCC_VALU")) {
        const int tilesY = (h->H + 15) / 16;
        // few frames: split the columns - into as many segments as keep every workgroup resident at once (256 CUs x 4): a
        // second, partly filled round costs a one-frame call more than the longer segments (21.0 -> 18.1 us at 1280x1024)
        const int want = h->WW * nb <= 512 ? 1024 : 2048;
        int nseg = std::min(tilesY, std::max(1, want == 1024 ? want / (h->WW * nb) : (want + h->WW * nb - 1) / (h->WW * nb)));
        if (VBS_KNOB("VBS_NCC_NSEG")) nseg = VBS_KNOB("VBS_NCC_NSEG");
        const int tps = (tilesY + nseg - 1) / nseg;
        nseg = (tilesY + tps - 1) / tps;
        dim3 grid(h->WW, nseg, nb);
#define NCC_GO(L_, LO_, U8)                                                                                      \
    VBS_LAUNCH(h, s, "k_ncc_mfma", (k_ncc_mfma<L_, LO_, U8>), grid, dim3(256), 0, s, h->area_bits, h->ncc_rx,    \
               h->ncc_ry, h->ncc_frags, h->ncc_tab, h->ncc_rowf, h->mask_bits, mask_u8, h->fstat, h->ncc_tot,   \
               h->H, h->W, h->WW, tps,                                                                           \
               VBS_KNOB("VBS_NCC_DBG"), std::max(NCC_REL, 1e-6f * (float)h->ncc_margin_ppm), h->ncc)
        if (!h->bp.small) { if (mask_u8) NCC_GO(80, -40, true); else NCC_GO(80, -40, false); }
        else { if (mask_u8) NCC_GO(33, -16, true); else NCC_GO(33, -16, false); }
#undef NCC_GO
        return;
    }
    dim3 grid(h->WW, (h->H + 63) / 64, nb);
    const int stop = VBS_KNOB("VBS_NCC_STOP");
    if (!h->bp.small) {
        VBS_LAUNCH(h, s, "k_ncc", (k_ncc<80, -40>), grid, dim3(256), 0, s, h->area_bits, h->ncc_rx, h->ncc_ry,
                   h->mask_bits, mask_u8, ncc_out, h->fstat, h->H, h->W, h->WW, stop, h->ncc);
    } else {
        VBS_LAUNCH(h, s, "k_ncc", (k_ncc<33, -16>), grid, dim3(256), 0, s, h->area_bits, h->ncc_rx, h->ncc_ry,
                   h->mask_bits, mask_u8, ncc_out, h->fstat, h->H, h->W, h->WW, stop, h->ncc);
    }
    launch_stat_accum(h, nb, s);                         // (k_ncc_mfma adds to the running totals itself)
}
