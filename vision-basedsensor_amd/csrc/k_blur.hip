// a3-a5: BGR->gray, two uint8 GaussianBlurs (OpenCV fixed-point model), DoG + 15 (mod 256), inRange.
// Reference: marker_detection.py:114-129.  Integer arithmetic throughout, so results are
// independent of summation order and bit-exact against oracle/stages.py:gaussian_blur_u8.
//
//   out(y,x) = ( sum_i ky[i] * ( sum_j kx[j] * p(y+i-c, x+j-c) ) + 2^15 ) >> 16,  taps in 1/256
//
// k_blur_mfma evaluates both passes of both blurs as banded-Toeplitz products on the int8 matrix cores.
#include <algorithm>
#include <cstdlib>

#include "common.h"

__device__ __forceinline__ int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return min(max(i, 0), n - 1);
}

// gray pixels of 4 consecutive BGR pixels (12 bytes as 3 dwords): cv2's fixed-point weights (common.h gray_coef)
__device__ __forceinline__ u32 bgr4_to_gray(u32 a, u32 b, u32 c, const GrayCoef& gc) {
    const u32 g0 = (gc.cb * (a & 255u) + gc.cg * ((a >> 8) & 255u) + gc.cr * ((a >> 16) & 255u) + gc.half) >> gc.shift;
    const u32 g1 = (gc.cb * (a >> 24) + gc.cg * (b & 255u) + gc.cr * ((b >> 8) & 255u) + gc.half) >> gc.shift;
    const u32 g2 = (gc.cb * ((b >> 16) & 255u) + gc.cg * (b >> 24) + gc.cr * (c & 255u) + gc.half) >> gc.shift;
    const u32 g3 = (gc.cb * ((c >> 8) & 255u) + gc.cg * ((c >> 16) & 255u) + gc.cr * (c >> 24) + gc.half) >> gc.shift;
    return g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
}

// cvtColor(BGR2GRAY) (or a plain copy for 1 channel) into the pitched gray plane the blur reads: a streaming kernel,
// 16 pixels per thread (three 16-byte loads, one 16-byte store) when the rows are 16-byte aligned: 0.85-1.0 us per
// 1280x1024 frame (5.3-6.2 TB/s of its 5.2 MB).  It runs in line in front of the blur; VBS_OPT_GRAY_SIDE_STREAM moves it one
// internal pass ahead onto the handle's own stream (api.hip), which measured no faster: the matrix-core kernels leave
// the HBM idle but not the registers, LDS and issue slots a conversion workgroup needs next to them.
// (Converting inside the blur's own loader was built and measured: LDS-DMA staging of the raw bytes kept the matrix
//  operands in registers only at the price of 50 spilled VGPRs - 5 us per frame against 1.45.)
__global__ __launch_bounds__(256) void k_gray(const u8* __restrict__ frames, int channels, int64_t stride_n,
                                              int64_t stride_row, u8* __restrict__ gray, int H, int W, int P, GrayCoef gc,
                                              int vec_ok, int flat) {
    __shared__ __align__(16) uint4 raw[2][3 * 256];     // 24 KB: the 48 bytes of each thread's 16 pixels, loaded coalesced
    const int n = blockIdx.z;
    if (flat) {
        // dense BGR frame (row stride 3 W, gray pitch W): one run of H W pixels, 2 x 4096 per block, loaded as consecutive
        // 16-byte pieces by consecutive lanes (both halves in flight together, streamed past the caches) and handed to
        // their owners through LDS
        const int64_t npx = (int64_t)H * W, pb = (int64_t)blockIdx.x * 8192;
        const u8* src = frames + (int64_t)n * stride_n;
        u8* dstf = gray + (int64_t)n * H * P;
        if (pb + 8192 <= npx) {                          // block-uniform
            const uint4* s4 = reinterpret_cast<const uint4*>(src + pb * 3);
            uint4 v[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                typedef u32 u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(s4 + q * 256 + threadIdx.x));
                v[q] = make_uint4(t.x, t.y, t.z, t.w);
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) raw[q / 3][(q % 3) * 256 + threadIdx.x] = v[q];
            __syncthreads();
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const uint4 r0 = raw[hf][3 * threadIdx.x], r1 = raw[hf][3 * threadIdx.x + 1], r2 = raw[hf][3 * threadIdx.x + 2];
                *reinterpret_cast<uint4*>(dstf + pb + 4096 * hf + threadIdx.x * 16) =
                    make_uint4(bgr4_to_gray(r0.x, r0.y, r0.z, gc), bgr4_to_gray(r0.w, r1.x, r1.y, gc),
                               bgr4_to_gray(r1.z, r1.w, r2.x, gc), bgr4_to_gray(r2.y, r2.z, r2.w, gc));
            }
        } else {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int64_t p0 = pb + 4096 * hf + threadIdx.x * 16;
                if (p0 >= npx) return;                   // (H W is a multiple of 16 in flat mode)
                const uint4* s4 = reinterpret_cast<const uint4*>(src + p0 * 3);
                const uint4 r0 = s4[0], r1 = s4[1], r2 = s4[2];
                *reinterpret_cast<uint4*>(dstf + p0) =
                    make_uint4(bgr4_to_gray(r0.x, r0.y, r0.z, gc), bgr4_to_gray(r0.w, r1.x, r1.y, gc),
                               bgr4_to_gray(r1.z, r1.w, r2.x, gc), bgr4_to_gray(r2.y, r2.z, r2.w, gc));
            }
        }
        return;
    }
    // rows with their own stride (a crop view, a pitched plane): thread = (row, 16-pixel piece), pieces of consecutive rows
    // side by side in a block - a 480-pixel row fills 30 of a block's 256 threads when every block takes one row
    // (0.73 us per 480x450 BGR frame of the reference's cropped configuration, the largest kernel of that workload)
    const int ppr = P >> 4, idx = blockIdx.x * 256 + threadIdx.x;
    const int y = idx / ppr, x0 = (idx - y * ppr) * 16;
    if (y >= H) return;
    const u8* src = frames + (int64_t)n * stride_n + (int64_t)y * stride_row;
    u8* dst = gray + ((int64_t)n * H + y) * P + x0;
    if (vec_ok && channels == 3 && x0 + 16 <= W) {
        const uint4* s4 = reinterpret_cast<const uint4*>(src + (int64_t)x0 * 3);
        const uint4 r0 = s4[0], r1 = s4[1], r2 = s4[2];
        *reinterpret_cast<uint4*>(dst) = make_uint4(bgr4_to_gray(r0.x, r0.y, r0.z, gc), bgr4_to_gray(r0.w, r1.x, r1.y, gc),
                                                    bgr4_to_gray(r1.z, r1.w, r2.x, gc), bgr4_to_gray(r2.y, r2.z, r2.w, gc));
        return;
    }
    u32 out[4] = {0, 0, 0, 0};
    for (int k = 0; k < 16; ++k) {
        const int x = x0 + k;
        u32 v = 0;
        if (x < W) {
            if (channels == 1) {
                v = src[x];
            } else {   // cv2 8-bit BGR2GRAY, fixed point (coefficient set: common.h gray_coef)
                const u8* p = src + (int64_t)x * channels;
                v = (gc.cb * p[0] + gc.cg * p[1] + gc.cr * p[2] + gc.half) >> gc.shift;
            }
        }
        out[k >> 2] |= v << (8 * (k & 3));
    }
    *reinterpret_cast<uint4*>(dst) = make_uint4(out[0], out[1], out[2], out[3]);
}

// k_gray's dense path without LDS (48 contiguous bytes per thread, 3 KB per wave): the form the side stream launches, so
// that its workgroups fit next to k_ncc_mfma's, which leave registers and wave slots but no LDS
__global__ __launch_bounds__(256) void k_gray_flat(const u8* __restrict__ frames, int64_t stride_n, u8* __restrict__ gray,
                                                   int64_t npx, GrayCoef gc) {
    const int n = blockIdx.z;
    const int64_t p0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (p0 >= npx) return;
    const uint4* s4 = reinterpret_cast<const uint4*>(frames + (int64_t)n * stride_n + p0 * 3);
    const uint4 r0 = s4[0], r1 = s4[1], r2 = s4[2];
    *reinterpret_cast<uint4*>(gray + (int64_t)n * npx + p0) =
        make_uint4(bgr4_to_gray(r0.x, r0.y, r0.z, gc), bgr4_to_gray(r0.w, r1.x, r1.y, gc),
                   bgr4_to_gray(r1.z, r1.w, r2.x, gc), bgr4_to_gray(r2.y, r2.z, r2.w, gc));
}

// the cvtColor stage on its own (vbs_bgr2gray): dense [n,H,W] output, one pixel per thread
__global__ void k_gray_dense(const u8* __restrict__ frames, int64_t stride_n, int64_t stride_row,
                             u8* __restrict__ out, int H, int W, GrayCoef gc) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, n = blockIdx.z;
    if (x >= W) return;
    const u8* p = frames + (int64_t)n * stride_n + (int64_t)y * stride_row + (int64_t)x * 3;
    out[((int64_t)n * H + y) * W + x] = (u8)((gc.cb * p[0] + gc.cg * p[1] + gc.cr * p[2] + gc.half) >> gc.shift);
}

void launch_gray_dense(vbs_handle* h, const u8* frames, int nb, int64_t stride_n, int64_t stride_row, u8* out,
                       hipStream_t s) {
    dim3 grid((h->W + 255) / 256, h->H, nb);
    VBS_LAUNCH(h, s, "k_gray_dense", k_gray_dense, grid, dim3(256), 0, s, frames, stride_n, stride_row, out, h->H, h->W,
               gray_coef(h->gray_bits));
}

// ---- MFMA path -------------------------------------------------------------------------------------
// A 101-tap separable blur is 140 MACs per pixel per pass: compute-bound on the vector ALU (v_dot4 at half
// rate), but a banded-Toeplitz matrix product for the matrix cores, and exact there: taps < 128 and
// p - 128 are int8, v_mfma_i32_32x32x32_i8 accumulates in int32.
//
//   horizontal  Hs[y][x]  = sum_k (p[y][xw+k] - 128) * tap[k - x - (LEFT - R)]        (A = image rows from LDS,
//                                                                                       B = Toeplitz, constant)
//               H = Hs + 128 * 256                                                      (taps sum to 256)
//   vertical    V[x][y]   = sum_k Hs_hi[k][x] * tap[..] * 256 + sum_k (Hs_lo[k][x] - 128) * tap[..] + const
//
// One wave owns a 32-column strip and slides down it 32 rows per step.  The horizontal result tile (column
// on the lane, 16 rows in the accumulator registers) is split into its signed high byte and its low byte
// (offset by 128), packed four rows to a dword and used directly as the A operand of the vertical product
// (X^T * T^T sums over the accumulator's row index, so no lane movement and no LDS).  The last NK tiles are
// kept in a register ring, so every horizontal tile is computed once.  The vertical result has the output
// row on the lane and 16 columns in registers: each lane assembles its row's 16 mask bits from sign bits
// and the two lane halves are OR-ed into the 32-bit half word of the bit image.
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void pack_tile(const v16i& acc, v4i& hi, v4i& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        u32 t01 = __builtin_amdgcn_perm((u32)acc[4 * q + 1], (u32)acc[4 * q + 0], 0x05010400u);
        u32 t23 = __builtin_amdgcn_perm((u32)acc[4 * q + 3], (u32)acc[4 * q + 2], 0x05010400u);
        lo[q] = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
        hi[q] = (int)__builtin_amdgcn_perm(t23, t01, 0x07060302u);
    }
}

// reflect-101 bytes of one 16-byte chunk that touches the image border or is not dword aligned
__device__ __forceinline__ uint4 fetch_chunk_slow(const u8* src, int px, int W) {
    u32 w[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        u32 v = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) v |= (u32)src[reflect101(px + 4 * d + b, W)] << (8 * b);
        w[d] = v;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int NK, int SA0, int NKA, bool U8OUT>
__global__ __launch_bounds__(256, 2) void k_blur_mfma(const u8* __restrict__ gray, int64_t gstride_n,
                                                      int64_t gstride_row, const uint4* __restrict__ frags,
                                                      u64* __restrict__ bits, u8* __restrict__ area_u8,
                                                      u32* __restrict__ fstat, int H, int W, int WW,
                                                      int tiles_per_seg, int k3, int k8, int span_i, int dbg_arg) {
#ifdef VBS_DEBUG_KNOBS
    const int dbg = dbg_arg;                            // tools/gpu_ncc_phase.py: phase timing by early exit
#else
    constexpr int dbg = 0;
#endif
    constexpr int LEFT = 32 * ((NK - 1) / 2);
    constexpr int ROWB = 128 + 32 * (NK - 1);          // bytes staged per image row
    constexpr int CH = ROWB / 16;
    constexpr int STRIDE = ROWB + 16;                  // 68 (52) dwords: 16 consecutive rows hit all banks
    constexpr int NIT = (32 * CH + 255) / 256;
    __shared__ __align__(16) u8 tile[2][32 * STRIDE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // uniform, and known to the compiler to be
    const int hh = lane >> 5, m = lane & 31;
    const int X0 = blockIdx.x * 128, n = blockIdx.z;
    const int tilesY = (H + 31) / 32;
    const int tile0 = blockIdx.y * tiles_per_seg;
    const int ntiles = min(tiles_per_seg, tilesY - tile0);
    if (ntiles <= 0) return;
    const int Y0 = tile0 * 32, nsteps = ntiles + NK - 1;
    const u8* g = gray + (int64_t)n * gstride_n;
    const bool aligned = ((gstride_row & 3) == 0) && ((gstride_n & 3) == 0) && ((reinterpret_cast<uintptr_t>(gray) & 3) == 0);
    const bool rows24 = gstride_row > 0 && gstride_row < (1 << 24) && (int64_t)H * gstride_row < (1ll << 31);   // (uniform)

    v4i bh[NK], bha[NKA], tv[NK], tva[NKA];
#pragma unroll
    for (int s = 0; s < NK; ++s) {
        uint4 a = frags[(0 * NK + s) * 64 + lane], b = frags[(1 * NK + s) * 64 + lane];
        bh[s] = v4i{(int)a.x, (int)a.y, (int)a.z, (int)a.w};
        tv[s] = v4i{(int)b.x, (int)b.y, (int)b.z, (int)b.w};
    }
#pragma unroll
    for (int s = 0; s < NKA; ++s) {
        uint4 a = frags[(2 * NK + s) * 64 + lane], b = frags[(2 * NK + NKA + s) * 64 + lane];
        bha[s] = v4i{(int)a.x, (int)a.y, (int)a.z, (int)a.w};
        tva[s] = v4i{(int)b.x, (int)b.y, (int)b.z, (int)b.w};
    }

    // this thread's chunks of the staged tile: row, pixel offset, LDS offset; fast = plain 16-byte load
    int c_row[NIT], c_px[NIT], c_lds[NIT];
    bool c_on[NIT], c_fast[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int c = tid + 256 * it;
        c_on[it] = c < 32 * CH;
        c_row[it] = c / CH;
        int ch = c - c_row[it] * CH;
        c_px[it] = X0 - LEFT + 16 * ch;
        c_lds[it] = c_row[it] * STRIDE + 16 * ch;
        c_fast[it] = aligned && c_px[it] >= 0 && c_px[it] + 16 <= W;
    }
    // plain chunks are loaded a step ahead into registers and written to LDS at the end of the step; border chunks
    // (few, only in the first / last workgroup of a row) are gathered byte by byte at commit time.  (Two steps ahead,
    // paid for by reading the small kernel's vertical fragments from LDS: measured slower, 1.65 against 1.44 us.)
    uint4 stage[NIT];
    auto row_of = [&](int t, int it) { return g + (int64_t)reflect101(Y0 - LEFT + 32 * t + c_row[it], H) * gstride_row; };
    auto fetch = [&](int t) {
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (c_on[it] && c_fast[it]) {
                // (rows24: the row pitch and a frame's size fit 24 / 31 bits - one full-rate multiply and a 32-bit
                //  offset from the scalar frame pointer instead of a 64-bit multiply per chunk and step)
                const u32* s32 = rows24 ? reinterpret_cast<const u32*>(g + (u32)(__mul24(reflect101(Y0 - LEFT + 32 * t + c_row[it], H), (int)gstride_row) + c_px[it]))
                                        : reinterpret_cast<const u32*>(row_of(t, it) + c_px[it]);
                stage[it] = make_uint4(s32[0], s32[1], s32[2], s32[3]);
            }
    };
    auto commit = [&](int t, int buf) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (!c_on[it]) continue;
            if (c_fast[it]) {
                uint4 v = stage[it];
                v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
                *reinterpret_cast<uint4*>(&tile[buf][c_lds[it]]) = v;
            } else {
                uint4 v = fetch_chunk_slow(row_of(t, it), c_px[it], W);
                v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
                *reinterpret_cast<uint4*>(&tile[buf][c_lds[it]]) = v;
            }
        }
    };

    v4i rLh[NK], rLl[NK], rSh[NK], rSl[NK];            // ring of horizontal tiles: large / small kernel, hi / lo bytes
#pragma unroll
    for (int s = 0; s < NK; ++s) rLh[s] = rLl[s] = rSh[s] = rSl[s] = v4i{0, 0, 0, 0};
    const int xw = X0 + 32 * wave;                     // first column of this wave's strip
    u32* const bits32 = reinterpret_cast<u32*>(bits) + ((int64_t)n * H * WW + (xw >> 6)) * 2 + ((xw >> 5) & 1);   // (uniform)
    const u32 colmask = xw + 32 <= W ? 0xFFFFFFFFu : (xw >= W ? 0u : ((1u << (W - xw)) - 1u));
    // sum tap*H = 256*Dhi + Dlo + 256*(128 + 32768); + 2^15 to round; the large kernel also carries
    // (15 - thresh) << 16 so that its high word is im_blur_8 + 15 - thresh (mod 2^16)
    // (host computes k3 = 256*(128+32768) + 2^15, k8 = k3 + (15 - thresh) << 16, span = hi - thresh)
    const u32 span = (u32)span_i;
    u32 total = 0, pend_off = 0xFFFFFFFFu, pend_full = 0;

    fetch(0);
    commit(0, 0);
    // every load issued so far (the operand fragments above all) has landed: without this the loop's first uses
    // keep a vmcnt wait that, in steady state, stalls on the prefetch of the next tile instead
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0)
    __syncthreads();
    for (int t0 = 0; t0 < nsteps; t0 += NK) {
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            const int t = t0 + u;
            if (t >= nsteps) break;                    // uniform
            const bool more = t + 1 < nsteps;
            if (more) fetch(t + 1);
            const u8* tb = &tile[t & 1][m * STRIDE + 32 * wave + 16 * hh];
            v4i a[NK];
#pragma unroll
            for (int s = 0; s < NK; ++s) a[s] = *reinterpret_cast<const v4i*>(tb + 32 * s);
            {
                v16i acc = {};
#pragma unroll
                for (int s = 0; s < NK; ++s) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], bh[s], acc, 0, 0, 0);
                pack_tile(acc, rLh[u], rLl[u]);
            }
            {
                v16i acc = {};
#pragma unroll
                for (int s = 0; s < NKA; ++s) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[SA0 + s], bha[s], acc, 0, 0, 0);
                pack_tile(acc, rSh[u], rSl[u]);
            }
            if (t >= NK - 1 && dbg != 2) {
                v16i d8 = {}, d3 = {};
#pragma unroll
                for (int o = 0; o < NK; ++o) d8 = __builtin_amdgcn_mfma_i32_32x32x32_i8(rLh[(u + 1 + o) % NK], tv[o], d8, 0, 0, 0);
#pragma unroll
                for (int o = 0; o < NKA; ++o) d3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(rSh[(u + 1 + SA0 + o) % NK], tva[o], d3, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) d8[i] = (d8[i] << 8) + k8;
#pragma unroll
                for (int o = 0; o < NK; ++o) d8 = __builtin_amdgcn_mfma_i32_32x32x32_i8(rLl[(u + 1 + o) % NK], tv[o], d8, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; ++i) d3[i] = (d3[i] << 8) + k3;
#pragma unroll
                for (int o = 0; o < NKA; ++o) d3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(rSl[(u + 1 + SA0 + o) % NK], tva[o], d3, 0, 0, 0);
                if (dbg == 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("" :: "v"(d8[i]), "v"(d3[i]));
                } else {
                u32 sgn = 0;                           // bit i = 1 when register i is OUT of range
#pragma unroll
                for (int i = 15; i >= 0; --i) {
                    u32 dg = (((u32)d8[i] >> 16) - ((u32)d3[i] >> 16)) & 255u;   // (blur_8 - blur_3 + 15 - thresh) mod 256 (:128)
                    sgn = __builtin_amdgcn_alignbit(sgn, span - dg, 31);
                }
                u32 w16 = ~sgn & 0xFFFFu;              // register i = column (i&3) + 8(i>>2) + 4*half
                u32 w32 = ((w16 & 0xFu) | ((w16 & 0xF0u) << 4) | ((w16 & 0xF00u) << 8) | ((w16 & 0xF000u) << 12)) << (4 * hh);
                const int y = Y0 + 32 * (t - (NK - 1)) + m;
                w32 = (y < H) ? (w32 & colmask) : 0u;
                u32 full = w32 | (u32)__shfl_xor((int)w32, 32);
                if (hh == 0 && y < H && (xw >> 6) < WW) {    // stored behind this step's commit (below)
                    pend_off = (u32)__mul24(y, 2 * WW);
                    pend_full = full;
                    total += __popc(full);
                }
                if (U8OUT && y < H) {                    // uint8 image for the staged API: 4 pixels per store where possible
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int x = xw + 8 * q + 4 * hh;
                        const u32 nib = (w32 >> (8 * q + 4 * hh)) & 15u;
                        const u32 four = ((nib * 0x00204081u) & 0x01010101u) * 0xFFu;      // bit r -> byte r = 0 / 255
                        u8* dst = area_u8 + ((int64_t)n * H + y) * W + x;
                        if (((W & 3) == 0) && x + 3 < W && ((reinterpret_cast<uintptr_t>(area_u8) & 3) == 0)) {
                            *reinterpret_cast<u32*>(dst) = four;
                        } else {
                            for (int r = 0; r < 4; ++r)
                                if (x + r < W) dst[r] = (u8)(four >> (8 * r));
                        }
                    }
                }
                }                                      // (dbg != 1)
            }
            if (more) commit(t + 1, (t + 1) & 1);
            // The mask word goes out only now: issued before the commit, its acknowledgement would be part of the commit's
            // wait for the prefetched rows, every step and for all four waves at the barrier.
            if (pend_off != 0xFFFFFFFFu) { bits32[pend_off] = pend_full; pend_off = 0xFFFFFFFFu; }
            __syncthreads();
        }
    }
#pragma unroll
    for (int off = 32; off; off >>= 1) total += __shfl_xor((int)total, off);
    if (lane == 0 && total) atomicAdd(&fstat[n * 8 + 0], total);
}

// Toeplitz operand fragments in the lane layout of v_mfma_i32_32x32x32_i8 (lane = 32*half + column; a
// lane's 16 bytes pair with the other operand's 16 bytes of the same half, so only the pairing matters):
//   horizontal (B operand, k = pixel of the staged window): byte e of half h is window pixel 32s + 16h + e
//   vertical   (B operand, k = row of ring tile o):         byte 4q + r of half h is tile row 8q + 4h + r
std::vector<u32> blur_mfma_fragments(const std::vector<int>& taps_a, const std::vector<int>& taps_b, int nk,
                                     int sa0, int nka) {
    const int left = 32 * ((nk - 1) / 2);
    std::vector<u32> out((size_t)(2 * nk + 2 * nka) * 64 * 4, 0);
    auto fill = [&](int frag, const std::vector<int>& taps, int kbase, bool vertical) {
        const int R = (int)taps.size() / 2;
        for (int lane = 0; lane < 64; ++lane) {
            const int h = lane >> 5, col = lane & 31;
            for (int e = 0; e < 16; ++e) {
                int k = kbase + (vertical ? 8 * (e >> 2) + 4 * h + (e & 3) : 16 * h + e);
                int idx = k - col - (left - R);
                u32 v = (idx >= 0 && idx <= 2 * R) ? (u32)taps[idx] : 0u;
                out[((size_t)frag * 64 + lane) * 4 + (e >> 2)] |= v << (8 * (e & 3));
            }
        }
    };
    for (int s = 0; s < nk; ++s) { fill(0 * nk + s, taps_b, 32 * s, false); fill(1 * nk + s, taps_b, 32 * s, true); }
    for (int s = 0; s < nka; ++s) {
        fill(2 * nk + s, taps_a, 32 * (sa0 + s), false);
        fill(2 * nk + nka + s, taps_a, 32 * (sa0 + s), true);
    }
    return out;
}


// ---- strips of 16 columns ---------------------------------------------------------------------------
// The same two products on v_mfma_i32_16x16x64_i8, one wave per 16-column strip sliding down 16 rows per step, for the
// large branch (101 / 39 taps) on frames whose rows can be loaded as aligned dwords.  Against k_blur_mfma:
//   * a 16 + 100 pixel window fits K = 128 (32-column strips: 160) and a 16 + 100 row window eight 16-row tiles
//     (32-row tiles: five of 32), so a quarter of the matrix work on the Toeplitz zero band is gone
//   * the ring of horizontal tiles is 8 + 4 dwords per byte plane instead of 80 registers: 100 registers per lane, 4
//     waves per SIMD instead of 2
//   * reflect-101 at the left / right border is folded into the strip's own horizontal fragments (a pixel that the
//     border mirrors onto carries the sum of the taps that reach it; the window is shifted to stay inside the row), so
//     nothing is ever gathered byte by byte at the image border
//   * the vertical result has the column on the lane and four rows in registers: the range test is one subtraction
//     of byte 2 and one compare per register, whose lane mask IS 4 rows x 16 mask bits
//   * no workgroup barrier in the loop, and the strips never load a row: see below
// A workgroup = seven strips (waves 0..6) and ONE LOADER WAVE (wave 7).  The workgroup's window is 240 bytes of each row,
// [X0, X0 + 240) with X0 = 112 bx - 64 shifted to stay inside the row; the loader keeps four tiles (16 rows x 15 pieces
// of 16 bytes, four per lane) on their way from memory, stages a tile into one of eight ring slots in LDS once every strip
// has ticked off the tile that was there, and announces it by counting `staged` up; a strip keeps the last count it saw and
// asks again only when it needs a tile beyond it (the loader runs tiles ahead: every few steps), reads its own 128-byte
// window [L0, L0 + 128) out of the slot as operand P = window bytes 0..31 and 96..127 and operand Q = bytes 32..95 (all the 39-tap kernel needs away
// from the border), and ticks the slot off one step later.  The first touch of a row from HBM - which every strip of the
// frame used to wait for at about the same time - is the loader's business four tiles ahead of anybody's need, and the
// L1 sees 64 tag lookups per tile instead of 8 x 62.
// (On the way here, us per frame at 1280x1024: every wave loading its own window straight into the operand layout 1.87 -
// every lane of a load quad on another cache line; four lanes per 64 bytes of a row and a per-wave LDS hop 1.41; + a
// frame's workgroups on one XCD 1.25, + eight strips per workgroup 1.21 = round 3's first product form; the shared
// window with a barrier per tile 1.55, with tick counters in every wave 1.7, with this loader wave 3.4 - all three because
// the pieces past the image border were gathered byte by byte behind a full wait for memory, in two workgroups of every
// frame; with the border in the fragments instead, as above: 1.06-1.13; the strips re-reading the loader's count only
// when they need a tile beyond the last one seen: 0.98-1.07; eight slots and four tiles in flight: 0.97-1.00.  With no row
// loads at all: 0.78-1.06.)
// Horizontal tile t = rows Y0 - 56 + 16 t ..: the output tile of step t (rows Y0 + 16 (t - 7) ..) reads tiles t-7 .. t of
// the large kernel and tiles t-5 .. t-2 of the small one.  Ring slot = t mod 8 (mod 4), the step loop is unrolled by 8,
// and what changes with the phase is the vertical fragment: 8 + 4 variants in LDS.
// SMALL BRANCH (round 4; 35 / 21 taps, the reference's real configuration: 640x480 through the default crop = 480 x 450):
// the same kernel with SB = true.  A 16 + 34 pixel window fits ONE K = 64 operand (two horizontal products per step instead
// of three or four), a 16 + 34 row window four 16-row tiles, which both kernels share (tiles t-3 .. t: one 4-tile ring per
// byte plane, four vertical products), the workgroup's window is 176 bytes of a row (11 pieces) and the output tile of step
// t is tile t - 3.  Everything else - loader wave, ring of eight slots, hand-shake, range test, mask stores - is the code
// below, unchanged.
// WIDTHS THAT ARE A MULTIPLE OF 4 BUT NOT OF 8 (round 4): the workgroup at the right end of a row has its window shifted
// to X0 = W - 240, which is then 4 (mod 8), while the strips read their operands with 8-byte LDS loads.  The staged rows
// keep an 8-aligned origin XB = X0 & ~7 instead (the loader stores that workgroup's pieces as dwords at byte X0 - XB),
// so every strip's offset L0 - XB stays a multiple of 8.
#define B16_NSW 7                                        // strips (compute waves) per workgroup; wave B16_NSW is the loader
#define B16_NS 8                                         // tiles of rows in LDS (ring slots; 4: +3 %, 6: +2 % on the kernel's time)
#define B16_LD 4                                         // tiles the loader keeps on their way from memory
template <bool SB> struct B16 {
    static constexpr int LEFT = SB ? 24 : 56;            // window / tile origin left of (above) the strip's first column (row)
    static constexpr int WIN = SB ? 64 : 128;            // a strip's window
    static constexpr int NP = SB ? 11 : 15;              // 16-byte pieces of the workgroup's window per row: 112 + WIN columns
    static constexpr int ROWB = SB ? 208 : 272;          // LDS bytes per staged row: pieces + pad (+ 4 for the shifted origin)
    static constexpr int DEPTH = SB ? 3 : 7;             // the output tile of step t is tile t - DEPTH
    static constexpr int NVF = SB ? 8 : 12;              // vertical fragment variants (ring phases of the two kernels)
    static constexpr int WG_LEFT = SB ? 32 : 64;         // the workgroup's window starts this far left of its first strip
};
#define B16_LEFT 56                                      // (large branch, host side)

__device__ __forceinline__ void pack16(const v4i& acc, int& hi, int& lo) {
    const u32 t01 = __builtin_amdgcn_perm((u32)acc[1], (u32)acc[0], 0x05010400u);
    const u32 t23 = __builtin_amdgcn_perm((u32)acc[3], (u32)acc[2], 0x05010400u);
    lo = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u);
    hi = (int)__builtin_amdgcn_perm(t23, t01, 0x07060302u);
}

template <bool U8OUT, bool SB, bool SHIFT>               // SHIFT: widths that are 4 (mod 8) - the last workgroup's shifted origin
__global__ __launch_bounds__(64 * (B16_NSW + 1), 4) void k_blur16(const u8* __restrict__ gray, int64_t gstride_n, int gstride_row,
                                                   const uint4* __restrict__ hfrag, const uint4* __restrict__ vfrag,
                                                   u64* __restrict__ bits, u8* __restrict__ area_u8,
                                                   u32* __restrict__ fstat, int H, int W, int WW, int tiles_per_seg,
                                                   int k3, int k8, int span_i, int nframes, int gx, int gy, int dbg_drop) {
    typedef B16<SB> G;
    constexpr int B16_LEFT_ = G::LEFT, B16_NP = G::NP, B16_ROWB = G::ROWB, DEPTH = G::DEPTH;
    __shared__ uint4 vf[G::NVF * 64];
    __shared__ __align__(16) u8 stg[B16_NS][16 * G::ROWB];
    __shared__ u32 staged;                               // tiles the loader has staged so far
    __shared__ u32 done[B16_NS];                         // done[s]: reads of slot s the strips have finished, ever
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, q = lane & 15;
    // Workgroups are dealt round-robin over the 8 XCDs (observed, not promised: it only matters for speed).  With many
    // frames in the launch, workgroup b works on frame 8 (b / 8 / per_frame) + b % 8: the workgroups of one frame - whose
    // windows overlap - then share one XCD's L2 and the frame crosses the fabric once.
    int n, bx, by;
    if (nframes) {
        const int b = blockIdx.x, per = gx * gy, j = b >> 3;
        n = 8 * (j / per) + (b & 7);
        const int w = j % per;
        bx = w % gx; by = w / gx;
        if (n >= nframes) return;
    } else { n = blockIdx.z; bx = blockIdx.x; by = blockIdx.y; }
    const int tilesY = (H + 15) / 16;
    const int tile0 = by * tiles_per_seg;
    const int ntiles = min(tiles_per_seg, tilesY - tile0);
    if (ntiles <= 0) return;
    for (int i = tid; i < G::NVF * 64; i += 64 * (B16_NSW + 1)) vf[i] = vfrag[i];
    const bool is_loader = wave == B16_NSW;              // (uniform)
    const int strip = bx * B16_NSW + wave, xw = 16 * strip;
    const int Y0 = tile0 * 16, nsteps = ntiles + DEPTH;
    const bool live = !is_loader && xw < W;              // (uniform) else a strip in the padding of the last mask word
    if (!is_loader && !live && xw < 64 * WW)
        for (int y = Y0 + lane; y < min(Y0 + 16 * ntiles, H); y += 64)
            reinterpret_cast<unsigned short*>(bits)[((int64_t)n * H + y) * WW * 4 + strip] = 0;
    const int L0 = min(max(xw - B16_LEFT_, 0), W - G::WIN);   // this strip's window [L0, L0 + WIN): shifted to stay inside the row
    const bool edge = !SB && L0 != xw - B16_LEFT_;       // (uniform) the 39-tap window is not all inside Q
    v4i lp, lq, sq, sp = {0, 0, 0, 0};
    {
        const uint4* hf = hfrag + (size_t)min(strip, (W + 15) / 16 - 1) * 4 * 64 + lane;
        const uint4 b = hf[64], c = hf[128];
        lq = v4i{(int)b.x, (int)b.y, (int)b.z, (int)b.w};
        sq = v4i{(int)c.x, (int)c.y, (int)c.z, (int)c.w};
        lp = v4i{0, 0, 0, 0};
        if (!SB) { const uint4 a = hf[0]; lp = v4i{(int)a.x, (int)a.y, (int)a.z, (int)a.w}; }
        if (edge) { const uint4 d = hf[192]; sp = v4i{(int)d.x, (int)d.y, (int)d.z, (int)d.w}; }
    }
    const u8* gf = gray + (int64_t)n * gstride_n;        // (uniform)
    const int X0 = min(max(16 * B16_NSW * bx - G::WG_LEFT, 0), W - 16 * B16_NP);   // the workgroup's window [X0, X0 + 16 NP): inside the row
    const int XB = SHIFT ? (X0 & ~7) : X0;               // origin of the staged rows (see "widths that are a multiple of 4" above)
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // operand fragments landed (see k_blur_mfma)
    if (tid < B16_NS) done[tid] = 0;
    if (tid == B16_NS) staged = 0;
    __syncthreads();                                     // (the only barrier: fragments and counters are in place)

    // ================================ the loader wave ===============================================================
    // Tile tau = rows Y0 - 56 + 16 tau ..: 16 rows x 15 pieces of 16 bytes, four per lane (id = 64 k + lane: row id / 15,
    // piece id % 15).  Four tiles are on their way at any time (four register sets, the loop unrolled by four), a tile
    // is staged into ring slot tau % B16_NS once every strip has ticked off its reads of tile tau - B16_NS, and
    // announced by counting `staged` up.  The strips never load a row: nothing of theirs queues behind a first touch of
    // HBM, and this wave sees that latency four tiles deep.  Loads and waits are inline assembly: "at most 12
    // outstanding" = the three younger tiles' loads (always issued, also past the last tile: clamped rows nobody reads),
    // loads return in order.  The window lies inside the row, so every piece is a plain 16-byte load; every lane always issues
    // its four loads (the 16 lanes without a fourth piece repeat piece 0), so the count of operations in flight is fixed.
    if (is_loader) {
        int prow[4];
        bool pval[4];
        u32 poff[4];
        u8* pdst[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int id = 64 * k + lane;
            pval[k] = id < 16 * B16_NP;
            prow[k] = pval[k] ? id / B16_NP : 0;
            const int pc = pval[k] ? id - B16_NP * prow[k] : 0;
            poff[k] = (u32)(__mul24(prow[k], gstride_row) + X0 + 16 * pc);
            pdst[k] = &stg[0][0] + B16_ROWB * prow[k] + 16 * pc;      // (16-byte aligned; + X0 - XB where the origin is shifted)
        }
        u32x4 R[B16_LD][4];
#pragma unroll
        for (int j = 0; j < B16_LD; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) R[j][k] = u32x4{0, 0, 0, 0};
        auto row_of = [&](int t, int r) { return reflect101(Y0 - B16_LEFT_ + 16 * t + r, H); };
        auto issue = [&](int t, u32x4 (&Rt)[4]) {
            const int yt = Y0 - B16_LEFT_ + 16 * t;      // (uniform)
            u32 o[4];
            if (yt >= 0 && yt + 15 < H) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (u32)(yt * gstride_row) + poff[k];
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (u32)(__mul24(row_of(t, prow[k]) - prow[k], gstride_row)) + poff[k];
            }
            asm volatile("global_load_dwordx4 %0, %4, %8\n\tglobal_load_dwordx4 %1, %5, %8\n\t"
                         "global_load_dwordx4 %2, %6, %8\n\tglobal_load_dwordx4 %3, %7, %8"
                         : "=&v"(Rt[0]), "=&v"(Rt[1]), "=&v"(Rt[2]), "=&v"(Rt[3])
                         : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "s"(gf) : "memory");
        };
#pragma unroll
        for (int j = 0; j < B16_LD; ++j) issue(j, R[j]);
        int slot = 0, gen = 0;                           // slot = tau % B16_NS, gen = tau / B16_NS
        bool lost = false;                               // a bounded wait of this wave has expired
        for (int t0 = 0; t0 < nsteps; t0 += B16_LD) {
#pragma unroll
            for (int j = 0; j < B16_LD; ++j) {
                const int t = t0 + j;
                if (t >= nsteps) break;                  // uniform
                if (gen > 0) {                           // the slot's last tile read by every strip?  (bounded spin)
                    const u32 want = (u32)(B16_NSW * gen);
                    bool ok = false;
                    for (int spin = 0; spin < (lost ? 1 : (1 << 20)); ++spin) {
                        const u32 have = __hip_atomic_load(&done[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (__builtin_amdgcn_readfirstlane(have) >= want) { ok = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    // expired: the frame is reported, never continued silently (and later waits of this wave give up at once)
                    if (!ok && !lost) { lost = true; if (lane == 0) atomicMin((int*)&fstat[n * 8 + 2], VBS_EINTERNAL); }
                }
                static_assert(B16_LD == 4, "the wait below leaves the B16_LD - 1 younger tiles' loads outstanding");
                asm volatile("s_waitcnt vmcnt(12)" : "+v"(R[j][0]), "+v"(R[j][1]), "+v"(R[j][2]), "+v"(R[j][3]) :: "memory");
                const int sb = slot * (16 * B16_ROWB);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    u8* dst = pdst[k] + sb;
                    const uint4 v = make_uint4(R[j][k].x ^ 0x80808080u, R[j][k].y ^ 0x80808080u,
                                               R[j][k].z ^ 0x80808080u, R[j][k].w ^ 0x80808080u);
                    // (The loader is as heavy as a strip and sits on the kernel's critical path: every instruction added per
                    //  tile shows.  Widths that are a multiple of 8 - SHIFT = false - compile to the four plain 16-byte stores;
                    //  the pointer must be KNOWN to be aligned, or the compiler splits each store into 12 + 4 bytes: +20 %
                    //  on the kernel, and a run-time test per piece cost +17 %.)
                    if (!SHIFT || X0 == XB) {            // (uniform)
                        if (pval[k]) *reinterpret_cast<uint4*>(__builtin_assume_aligned(dst, 16)) = v;
                    } else if (pval[k]) {                // the shifted origin: 4-byte aligned pieces
                        u32* d32 = reinterpret_cast<u32*>(dst + (X0 - XB));
                        d32[0] = v.x; d32[1] = v.y; d32[2] = v.z; d32[3] = v.w;
                    }
                }
                // (release: the rows above are in LDS before the tick is)
#ifdef VBS_DEBUG_KNOBS
                // tests/: VBS_BLUR16_DROP = tile whose tick the loader of workgroup 0 "forgets" - the strips' wait for it
                // must expire into the frame's status word
                if (!(dbg_drop > 0 && t == dbg_drop - 1 && bx == 0 && by == 0))
#endif
                if (lane == 0) __hip_atomic_fetch_add(&staged, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                issue(t + B16_LD, R[j]);
                if (++slot == B16_NS) { slot = 0; ++gen; }
            }
        }
#pragma unroll
        for (int j = 0; j < B16_LD; ++j)                 // (the loads past the last tile, never used, have landed)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(R[j][0]), "+v"(R[j][1]), "+v"(R[j][2]), "+v"(R[j][3]) :: "memory");
        return;
    }

    // ================================ the strips ======================================================================
    // LDS row: pixel x of the workgroup's window at byte 8 + (x - X0), so that every strip's operand window (it starts
    // 8 + 16 wave pixels in) is 16-byte aligned.  Lane (g, q) takes row q: P = window bytes 16 g (+ 64 for g >= 2), Q = 32 + 16 g.
    const u8* const rP = &stg[0][0] + B16_ROWB * q + (L0 - XB) + 16 * g + (g >= 2 ? 64 : 0);
    const u8* const rQ = &stg[0][0] + B16_ROWB * q + (L0 - XB) + (SB ? 0 : 32) + 16 * g;
    // A tile is read once the loader has counted it into `staged` (the strip remembers the last count it saw and asks again
    // only beyond it: the loader runs tiles ahead), and ticked off a step later, when its operands have been multiplied.  Bounded spins: a logic error cannot hang the GPU;
    // a wait that expires sets the frame's status word (VBS_EINTERNAL in counts[]).
    int rslot = 0;                                       // ring slot of the NEXT tile to read
    u32 rt = 0, known = 0;                               // its index; tiles known to be staged (asked for again only beyond it)
    bool lost = false;                                   // a bounded wait of this wave has expired
    auto read_ops = [&](uint4& p, uint4& qq) {
        if (rt >= known) {
            for (int spin = 0; spin < (lost ? 1 : (1 << 20)); ++spin) {
                known = (u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&staged, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (rt < known) break;
                __builtin_amdgcn_s_sleep(1);
            }
            // expired: reported in the frame's status word (k_finalize hands it to counts[]), never continued silently
            if (rt >= known && !lost) { lost = true; if (lane == 0) atomicMin((int*)&fstat[n * 8 + 2], VBS_EINTERNAL); }
        }
        const uint2* b = reinterpret_cast<const uint2*>(rQ + rslot * (16 * B16_ROWB));   // (8-byte aligned: two halves each)
        const uint2 b0 = b[0], b1 = b[1];
        qq = make_uint4(b0.x, b0.y, b1.x, b1.y);
        if (!SB) {
            const uint2* a = reinterpret_cast<const uint2*>(rP + rslot * (16 * B16_ROWB));
            const uint2 a0 = a[0], a1 = a[1];
            p = make_uint4(a0.x, a0.y, a1.x, a1.y);
        }
        ++rt;
        if (++rslot == B16_NS) rslot = 0;
    };
    int dslot = 0;                                       // ring position of the next tile to tick off
    auto tick_done = [&]() {
        if (lane == 0) __hip_atomic_fetch_add(&done[dslot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (++dslot == B16_NS) dslot = 0;
    };
    v4i LhA = {0, 0, 0, 0}, LhB = {0, 0, 0, 0}, LlA = {0, 0, 0, 0}, LlB = {0, 0, 0, 0};   // large kernel: ring slots 0-3 / 4-7
    v4i S4h = {0, 0, 0, 0}, S4l = {0, 0, 0, 0};          // small kernel: tiles t-5 .. t-2
    int Sdh[4] = {0, 0, 0, 0}, Sdl[4] = {0, 0, 0, 0};    // small kernel: the last four tiles (slot t mod 4)
    const u32 span = (u32)span_i;
    const u32 m16 = xw + 16 <= W ? 0xFFFFu : (live ? ((1u << (W - xw)) - 1u) : 0u);
    const u64 colmask = (u64)m16 * 0x0001000100010001ull;
    unsigned short* mb16 = reinterpret_cast<unsigned short*>(bits) + (int64_t)n * H * WW * 4 + strip;   // (uniform)
    const u32 sel1 = 0u - ((u32)lane & 1u), sel2 = 0u - (((u32)lane >> 1) & 1u);
    // Mask rows leave once per eight steps (lane j < 16 keeps the 16-bit pieces of rows yo + j of the eight tiles in four
    // registers).
    u32 total = 0, pp[4] = {0, 0, 0, 0};
    auto flush_rows = [&](int tg) {                      // tiles of steps tg .. tg + 7
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int y = Y0 + 16 * (tg + k - DEPTH) + lane;
            if (live && lane < 16 && tg + k >= DEPTH && tg + k < nsteps && y < H)
                mb16[(u32)__mul24(y, 4 * WW)] = (unsigned short)(pp[k >> 1] >> (16 * (k & 1)));
        }
    };
    // range test of a finished tile (d8 / d3: the two blurs' vertical results, rows yo + 4 g + i of column xw + q) and its
    // 16-bit mask pieces into pp[] (slot u of the eight-step group)
    auto finish_tile = [&](const v4i& d8, const v4i& d3, int yo, int u) {
            u64 pw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32 dg = (((u32)d8[i] >> 16) - ((u32)d3[i] >> 16)) & 255u;     // (blur_8 - blur_3 + 15 - thresh) mod 256 (:128)
                pw[i] = __ballot(dg <= span) & colmask;
            }
            if (yo + 15 >= H) {                          // uniform: the last tile sticks out of the image
#pragma unroll
                for (int i = 0; i < 4; ++i) pw[i] &= __ballot(yo + 4 * g + i < H);
            }
            total += (u32)(__popcll(pw[0]) + __popcll(pw[1]) + __popcll(pw[2]) + __popcll(pw[3]));
            if (U8OUT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int y = yo + 4 * g + i, x = xw + q;
                    if (y < H && x < W) area_u8[((int64_t)n * H + y) * W + x] = (u8)(((pw[i] >> lane) & 1ull) ? 255 : 0);
                }
            }
            {
                // lane j < 16 keeps row yo + j: quarter j >> 2 of pw[j & 3] (as k_ncc_mfma does)
                const u32 x01l = (u32)pw[0] ^ (u32)pw[1], x01h = (u32)(pw[0] >> 32) ^ (u32)(pw[1] >> 32);
                const u32 x23l = (u32)pw[2] ^ (u32)pw[3], x23h = (u32)(pw[2] >> 32) ^ (u32)(pw[3] >> 32);
                const u32 t0l = (x01l & sel1) ^ (u32)pw[0], t0h = (x01h & sel1) ^ (u32)(pw[0] >> 32);
                const u32 t1l = (x23l & sel1) ^ (u32)pw[2], t1h = (x23h & sel1) ^ (u32)(pw[2] >> 32);
                const u32 vl = ((t0l ^ t1l) & sel2) ^ t0l, vh = ((t0h ^ t1h) & sel2) ^ t0h;
                const u32 piece = ((lane & 8 ? vh : vl) >> (16 * ((lane >> 2) & 1))) & 0xFFFFu;
                pp[u >> 1] = (u & 1) ? (pp[u >> 1] | (piece << 16)) : piece;
            }
    };
    uint4 aP_ = make_uint4(0, 0, 0, 0), aQ_;
    read_ops(aP_, aQ_);                                  // tile 0
    for (int t0 = 0; t0 < nsteps; t0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {                    // u = ring slot of step t: a constant of this copy of the body
            const int t = t0 + u;
            if (t >= nsteps) break;                      // uniform
            const v4i aP = v4i{(int)aP_.x, (int)aP_.y, (int)aP_.z, (int)aP_.w}, aQ = v4i{(int)aQ_.x, (int)aQ_.y, (int)aQ_.z, (int)aQ_.w};
            if (!SB) {
            // this step's vertical fragments, asked for here and used after the horizontal products
            const uint4 fa_ = vf[u * 64 + lane], fb_ = vf[((u + 4) & 7) * 64 + lane], fs_ = vf[(8 + (u & 3)) * 64 + lane];
            // ---- horizontal tile t ----
            v4i accL = {0, 0, 0, 0}, accS = {0, 0, 0, 0};
            accL = __builtin_amdgcn_mfma_i32_16x16x64_i8(aP, lp, accL, 0, 0, 0);
            accS = __builtin_amdgcn_mfma_i32_16x16x64_i8(aQ, sq, accS, 0, 0, 0);
            accL = __builtin_amdgcn_mfma_i32_16x16x64_i8(aQ, lq, accL, 0, 0, 0);
            if (edge) accS = __builtin_amdgcn_mfma_i32_16x16x64_i8(aP, sp, accS, 0, 0, 0);
            // behind the products: tile t's slot is free (its operands have been multiplied), tile t + 1 into next step's operands
            tick_done();
            if (t + 1 < nsteps) read_ops(aP_, aQ_);
            {
                int hi, lo;
                pack16(accL, hi, lo);
                if (u < 4) { LhA[u & 3] = hi; LlA[u & 3] = lo; } else { LhB[u & 3] = hi; LlB[u & 3] = lo; }
                pack16(accS, hi, lo);
                Sdh[u & 3] = hi; Sdl[u & 3] = lo;
                S4h[(u + 2) & 3] = Sdh[(u + 2) & 3];     // tile t - 2 (zeros for t < 2) takes the place of tile t - 6
                S4l[(u + 2) & 3] = Sdl[(u + 2) & 3];
            }
            if (t >= DEPTH) {
            // ---- vertical: output rows yo .. yo + 15, lane (g, q) gets rows yo + 4 g + i of column xw + q ----
            const v4i fa = v4i{(int)fa_.x, (int)fa_.y, (int)fa_.z, (int)fa_.w}, fb = v4i{(int)fb_.x, (int)fb_.y, (int)fb_.z, (int)fb_.w};
            const v4i fs = v4i{(int)fs_.x, (int)fs_.y, (int)fs_.z, (int)fs_.w};
            v4i d8 = {0, 0, 0, 0}, d3 = {0, 0, 0, 0};
            d8 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, LhA, d8, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fs, S4h, d3, 0, 0, 0);
            d8 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb, LhB, d8, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) d3[i] = (d3[i] << 8) + k3;
            d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fs, S4l, d3, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) d8[i] = (d8[i] << 8) + k8;
            d8 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, LlA, d8, 0, 0, 0);
            d8 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fb, LlB, d8, 0, 0, 0);
            finish_tile(d8, d3, Y0 + 16 * (t - DEPTH), u);
            }
            } else {
            // small branch: one window operand, both kernels over the same four tiles (ring slot u & 3; LhA / LlA hold the
            // 35-tap kernel's byte planes, S4h / S4l the 21-tap kernel's)
            const uint4 fa_ = vf[(u & 3) * 64 + lane], fs_ = vf[(4 + (u & 3)) * 64 + lane];
            v4i accL = {0, 0, 0, 0}, accS = {0, 0, 0, 0};
            accL = __builtin_amdgcn_mfma_i32_16x16x64_i8(aQ, lq, accL, 0, 0, 0);
            accS = __builtin_amdgcn_mfma_i32_16x16x64_i8(aQ, sq, accS, 0, 0, 0);
            tick_done();
            if (t + 1 < nsteps) read_ops(aP_, aQ_);
            {
                int hi, lo;
                pack16(accL, hi, lo);
                LhA[u & 3] = hi; LlA[u & 3] = lo;
                pack16(accS, hi, lo);
                S4h[u & 3] = hi; S4l[u & 3] = lo;
            }
            if (t >= DEPTH) {
            const v4i fa = v4i{(int)fa_.x, (int)fa_.y, (int)fa_.z, (int)fa_.w}, fs = v4i{(int)fs_.x, (int)fs_.y, (int)fs_.z, (int)fs_.w};
            v4i d8 = {0, 0, 0, 0}, d3 = {0, 0, 0, 0};
            d8 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, LhA, d8, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fs, S4h, d3, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) d8[i] = (d8[i] << 8) + k8;
#pragma unroll
            for (int i = 0; i < 4; ++i) d3[i] = (d3[i] << 8) + k3;
            d8 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa, LlA, d8, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(fs, S4l, d3, 0, 0, 0);
            finish_tile(d8, d3, Y0 + 16 * (t - DEPTH), u);
            }
            }
            if (u == 7) flush_rows(t0);
        }
    }
    if (nsteps & 7) flush_rows(nsteps & ~7);
    if (live && lane == 0 && total) atomicAdd(&fstat[n * 8 + 0], total);
}

// Operand fragments of k_blur16 (lane = 16 g + index, byte e of a lane <-> k = 16 g + e):
//   horizontal, per strip: LP, LQ, SQ, SP - B operands over the window pixels of P / Q, output column on the lane; the
//     entry for window pixel x_in and output column x is the sum of the taps j with reflect101(x + j - R) = x_in
//   vertical: A operands, output row on the lane; byte 4 pos + i of lane group g is row 4 g + i of the tile in ring slot
//     pos; variant psi for the phase of the ring (8 for the large kernel's two groups, 4 for the small kernel's)
void blur16_fragments(const std::vector<int>& taps_s, const std::vector<int>& taps_l, int W, bool small, std::vector<u32>* hfrag,
                      std::vector<u32>* vfrag) {
    const int nstrips = (W + 15) / 16;
    const int LEFT = small ? B16<true>::LEFT : B16<false>::LEFT, WIN = small ? B16<true>::WIN : B16<false>::WIN;
    auto refl = [&](int i) { if (i < 0) i = -i; if (i >= W) i = 2 * (W - 1) - i; return std::min(std::max(i, 0), W - 1); };
    hfrag->assign((size_t)nstrips * 4 * 64 * 4, 0);
    for (int s = 0; s < nstrips; ++s) {
        const int xw = 16 * s, L0 = std::min(std::max(xw - LEFT, 0), W - WIN);
        for (int f = 0; f < 4; ++f) {
            // large branch: LP, LQ, SQ, SP over the two operands of the 128-pixel window; small branch: slots 1 and 2 are
            // the 35- and the 21-tap kernel over the ONE 64-pixel operand, slots 0 and 3 stay empty
            if (small && (f == 0 || f == 3)) continue;
            const std::vector<int>& taps = f < 2 ? taps_l : taps_s;
            const int R = (int)taps.size() / 2;
            const bool useP = (f == 0 || f == 3);
            for (int lane = 0; lane < 64; ++lane) {
                const int g = lane >> 4, x = std::min(xw + (lane & 15), W - 1);
                for (int e = 0; e < 16; ++e) {
                    const int k = 16 * g + e, w = small ? k : (useP ? (k < 32 ? k : k + 64) : 32 + k), xin = L0 + w;
                    int c = 0;
                    for (int j = 0; j <= 2 * R; ++j) if (refl(x + j - R) == xin) c += taps[j];
                    (*hfrag)[(((size_t)s * 4 + f) * 64 + lane) * 4 + (e >> 2)] |= (u32)(c & 255) << (8 * (e & 3));
                }
            }
        }
    }
    // vertical variants: ring phase psi; large branch 8 (101 taps over eight tiles) + 4 (39 taps over tiles t-5 .. t-2),
    // small branch 4 + 4 (both kernels over tiles t-3 .. t)
    const int nv = small ? 8 : 12, nl = small ? 4 : 8;
    vfrag->assign((size_t)nv * 64 * 4, 0);
    for (int v = 0; v < nv; ++v) {
        const std::vector<int>& taps = v < nl ? taps_l : taps_s;
        const int R = (int)taps.size() / 2;
        for (int lane = 0; lane < 64; ++lane) {
            const int g = lane >> 4, m = lane & 15;
            for (int e = 0; e < 16; ++e) {
                const int pos = e >> 2, i = e & 3;
                int a;                                   // age of the tile in ring slot `pos`: 0 = this step's
                if (small) a = (((v < nl ? v : v - nl) - pos) % 4 + 4) % 4;
                else a = v < 8 ? ((v - pos) % 8 + 8) % 8 : 2 + (((v - 8) - pos - 2) % 4 + 4) % 4;
                const int k = R + LEFT - 16 * a + 4 * g + i - m;
                const u32 c = (k >= 0 && k <= 2 * R) ? (u32)taps[k] : 0u;
                (*vfrag)[((size_t)v * 64 + lane) * 4 + (e >> 2)] |= c << (8 * (e & 3));
            }
        }
    }
}

// the strips kernel takes frames whose rows load as aligned dwords (else k_blur_mfma): large branch from 240 columns,
// small branch from 176; widths that are a multiple of 4
static bool blur16_takes(const vbs_handle* h, const u8* gray, int64_t gstride_n, int64_t gstride_row) {
    const int minw = 16 * (h->bp.small ? B16<true>::NP : B16<false>::NP);
    return h->blur_impl == 0 && h->blur16_h && h->W >= minw && (h->W & 3) == 0 && h->H >= 64 &&
           (reinterpret_cast<uintptr_t>(gray) & 3) == 0 && (gstride_n & 3) == 0 && (gstride_row & 3) == 0 &&
           gstride_row >= h->W && gstride_row < (1 << 23) && (int64_t)h->H * gstride_row < (1ll << 31);
}

void launch_gray(vbs_handle* h, const u8* frames, int nb, int channels, int64_t stride_n,
                 int64_t stride_row, u8* gray, hipStream_t s) {
    const int vec_ok = (reinterpret_cast<uintptr_t>(frames) % 16 == 0) && (stride_n % 16 == 0) && (stride_row % 16 == 0);
    const int flat = vec_ok && channels == 3 && stride_row == (int64_t)h->W * 3 && h->P == h->W && ((int64_t)h->H * h->W) % 16 == 0;
    if (flat && s == h->side) {
        const int64_t npx = (int64_t)h->H * h->W;
        VBS_LAUNCH(h, s, "k_gray", k_gray_flat, dim3((unsigned)((npx / 16 + 255) / 256), 1, nb), dim3(256), 0, s, frames, stride_n,
                   gray, npx, gray_coef(h->gray_bits));
        return;
    }
    dim3 grid = flat ? dim3((unsigned)(((int64_t)h->H * h->W + 8191) / 8192), 1, nb)
                     : dim3((unsigned)(((int64_t)h->H * (h->P / 16) + 255) / 256), 1, nb);
    VBS_LAUNCH(h, s, "k_gray", k_gray, grid, dim3(256), 0, s, frames, channels, stride_n, stride_row, gray,
                       h->H, h->W, h->P, gray_coef(h->gray_bits), vec_ok, flat);
}

void launch_blur(vbs_handle* h, const u8* gray, int64_t gstride_n, int64_t gstride_row, int nb,
                 u8* area_u8, hipStream_t s) {
    const int k3 = 256 * (128 + 32768) + 32768, k8 = k3 + (15 - h->bp.thresh) * 65536;
    if (blur16_takes(h, gray, gstride_n, gstride_row)) {
        const int gx16 = (4 * h->WW + B16_NSW - 1) / B16_NSW, tiles16 = (h->H + 15) / 16;
        int nseg = std::min(tiles16 / 8, std::max(1, (2048 + gx16 * nb - 1) / (gx16 * nb)));     // few frames: split the columns
        nseg = std::max(nseg, 1);
        if (VBS_KNOB("VBS_BLUR16_NSEG")) nseg = VBS_KNOB("VBS_BLUR16_NSEG");
        const int tps = (tiles16 + nseg - 1) / nseg;
        nseg = (tiles16 + tps - 1) / tps;
        // many frames: a 1-D grid that the kernel maps to (frame, strip group, segment) with a frame's workgroups on one XCD
        const int xcd = nb >= 32 ? nb : 0;
        dim3 grid16 = xcd ? dim3((unsigned)((nb + 7) / 8 * 8 * gx16 * nseg)) : dim3(gx16, nseg, nb);
#define B16_GO(U8, SB_, SH_)                                                                                                  \
    VBS_LAUNCH(h, s, "k_blur16", (k_blur16<U8, SB_, SH_>), grid16, dim3(64 * (B16_NSW + 1)), 0, s, gray, gstride_n,              \
               (int)gstride_row, h->blur16_h, h->blur16_v, h->area_bits, area_u8, h->fstat, h->H, h->W, h->WW, tps, k3, k8,     \
               h->bp.hi - h->bp.thresh, xcd, gx16, nseg, VBS_KNOB("VBS_BLUR16_DROP"))
#define B16_GO2(U8, SB_) do { if (h->W & 7) B16_GO(U8, SB_, true); else B16_GO(U8, SB_, false); } while (0)
        if (h->bp.small) { if (area_u8) B16_GO2(true, true); else B16_GO2(false, true); }
        else { if (area_u8) B16_GO2(true, false); else B16_GO2(false, false); }
#undef B16_GO2
#undef B16_GO
        return;
    }
    const int gx = (h->P + 127) / 128, tilesY = (h->H + 31) / 32;
    int nseg = std::min(tilesY, std::max(1, (1024 + gx * nb - 1) / (gx * nb)));     // few frames: split columns
    const int tps = (tilesY + nseg - 1) / nseg;
    nseg = (tilesY + tps - 1) / tps;
    dim3 grid(gx, nseg, nb);
#define BLUR_GO(NK, SA0, NKA, U8)                                                                            \
    VBS_LAUNCH(h, s, "k_blur_mfma", (k_blur_mfma<NK, SA0, NKA, U8>), grid, dim3(256), 0, s, gray, gstride_n, \
               gstride_row, h->blur_frags, h->area_bits, area_u8, h->fstat, h->H, h->W, h->WW, tps, k3, k8,  \
               h->bp.hi - h->bp.thresh, VBS_KNOB("VBS_BLUR_DBG"))
    if (!h->bp.small) { if (area_u8) BLUR_GO(5, 1, 3, true); else BLUR_GO(5, 1, 3, false); }
    else { if (area_u8) BLUR_GO(3, 0, 3, true); else BLUR_GO(3, 0, 3, false); }
#undef BLUR_GO
}
