// a3-a5: BGR->gray, two uint8 GaussianBlurs (OpenCV fixed-point model), DoG + 15 (mod 256), inRange.
// Reference: marker_detection.py:114-129.  Integer arithmetic throughout, so results are
// independent of summation order and bit-exact against oracle/stages.py:gaussian_blur_u8.
//
//   out(y,x) = ( sum_i ky[i] * ( sum_j kx[j] * p(y+i-c, x+j-c) ) + 2^15 ) >> 16,  taps in 1/256
//
// Horizontal pass (k_blur_h): one wave per 4 rows x 256 px.  Rows are staged in LDS with the
// reflect-101 border already applied; each lane produces 4 px of each of 4 rows for both kernels
// with v_dot4_u32_u8 against phase-shifted tap words held in SGPRs.  The 16-bit row sums are split
// into hi / lo byte planes packed four ROWS to a dword, so the vertical pass can use the same dot4
// trick down the columns (k_blur_v), finishing with a wave ballot that emits 64 mask bits per row.
#include "common.h"

__device__ __forceinline__ int reflect101(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return min(max(i, 0), n - 1);
}

__global__ void k_gray(const u8* __restrict__ frames, int channels, int64_t stride_n,
                       int64_t stride_row, u8* __restrict__ gray, int H, int W, int P) {
    int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    int y = blockIdx.y, n = blockIdx.z;
    if (x4 >= P) return;
    const u8* src = frames + (int64_t)n * stride_n + (int64_t)y * stride_row;
    u32 out = 0;
    for (int k = 0; k < 4; ++k) {
        int x = x4 + k;
        u32 v = 0;
        if (x < W) {
            if (channels == 1) {
                v = src[x];
            } else {   // cv2 8-bit BGR2GRAY: (1868 B + 9617 G + 4899 R + 2^13) >> 14
                const u8* p = src + (int64_t)x * channels;
                v = (1868u * p[0] + 9617u * p[1] + 4899u * p[2] + 8192u) >> 14;
            }
        }
        out |= v << (8 * k);
    }
    *reinterpret_cast<u32*>(gray + ((int64_t)n * H + y) * P + x4) = out;
}

template <int NWA, int NWB, int C4A, int C4B>
__global__ __launch_bounds__(64) void k_blur_h(const u8* __restrict__ gray, int64_t gstride_n,
                                               int64_t gstride_row, u32* __restrict__ planes,
                                               int H, int W, int P, int QE, BlurTaps taps) {
    constexpr int ROWB = 256 + 2 * C4B;              // bytes staged per row
    __shared__ u32 rowbuf[4][ROWB / 4 + 1];
    const int lane = threadIdx.x;
    const int tile_x0 = blockIdx.x * 256;
    const int qe = blockIdx.y, n = blockIdx.z;
    const u8* g = gray + (int64_t)n * gstride_n;
    // interior tiles of 4-byte aligned rows are staged with dword loads; tiles that touch the left /
    // right border (reflect-101) or unaligned inputs (crops) go byte by byte
    const bool fast = (tile_x0 - C4B >= 0) && (tile_x0 - C4B + ROWB <= W) && ((gstride_row & 3) == 0) &&
                      ((gstride_n & 3) == 0) && ((reinterpret_cast<uintptr_t>(gray) & 3) == 0);
    for (int r = 0; r < 4; ++r) {
        int e = 4 * qe - C4B + r;
        int sy = reflect101(e, H);
        const u8* row = g + (int64_t)sy * gstride_row;
        if (fast) {
            const u32* row32 = reinterpret_cast<const u32*>(row + tile_x0 - C4B);
            for (int i = lane; i < ROWB / 4; i += 64) rowbuf[r][i] = row32[i];
        } else {
            u8* dst = reinterpret_cast<u8*>(&rowbuf[r][0]);
            for (int i = lane; i < ROWB; i += 64) dst[i] = row[reflect101(tile_x0 - C4B + i, W)];
        }
    }
    __syncthreads();
    // tap word outermost: its 4 phase variants are fetched once (SGPRs) and serve all 4 rows
    u32 oa[4][4], ob[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) oa[r][s] = ob[r][s] = 0;
#pragma unroll
    for (int q = 0; q < NWB; ++q) {
        u32 pw[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[r] = rowbuf[r][lane + q];
        constexpr int QA0 = (C4B - C4A) / 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const u32 tb = taps.b[s][q];
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[r][s] = __builtin_amdgcn_udot4(pw[r], tb, ob[r][s], false);
            if (q >= QA0 && q < QA0 + NWA) {
                const u32 ta = taps.a[s][q - QA0];
#pragma unroll
                for (int r = 0; r < 4; ++r) oa[r][s] = __builtin_amdgcn_udot4(pw[r], ta, oa[r][s], false);
            }
        }
    }
    u32 hiA[4] = {0, 0, 0, 0}, loA[4] = {0, 0, 0, 0}, hiB[4] = {0, 0, 0, 0}, loB[4] = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            hiA[s] |= (oa[r][s] >> 8) << (8 * r);
            loA[s] |= (oa[r][s] & 255u) << (8 * r);
            hiB[s] |= (ob[r][s] >> 8) << (8 * r);
            loB[s] |= (ob[r][s] & 255u) << (8 * r);
        }
    const int x0 = tile_x0 + 4 * lane;
    if (x0 < P) {
        int64_t plane_sz = (int64_t)QE * P;
        u32* base = planes + (int64_t)n * 4 * plane_sz + (int64_t)qe * P + x0;
        *reinterpret_cast<uint4*>(base + 0 * plane_sz) = make_uint4(hiA[0], hiA[1], hiA[2], hiA[3]);
        *reinterpret_cast<uint4*>(base + 1 * plane_sz) = make_uint4(loA[0], loA[1], loA[2], loA[3]);
        *reinterpret_cast<uint4*>(base + 2 * plane_sz) = make_uint4(hiB[0], hiB[1], hiB[2], hiB[3]);
        *reinterpret_cast<uint4*>(base + 3 * plane_sz) = make_uint4(loB[0], loB[1], loB[2], loB[3]);
    }
}

// Vertical pass: lane = (column x, TQ consecutive row-quads).  A plane word loaded for the window of one
// output quad is also tap word q-1 of the next one, so two quads per lane nearly halve the loads.
template <int NWA, int NWB, int C4A, int C4B>
__global__ __launch_bounds__(256) void k_blur_v(const u32* __restrict__ planes, u64* __restrict__ bits,
                                                u8* __restrict__ area_u8, u32* __restrict__ fstat,
                                                int H, int W, int P, int WW, int QE, int thresh,
                                                int hi, BlurTaps taps) {
    constexpr int TQ = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane;
    const int yq0 = (blockIdx.y * 4 + wave) * TQ;
    const int n = blockIdx.z;
    if (4 * yq0 >= H) return;                      // wave-uniform
    const int64_t plane_sz = (int64_t)QE * P;
    const u32* pl = planes + (int64_t)n * 4 * plane_sz + x;
    u32 va_hi[TQ][4], va_lo[TQ][4], vb_hi[TQ][4], vb_lo[TQ][4];
#pragma unroll
    for (int o = 0; o < TQ; ++o)
#pragma unroll
        for (int s = 0; s < 4; ++s) va_hi[o][s] = va_lo[o][s] = vb_hi[o][s] = vb_lo[o][s] = 0;
#pragma unroll
    for (int q = 0; q < NWA + TQ - 1; ++q) {
        int qq = min(yq0 + (C4B - C4A) / 4 + q, QE - 1);
        int64_t off = (int64_t)qq * P;
        u32 ph = pl[0 * plane_sz + off], plo = pl[1 * plane_sz + off];
#pragma unroll
        for (int o = 0; o < TQ; ++o) {
            const int t = q - o;
            if (t >= 0 && t < NWA) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    va_hi[o][s] = __builtin_amdgcn_udot4(ph, taps.a[s][t], va_hi[o][s], false);
                    va_lo[o][s] = __builtin_amdgcn_udot4(plo, taps.a[s][t], va_lo[o][s], false);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NWB + TQ - 1; ++q) {
        int qq = min(yq0 + q, QE - 1);
        int64_t off = (int64_t)qq * P;
        u32 ph = pl[2 * plane_sz + off], plo = pl[3 * plane_sz + off];
#pragma unroll
        for (int o = 0; o < TQ; ++o) {
            const int t = q - o;
            if (t >= 0 && t < NWB) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    vb_hi[o][s] = __builtin_amdgcn_udot4(ph, taps.b[s][t], vb_hi[o][s], false);
                    vb_lo[o][s] = __builtin_amdgcn_udot4(plo, taps.b[s][t], vb_lo[o][s], false);
                }
            }
        }
    }
    u32 total = 0;
#pragma unroll
    for (int o = 0; o < TQ; ++o) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            int y = 4 * (yq0 + o) + s;
            u32 b3 = (((va_hi[o][s] << 8) + va_lo[o][s]) + 32768u) >> 16;    // im_blur_3 (small kernel)
            u32 b8 = (((vb_hi[o][s] << 8) + vb_lo[o][s]) + 32768u) >> 16;    // im_blur_8 (large kernel)
            u32 dog = (b8 - b3 + 15u) & 255u;                                // uint8 arithmetic wraps (:128)
            bool pred = (dog >= (u32)thresh) && (dog <= (u32)hi) && (x < W) && (y < H);
            u64 word = __ballot(pred);
            if (y < H) {
                if (lane == 0) bits[((int64_t)n * H + y) * WW + blockIdx.x] = word;
                if (area_u8 && x < W) area_u8[((int64_t)n * H + y) * W + x] = pred ? 255 : 0;
                total += __popcll(word);
            }
        }
    }
    if (lane == 0 && total) atomicAdd(&fstat[n * 8 + 0], total);
}

void launch_gray(vbs_handle* h, const u8* frames, int nb, int channels, int64_t stride_n,
                 int64_t stride_row, hipStream_t s) {
    dim3 grid((h->P / 4 + 255) / 256, h->H, nb);
    VBS_LAUNCH(h, s, "k_gray", k_gray, grid, dim3(256), 0, s, frames, channels, stride_n, stride_row, h->gray,
                       h->H, h->W, h->P);
}

void launch_blur(vbs_handle* h, const u8* gray, int64_t gstride_n, int64_t gstride_row, int nb,
                 u8* area_u8, hipStream_t s) {
    dim3 gh((h->P + 255) / 256, h->QE, nb);
    dim3 gv(h->WW, (h->H + 31) / 32, nb);
    if (!h->bp.small) {
        VBS_LAUNCH(h, s, "k_blur_h", (k_blur_h<11, 27, 20, 52>), gh, dim3(64), 0, s, gray, gstride_n, gstride_row,
                           h->planes, h->H, h->W, h->P, h->QE, h->taps);
        VBS_LAUNCH(h, s, "k_blur_v", (k_blur_v<11, 27, 20, 52>), gv, dim3(256), 0, s, h->planes, h->area_bits,
                           area_u8, h->fstat, h->H, h->W, h->P, h->WW, h->QE, h->bp.thresh, h->bp.hi,
                           h->taps);
    } else {
        VBS_LAUNCH(h, s, "k_blur_h", (k_blur_h<7, 11, 12, 20>), gh, dim3(64), 0, s, gray, gstride_n, gstride_row,
                           h->planes, h->H, h->W, h->P, h->QE, h->taps);
        VBS_LAUNCH(h, s, "k_blur_v", (k_blur_v<7, 11, 12, 20>), gv, dim3(256), 0, s, h->planes, h->area_bits,
                           area_u8, h->fstat, h->H, h->W, h->P, h->WW, h->QE, h->bp.thresh, h->bp.hi,
                           h->taps);
    }
}
