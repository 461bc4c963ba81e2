// a9-a13: _marker_center (marker_detection.py:166-249) on bit-packed masks.
//
//   k_threshold : uint8 mask / area_mask -> 1 bit per pixel (the HBM-streaming stage: 16 px per lane
//                 per load, SWAR non-zero test, v_dot4 bit gather, 4 lanes -> one 64-bit word)
//   k_morph     : band = mask & ~erode_ns(mask)   (maximum/minimum_filter :171-174, window -ns/2..ns/2-1,
//                 pixels outside the image ignored == scipy 'reflect' for a min/max filter)
//                 open = dilate5(erode5(area))    (cv2.morphologyEx MORPH_OPEN 5x5 :195)
//   k_label     : one workgroup per (frame, mask): runs of 1-bits are the union-find nodes, parents
//                 live in LDS, unions between adjacent rows with LDS atomicMin; the root of a
//                 component is its first run in raster order, so component ids come out in
//                 ndimage.label order (:176) and reversed they are cv2.findContours' order (:196).
//                 band mask (4-connectivity): count / sum x / sum y per component (center_of_mass :181)
//                 open mask (8-connectivity): integer moments up to order 4 of the CHAIN_APPROX_SIMPLE
//                 contour vertices, classified per border pixel from its 8-neighbourhood by a LUT.
//   k_finalize  : per frame: centroids, fitEllipse (:208) from the vertex moments via two normal-
//                 equation solves in float64, then the sequential contour <-> centre matching (:203-243).
#include <algorithm>
#include <cstdlib>

#include "ccl_common.h"

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 nz4(u32 x) {       // 4 bytes -> 4 bits (byte != 0)
    u32 t = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) >> 7;
    return __builtin_amdgcn_udot4(t & 0x01010101u, 0x08040201u, 0u, false);
}

__device__ __forceinline__ u32 nz16(uint4 v) {
    return nz4(v.x) | (nz4(v.y) << 4) | (nz4(v.z) << 8) | (nz4(v.w) << 12);
}

__global__ __launch_bounds__(256) void k_threshold(const u8* __restrict__ mask,
                                                   const u8* __restrict__ area,
                                                   u64* __restrict__ mbits, u64* __restrict__ abits,
                                                   int nb, int H, int W, int P, int WW, int vec_ok) {
    // frames are folded into one index space so that every wave is full; no early return because
    // the 4-lane word assembly below shuffles across lanes.
    const int per_row = P / 16;
    int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = gid < (int64_t)nb * H * per_row;
    int n = 0, y = 0, t = 0;
    if (live) {
        n = (int)(gid / ((int64_t)H * per_row));
        int64_t r = gid - (int64_t)n * H * per_row;
        y = (int)(r / per_row);
        t = (int)(r - (int64_t)y * per_row);
    }
    const int x0 = t * 16;
    u32 bm = 0, ba = 0;
    if (live && x0 < W) {
        int64_t off = ((int64_t)n * H + y) * W + x0;
        if (vec_ok && x0 + 16 <= W) {
            bm = nz16(*reinterpret_cast<const uint4*>(mask + off));
            ba = nz16(*reinterpret_cast<const uint4*>(area + off));
        } else {
            for (int k = 0; k < 16 && x0 + k < W; ++k) {
                bm |= (u32)(mask[off + k] != 0) << k;
                ba |= (u32)(area[off + k] != 0) << k;
            }
        }
    }
    u64 wm = (u64)bm | ((u64)__shfl_down(bm, 1) << 16) | ((u64)__shfl_down(bm, 2) << 32) |
             ((u64)__shfl_down(bm, 3) << 48);
    u64 wa = (u64)ba | ((u64)__shfl_down(ba, 1) << 16) | ((u64)__shfl_down(ba, 2) << 32) |
             ((u64)__shfl_down(ba, 3) << 48);
    if (live && (t & 3) == 0) {
        int64_t o = ((int64_t)n * H + y) * WW + (t >> 2);
        mbits[o] = wm;
        abits[o] = wa;
    }
}

void launch_threshold(vbs_handle* h, const u8* mask, const u8* area, int nb, hipStream_t s) {
    int64_t total = (int64_t)nb * h->H * (h->P / 16);
    int vec_ok = (h->W % 16 == 0) && (((uintptr_t)mask | (uintptr_t)area) % 16 == 0);
    VBS_LAUNCH(h, s, "k_threshold", k_threshold, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, mask, area,
                       h->mask_bits, h->area_bits, nb, h->H, h->W, h->P, h->WW, vec_ok);
}

// ------------------------------------------------------------------------------------------------
// horizontal window AND / OR of one row word over dx in [lo, hi] (lo <= 0 <= hi, hi - lo < 64) given its left / right
// neighbour words.  The 128 bits from position lo on are combined with themselves shifted by 1, 2, 4, ... (AND and OR
// are idempotent, so the last shift may overlap): log2(window) steps instead of one per offset.
template <bool ERODE>
__device__ __forceinline__ u64 hmorph(u64 wl, u64 wc, u64 wr, int lo, int hi) {
    const int pre = -lo, n = hi - lo + 1;               // bit p of (ulo, uhi) = pixel p - pre
    u64 ulo = pre ? ((wl >> (64 - pre)) | (wc << pre)) : wc;
    u64 uhi = pre ? ((wc >> (64 - pre)) | (wr << pre)) : wr;
    int have = 1;
    while (have < n) {
        const int s = min(have, n - have);
        const u64 slo = (ulo >> s) | (uhi << (64 - s)), shi = uhi >> s;      // (positions past the 128 bits are never used)
        ulo = ERODE ? (ulo & slo) : (ulo | slo);
        uhi = ERODE ? (uhi & shi) : (uhi | shi);
        have += s;
    }
    return ulo;                                          // bit i = AND / OR of pixels i + lo .. i + hi
}

// band = mask & ~erode_ns(mask) and open = dilate5(erode5(area)), separably, as a stream down the image:
// a wave holds G = 64 / WW strips of rows side by side (lane = strip * WW + word column) and takes one image row per
// step; the horizontal passes get their neighbour words by DPP lane shifts, the vertical passes are delay lines in
// registers (the ns-row AND by doubling: 2, 4, 8, ns rows).  No LDS, no barrier; a strip re-reads only the ns - 1 rows
// above / below it.  (The first version staged 32-row tiles in LDS behind four barriers: 0.31 us per 1280x1024 frame.)
// Outside the image erosion sees 1s (pixels ignored), dilation sees 0s - scipy 'reflect' / cv2's default border.
// one wave: the G strips `wv` G .. of frame n
template <int NS14>                                      // ns = 14 (large frames) or 8 (small)
__device__ __forceinline__ void morph_wave(const u64* __restrict__ mbits, const u64* __restrict__ abits,
                                           u64* __restrict__ band, u64* __restrict__ opn, int H, int W, int WW, int G, int strips,
                                           int rows_per_strip, int n, int wv) {
    const int lane = threadIdx.x & 63;
    const int sidx = wv * G + lane / WW, j = lane % WW;
    const bool act = lane < G * WW && sidx < strips;
    const int ra = min(sidx * rows_per_strip, H), rb = min(ra + rows_per_strip, H);
    const int64_t fo = (int64_t)n * H * WW;
    const u64* M = mbits + fo;
    const u64* A = abits + fo;
    const u64 vm = valid_mask(j, W);
    const bool hasl = j > 0, hasr = j + 1 < WW;
    constexpr int LO = -(NS14 / 2), HI = NS14 / 2 - 1;    // window rows / columns y + LO .. y + HI
    // every lane runs the same number of steps (DPP moves need all lanes): the longest strip of the wave
    // input rows ra + LO .. : the band row yb needs rows up to yb + HI, the opened row yo rows up to yo + 4
    const int nsteps = rows_per_strip + (NS14 - 1 > 4 - LO ? NS14 - 1 : 4 - LO);
    // delay lines (index 0 = newest)
    u64 h1 = ~0ull, a2[2] = {~0ull, ~0ull}, a4[4] = {~0ull, ~0ull, ~0ull, ~0ull}, a8[6] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
    u64 e5[4] = {~0ull, ~0ull, ~0ull, ~0ull}, d5[4] = {0, 0, 0, 0};
    // rows are loaded three steps ahead of their use (nothing else hides the load latency: there is no other work between
    // two steps of a wave)
    auto ld = [&](const u64* src, int row, bool ok) { return (ok && row >= 0 && row < H) ? src[(int64_t)row * WW + j] : 0ull; };
    auto ldc = [&](int row) { return (act && row >= ra && row < rb) ? M[(int64_t)row * WW + j] : 0ull; };
    const int tb = ra + LO;
    u64 mq0 = ld(M, tb, act), mq1 = ld(M, tb + 1, act), mq2 = ld(M, tb + 2, act);
    u64 aq0 = ld(A, tb, act), aq1 = ld(A, tb + 1, act), aq2 = ld(A, tb + 2, act);
    u64 cq0 = ldc(tb - HI), cq1 = ldc(tb + 1 - HI), cq2 = ldc(tb + 2 - HI);
    for (int k = 0; k < nsteps; ++k) {
        const int t = tb + k;                            // input row of this step
        const bool tin = act && t >= 0 && t < H;
        const u64 mw = mq0, aw = aq0, mc = cq0;
        mq0 = mq1; mq1 = mq2; mq2 = ld(M, t + 3, act);
        aq0 = aq1; aq1 = aq2; aq2 = ld(A, t + 3, act);
        cq0 = cq1; cq1 = cq2; cq2 = ldc(t + 3 - HI);
        const int yb = t - HI;                           // band row completed by this step (window yb + LO .. yb + HI = t)
        const bool bout = act && yb >= ra && yb < rb;
        // ---- horizontal erosions (neighbour words by lane shift; rows outside the image are all ones) ----
        u64 hm, ha;
        {
            const u64 wc = tin ? (mw | ~vm) : ~0ull, wl_ = dpp_shr1(wc), wr_ = dpp_shl1(wc);
            hm = hmorph<true>(hasl ? wl_ : ~0ull, wc, hasr ? wr_ : ~0ull, LO, HI);
            const u64 ac = tin ? (aw | ~vm) : ~0ull, al_ = dpp_shr1(ac), ar_ = dpp_shl1(ac);
            ha = hmorph<true>(hasl ? al_ : ~0ull, ac, hasr ? ar_ : ~0ull, -2, 2);
        }
        // ---- vertical erosion over NS14 rows by doubling: a2[t-1], a4[t-3], a8[t-7], then rows t-NS14+1 .. t ----
        u64 e14;
        {
            const u64 n2 = h1 & hm;                      // rows t-1, t
            const u64 n4 = a2[1] & n2;                   // rows t-3 .. t      (a2[1] = rows t-3, t-2)
            if (NS14 == 14) {
                const u64 n8 = a4[3] & n4;               // rows t-7 .. t      (a4[3] = rows t-7 .. t-4)
                e14 = a8[5] & n8;                        // rows t-13 .. t     (a8[5] = rows t-13 .. t-6)
                a8[5] = a8[4]; a8[4] = a8[3]; a8[3] = a8[2]; a8[2] = a8[1]; a8[1] = a8[0]; a8[0] = n8;
            } else {
                e14 = a4[3] & n4;                        // ns = 8: rows t-7 .. t
            }
            a4[3] = a4[2]; a4[2] = a4[1]; a4[1] = a4[0]; a4[0] = n4;
            a2[1] = a2[0]; a2[0] = n2;
            h1 = hm;
        }
        if (bout) band[fo + (int64_t)yb * WW + j] = mc & ~e14 & vm;
        // ---- open: vertical erosion over 5 rows -> row t-2 (0 outside the image), horizontal dilation, vertical
        //      dilation over 5 rows -> row t-4 ----
        {
            const int ye = t - 2;
            u64 ve = ha & e5[0] & e5[1] & e5[2] & e5[3];
            e5[3] = e5[2]; e5[2] = e5[1]; e5[1] = e5[0]; e5[0] = ha;
            ve = (act && ye >= 0 && ye < H) ? (ve & vm) : 0ull;
            const u64 vl_ = dpp_shr1(ve), vr_ = dpp_shl1(ve);
            const u64 hd = hmorph<false>(hasl ? vl_ : 0ull, ve, hasr ? vr_ : 0ull, -2, 2);
            const u64 o = hd | d5[0] | d5[1] | d5[2] | d5[3];
            d5[3] = d5[2]; d5[2] = d5[1]; d5[1] = d5[0]; d5[0] = hd;
            const int yo = t - 4;
            if (act && yo >= ra && yo < rb) opn[fo + (int64_t)yo * WW + j] = o & vm;
        }
    }
}

template <int NS14>
__global__ __launch_bounds__(256) void k_morph(const u64* __restrict__ mbits, const u64* __restrict__ abits,
                                               u64* __restrict__ band, u64* __restrict__ opn, const u32* __restrict__ only,
                                               const u32* __restrict__ nslow,
                                               int nb, int H, int W, int WW, int G, int strips, int rows_per_strip,
                                               int waves_per_frame) {
    if (nslow && *nslow == 0) return;                    // the fused kernel handed no frame on
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = gw / waves_per_frame;
    if (n >= nb || (only && !only[n])) return;           // wave-uniform (`only`: just the frames the fused path handed on)
    morph_wave<NS14>(mbits, abits, band, opn, H, W, WW, G, strips, rows_per_strip, n, gw - n * waves_per_frame);
}

static void launch_morph(vbs_handle* h, int nb, const u32* only, hipStream_t s) {
    const int G = 64 / h->WW;                            // strips per wave (WW <= 64)
    // strips per frame: enough waves to fill the chip several times over, but strips much longer than the ns - 1 rows
    // each re-reads
    // one round of resident waves when the batch is large (94 / 78 VGPRs: 5 / 6 waves per SIMD on 1024 SIMDs; with 6144
    // waves the large branch ran a full round and then a 20 % one), else as many as the strip length allows
    const int resident = (h->bp.ns == 14 ? 5 : 6) * 1024;
    int wpf = resident / std::max(nb, 1);
    if (wpf < 4) wpf = (2 * resident + nb - 1) / nb;     // small strips would dominate: take two rounds instead
    if (VBS_KNOB("VBS_MORPH_WPF")) wpf = VBS_KNOB("VBS_MORPH_WPF");
    wpf = std::max(1, std::min(wpf, h->H / (2 * h->bp.ns) / G));      // strips of at least 2 ns rows
    const int strips = wpf * G, rps = (h->H + strips - 1) / strips;
    const int waves = nb * wpf;
    dim3 grid((waves + 3) / 4);
    if (h->bp.ns == 14)
        VBS_LAUNCH(h, s, "k_morph", k_morph<14>, grid, dim3(256), 0, s, h->mask_bits, h->area_bits, h->band_bits, h->open_bits,
                   only, only ? h->slow_total : nullptr, nb, h->H, h->W, h->WW, G, strips, rps, wpf);
    else
        VBS_LAUNCH(h, s, "k_morph", k_morph<8>, grid, dim3(256), 0, s, h->mask_bits, h->area_bits, h->band_bits, h->open_bits,
                   only, only ? h->slow_total : nullptr, nb, h->H, h->W, h->WW, G, strips, rps, wpf);
}

// ------------------------------------------------------------------------------------------------
// CHAIN_APPROX_SIMPLE vertex multiplicity of a border pixel from its 8-neighbourhood (bit d = neighbour
// in chain direction d is foreground).  The outer border visits the pixel once per maximal arc of
// background neighbours that contains a 4-neighbour (an arc made of one diagonal pixel is stepped
// over diagonally); arriving from the foreground neighbour that precedes the arc and leaving to the
// one that follows it, the point is kept iff the two step directions differ.
void make_contour_lut(u8 out[256]) {
    for (int p = 0; p < 256; ++p) {
        int cnt = 0;
        if (p == 0) {
            cnt = 1;                                     // isolated pixel: written once
        } else if (p != 255) {
            for (int a = 0; a < 8; ++a) {
                // arc starts at direction a: a is background, a-1 is foreground
                if (((p >> a) & 1) || !((p >> ((a + 7) & 7)) & 1)) continue;
                int b = a;
                bool has4 = false;
                while (!((p >> (b & 7)) & 1)) {
                    if (((b & 7) & 1) == 0) has4 = true;
                    ++b;
                }
                if (!has4) continue;
                int q = (a + 7) & 7;                      // neighbour before the arc
                int r = b & 7;                            // neighbour after the arc
                int dir_in = (q + 4) & 7;                 // step q -> p
                int dir_out = r;                          // step p -> r
                if (dir_in != dir_out) ++cnt;
            }
        }
        out[p] = (u8)cnt;
    }
}

// ------------------------------------------------------------------------------------------------
// find with path halving.  Parents only ever move to a smaller-indexed member of the same set (hooking is an
// atomicMin on a root, halving stores an ancestor), so the racy 32-bit stores are benign: a lost hook is
// re-established by uf_union's retry with the value atomicMin returned.
__device__ __forceinline__ u32 uf_find(volatile u32* parent, u32 x) {
    for (;;) {
        u32 p = parent[x];
        if (p == x) return x;
        u32 gp = parent[p];
        if (gp == p) return p;
        parent[x] = gp;
        x = gp;
    }
}

// read-only walk to the root (flatten phase: there, every store must be a final root)
__device__ __forceinline__ u32 uf_root(volatile u32* parent, u32 x) {
    u32 p;
    while ((p = parent[x]) != x) x = p;
    return x;
}

__device__ __forceinline__ void uf_union(u32* parent, u32 a, u32 b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { u32 t = a; a = b; b = t; }
        u32 old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;
    }
}

// node (run) index of the run that contains bit k of word j in `row`
__device__ __forceinline__ u32 node_of(const u64* __restrict__ row, const u32* __restrict__ wb, int j,
                                       int k) {
    int jj = j, kk = k, start;
    for (;;) {
        u64 w = row[jj];
        u64 below = (kk == 63) ? ~0ull : ((1ull << (kk + 1)) - 1ull);
        u64 z = ~w & below;
        if (z) { start = 64 - __clzll(z); break; }
        if (jj == 0 || !(row[jj - 1] >> 63)) { start = 0; break; }
        --jj;
        kk = 63;
    }
    u64 w = row[jj];
    u64 prev = (jj > 0) ? (row[jj - 1] >> 63) : 0ull;
    u64 starts = w & ~((w << 1) | prev);
    u64 lowmask = start ? ((1ull << start) - 1ull) : 0ull;
    return wb[jj] + (u32)__popcll(starts & lowmask);
}

__device__ __forceinline__ u32 block_exclusive_scan(u32 v, u32* tmp, u32* total) {
    // 1024 threads = 16 waves; tmp has >= 17 entries
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    if (wave == 0) {                                   // scan of the (<= 16) wave totals by one wave
        const int nw = blockDim.x >> 6;
        u32 t = lane < nw ? tmp[lane] : 0u, ti = t;
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            u32 o = __shfl_up(ti, d);
            if (lane >= d) ti += o;
        }
        if (lane < nw) tmp[lane] = ti - t;
        if (lane == nw - 1) tmp[16] = ti;
    }
    __syncthreads();
    u32 ex = inc - v + tmp[wave];
    *total = tmp[16];
    __syncthreads();
    return ex;
}

// moment index of x^a y^b, a+b <= 4:  (0,0) (1,0) (0,1) (2,0) (1,1) (0,2) (3,0) (2,1) (1,2) (0,3) (4,0) (3,1) (2,2) (1,3) (0,4)
#define NONE32 0xFFFFFFFFu

__device__ __forceinline__ u32 wave_scan_incl(u32 x) {   // inclusive prefix sum over the 64 lanes
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return x;
}
__device__ __forceinline__ u32 wave_last(u32 x) { return (u32)__builtin_amdgcn_readlane((int)x, 63); }


// Per-lane view of G = 64 / WW consecutive image rows ("a step"): lane = g * WW + j holds word j (64 px) of row
// y0 + g, the run-start bits in it, the node index of the first run that starts in it, and the node of the run that
// enters it from the left (NONE32 if none).  Lanes are in raster order, so one wave-wide prefix sum numbers the runs.
struct RowState {
    u64 w, st;
    u32 base, cin;
};

// Build the state of a step from its words (lanes outside the step hold 0).  `j` = word column of the lane,
// `rowbase` = node index of the step's first run (wave-uniform), advanced past the step.
__device__ __forceinline__ RowState make_row_state(u64 w, int j, u32& rowbase) {
    RowState s;
    s.w = w;
    // (cross-lane moves run with all lanes enabled and are selected afterwards: a DPP read from an exec-masked
    //  lane returns 0, so they must never sit inside a conditional)
    const u64 msb_all = (u64)(dpp_shr1((u32)(w >> 32)) >> 31);
    const u64 msb = j ? msb_all : 0ull;                   // bit 63 of the word to the left in the row
    s.st = w & ~((w << 1) | msb);
    const u32 c = __popcll(s.st);
    const u32 inc = wave_scan_incl(c);
    s.base = rowbase + inc - c;
    rowbase += wave_last(inc);
    // node entering from the left = last run of word j-1, which is that word's last start, or - when word j-1 is
    // all ones inside one long run - the run entering IT (resolved by iterating; one pass unless runs span > 64 px)
    const bool cont = (w & 1ull) && msb;
    const bool full = cont && (w == ~0ull);
    const u32 last_here = c ? s.base + c - 1 : NONE32;
    u32 lastnode = last_here;
    s.cin = NONE32;
    for (int it = 0; it < 64; ++it) {
        u32 upn = dpp_shr1(lastnode);
        u32 ncin = cont ? upn : NONE32;                  // cont implies j > 0
        u32 nlast = c ? last_here : (full ? ncin : NONE32);
        bool changed = (ncin != s.cin) || (nlast != lastnode);
        s.cin = ncin;
        lastnode = nlast;
        if (!__any(changed)) break;
    }
    return s;
}

__device__ __forceinline__ RowState zero_state() { RowState z = {0ull, 0ull, 0u, NONE32}; return z; }

// state of the word to the left / right in the same row (zeros at the row ends)
__device__ __forceinline__ RowState state_left(const RowState& s, int j) {
    RowState r;
    r.w = dpp_shr1(s.w); r.st = dpp_shr1(s.st); r.base = dpp_shr1(s.base); r.cin = dpp_shr1(s.cin);
    return j ? r : zero_state();
}
__device__ __forceinline__ RowState state_right(const RowState& s, int j, int WW) {
    RowState r;
    r.w = dpp_shl1(s.w); r.st = dpp_shl1(s.st); r.base = dpp_shl1(s.base); r.cin = dpp_shl1(s.cin);
    return (j + 1 < WW) ? r : zero_state();
}

// state held by another lane (ds_bpermute; executed by every lane)
__device__ __forceinline__ RowState state_from(const RowState& s, int src) {
    RowState r;
    r.w = __shfl(s.w, src); r.st = __shfl(s.st, src); r.base = __shfl(s.base, src); r.cin = __shfl(s.cin, src);
    return r;
}

// node (run) index of the run containing bit k of this lane's word
__device__ __forceinline__ u32 node_in_row(const RowState& s, int k) {
    const u64 below = k ? ((1ull << k) - 1ull) : 0ull;
    const u64 z = ~s.w & below;
    if (z) {
        const int stpos = 64 - __clzll(z);
        return s.base + (u32)__popcll(s.st & ((1ull << stpos) - 1ull));
    }
    if ((s.w & 1ull) && s.cin != NONE32) return s.cin;
    return s.base;
}

// unions of the runs of this lane's word (`cur`) with the runs of the row above (`prev` = same column, one row up):
// 4-connectivity (m = 0) or 8 (m = 1).  Cross-lane moves first, by every lane; the loops after them may diverge.
__device__ __forceinline__ void link_rows(u32* parent, const RowState& cur, const RowState& prev, int j, int WW, int m) {
    const RowState pl = state_left(prev, j), pr = state_right(prev, j, WW);
    const u64 B = cur.w, A = prev.w;
    u64 adj = A;
    if (m == 1) adj |= (A << 1) | (A >> 1) | (pl.w >> 63) | (pr.w << 63);
    u64 mB = (B & adj) ? B : 0ull;
    while (mB) {                                       // groups of consecutive 1s of B inside this word
        u64 lowbit = mB & (~mB + 1ull);
        u64 t = mB + lowbit;
        u64 g = mB & ~t;
        mB &= t;
        if (!(g & adj)) continue;
        const u32 nb_ = node_in_row(cur, __ffsll((long long)g) - 1);
        u64 rm = g;
        if (m == 1) rm |= (g << 1) | (g >> 1);
        u64 mA = A & rm;
        while (mA) {
            u64 lb = mA & (~mA + 1ull);
            u64 t2 = mA + lb;
            u64 ga = mA & ~t2;
            mA &= t2;
            uf_union(parent, node_in_row(prev, __ffsll((long long)ga) - 1), nb_);
        }
        if (m == 1) {
            if ((g & 1ull) && (pl.w >> 63)) uf_union(parent, node_in_row(pl, 63), nb_);
            if ((g >> 63) && (pr.w & 1ull)) uf_union(parent, node_in_row(pr, 0), nb_);
        }
    }
}

// component id of pixel (x, y) of the opened mask from the general path's tables, 0xFFFF when it is not foreground
__device__ __forceinline__ u32 comp_at(const u64* __restrict__ bits, const u32* __restrict__ wbase,
                                       const u32* __restrict__ node_comp, int H, int W, int WW, int x, int y) {
    if (x < 0 || y < 0 || x >= W || y >= H) return 0xFFFFu;
    const u64* row = bits + (int64_t)y * WW;
    if (!((row[x >> 6] >> (x & 63)) & 1ull)) return 0xFFFFu;
    return node_comp[node_of(row, wbase + (int64_t)y * WW, x >> 6, x & 63)];
}

// 4-deep register ring of step words: PF_INIT issues the loads of steps y0, y0+G, .., PF_NEXT hands out the step at
// y and issues the one at y + 4 G, so a step never waits for a load it has just issued.  Lane (g, j) holds word j of
// row y + g; rows >= ylim read as 0.
#define PF_LOAD(ptr, yy, ylim) ((act && (yy) < (ylim)) ? (inv ? (~(ptr)[(int64_t)(yy) * WW + j] & vmask) : (ptr)[(int64_t)(yy) * WW + j]) : 0ull)
#define PF_INIT(ptr, y0, ylim)                               \
    u64 pf0 = PF_LOAD(ptr, (y0) + g, ylim);                  \
    u64 pf1 = PF_LOAD(ptr, (y0) + G + g, ylim);              \
    u64 pf2 = PF_LOAD(ptr, (y0) + 2 * G + g, ylim);          \
    u64 pf3 = PF_LOAD(ptr, (y0) + 3 * G + g, ylim);
#define PF_NEXT(ptr, y, ylim, out)                           \
    out = pf0; pf0 = pf1; pf1 = pf2; pf2 = pf3;              \
    pf3 = PF_LOAD(ptr, (y) + 4 * G + g, ylim);

// One workgroup (16 waves) per (frame, mask).  Wave w owns the strip of rows [w R, (w+1) R), R = ceil(H / 16), and
// walks it top-down G = 64 / WW rows at a time (3 at W = 1280), lane = g * WW + j holding word j of row y0 + g:
// run starts, node indices (one wave prefix sum, raster order), the run entering from the left and the links to the
// row above come from registers, DPP moves and a few ds_bpermute; the union-find table lives in LDS (path halving,
// atomicMin hooking).  The 15 strip boundaries are linked at the end.  Node indices follow raster order, so the
// root (minimum) of a component is its first run and component ids come out in ndimage.label order.
// mode 0: label the band mask (blockIdx.y = 0, 4-conn) and the opened mask (1, 8-conn).
// mode 1: only for frames whose opened mask has holes (Euler check of mode 0): label its COMPLEMENT (4-conn), find the
//         background components that do not touch the image border (= holes) and fill them in open_bits: outer
//         contours, "inside the contour" and RETR_EXTERNAL's nesting rule are invariant under hole filling, and after
//         it every border is an outer border, which is what the per-pixel vertex table assumes.
// mode 2: relabel the (now hole-free) opened mask of those frames.
// MORPH = 14 | 8 (the few-frames path): the workgroup first makes the frame's band / opened planes itself (k_morph's waves,
// its sixteen in turn) - one launch less on a path where a launch costs as much as a kernel; 0: the planes are there
template <int MORPH>
// (band_bits / open_bits: plain pointers - the MORPH instances WRITE the planes (morph_wave) and read them back later in the
//  kernel; a write through a restrict-qualified pointer-to-const was undefined behaviour)
__global__ __launch_bounds__(1024) void k_label(u64* band_bits,
                                                u64* open_bits,
                                                u32* __restrict__ wbase_all, u32* __restrict__ node_pos_all,
                                                u32* __restrict__ node_comp_all, u32* __restrict__ ncomp_all,
                                                u32* __restrict__ band_first, u64* __restrict__ band_sums,
                                                u32* __restrict__ area_first, i64* __restrict__ area_sums,
                                                u32* __restrict__ fstat, const u8* __restrict__ lut_g,
                                                const u32* __restrict__ slow_flag, const u32* __restrict__ nslow,
                                                unsigned short* __restrict__ probe_all, int nb, int all, int H, int W, int WW,
                                                int maxm, int stop, const u64* __restrict__ mbits, const u64* __restrict__ abits,
                                                int mG, int mstrips, int mrps, int mwpf) {
    __shared__ u32 parent[VBS_RUN_CAP];                // union-find parents; later [m=1] the moment accumulators
    __shared__ u64 bnd_w[16][64];                      // last row of every strip: words,
    __shared__ u32 bnd_base[16][64], bnd_cin[16][64];  //   first-node indices, entering nodes
    __shared__ u32 wtot[16], wfirst[16];               // runs per strip, node index of the strip's first run
    __shared__ u32 acc_cnt[1024];                      // band: pixel counts;  open: first pixel (anchor) per component
    __shared__ u64 acc_sx[1024], acc_sy[1024];
    __shared__ u32 tmp[32];
    __shared__ u8 lut[256];
    __shared__ int euler4;                             // 4 x Euler number of the opened mask (bit quads)
    // Only the frames the fast path (k_ccl.hip) handed on (slow_flag; all of them when `all`): a workgroup takes such a
    // frame through the four stages in turn - label the band mask, label the opened mask, fill its holes (if any),
    // relabel it (if any was filled) - and then writes the probes k_finalize's polygon test reads.  Every exit of a
    // stage's body is workgroup-uniform, so it is a `continue` of the stage loop.
    if (nslow && *nslow == 0) return;                    // the fused kernel handed no frame on
    for (int n = blockIdx.x; n < nb; n += gridDim.x) {
    if (!all && !slow_flag[n]) continue;
    if (MORPH) {                                        // (the stage loop's first fence + barrier publish the planes)
        for (int wv = (int)(threadIdx.x >> 6); wv < mwpf; wv += 16)
            morph_wave<(MORPH ? MORPH : 14)>(mbits, abits, band_bits, open_bits, H, W, WW, mG, mstrips, mrps, n, wv);
    }
    for (int stage = 0; stage < 4; ++stage) {
    __threadfence();                                    // the previous stage's global writes (holes, filled bits, tables)
    __syncthreads();                                    // are visible; its readers of the LDS tables are done
    const int mode = stage < 2 ? 0 : stage - 1;
    const int m = stage == 0 ? 0 : (stage == 2 ? 2 : 1);    // 0 band (4-conn), 1 open (8-conn), 2 background of open
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int NW = H * WW;
    if (mode == 1 && fstat[n * 8 + 4] == 0) continue;    // no holes in this frame
    if (mode == 2 && fstat[n * 8 + 7] == 0) continue;    // nothing was filled
    const bool inv = (m == 2);                          // walk the complement of the opened mask
    const int mslot = (m == 2) ? 0 : m;                 // the background pass borrows the band pass's scratch tables
    const u64* bits = (m == 0 ? band_bits : open_bits) + (int64_t)n * NW;
    u32* wbase = wbase_all + ((int64_t)n * 2 + mslot) * NW;
    u32* node_pos = node_pos_all + ((int64_t)n * 2 + mslot) * VBS_RUN_CAP;
    u32* node_comp = node_comp_all + ((int64_t)n * 2 + mslot) * VBS_RUN_CAP;
    if (stop == 9) continue;
    if (tid < 256) lut[tid] = lut_g[tid];
    if (tid == 0) euler4 = 0;
    for (int i = tid; i < 1024; i += nthr) { acc_cnt[i] = 0; acc_sx[i] = 0; acc_sy[i] = 0; }
    const int R = (H + 15) / 16;
    const int ya = min(H, wave * R), yb = min(H, ya + R);                  // this wave's rows
    const int G = 64 / WW;                                                 // rows per step (WW <= 64)
    const int g = lane / WW, j = lane - g * WW;
    const bool act = g < G;
    const u64 vmask = valid_mask(j, W);
    const int up_src = g ? lane - WW : lane + (G - 1) * WW;                // lane holding the word one row up
    bnd_w[wave][lane] = 0; bnd_base[wave][lane] = 0; bnd_cin[wave][lane] = NONE32;

    // ---- 1: runs per strip -> first node index of every strip ---------------------------------------
    {
        u32 c = 0;
        PF_INIT(bits, ya, yb)
        for (int y = ya; y < yb; y += G) {
            u64 w; PF_NEXT(bits, y, yb, w)
            const u64 msb_all = (u64)(dpp_shr1((u32)(w >> 32)) >> 31);
            const u64 msb = j ? msb_all : 0ull;
            c += __popcll(w & ~((w << 1) | msb));
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
        if (lane == 0) wtot[wave] = c;
    }
    __syncthreads();
    u32 nruns = 0, mybase = 0;
    for (int w = 0; w < 16; ++w) { if (w == wave) mybase = nruns; nruns += wtot[w]; }
    if (lane == 0) wfirst[wave] = mybase;
    if (nruns > VBS_RUN_CAP) {                          // block-uniform
        if (tid == 0) { if (m != 2) ncomp_all[n * 2 + m] = 0; atomicMin((int*)&fstat[n * 8 + 2], VBS_ECAPACITY); }
        continue;
    }

    // ---- 2: label the strip, G rows per step ------------------------------------------------------------
    RowState first_row = zero_state();
    {
        u32 rowbase = mybase;
        RowState last = zero_state();                   // previous step
        PF_INIT(bits, ya, yb)
        for (int y = ya; y < yb; y += G) {
            u64 w; PF_NEXT(bits, y, yb, w)
            const int yr = y + g;                       // this lane's row
            if (!__any(w != 0ull)) {                    // empty step (a third of a marker frame's rows): no run starts,
                if (act && yr < yb) wbase[(int64_t)yr * WW + j] = rowbase;     // nothing to link below either
                last = zero_state();
                continue;
            }
            RowState cur = make_row_state(w, j, rowbase);
            if (act && yr < yb) wbase[(int64_t)yr * WW + j] = cur.base;
            u64 st = cur.st;
            u32 nd = cur.base;
            while (st) {
                int k = __ffsll((long long)st) - 1;
                st &= st - 1;
                parent[nd] = nd;
                node_pos[nd] = (u32)(yr * W + 64 * j + k);
                ++nd;
            }
            // the row above: same step (g > 0) or the last row of the previous step (g == 0)
            const RowState a = state_from(cur, up_src), b = state_from(last, up_src);
            RowState prev = g ? a : b;
            if (!act) prev = zero_state();
            if (y == ya && g == 0) { first_row = cur; prev = zero_state(); }    // strip boundary: linked in pass 3
            __builtin_amdgcn_wave_barrier();            // the new nodes' parents are initialised before any union
            if (!(stop & 32)) link_rows(parent, cur, prev, j, WW, m == 1 ? 1 : 0);
            last = cur;
            if (y + G >= yb) {                          // last step: keep the strip's last row for the wave below
                const int gl = (yb - 1 - y);
                if (act && g == gl) { bnd_w[wave][j] = cur.w; bnd_base[wave][j] = cur.base; bnd_cin[wave][j] = cur.cin; }
            }
        }
    }
    __syncthreads();
    // ---- 3: link every strip's first row with the last row of the strip above --------------------------
    if (wave > 0 && ya < yb && !(stop & 32)) {
        RowState cur = (act && g == 0) ? first_row : zero_state();
        RowState prev = zero_state();
        if (act && g == 0) { prev.w = bnd_w[wave - 1][j]; prev.base = bnd_base[wave - 1][j]; prev.cin = bnd_cin[wave - 1][j]; }
        const u64 msb_all = (u64)(dpp_shr1((u32)(prev.w >> 32)) >> 31);
        const u64 msb = j ? msb_all : 0ull;
        prev.st = prev.w & ~((prev.w << 1) | msb);
        link_rows(parent, cur, prev, j, WW, m == 1 ? 1 : 0);
    }
    __syncthreads();
    if ((stop & 15) == 2) continue;

    // ---- D: flatten -------------------------------------------------------------------------------
    for (u32 i = tid; i < nruns; i += nthr) {
        u32 r = uf_root(parent, i);                    // no halving here: a late halving store could
        if (r != i) parent[i] = r;                     // overwrite another thread's final root
    }
    __syncthreads();

    // ---- E: component ids = rank of the root in raster order ---------------------------------------
    const int chunk2 = (nruns + nthr - 1) / nthr;
    const u32 r0 = tid * chunk2, r1 = min(r0 + (u32)chunk2, nruns);
    u32 nroot = 0;
    for (u32 i = r0; i < r1; ++i) nroot += (parent[i] == i);
    u32 ncomp;
    u32 cbase = block_exclusive_scan(nroot, tmp, &ncomp);
    if ((m != 2 && ncomp > (u32)maxm) || ncomp > 1024u || (m == 1 && ncomp * NMOM * 8u > sizeof(parent) / 2)) {
        if (tid == 0) { if (m != 2) ncomp_all[n * 2 + m] = 0; atomicMin((int*)&fstat[n * 8 + 2], VBS_ECAPACITY); }
        continue;
    }
    u32* first = (m == 0 ? band_first : area_first) + (int64_t)n * maxm;
    for (u32 i = r0; i < r1; ++i) {
        if (parent[i] == i) {
            const u32 pos = node_pos[i];                // first pixel of the component = start of its root run
            if (m != 2) first[cbase] = pos;
            if (m == 1) acc_cnt[cbase] = pos;          // LDS copy of the anchor for phase F
            parent[i] = i | ((cbase + 1) << 16);       // root: id in the high half
            ++cbase;
        }
    }
    __syncthreads();
    // component id of every node: to global (k_finalize's polygon test) and, as uint16, into the lower half of
    // the parent table (read the ids into registers first: the uint16 table overwrites the parents in place)
    constexpr int PER = (VBS_RUN_CAP + 1023) / 1024;
    unsigned short cids[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        u32 i = tid + k * 1024;
        u32 cid = 0;
        if (i < nruns) {
            u32 root = parent[i] & 0xFFFFu;
            cid = (parent[root] >> 16) - 1;
            node_comp[i] = cid;
        }
        cids[k] = (unsigned short)cid;
    }
    if (tid == 0 && m != 2) { ncomp_all[n * 2 + m] = ncomp; fstat[n * 8 + 5 + m] = ncomp; }
    __syncthreads();
    unsigned short* cid16 = reinterpret_cast<unsigned short*>(parent);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        u32 i = tid + k * 1024;
        if (i < nruns) cid16[i] = cids[k];
    }
    u64* acc = reinterpret_cast<u64*>(parent) + VBS_RUN_CAP / 4;           // upper half: [ncomp][15] moments (m = 1)
    if (m == 1) for (u32 c = tid; c < ncomp * NMOM; c += nthr) acc[c] = 0;
    __syncthreads();
    if ((stop & 15) == 4) continue;

    // ---- F: per-component sums, same walk; sums stay in registers and are flushed with LDS atomics when the
    //         component under the lane changes (a lane sees rows y0 + g, y0 + g + G, ... of one word column) --------
    if (m == 2) {
        // background components that touch the image border are the outside; the others are holes: fill them
        for (int pass = 0; pass < 2; ++pass) {
            u32 rowbase = wfirst[wave];
            PF_INIT(bits, ya, yb)
            for (int y = ya; y < yb; y += G) {
                u64 wv; PF_NEXT(bits, y, yb, wv)
                const int yr = y + g;
                RowState cur = make_row_state(wv, j, rowbase);
                u64 w = wv, fill = 0;
                while (w) {
                    u64 lowbit = w & (~w + 1ull);
                    u64 t = w + lowbit;
                    u64 gg = w & ~t;
                    w &= t;
                    const u32 cid = cid16[node_in_row(cur, __ffsll((long long)gg) - 1)];
                    if (pass == 0) {
                        const int xl = W - 1 - 64 * j;                   // bit of the last image column, if in this word
                        const bool edge = (yr == 0) || (yr == H - 1) || (j == 0 && (gg & 1ull)) ||
                                          (xl >= 0 && xl < 64 && ((gg >> xl) & 1ull));
                        if (edge && acc_cnt[cid] == 0) acc_cnt[cid] = 1;
                    } else if (acc_cnt[cid] == 0) {
                        fill |= gg;
                    }
                }
                if (pass == 1 && fill)
                    open_bits[(int64_t)n * NW + (int64_t)yr * WW + j] = (~wv & vmask) | fill;
            }
            __syncthreads();
        }
        u32 nholes = 0;
        for (u32 c = tid; c < ncomp; c += nthr) nholes += (acc_cnt[c] == 0);
        if (nholes) atomicAdd(&fstat[n * 8 + 7], nholes);
        continue;
    }
    // One work item per RUN (node): its position comes back from node_pos, its extent from the bits; every lane has
    // work (a word walk leaves most lanes idle on a marker frame) and the sums go to the component's LDS accumulators.
    if (m == 0) {
        for (u32 i = tid; i < nruns; i += nthr) {
            const u32 cid = cid16[i], pos = node_pos[i];
            const u32 y = pos / (u32)W, x0 = pos - y * (u32)W;
            const u64* row = bits + (int64_t)y * WW;
            int jw = (int)(x0 >> 6);
            const int b0 = (int)(x0 & 63);
            const u64 t = ~(row[jw] >> b0);
            u32 len = t ? (u32)(__ffsll((long long)t) - 1) : 64u;
            if (b0 + (int)len == 64)                     // reaches the word's last bit: continues while the next words start with 1s
                for (++jw; jw < WW; ++jw) {
                    const u64 nw = ~row[jw];
                    const u32 tz = nw ? (u32)(__ffsll((long long)nw) - 1) : 64u;
                    len += tz;
                    if (tz < 64) break;
                }
            atomicAdd(&acc_cnt[cid], len);
            atomicAdd(&acc_sx[cid], (u64)len * x0 + (u64)len * (len - 1) / 2);
            atomicAdd(&acc_sy[cid], (u64)len * (u64)y);
        }
        __syncthreads();
        u64* bs = band_sums + (int64_t)n * maxm * 4;
        for (u32 c = tid; c < ncomp; c += nthr) {
            bs[c * 4 + 0] = acc_cnt[c];
            bs[c * 4 + 1] = acc_sx[c];
            bs[c * 4 + 2] = acc_sy[c];
        }
    } else {
        // Euler number by bit quads (8-connected foreground): E = (Q1 - Q3 - 2 QD) / 4 over all 2x2 windows of the
        // zero-padded image; window (x, y) = pixels (x..x+1, y..y+1).  A work item counts the windows whose top row is
        // its word's row (and, for row 0, the padding row above it).  holes = components - E.
        int e4 = 0;
        for (int idx = tid; idx < NW; idx += nthr) {
            const int yr = idx / WW, jc = idx - yr * WW;
            const u64 wv = bits[idx];
            const u64 dn = yr + 1 < H ? bits[idx + WW] : 0ull;
            if (!(wv | dn) && !(yr == 0)) {
                if (jc + 1 >= WW) continue;
                if (!((bits[idx + 1] | (yr + 1 < H ? bits[idx + 1 + WW] : 0ull)) & 1ull)) continue;
            }
            const u64 wn_ = jc + 1 < WW ? bits[idx + 1] : 0ull;
            const u64 dn_ = (jc + 1 < WW && yr + 1 < H) ? bits[idx + 1 + WW] : 0ull;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q == 1 && yr != 0) continue;
                const u64 a = q ? 0ull : wv, an = q ? 0ull : wn_, bq = q ? wv : dn, bn = q ? wn_ : dn_;
                if (a | bq | (an & 1ull) | (bn & 1ull)) {
                    u64 a1 = (a >> 1) | (an << 63), b1 = (bq >> 1) | (bn << 63);
                    u64 x2 = (a ^ a1) ^ (bq ^ b1);
                    u64 pairs = (a & a1) | (a & bq) | (a & b1) | (a1 & bq) | (a1 & b1) | (bq & b1);
                    u64 qd = (a & b1 & ~a1 & ~bq) | (a1 & bq & ~a & ~b1);
                    e4 += __popcll(x2 & ~pairs) - __popcll(x2 & pairs) - 2 * __popcll(qd);
                    if (jc == 0) e4 += (int)((a ^ bq) & 1ull);               // window x = -1: only (0,y), (0,y+1)
                }
            }
        }
        for (u32 i = tid; i < nruns; i += nthr) {
            const u32 cid = cid16[i], pos = node_pos[i];
            const int y = (int)(pos / (u32)W), x0 = (int)(pos - (u32)y * (u32)W);
            const u32 fp = acc_cnt[cid];
            const int ay = (int)(fp / (u32)W), ax = (int)(fp - (u32)ay * (u32)W);
            const u64* rowm = bits + (int64_t)y * WW;
            const bool hasu = y > 0, hasd = y + 1 < H;
            i64 s[NMOM];
#pragma unroll
            for (int q = 0; q < NMOM; ++q) s[q] = 0;
            bool more = true;
            for (int jw = x0 >> 6; more && jw < WW; ++jw) {      // word segments of the run
                const u64 w = rowm[jw];
                const int lo = jw == (x0 >> 6) ? (x0 & 63) : 0;
                const u64 t = ~(w >> lo);
                const int len = t ? __ffsll((long long)t) - 1 : 64;
                if (len == 0) break;                             // (the run ended exactly at the previous word's last bit)
                const int hi = lo + len - 1;
                more = hi == 63;
                const u64 gg = (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull);
                const bool contR = more && jw + 1 < WW && (rowm[jw + 1] & 1ull);     // the run goes on in the next word
                const u64 up = hasu ? rowm[jw - WW] : 0ull, dn = hasd ? rowm[jw + WW] : 0ull;
                u64 upL = 0, upR = 0, dnL = 0, dnR = 0;  // edge bits of the neighbouring words, only where a pixel needs them
                if (lo == 0 && jw > 0) { upL = hasu ? rowm[jw - 1 - WW] >> 63 : 0ull; dnL = hasd ? rowm[jw - 1 + WW] >> 63 : 0ull; }
                if (hi == 63 && jw + 1 < WW) { upR = hasu ? rowm[jw + 1 - WW] & 1ull : 0ull; dnR = hasd ? rowm[jw + 1 + WW] & 1ull : 0ull; }
                const u64 E = (gg >> 1) | (contR ? 1ull << 63 : 0ull);
                const u64 Wd = (gg << 1) | (jw > (x0 >> 6) ? 1ull : 0ull);
                const u64 NE = (up >> 1) | (upR << 63), NWd = (up << 1) | upL;
                const u64 SE = (dn >> 1) | (dnR << 63), SW = (dn << 1) | dnL;
                u64 bg = gg & ~(up & dn & E & Wd);
                // pixels inside a straight horizontal edge (patterns 241 / 31: no vertex, lut = 0) are dropped by bit
                // operations, so the long top / bottom rows of a blob do not cost one loop turn per pixel
                bg &= ~(E & Wd & ((~up & ~NE & ~NWd & dn & SE & SW) | (up & NE & NWd & ~dn & ~SE & ~SW)));
                while (bg) {
                    const int k = __ffsll((long long)bg) - 1;
                    bg &= bg - 1;
                    const u32 pat = (u32)((E >> k) & 1ull) | ((u32)((NE >> k) & 1ull) << 1) |
                                    ((u32)((up >> k) & 1ull) << 2) | ((u32)((NWd >> k) & 1ull) << 3) |
                                    ((u32)((Wd >> k) & 1ull) << 4) | ((u32)((SW >> k) & 1ull) << 5) |
                                    ((u32)((dn >> k) & 1ull) << 6) | ((u32)((SE >> k) & 1ull) << 7);
                    const int mult = lut[pat];
                    if (!mult) continue;
                    const int dx = 64 * jw + k - ax, dy = y - ay;
                    if (max(abs(dx), abs(dy)) <= 150) {          // 4 * 150^4 < 2^31: products in 32 bits, sums in 64
                        const int x2 = dx * dx, y2 = dy * dy, mx = mult * dx, my = mult * dy;
                        s[0] += mult;
                        s[1] += mx;                 s[2] += my;
                        s[3] += mx * dx;            s[4] += mx * dy;            s[5] += my * dy;
                        s[6] += mx * x2;            s[7] += my * x2;            s[8] += mx * y2;
                        s[9] += my * y2;
                        s[10] += mult * x2 * x2;    s[11] += mx * x2 * dy;      s[12] += mult * x2 * y2;
                        s[13] += mx * dy * y2;      s[14] += mult * y2 * y2;
                    } else {
                        const i64 ml = mult, dl = dx, el = dy, x2 = dl * dl, y2 = el * el;
                        s[0] += ml;
                        s[1] += ml * dl;            s[2] += ml * el;
                        s[3] += ml * x2;            s[4] += ml * dl * el;       s[5] += ml * y2;
                        s[6] += ml * x2 * dl;       s[7] += ml * x2 * el;       s[8] += ml * dl * y2;
                        s[9] += ml * y2 * el;
                        s[10] += ml * x2 * x2;      s[11] += ml * x2 * dl * el; s[12] += ml * x2 * y2;
                        s[13] += ml * dl * el * y2; s[14] += ml * y2 * y2;
                    }
                }
                if (!contR) break;
            }
#pragma unroll
            for (int q = 0; q < NMOM; ++q)
                if (s[q]) atomicAdd(&acc[cid * NMOM + q], (u64)s[q]);
        }
        if (e4) atomicAdd(&euler4, e4);
        __syncthreads();
        i64* as = area_sums + (int64_t)n * maxm * VBS_AREA_SUMS;
        for (u32 c = tid; c < ncomp * NMOM; c += nthr) as[(c / NMOM) * VBS_AREA_SUMS + (c % NMOM)] = (i64)acc[c];
        if (tid == 0) fstat[n * 8 + 4] = (u32)((int)ncomp - euler4 / 4);       // holes in the opened mask
    }
    }                                                   // stages
    // probes: component ids of the 2x2 pixel cell around every band centroid (the fast path writes them itself)
    __threadfence();
    __syncthreads();
    if ((int)fstat[n * 8 + 2] == 0) {
        const int NW = H * WW;
        const u64* obits = open_bits + (int64_t)n * NW;
        const u32* wb1 = wbase_all + ((int64_t)n * 2 + 1) * NW;
        const u32* nc1 = node_comp_all + ((int64_t)n * 2 + 1) * VBS_RUN_CAP;
        const u64* bs = band_sums + (int64_t)n * maxm * 4;
        unsigned short* pr = probe_all + (int64_t)n * maxm * 4;
        const u32 nband = ncomp_all[n * 2 + 0];
        for (u32 i = threadIdx.x; i < nband; i += blockDim.x) {
            const double cn = (double)bs[i * 4 + 0];
            const float xf = (float)((double)bs[i * 4 + 1] / cn), yf = (float)((double)bs[i * 4 + 2] / cn);
            const int ix = (int)floorf(xf), iy = (int)floorf(yf);
            pr[i * 4 + 0] = (unsigned short)comp_at(obits, wb1, nc1, H, W, WW, ix, iy);
            pr[i * 4 + 1] = (unsigned short)comp_at(obits, wb1, nc1, H, W, WW, ix + 1, iy);
            pr[i * 4 + 2] = (unsigned short)comp_at(obits, wb1, nc1, H, W, WW, ix, iy + 1);
            pr[i * 4 + 3] = (unsigned short)comp_at(obits, wb1, nc1, H, W, WW, ix + 1, iy + 1);
        }
    }
    }                                                   // frames
}

bool launch_ccl(vbs_handle* h, int nb, hipStream_t s);   // false: geometry outside the round-2 fast path
bool launch_stage(vbs_handle* h, int nb, hipStream_t s);  // false: geometry outside the fused path
bool launch_stage_lat(vbs_handle* h, int nb, hipStream_t s);   // k_stage_lat.hip; false: not for this pass

// a9-a12: band / opened planes, labelling, per-component sums.  The fused kernel (k_stage.hip) takes the pass; k_morph and
// the general kernel then run over the frames it handed on (none on marker frames: their waves / workgroups find no
// flagged frame and exit).  Geometries outside the fused path - and VBS_OPT_STAGE_IMPL = 1 - take the round-2 kernels:
// k_morph over every frame, k_ccl<0|1>, the general kernel over what those hand on.
void launch_labelling(vbs_handle* h, int nb, hipStream_t s) {
    // a pass of a few frames (MarkerTracker.process: ONE) spreads each frame over several workgroups: k_stage_lat.hip
    const bool fused = h->stage_impl == 0 || h->stage_impl >= 3;
    const bool lat = fused && nb <= h->lat_frames && nb <= h->lat_slots;
    if (h->pass_cleared) h->pass_cleared = false;        // (detect_pass cleared them with the frame statistics: one launch less)
    else if (lat) launch_fill(h->lat_hdr, 0u, (size_t)VBS_LAT_MAXN * VBS_LAT_HDR + nb + 4, s);     // its headers, the counter and the flags
    else launch_fill(h->slow_total, 0u, (size_t)nb + 4, s);                       // the counter and the flags
    int all = 0;
    const u32* nslow = nullptr;
    if (lat && launch_stage_lat(h, nb, s)) {
        // what it hands on: planes and labels by ONE more kernel (k_label<ns> makes the planes itself); no frame on marker frames
        const int G = 64 / h->WW, wpf = std::max(1, std::min(64, h->H / (2 * h->bp.ns) / G));
        const int strips = wpf * G, rps = (h->H + strips - 1) / strips;
#define LABEL_M(NS_)                                                                                                              \
        VBS_LAUNCH(h, s, "k_label", k_label<NS_>, dim3(nb), dim3(1024), 0, s, h->band_bits, h->open_bits, h->wbase, h->node_pos,    \
                   h->node_comp, h->ncomp, h->band_first, h->band_sums, h->area_first, h->area_sums, h->fstat, h->lut, h->slow_flag, \
                   h->slow_total, h->probe, nb, 0, h->H, h->W, h->WW, h->maxm, 0, h->mask_bits, h->area_bits, G, strips, rps, wpf)
        if (h->bp.ns == 14) LABEL_M(14); else LABEL_M(8);
#undef LABEL_M
        return;
    }
    if (fused && launch_stage(h, nb, s)) {
        launch_morph(h, nb, h->slow_flag, s);
        nslow = h->slow_total;
    } else {
        launch_morph(h, nb, nullptr, s);
        all = (h->stage_impl != 2 && launch_ccl(h, nb, s)) ? 0 : 1;       // (2: the general kernel labels EVERY frame - its rate, tests)
    }
    VBS_LAUNCH(h, s, "k_label", k_label<0>, dim3(nb < 64 ? nb : 64), dim3(1024), 0, s, h->band_bits, h->open_bits, h->wbase,
               h->node_pos, h->node_comp, h->ncomp, h->band_first, h->band_sums, h->area_first, h->area_sums, h->fstat,
               h->lut, h->slow_flag, nslow, h->probe, nb, all, h->H, h->W, h->WW, h->maxm, VBS_KNOB("VBS_LABEL_STOP"),
               (const u64*)nullptr, (const u64*)nullptr, 0, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// fitEllipse from vertex moments (see oracle/stages.py:fit_ellipse for the algorithm being followed)
// Gaussian elimination with partial pivoting, fully unrolled: the row swap is a chain of predicated exchanges instead of
// a run-time row index, so the system stays in registers (indexed by a run-time pivot row it lived in scratch memory:
// 288 bytes per lane).  Same operations in the same order as the rolled form.
template <int N>
__device__ __forceinline__ bool solve_sym(double (&A)[N * N], double (&b)[N]) {
#pragma unroll
    for (int c = 0; c < N; ++c) {
        int p = c;
        double best = fabs(A[c * N + c]);
#pragma unroll
        for (int r = c + 1; r < N; ++r)
            if (fabs(A[r * N + c]) > best) { best = fabs(A[r * N + c]); p = r; }
        if (!(best > 1e-300)) return false;
#pragma unroll
        for (int r = c + 1; r < N; ++r) {
            if (p == r) {
#pragma unroll
                for (int k = 0; k < N; ++k) { const double t = A[c * N + k]; A[c * N + k] = A[r * N + k]; A[r * N + k] = t; }
                const double t = b[c]; b[c] = b[r]; b[r] = t;
            }
        }
#pragma unroll
        for (int r = c + 1; r < N; ++r) {
            const double f = A[r * N + c] / A[c * N + c];
#pragma unroll
            for (int k = c; k < N; ++k) A[r * N + k] -= f * A[c * N + k];
            b[r] -= f * b[c];
        }
    }
#pragma unroll
    for (int c = N - 1; c >= 0; --c) {
        double v = b[c];
#pragma unroll
        for (int k = c + 1; k < N; ++k) v -= A[c * N + k] * b[k];
        b[c] = v / A[c * N + c];
    }
    return true;
}

// m[a][b] (a+b<=4) about a point shifted by (sx, sy): sum (x-sx)^a (y-sy)^b
__device__ __forceinline__ void shift_moments(const double (&in)[5][5], double sx, double sy, double (&out)[5][5]) {
    const double C[5][5] = {{1, 0, 0, 0, 0}, {1, 1, 0, 0, 0}, {1, 2, 1, 0, 0}, {1, 3, 3, 1, 0}, {1, 4, 6, 4, 1}};
    const double px[5] = {1, -sx, sx * sx, -sx * sx * sx, sx * sx * sx * sx};
    const double py[5] = {1, -sy, sy * sy, -sy * sy * sy, sy * sy * sy * sy};
    // (every loop has constant bounds and is unrolled: the tables stay in registers)
#pragma unroll
    for (int a = 0; a <= 4; ++a)
#pragma unroll
        for (int b = 0; b <= 4; ++b) {
            if (a + b > 4) continue;
            double v = 0;
#pragma unroll
            for (int i = 0; i <= 4; ++i)
#pragma unroll
                for (int j = 0; j <= 4; ++j)
                    if (i <= a && j <= b) v += C[a][i] * C[b][j] * px[a - i] * py[b - j] * in[i][j];
            out[a][b] = v;
        }
}

// out: cx, cy, w, h, angle (float32-rounded, w <= h), nvert, ok
__device__ void fit_ellipse_moments(const i64* S, int ax, int ay, double* out) {
    const double PI = 3.14159265358979323846;
    double n = (double)S[0];
    out[5] = n;
    out[6] = 0.0;
    if (S[0] < 5) return;
    // float32 mean of the absolute coordinates, like Point2f accumulation in cv2
    float cx32 = (float)((double)S[0] * ax + (double)S[1]) / (float)n;
    float cy32 = (float)((double)S[0] * ay + (double)S[2]) / (float)n;
    double M0[5][5] = {{0}}, M[5][5];
    M0[0][0] = (double)S[0];
    M0[1][0] = (double)S[1];  M0[0][1] = (double)S[2];
    M0[2][0] = (double)S[3];  M0[1][1] = (double)S[4];  M0[0][2] = (double)S[5];
    M0[3][0] = (double)S[6];  M0[2][1] = (double)S[7];  M0[1][2] = (double)S[8];  M0[0][3] = (double)S[9];
    M0[4][0] = (double)S[10]; M0[3][1] = (double)S[11]; M0[2][2] = (double)S[12]; M0[1][3] = (double)S[13];
    M0[0][4] = (double)S[14];
    shift_moments(M0, (double)cx32 - ax, (double)cy32 - ay, M);
    double r2 = (M[2][0] + M[0][2]) / n;
    if (!(r2 > 0.0)) return;
    double scale = 100.0 / (n * sqrt(r2) * 1.2732395447351628);
    double sp[5] = {1, scale, scale * scale, scale * scale * scale, scale * scale * scale * scale};
    double m[5][5];
#pragma unroll
    for (int a = 0; a <= 4; ++a)
#pragma unroll
        for (int b = 0; b <= 4; ++b)
            if (a + b <= 4) m[a][b] = M[a][b] * sp[a + b];
    double A[25] = {
        m[4][0],  m[2][2],  m[3][1],  -m[3][0], -m[2][1],
        m[2][2],  m[0][4],  m[1][3],  -m[1][2], -m[0][3],
        m[3][1],  m[1][3],  m[2][2],  -m[2][1], -m[1][2],
        -m[3][0], -m[1][2], -m[2][1], m[2][0],  m[1][1],
        -m[2][1], -m[0][3], -m[1][2], m[1][1],  m[0][2]};
    double g[5] = {-1e4 * m[2][0], -1e4 * m[0][2], -1e4 * m[1][1], 1e4 * m[1][0], 1e4 * m[0][1]};
    // conditioning guard, in the spirit of cv2's singular-value test (w[0]*FLT_EPSILON > w[4])
    double tr = A[0] + A[6] + A[12] + A[18] + A[24];
    if (!solve_sym<5>(A, g)) return;
#pragma unroll
    for (int i = 0; i < 5; ++i) if (!isfinite(g[i])) return;
    (void)tr;
    double det = 4.0 * g[0] * g[1] - g[2] * g[2];
    if (!(fabs(det) > 1e-300)) return;
    double rp0 = (2.0 * g[1] * g[3] - g[2] * g[4]) / det;
    double rp1 = (2.0 * g[0] * g[4] - g[2] * g[3]) / det;
    double mu[5][5];
    shift_moments(m, rp0, rp1, mu);
    double A3[9] = {mu[4][0], mu[2][2], mu[3][1], mu[2][2], mu[0][4], mu[1][3], mu[3][1], mu[1][3], mu[2][2]};
    double g3[3] = {mu[2][0], mu[0][2], mu[1][1]};
    if (!solve_sym<3>(A3, g3)) return;
    const double min_eps = 1e-8;
    double ang = -0.5 * atan2(g3[2], g3[1] - g3[0]);
    double t;
    if (fabs(g3[2]) > min_eps) t = g3[2] / sin(-2.0 * ang);
    else t = g3[1] - g3[0];
    double r_2 = fabs(g3[0] + g3[1] - t);
    if (r_2 > min_eps) r_2 = sqrt(2.0 / r_2);
    double r_3 = fabs(g3[0] + g3[1] + t);
    if (r_3 > min_eps) r_3 = sqrt(2.0 / r_3);
    float ecx = (float)(rp0 / scale) + cx32;
    float ecy = (float)(rp1 / scale) + cy32;
    float wd = (float)(r_2 * 2.0 / scale);
    float ht = (float)(r_3 * 2.0 / scale);
    float fang = (float)(ang * 180.0 / PI);
    if (wd > ht) {
        float tt = wd; wd = ht; ht = tt;
        fang = (float)(90.0 + ang * 180.0 / PI);
    }
    if (fang < -180.f) fang += 360.f;
    if (fang > 360.f) fang -= 360.f;
    if (!(isfinite(wd) && isfinite(ht) && isfinite(ecx) && isfinite(ecy))) return;
    out[0] = ecx; out[1] = ecy; out[2] = wd; out[3] = ht; out[4] = fang;
    out[6] = 1.0;
}

// cv2.pointPolygonTest(contour, pt, False) >= 0 for the outer border polygon of component cid, decided from the 2x2
// pixel cell around the (float32-rounded) point; pr = component ids of the cell's pixels (x, y), (x+1, y), (x, y+1),
// (x+1, y+1) as left by k_ccl / k_probe_slow (0xFFFF = background or outside the image).
__device__ bool inside_polygon(const unsigned short* __restrict__ pr, double px, double py, u32 cid) {
    const float xf = (float)px, yf = (float)py;
    const float fx = xf - floorf(xf), fy = yf - floorf(yf);
    const bool c00 = pr[0] == cid;
    if (fx == 0.f && fy == 0.f) return c00;
    if (fy == 0.f) return c00 && pr[1] == cid;
    if (fx == 0.f) return c00 && pr[2] == cid;
    const bool c10 = pr[1] == cid, c01 = pr[2] == cid, c11 = pr[3] == cid;
    const int cnt = (int)c00 + c10 + c01 + c11;
    if (cnt == 4) return true;
    if (cnt == 3) {
        if (!c11) return fx + fy <= 1.f;
        if (!c00) return fx + fy >= 1.f;
        if (!c10) return fy >= fx;
        return fx >= fy;
    }
    if (cnt == 2) {
        if (c00 && c11) return fx == fy;
        if (c10 && c01) return fx + fy == 1.f;
    }
    return false;
}

// frame n, by one workgroup of 256 threads
__device__ __forceinline__ void finalize_frame(int n, const u32* __restrict__ ncomp_all,
                                               const u64* __restrict__ band_sums,
                                               const u32* __restrict__ area_first,
                                               const i64* __restrict__ area_sums,
                                               const unsigned short* __restrict__ probe_all,
                                               const u32* __restrict__ fstat, double* __restrict__ ell_all,
                                               double* __restrict__ det64, int32_t* __restrict__ cnt64,
                                               double* __restrict__ det32, int32_t* __restrict__ cnt32,
                                               int H, int W, int WW, int maxm, int stop, int force_seq) {
    __shared__ double bx[1024], by[1024];
    __shared__ u8 unmatched[1024];
    __shared__ int claim[1024], wsum[4];
    // per opened component (at most CCL_OPEN_COMPS = 512 of them: k_stage, k_stage_lat and k_label all stop there)
    __shared__ int best_of[CCL_OPEN_COMPS];
    __shared__ double thr_s[CCL_OPEN_COMPS], ecx_s[CCL_OPEN_COMPS], ecy_s[CCL_OPEN_COMPS];   // (:219) threshold, ellipse centre
    __shared__ u64 best_d[CCL_OPEN_COMPS];
    __shared__ int dup_s;
    const int tid = threadIdx.x;
    int status = (int)fstat[n * 8 + 2];
    if (status != 0) {
        if (tid == 0) { cnt64[n] = status; if (cnt32) cnt32[n] = status; }
        return;
    }
    const int nb_ = min((int)ncomp_all[n * 2 + 0], 1024), na = min((int)ncomp_all[n * 2 + 1], CCL_OPEN_COMPS);   // (never past the tables)
    const u64* bs = band_sums + (int64_t)n * maxm * 4;
    for (int i = tid; i < nb_; i += blockDim.x) {
        double c = (double)bs[i * 4 + 0];
        bx[i] = (double)bs[i * 4 + 1] / c;             // center_of_mass: integer sums, one division
        by[i] = (double)bs[i * 4 + 2] / c;
        unmatched[i] = 1;
    }
    double* ell = ell_all + (int64_t)n * maxm * 8;
    const u32* af = area_first + (int64_t)n * maxm;
    for (int i = tid; i < na; i += blockDim.x) {
        u32 fp = af[i];
        fit_ellipse_moments(area_sums + ((int64_t)n * maxm + i) * VBS_AREA_SUMS, fp % W, fp / W, ell + i * 8);
    }
    __syncthreads();
    if (stop == 1) return;
    const unsigned short* probe = probe_all + (int64_t)n * maxm * 4;
    double* d64 = det64 + (int64_t)n * maxm * 6;
    double* d32 = det32 ? det32 + (int64_t)n * maxm * 6 : nullptr;

    // ---- matching (:203-243).  The reference walks the contours in order and gives each the nearest
    // still-unmatched centre inside it.  Every contour first gets its nearest admissible centre among
    // ALL centres, in parallel; if no centre is claimed twice, the sequential walk would have made exactly
    // these choices (a contour loses its first choice only to an earlier contour with the same choice).
    // Otherwise (never seen on marker frames) one wave replays the reference's sequential loop.
    // A centre can only lie inside the polygon of a component that owns a pixel of the 2x2 cell around it (every
    // accepting branch of inside_polygon needs one), so the search runs from the centres: a thread per centre tries the
    // at most four components of its probe cell, and a contour keeps the smallest (distance, index) offered to it - the
    // order the reference's strict "<" over ascending indices produces - through two LDS atomic minima: the distance's
    // bit pattern (monotone for non-negative doubles), then the index among the centres at that distance.
    for (int i = tid; i < nb_; i += blockDim.x) claim[i] = 0;
    for (int c = tid; c < na; c += blockDim.x) {
        const double* e = ell + c * 8;
        double thr = -1.0;
        if (e[6] != 0.0 && e[5] >= 5.0) {               // len(contour) >= 5 (:204) and a valid fit
            const double w = e[2], hh = e[3], minor = (w > hh) ? hh : w;
            if (!(minor < 5.0)) thr = (minor / 10.0) * (minor / 10.0);     // (:219)
        }
        thr_s[c] = thr;
        ecx_s[c] = e[0]; ecy_s[c] = e[1];               // (the matching below reads the centres a few times: not from memory)
        best_d[c] = ~0ull;
        best_of[c] = 0x7FFFFFFF;
    }
    if (tid == 0) dup_s = force_seq;
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = tid; i < nb_; i += blockDim.x) {
            const unsigned short* pr = probe + i * 4;
            const double cx = bx[i], cy = by[i];
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const u32 cid = pr[q4];
                if (cid >= (u32)na) continue;
                bool seen = false;
#pragma unroll
                for (int q5 = 0; q5 < 4; ++q5) seen |= (q5 < q4) && (pr[q5] == cid);
                if (seen) continue;
                const double thr = thr_s[cid];
                if (!(thr >= 0.0)) continue;
                const double dx = cx - ecx_s[cid], dy = cy - ecy_s[cid], d = dx * dx + dy * dy;
                if (!(d < thr) || !inside_polygon(pr, cx, cy, cid)) continue;
                const u64 key = (u64)__double_as_longlong(d);
                if (pass == 0) atomicMin(&best_d[cid], key);
                else if (key == best_d[cid]) atomicMin(&best_of[cid], i);
            }
        }
        __syncthreads();
    }
    for (int c = tid; c < na; c += blockDim.x) {
        const int bi = best_of[c] == 0x7FFFFFFF ? -1 : best_of[c];
        best_of[c] = bi;
        if (bi >= 0 && atomicAdd(&claim[bi], 1) > 0) dup_s = 1;
    }
    __syncthreads();
    if (!dup_s) {
        // output order = contour order = descending component id; rank by a block scan over reversed ids
        const int per = (na + blockDim.x - 1) / blockDim.x;
        const int r0 = tid * per, r1 = min(r0 + per, na);
        int mine = 0;
        for (int r = r0; r < r1; ++r) mine += (best_of[na - 1 - r] >= 0);
        int inc = mine;
        const int lane = tid & 63, wave = tid >> 6;
        for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = inc - mine;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        for (int r = r0; r < r1; ++r) {
            int ci = na - 1 - r, bi = best_of[ci];
            if (bi < 0) continue;
            const double* e = ell + ci * 8;
            double w = e[2], hh = e[3], ang = e[4], major, minor, eang;
            if (w > hh) { major = w; minor = hh; eang = ang; }
            else { major = hh; minor = w; eang = ang + 90.0; }
            double* o = d64 + base * 6;
            o[0] = bx[bi]; o[1] = by[bi]; o[2] = major; o[3] = minor; o[4] = eang; o[5] = bi + 1;
            if (d32) {
                double* f = d32 + base * 6;
                f[0] = bx[bi]; f[1] = by[bi]; f[2] = major; f[3] = minor; f[4] = eang; f[5] = bi + 1;
            }
            ++base;
        }
        if (tid == (int)blockDim.x - 1) { cnt64[n] = base; if (cnt32) cnt32[n] = base; }
        return;
    }
    if (tid >= 64) return;
    // ---- sequential replay in cv2 contour order (last component found first), one wave -------------
    int count = 0;
    for (int ci = na - 1; ci >= 0; --ci) {
        const double* e = ell + ci * 8;
        if (e[6] == 0.0 || e[5] < 5.0) continue;       // len(contour) < 5 (:204) or no fit
        double ecx = e[0], ecy = e[1], w = e[2], hh = e[3], ang = e[4];
        double major, minor, eang;
        if (w > hh) { major = w; minor = hh; eang = ang; }
        else { major = hh; minor = w; eang = ang + 90.0; }
        if (minor < 5.0) continue;                      // (:219)
        double thr = (minor / 10.0) * (minor / 10.0);
        double best = 1e300;
        int bi = -1;
        for (int i = tid; i < nb_; i += 64) {
            if (!unmatched[i]) continue;
            double dx = bx[i] - ecx, dy = by[i] - ecy;
            double d = dx * dx + dy * dy;
            if (d < thr && d < best &&
                inside_polygon(probe + i * 4, bx[i], by[i], (u32)ci)) {
                best = d; bi = i;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            double ob = __shfl_xor(best, off);
            int oi = __shfl_xor(bi, off);
            if (oi >= 0 && (bi < 0 || ob < best || (ob == best && oi < bi))) { best = ob; bi = oi; }
        }
        if (bi >= 0) {
            if (tid == 0) {
                unmatched[bi] = 0;
                double* o = d64 + count * 6;
                o[0] = bx[bi]; o[1] = by[bi]; o[2] = major; o[3] = minor; o[4] = eang; o[5] = bi + 1;
                if (d32) {
                    double* f = d32 + count * 6;
                    f[0] = bx[bi]; f[1] = by[bi]; f[2] = major; f[3] = minor; f[4] = eang; f[5] = bi + 1;
                }
            }
            ++count;
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (tid == 0) { cnt64[n] = count; if (cnt32) cnt32[n] = count; }
}

__global__ __launch_bounds__(256) void k_finalize(const u32* __restrict__ ncomp_all, const u64* __restrict__ band_sums,
                                                  const u32* __restrict__ area_first, const i64* __restrict__ area_sums,
                                                  const unsigned short* __restrict__ probe_all,
                                                  const u32* __restrict__ fstat, double* __restrict__ ell_all,
                                                  double* __restrict__ det64, int32_t* __restrict__ cnt64,
                                                  double* __restrict__ det32, int32_t* __restrict__ cnt32,
                                                  int H, int W, int WW, int maxm, int stop, int force_seq) {
    finalize_frame(blockIdx.x, ncomp_all, band_sums, area_first, area_sums, probe_all, fstat, ell_all, det64, cnt64, det32, cnt32,
                   H, W, WW, maxm, stop, force_seq);
}

void launch_finalize(vbs_handle* h, int nb, double* det, int32_t* counts, hipStream_t s) {
    VBS_LAUNCH(h, s, "k_finalize", k_finalize, dim3(nb), dim3(256), 0, s, h->ncomp, h->band_sums, h->area_first,
                       h->area_sums, h->probe, h->fstat, h->ell, h->det64,
                       h->cnt, det, counts, h->H, h->W, h->WW, h->maxm, VBS_KNOB("VBS_FINAL_STOP"),
                       h->force_seq_match ? 1 : 0);                 // (vbs_set_option: exercises the sequential replay)
}

// The few-frames path: a13 and a15 (+ a19 / a20) of a frame in ONE launch - the same workgroup fits and matches the
// frame's detections, then tracks them against the reference IDs (a launch on a dependent stream costs ~ 5 us, as much as
// either kernel works on one frame).  Same code, same results as k_finalize followed by k_track.
#include "track_common.h"
__global__ __launch_bounds__(256) void k_finalize_track(const u32* __restrict__ ncomp_all, const u64* __restrict__ band_sums,
                                                        const u32* __restrict__ area_first, const i64* __restrict__ area_sums,
                                                        const unsigned short* __restrict__ probe_all,
                                                        const u32* __restrict__ fstat, double* __restrict__ ell_all,
                                                        double* __restrict__ det64, int32_t* __restrict__ cnt64,
                                                        double* __restrict__ det32, int32_t* __restrict__ cnt32,
                                                        int H, int W, int WW, int maxm, int force_seq,
                                                        const double* __restrict__ ref_xy, int m_ref, double min_dist,
                                                        float* __restrict__ table, int do3d, CamD cam, double min_size) {
    finalize_frame(blockIdx.x, ncomp_all, band_sums, area_first, area_sums, probe_all, fstat, ell_all, det64, cnt64, det32, cnt32,
                   H, W, WW, maxm, 0, force_seq);
    // (the frame's detections and count were written by THIS workgroup: the barrier's workgroup-scope release / acquire is
    //  all their readers need - an agent-scope fence here writes back and invalidates the XCD's L2 for nothing)
    __syncthreads();
    track_frame(blockIdx.x, det64, cnt64, maxm, ref_xy, m_ref, min_dist, table, do3d, cam, min_size);
}

void launch_finalize_track(vbs_handle* h, int nb, double* det, int32_t* counts, const double* ref_xy, int m_ref, double min_dist,
                           float* table, const vbs_camera* cam, double min_size, hipStream_t s) {
    CamD c{};
    if (cam) c = make_cam(*cam);
    VBS_LAUNCH(h, s, "k_finalize_track", k_finalize_track, dim3(nb), dim3(256), 0, s, h->ncomp, h->band_sums, h->area_first,
               h->area_sums, h->probe, h->fstat, h->ell, h->det64, h->cnt, det, counts, h->H, h->W, h->WW, h->maxm,
               h->force_seq_match ? 1 : 0, ref_xy, m_ref, min_dist, table, cam ? 1 : 0, c, min_size);
}
