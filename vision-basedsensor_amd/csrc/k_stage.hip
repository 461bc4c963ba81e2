// a9-a13, fused fast path (marker_detection.py:170-196): band = mask & ~erode(mask) and the 5x5 opening of the area
// mask, the connected components of both (4- / 8-connectivity) and the per-component sums k_finalize needs, in ONE
// kernel per pass, one workgroup of 1024 threads per frame.  Nothing but the two input bit planes is read from memory
// and neither the band / opened planes nor any list of pixels is written: a frame's planes live in the registers of its
// workgroup.
//
// A thread owns a COLUMN SEGMENT: word column j (64 px) x R consecutive rows (R = 22 at 1280x1024: 2 VGPRs per row).
// A wave holds G = 64 / WW such segments side by side per word column (lane = g * WW + j), so the words left and right
// of a lane's word are in the neighbouring lanes (DPP moves) and the rows above / below it are its own registers; the
// segment also keeps one halo row above and below (computed, not exchanged).  All passes over the rows are unrolled:
//   morph  the source rows a segment needs (R + NS, R + 10) are loaded once; the vertical 14- / 5-row AND / OR runs in
//          place by doubling (2, 4, 8, 14 rows), the horizontal one on 64-bit words by doubling with the neighbour
//          lane's word (v_alignbit funnel shifts)
//   A      runs that start in each word (+ bit-quad Euler number of the opened mask); the runs (union-find nodes) are
//          numbered in RASTER order: counts packed two rows per register, a segmented prefix sum over the lanes of a
//          row block, and a block prefix sum over the image rows in LDS.  Raster order makes the run that enters a word
//          from the left the node `word base - 1`, the root of a component its first run, and the rank of the roots
//          ndimage.label's / cv2.findContours' order.
//   B, C   links to the row above (ccl_common.h: first link = parent, further links -> pair list), pointer jumping,
//          the pair unions, flatten, rank the roots: as k_ccl.hip, on the node table in LDS
//   D      band: count / sum x / sum y; the sums of a segment stay in registers (two entries: first run of a word,
//          other runs) and go to the component's LDS accumulators when the component under the segment changes
//          open: CHAIN_APPROX_SIMPLE vertex multiplicity BIT-PARALLEL (the 256-entry table as boolean functions of
//          the eight shifted neighbour planes: an arc of background neighbours that starts at direction a counts unless
//          it has no 4-neighbour or is exactly {a, a+1, a+2}), so only real vertices are visited; per run the row sums
//          s_a = sum mult dx^a, then the 15 moments as s_a dy^b: orders 0-3 cached in registers like the band sums,
//          order 4 by LDS atomics per run; the component ids of the 2x2 cell around every band centroid ("probes")
//          are answered by the segment that owns the pixel (requests posted through a small LDS mailbox)
// Frames the fast path cannot take (more runs than the node table holds, too many components / root candidates / pair
// links, holes in the opened mask, a vertex of multiplicity > 2, a crowded mailbox) set their slow flag: k_morph and
// k_label (k_label.hip) redo them.
#include "ccl_common.h"

#define ST_NT 1024
#define ST_MB_CAP 8                // probe requests a segment can hold (one per centroid, row and word)
#define NONE32 0xFFFFFFFFu

struct StageGeom {
    int H, W, WW, G, NB, maxm;      // G row blocks per wave, NB = 16 G row blocks of R rows
    u32 node_cap, pair_cap;
    u32 off_rowb, off_acc, off_mb, off_tmp;      // byte offsets into the dynamic LDS
    int stop;                       // debug builds: leave after phase `stop`
};

__device__ __forceinline__ u64 mk64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// bit p of the result = bit p + s of the row (this word, then the right neighbour's); `fill` = what lies past the row
// end.  Every lane must execute it (DPP), s = 1 .. 31.
__device__ __forceinline__ u64 shift_from_right(u64 x, int s, bool hasr, u32 fill) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    u32 rlo = dpp_shl1(lo);
    if (!hasr) rlo = fill;
    return mk64(__builtin_amdgcn_alignbit(hi, lo, (u32)s), __builtin_amdgcn_alignbit(rlo, hi, (u32)s));
}
// bit p of the result = bit p - s of the row (the left neighbour's word, then this one)
__device__ __forceinline__ u64 shift_from_left(u64 x, int s, bool hasl, u32 fill) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    u32 lhi = dpp_shr1(hi);
    if (!hasl) lhi = fill;
    return mk64(__builtin_amdgcn_alignbit(lo, lhi, (u32)(32 - s)), __builtin_amdgcn_alignbit(hi, lo, (u32)(32 - s)));
}

// AND (ERODE) / OR over the window x - N/2 .. x - N/2 + N - 1 of a row held one word per lane (scipy's / cv2's anchor):
// the part of the window at and to the right of x by doubling with the right neighbour's word, the part at and to the
// left of x with the left neighbour's; what lies past either end of the row is `fill` (ones for an erosion: ignored)
template <int N, bool ERODE>
__device__ __forceinline__ u64 hwin(u64 v, bool hasl, bool hasr) {
    const u32 fill = ERODE ? ~0u : 0u;
    constexpr int NL = N / 2 + 1, NR = N - N / 2;        // pixels x - N/2 .. x and x .. x + NR - 1
    u64 f = v, b = v;
    int have = 1;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        if (have >= NR) break;
        const int s = have < NR - have ? have : NR - have;
        const u64 t = shift_from_right(f, s, hasr, fill);
        f = ERODE ? (f & t) : (f | t);
        have += s;
    }
    have = 1;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        if (have >= NL) break;
        const int s = have < NL - have ? have : NL - have;
        const u64 t = shift_from_left(b, s, hasl, fill);
        b = ERODE ? (b & t) : (b | t);
        have += s;
    }
    return ERODE ? (f & b) : (f | b);
}

// in-place vertical window over N consecutive entries of a[0 .. CNT): a[t] = op(a[t .. t + N - 1])
template <int N, int CNT, bool ERODE>
__device__ __forceinline__ void vwin(u64 (&a)[CNT]) {
#define VSTEP(S, LEN)                                                          \
    _Pragma("unroll") for (int t = 0; t + (S) < CNT; ++t) a[t] = ERODE ? (a[t] & a[t + (S)]) : (a[t] | a[t + (S)]);
    if (N >= 2) { VSTEP(1, 2) }
    if (N >= 4) { VSTEP(2, 4) }
    if (N == 5) { VSTEP(1, 5) }
    if (N >= 8) { VSTEP(4, 8) }
    if (N == 14) { VSTEP(6, 14) }
#undef VSTEP
}

#define WB(t) ((wbp[(t) >> 1] >> (((t) & 1) * 16)) & 0xFFFFu)

// Phases A - C for the plane in Bw (tile rows 0 .. R + 1 = image rows y0 - 1 .. y0 + R; row 0 and R + 1 are halos).
// Returns 0, or (workgroup-uniform) why the fast path cannot take the frame: 1 more runs than the node table holds,
// 2 pair list full, 3 too many components, 4 holes in the opened mask, 5 root list full.  On return P[node] = component id (bit 15
// marks the root run), wbp = node index before each word of tile rows 0 .. R (two per register).
template <int R, int MODE>
__device__ __forceinline__ int stage_label(const u64 (&Bw)[R + 2], u32 lm, u32 rm, u32 (&wbp)[(R + 2) / 2],
                                            unsigned short* P, unsigned short* rowb, unsigned char* accb, u32* tmp,
                                            int* misc, const StageGeom& geo, int y0, int j, bool act, u32& total_out,
                                            u32& ncomp_out) {
    constexpr int NP = (R + 2) / 2;
    const int tid = threadIdx.x;
    const int WW = geo.WW, W = geo.W;
    // ---- A: run starts per word, raster numbering (+ Euler number) ---------------------------------------------
    u32 cpk[NP];
    int e4 = 0;
#pragma unroll
    for (int k = 0; k < NP; ++k) cpk[k] = 0;
#pragma unroll
    for (int t = 0; t <= R; ++t) {
        const u64 B = Bw[t];
        const u32 c = (u32)__popcll(ccl_starts(B, (lm >> t) & 1u));
        cpk[t >> 1] |= c << ((t & 1) * 16);
        if (MODE == 1 && t >= 1) {
            // bit quads (8-connected foreground): E = (Q1 - Q3 - 2 QD) / 4 over all 2x2 windows of the zero-padded image;
            // a word counts the windows whose top row is its row (image row 0 also the padding row above it)
            const bool top = (y0 + t - 1) == 0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (q == 1 && !top) continue;
                const u64 a = q ? 0ull : B, bq = q ? B : Bw[t + 1];
                const u32 an = q ? 0u : ((rm >> t) & 1u), bn = q ? ((rm >> t) & 1u) : ((rm >> (t + 1)) & 1u);
                if (a | bq | an | bn) {
                    const u64 a1 = (a >> 1) | ((u64)an << 63), b1 = (bq >> 1) | ((u64)bn << 63);
                    const u64 x2 = (a ^ a1) ^ (bq ^ b1);
                    const u64 pairs = (a & a1) | (a & bq) | (a & b1) | (a1 & bq) | (a1 & b1) | (bq & b1);
                    const u64 qd = (a & b1 & ~a1 & ~bq) | (a1 & bq & ~a & ~b1);
                    e4 += __popcll(x2 & ~pairs) - __popcll(x2 & pairs) - 2 * __popcll(qd);
                    if (j == 0) e4 += (int)((a ^ bq) & 1ull);                   // window x = -1: only (0,y), (0,y+1)
                }
            }
        }
    }
    if (MODE == 1 && e4) atomicAdd(&misc[0], e4);
    // prefix sum over the word columns of each row block (lanes g WW .. g WW + WW - 1), two rows per register
    u32 inc[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) inc[k] = cpk[k];
    for (int d = 1; d < WW; d <<= 1) {                   // (uniform trip count)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const u32 tv = (u32)__shfl_up((int)inc[k], d);
            if (j >= d) inc[k] += tv;
        }
    }
    if (act && j == WW - 1) {
#pragma unroll
        for (int t = 1; t <= R; ++t) rowb[y0 + t - 1] = (unsigned short)((inc[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu);
    }
    __syncthreads();
    u32 total;
    {
        const int NROWS = geo.NB * R, K = (NROWS + ST_NT - 1) / ST_NT;
        const int i0 = min(tid * K, NROWS), i1 = min(i0 + K, NROWS);
        u32 s = 0;
        for (int i = i0; i < i1; ++i) s += rowb[i];
        u32 ex = ccl_scan(s, tmp, &total);
        for (int i = i0; i < i1; ++i) { const u32 c = rowb[i]; rowb[i] = (unsigned short)ex; ex += c; }
    }
    total_out = total;
    if (total > geo.node_cap) return 1;                  // workgroup-uniform
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        u32 rb = 0;
        const int ya = y0 + 2 * k - 1, yb = ya + 1;
        if (act && ya >= 0) rb = rowb[ya];
        if (act && 2 * k + 1 <= R) rb |= (u32)rowb[yb] << 16;
        wbp[k] = inc[k] - cpk[k] + rb;                   // (16-bit halves: no carry, every sum < 32768)
    }
    if (geo.stop == 2) return 0;

    // ---- B: parents, pointer jumping, the remaining links ----------------------------------------------------------
    CclLists L;
    L.pairs = reinterpret_cast<u32*>(accb);
    L.npairs = &misc[5];
    L.pair_cap = (int)geo.pair_cap;
    L.roots = MODE == 1 ? reinterpret_cast<u32*>(accb + CCL_MOM_COMPS * NMOM * 8 + CCL_OPEN_COMPS * 4) : nullptr;
    L.nroots = &misc[4];
    bool ovf = false;
#pragma unroll
    for (int t = 1; t <= R; ++t) {
        const u64 B = Bw[t];
        if (!__any(B != 0ull)) continue;                 // (wave-uniform)
        if (B) {
            const u32 pos0 = (u32)(y0 + t - 1) * (u32)W + 64u * (u32)j;
            ovf |= ccl_link_word<MODE, 0>(P, B, Bw[t - 1], (lm >> t) & 1u, (lm >> (t - 1)) & 1u, (rm >> (t - 1)) & 1u, WB(t),
                                          WB(t - 1), pos0, L);
        }
    }
    if (ovf) misc[6] = 1;                                // the pair list is full: hand the frame on
    __syncthreads();
    if (misc[6]) return 2;
    if (geo.stop == 7) return 0;
    // pointer jumping: every node ends on the root of its tree (parents only ever move to an ancestor, so the
    // unsynchronised reads inside a round are harmless); three flags in turn: the one cleared in round r was last read
    // before the barrier of round r - 1
    for (int f = 0;; f = f == 2 ? 0 : f + 1) {
        if (tid == 0) misc[1 + (f == 2 ? 0 : f + 1)] = 0;
        bool ch = false;
        for (u32 i = tid; i < total; i += ST_NT) {
            const u32 p = P[i], pp = P[p];
            if (pp != p) { P[i] = (unsigned short)pp; ch = true; }
        }
        if (ch) misc[1 + f] = 1;
        __syncthreads();
        if (!misc[1 + f]) break;
    }
    if (geo.stop == 8) return 0;
    {   // the further links, densely: one pair per thread
        const int np = misc[5];
        for (int i = tid; i < np; i += ST_NT) { const u32 pr = L.pairs[i]; ccl_union(P, pr >> 16, pr & 0xFFFFu); }
    }
    __syncthreads();
    if (geo.stop == 3) return 0;

    // ---- C: flatten, rank the roots in raster order, resolve every node to its component id ---------------------
    for (u32 i = tid; i < total; i += ST_NT) {
        u32 x = i, p;
        while ((p = ((volatile unsigned short*)P)[x]) != x) x = p;
        if (x != i) P[i] = (unsigned short)x;
    }
    __syncthreads();
    const u32 K2 = (total + ST_NT - 1) / ST_NT;
    const u32 r0 = min((u32)tid * K2, total), r1 = min(r0 + K2, total);
    u32 nroot = 0;
    for (u32 i = r0; i < r1; ++i) nroot += (P[i] == i);
    u32 ncomp;
    u32 cid0 = ccl_scan(nroot, tmp, &ncomp);
    ncomp_out = ncomp;
    if (ncomp > (u32)geo.maxm || ncomp > (MODE == 0 ? 1024u : (u32)CCL_OPEN_COMPS)) return 3;
    if (MODE == 1 && (int)ncomp - misc[0] / 4 != 0) return 4;       // holes: RETR_EXTERNAL needs the fill passes of the general path
    if (MODE == 1 && misc[4] > CCL_ROOT_LIST) return 5;
    for (u32 i = r0; i < r1; ++i)
        if (P[i] == i) P[i] = (unsigned short)(0x8000u | cid0++);
    __syncthreads();
    for (u32 i = tid; i < total; i += ST_NT) {
        const u32 v = P[i];
        if (!(v & 0x8000u)) P[i] = (unsigned short)(P[v] & 0x7FFFu);
    }
    __syncthreads();
    return 0;
}

struct BandEntry { u32 cid, cnt, sx, sy; };

__device__ __forceinline__ void band_flush(BandEntry& e, u32* acnt, u64* asx, u64* asy) {
    if (e.cnt) { atomicAdd(&acnt[e.cid], e.cnt); atomicAdd(&asx[e.cid], (u64)e.sx); atomicAdd(&asy[e.cid], (u64)e.sy); }
    e.cnt = 0; e.sx = 0; e.sy = 0;
}

struct MomEntry { u32 cid; int m[10]; };                 // moments of order 0 - 3 about the component's first pixel

__device__ __forceinline__ void mom_flush(MomEntry& e, u64* acc) {
    if (e.cid != NONE32) {
        u64* a = acc + e.cid * NMOM;
#pragma unroll
        for (int q = 0; q < 10; ++q)
            if (e.m[q]) atomicAdd(&a[q], (u64)(i64)e.m[q]);
    }
    e.cid = NONE32;
#pragma unroll
    for (int q = 0; q < 10; ++q) e.m[q] = 0;
}

template <int R, int NS>
__global__ __launch_bounds__(ST_NT, 4) void k_stage(const u64* __restrict__ mask_all, const u64* __restrict__ area_all,
                                                    u32* __restrict__ ncomp_all, u64* __restrict__ band_sums,
                                                    u32* __restrict__ area_first, i64* __restrict__ area_sums,
                                                    unsigned short* __restrict__ probe_all, u32* __restrict__ fstat,
                                                    u32* __restrict__ slow_flag, StageGeom geo) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned short* P = reinterpret_cast<unsigned short*>(smem);                          // [node_cap]
    unsigned short* rowb = reinterpret_cast<unsigned short*>(smem + geo.off_rowb);        // [NB R + 1] nodes before a row
    unsigned char* accb = smem + geo.off_acc;                                            // band sums | pairs | moments ..
    u32* mb_cnt = reinterpret_cast<u32*>(smem + geo.off_mb);                              // [ST_NT] probe requests
    u32* mb_req = mb_cnt + ST_NT;                                                         // [ST_NT][ST_MB_CAP]
    u32* tmp = reinterpret_cast<u32*>(smem + geo.off_tmp);                                // [32]
    int* misc = reinterpret_cast<int*>(tmp + 32);                                         // [16]
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = geo.H, W = geo.W, WW = geo.WW, G = geo.G, maxm = geo.maxm;
    const int g = lane / WW, j = lane - g * WW;
    const bool act = g < G;
    const int blk = wave * G + g, y0 = blk * R;
    const bool hasl = j > 0, hasr = j + 1 < WW;
    const u64 vm = act ? valid_mask(j, W) : 0ull;
    constexpr int NP = (R + 2) / 2;
    constexpr int LO = -(NS / 2);
    if (tid < 16) misc[tid] = 0;     // [0] Euler sum, [1..3] jumping flags, [4] roots, [5] pairs, [6] hand the frame on
    mb_cnt[tid] = 0;
    const int64_t fo = (int64_t)n * H * WW;
    u64 Bw[R + 2];
    u32 wbp[NP];
    u32 lm, rm;
    // ================================ band plane ====================================================================
    {
        const u64* M = mask_all + fo;
        {
            u64 a[R + NS];                               // source rows y0 - 1 + LO ..; outside the image: ones (ignored)
#pragma unroll
            for (int t = 0; t < R + NS; ++t) {
                const int ys = y0 - 1 + LO + t;
                a[t] = (act && ys >= 0 && ys < H) ? (M[(int64_t)ys * WW + j] | ~vm) : ~0ull;
            }
            vwin<NS, R + NS, true>(a);                   // a[t] = rows of the window of tile row t, t = 0 .. R
#pragma unroll
            for (int t = 0; t <= R; ++t) Bw[t] = hwin<NS, true>(a[t], hasl, hasr);
        }
#pragma unroll
        for (int t = 0; t <= R; ++t) {
            const int y = y0 - 1 + t;
            const u64 mc = (act && y >= 0 && y < H) ? M[(int64_t)y * WW + j] : 0ull;
            Bw[t] = mc & ~Bw[t] & vm;                    // :171-174  maxima = mask & (window holds a 0)
        }
        Bw[R + 1] = 0;
    }
    {
        u32 mym = 0, myl = 0;
#pragma unroll
        for (int t = 0; t <= R + 1; ++t) { mym |= (u32)(Bw[t] >> 63) << t; myl |= ((u32)Bw[t] & 1u) << t; }
        lm = dpp_shr1(mym); rm = dpp_shl1(myl);
        if (!hasl) lm = 0;
        if (!hasr) rm = 0;
    }
    __syncthreads();                                     // misc / mailbox cleared
    if (geo.stop == 1) return;
    u32 total, ncomp;
    if (const int why = stage_label<R, 0>(Bw, lm, rm, wbp, P, rowb, accb, tmp, misc, geo, y0, j, act, total, ncomp)) {
        if (tid == 0) slow_flag[n] = (u32)why;           // (the value says why: vbs_stage_tables)
        return;
    }
    if (geo.stop && geo.stop < 10) return;
    {
        // ---- D (band): count, sum x, sum y (center_of_mass :181) ------------------------------------------------
        u32* acnt = reinterpret_cast<u32*>(accb);                                        // [maxm]
        u64* asx = reinterpret_cast<u64*>(accb + 8 * ((maxm + 1) / 2));                   // [maxm]
        u64* asy = asx + maxm;                                                           // [maxm]
        for (u32 c = tid; c < ncomp; c += ST_NT) { acnt[c] = 0; asx[c] = 0; asy[c] = 0; }
        __syncthreads();
        BandEntry e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
#pragma unroll
        for (int t = 1; t <= R; ++t) {
            const u64 B = Bw[t];
            if (!__any(B != 0ull)) continue;
            if (B) {
                const u64 stB = ccl_starts(B, (lm >> t) & 1u);
                const u32 bc = WB(t), y = (u32)(y0 + t - 1);
                u64 mB = B;
                bool firstrun = true;
                while (mB) {
                    const u64 lowbit = mB & (~mB + 1ull);
                    const u64 t2 = mB + lowbit;
                    const u64 gg = mB & ~t2;
                    mB &= t2;
                    const u32 cid = P[bc + (u32)__popcll(stB & ((lowbit << 1) - 1ull)) - 1u] & 0x7FFFu;
                    const u32 len = (u32)__popcll(gg), x0 = 64u * (u32)j + (u32)(__ffsll((long long)gg) - 1);
                    const u32 sx = len * x0 + len * (len - 1) / 2;
                    if (firstrun) {
                        if (e0.cid != cid) { band_flush(e0, acnt, asx, asy); e0.cid = cid; }
                        e0.cnt += len; e0.sx += sx; e0.sy += len * y;
                    } else {
                        if (e1.cid != cid) { band_flush(e1, acnt, asx, asy); e1.cid = cid; }
                        e1.cnt += len; e1.sx += sx; e1.sy += len * y;
                    }
                    firstrun = false;
                }
            }
        }
        band_flush(e0, acnt, asx, asy);
        band_flush(e1, acnt, asx, asy);
        __syncthreads();
        // the sums go out; the probe requests (2x2 pixel cell around every centroid) go to the segments that own the pixels
        u64* bs = band_sums + (int64_t)n * maxm * 4;
        unsigned short* pr = probe_all + (int64_t)n * maxm * 4;
        for (u32 c = tid; c < ncomp; c += ST_NT) {
            const u32 cn_ = acnt[c];
            const u64 sx = asx[c], sy = asy[c];
            bs[c * 4 + 0] = cn_; bs[c * 4 + 1] = sx; bs[c * 4 + 2] = sy;
            const double cn = (double)cn_;
            const float xf = (float)((double)sx / cn), yf = (float)((double)sy / cn);
            const int ix = (int)floorf(xf), iy = (int)floorf(yf);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int px = ix + (q & 1), py = iy + (q >> 1);
                if (px < 0 || py < 0 || px >= W || py >= H) { pr[c * 4 + q] = (unsigned short)NONE16; continue; }
                // one request per row and word: the pixel (ix + 1, py) rides along when it lies in the same word
                const bool pair = (q & 1) == 0 && px + 1 < W && (px & 63) != 63;
                if ((q & 1) && px > 0 && (px & 63) != 0) continue;                   // rode along with (ix, py)
                const int ob = py / R, oi = py - ob * R + 1, ow = ob / G, og = ob - ow * G;
                const int owner = ow * 64 + og * WW + (px >> 6);
                const u32 slot = atomicAdd(&mb_cnt[owner], 1u);
                if (slot < ST_MB_CAP)
                    mb_req[owner * ST_MB_CAP + slot] = c | ((u32)q << 10) | ((u32)oi << 12) | ((u32)(px & 63) << 17) | ((u32)pair << 23);
                else misc[6] = 1;
            }
        }
        if (tid == 0) { ncomp_all[n * 2 + 0] = ncomp; fstat[n * 8 + 5] = ncomp; }
    }
    __syncthreads();
    if (misc[6]) {                                       // a crowded mailbox (many tiny band components in one segment)
        if (tid == 0) slow_flag[n] = 8;
        return;
    }
    if (geo.stop == 10) return;
    const u32 nband = ncomp;
    (void)nband;

    // ================================ opened area plane =============================================================
    {
        const u64* A = area_all + fo;
        u64 a[R + 10];                                   // source rows y0 - 5 .. y0 + R + 4
#pragma unroll
        for (int t = 0; t < R + 10; ++t) {
            const int ys = y0 - 5 + t;
            a[t] = (act && ys >= 0 && ys < H) ? (A[(int64_t)ys * WW + j] | ~vm) : ~0ull;
        }
        vwin<5, R + 10, true>(a);                        // a[t] = 5-row AND about row y0 - 3 + t, t = 0 .. R + 5
#pragma unroll
        for (int t = 0; t < R + 6; ++t) {
            const int ye = y0 - 3 + t;
            const u64 e = hwin<5, true>(a[t], hasl, hasr);
            a[t] = (act && ye >= 0 && ye < H) ? (e & vm) : 0ull;         // eroded rows; nothing outside the image
        }
        u64 (&d)[R + 10] = a;
        {   // 5-row OR about row y0 - 1 + t over eroded rows t .. t + 4 (only the first R + 6 entries are eroded rows)
#pragma unroll
            for (int t = 0; t + 1 < R + 6; ++t) d[t] |= d[t + 1];
#pragma unroll
            for (int t = 0; t + 3 < R + 6; ++t) d[t] |= d[t + 2];
#pragma unroll
            for (int t = 0; t + 4 < R + 6; ++t) d[t] |= d[t + 1];
        }
#pragma unroll
        for (int t = 0; t <= R + 1; ++t) {
            const int y = y0 - 1 + t;
            const u64 o = hwin<5, false>(d[t], hasl, hasr);
            Bw[t] = (act && y >= 0 && y < H) ? (o & vm) : 0ull;          // :195  morphologyEx(MORPH_OPEN, 5x5)
        }
    }
    {
        u32 mym = 0, myl = 0;
#pragma unroll
        for (int t = 0; t <= R + 1; ++t) { mym |= (u32)(Bw[t] >> 63) << t; myl |= ((u32)Bw[t] & 1u) << t; }
        lm = dpp_shr1(mym); rm = dpp_shl1(myl);
        if (!hasl) lm = 0;
        if (!hasr) rm = 0;
    }
    if (tid < 6) misc[tid] = 0;                          // (not [6]: a late reader of the check above must still see 0)
    __syncthreads();
    if (geo.stop == 11) return;
    if (const int why = stage_label<R, 1>(Bw, lm, rm, wbp, P, rowb, accb, tmp, misc, geo, y0, j, act, total, ncomp)) {
        if (tid == 0) slow_flag[n] = 16u + (u32)why;
        return;
    }
    if (geo.stop && geo.stop < 20) return;

    // ---- D (open) 0: the component's first pixel = start of its root run (the moments' origin), from the root list ---
    u32* anchor = reinterpret_cast<u32*>(accb + CCL_MOM_COMPS * NMOM * 8);               // [CCL_OPEN_COMPS]  (y << 16) | x
    const u32* roots = reinterpret_cast<const u32*>(accb + CCL_MOM_COMPS * NMOM * 8 + CCL_OPEN_COMPS * 4);
    u32* first = area_first + (int64_t)n * maxm;
    for (int i = tid; i < misc[4]; i += ST_NT) {
        const u32 v = P[roots[2 * i]];
        if (v & 0x8000u) {
            const u32 pos = roots[2 * i + 1], py = pos / (u32)W;
            anchor[v & 0x7FFFu] = (py << 16) | (pos - py * (u32)W);
            first[v & 0x7FFFu] = pos;
        }
    }
    // this segment's probe requests: which of its rows have any
    u32 rowmask = 0;
    const u32 nreq = min(mb_cnt[tid], (u32)ST_MB_CAP);
    for (u32 q = 0; q < nreq; ++q) rowmask |= 1u << ((mb_req[tid * ST_MB_CAP + q] >> 12) & 31u);
    unsigned short* pr = probe_all + (int64_t)n * maxm * 4;
    u64* acc = reinterpret_cast<u64*>(accb);
    i64* as = area_sums + (int64_t)n * maxm * VBS_AREA_SUMS;
    // ---- D (open) 1: contour-vertex moments, 256 components per pass ---------------------------------------------
    for (u32 c0 = 0; c0 < ncomp; c0 += CCL_MOM_COMPS) {
        const u32 nc = min((u32)CCL_MOM_COMPS, ncomp - c0);
        for (u32 c = tid; c < nc * NMOM; c += ST_NT) acc[c] = 0;
        __syncthreads();
        MomEntry e0, e1;
        e0.cid = NONE32; e1.cid = NONE32;
#pragma unroll
        for (int q = 0; q < 10; ++q) { e0.m[q] = 0; e1.m[q] = 0; }
#pragma unroll
        for (int t = 1; t <= R; ++t) {
            const u64 B = Bw[t];
            if (!__any(B != 0ull || ((rowmask >> t) & 1u))) continue;
            const u32 pB = (lm >> t) & 1u;
            const u64 stB = ccl_starts(B, pB);
            const u32 bc = WB(t);
            const int y = y0 + t - 1;
            if (c0 == 0 && ((rowmask >> t) & 1u)) {      // probes: component id of a pixel of this row
                for (u32 q = 0; q < nreq; ++q) {
                    const u32 rq = mb_req[tid * ST_MB_CAP + q];
                    if (((rq >> 12) & 31u) != (u32)t) continue;
                    const u32 q0 = (rq >> 10) & 3u;
                    for (u32 d = 0; d <= ((rq >> 23) & 1u); ++d) {
                        const u32 k = ((rq >> 17) & 63u) + d;
                        u32 cid = NONE16;
                        if ((B >> k) & 1ull) {
                            const u64 below = (k == 63) ? ~0ull : ((1ull << (k + 1)) - 1ull);
                            cid = P[bc + (u32)__popcll(stB & below) - 1u] & 0x7FFFu;
                        }
                        pr[(rq & 1023u) * 4 + q0 + d] = (unsigned short)cid;
                    }
                }
            }
            if (!B) continue;
            // the eight neighbour planes: bit k = the neighbour of pixel k in chain direction d is foreground
            const u64 An = Bw[t - 1], Sn = Bw[t + 1];
            const u64 D0 = (B >> 1) | ((u64)((rm >> t) & 1u) << 63), D4 = (B << 1) | (u64)pB;
            const u64 D2 = An, D1 = (An >> 1) | ((u64)((rm >> (t - 1)) & 1u) << 63), D3 = (An << 1) | (u64)((lm >> (t - 1)) & 1u);
            const u64 D6 = Sn, D7 = (Sn >> 1) | ((u64)((rm >> (t + 1)) & 1u) << 63), D5 = (Sn << 1) | (u64)((lm >> (t + 1)) & 1u);
            // a vertex per maximal arc of background neighbours that starts at direction a (a background, a - 1 foreground),
            // holds a 4-neighbour and is not exactly {a, a + 1, a + 2} (then the border passes straight through)
#define KEPT_EVEN(Da, Dm1, Dp1, Dp2, Dp3) (~(Da) & (Dm1) & ((Dp1) | (Dp2) | ~(Dp3)))
#define KEPT_ODD(Da, Dm1, Dp1, Dp2, Dp3) (~(Da) & (Dm1) & ~(Dp1) & ((Dp2) | ~(Dp3)))
            const u64 k0 = KEPT_EVEN(D0, D7, D1, D2, D3), k1 = KEPT_ODD(D1, D0, D2, D3, D4);
            const u64 k2 = KEPT_EVEN(D2, D1, D3, D4, D5), k3 = KEPT_ODD(D3, D2, D4, D5, D6);
            const u64 k4 = KEPT_EVEN(D4, D3, D5, D6, D7), k5 = KEPT_ODD(D5, D4, D6, D7, D0);
            const u64 k6 = KEPT_EVEN(D6, D5, D7, D0, D1), k7 = KEPT_ODD(D7, D6, D0, D1, D2);
#undef KEPT_EVEN
#undef KEPT_ODD
            const u64 iso = ~(D0 | D1 | D2 | D3 | D4 | D5 | D6 | D7);    // an isolated pixel is written once
            // V1 / V2 / V3: at least one / two / three of the nine planes
            u64 V1 = k0, V2 = 0, V3 = 0;
#define ADDP(K) { V3 |= V2 & (K); V2 |= V1 & (K); V1 |= (K); }
            ADDP(k1) ADDP(k2) ADDP(k3) ADDP(k4) ADDP(k5) ADDP(k6) ADDP(k7) ADDP(iso)
#undef ADDP
            V1 &= B; V2 &= B; V3 &= B;
            if (V3) misc[6] = 1;                         // multiplicity > 2: impossible after a 5x5 opening; general path
            if (!V1) continue;
            bool firstrun = true;
            u64 mB = B;
            while (mB) {
                const u64 lowbit = mB & (~mB + 1ull);
                const u64 t2 = mB + lowbit;
                const u64 gg = mB & ~t2;
                mB &= t2;
                u64 vg = gg & V1;
                if (!vg) continue;
                const u32 cid = (P[bc + (u32)__popcll(stB & ((lowbit << 1) - 1ull)) - 1u] & 0x7FFFu) - c0;
                if (cid >= nc) continue;                 // another pass's component
                const u32 fp = anchor[cid + c0];
                const int dy = y - (int)(fp >> 16), dx0 = 64 * j - (int)(fp & 0xFFFFu);
                u64* a = acc + cid * NMOM;
                if (dx0 >= -150 && dx0 + 63 <= 150 && abs(dy) <= 150) {
                    // row sums s_a = sum mult dx^a (24-bit multiplies: every factor < 2^23, every product < 2^31)
                    int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                    i64 s4 = 0;
                    while (vg) {
                        const int k = __ffsll((long long)vg) - 1;
                        vg &= vg - 1;
                        const int dx = dx0 + k, sh = (int)((V2 >> k) & 1ull);
                        const int mx = sh ? 2 * dx : dx, x2 = __mul24(dx, dx), mxx = __mul24(mx, dx);
                        s0 += 1 << sh; s1 += mx; s2 += mxx; s3 += __mul24(mx, x2);
                        s4 += (i64)__mul24(mxx, x2);
                    }
                    const int dy2 = __mul24(dy, dy), dy3 = __mul24(dy2, dy);
                    auto apply = [&](MomEntry& e) {
                        if (e.cid != cid || e.m[0] > 256) { mom_flush(e, acc); e.cid = cid; }
                        e.m[0] += s0;                  e.m[1] += s1;                  e.m[2] += __mul24(s0, dy);
                        e.m[3] += s2;                  e.m[4] += __mul24(s1, dy);     e.m[5] += __mul24(s0, dy2);
                        e.m[6] += s3;                  e.m[7] += __mul24(s2, dy);     e.m[8] += __mul24(s1, dy2);
                        e.m[9] += __mul24(s0, dy3);
                    };
                    if (firstrun) apply(e0); else apply(e1);
                    atomicAdd(&a[10], (u64)s4);
                    if (dy) {
                        atomicAdd(&a[11], (u64)((i64)s3 * dy));
                        atomicAdd(&a[12], (u64)((i64)s2 * dy2));
                        atomicAdd(&a[13], (u64)((i64)s1 * dy3));
                        atomicAdd(&a[14], (u64)((i64)__mul24(s0, dy2) * dy2));
                    }
                } else {
                    while (vg) {                         // a large component: 64-bit terms, one vertex at a time
                        const int k = __ffsll((long long)vg) - 1;
                        vg &= vg - 1;
                        const i64 ml_ = 1 + (i64)((V2 >> k) & 1ull), dl_ = dx0 + k, el_ = dy, x2 = dl_ * dl_, y2 = el_ * el_;
                        atomicAdd(&a[0], (u64)ml_);
                        atomicAdd(&a[1], (u64)(ml_ * dl_));             atomicAdd(&a[2], (u64)(ml_ * el_));
                        atomicAdd(&a[3], (u64)(ml_ * x2));              atomicAdd(&a[4], (u64)(ml_ * dl_ * el_));
                        atomicAdd(&a[5], (u64)(ml_ * y2));              atomicAdd(&a[6], (u64)(ml_ * x2 * dl_));
                        atomicAdd(&a[7], (u64)(ml_ * x2 * el_));        atomicAdd(&a[8], (u64)(ml_ * dl_ * y2));
                        atomicAdd(&a[9], (u64)(ml_ * y2 * el_));        atomicAdd(&a[10], (u64)(ml_ * x2 * x2));
                        atomicAdd(&a[11], (u64)(ml_ * x2 * dl_ * el_)); atomicAdd(&a[12], (u64)(ml_ * x2 * y2));
                        atomicAdd(&a[13], (u64)(ml_ * dl_ * el_ * y2)); atomicAdd(&a[14], (u64)(ml_ * y2 * y2));
                    }
                }
                firstrun = false;
            }
        }
        mom_flush(e0, acc);
        mom_flush(e1, acc);
        __syncthreads();
        for (u32 c = tid; c < nc * NMOM; c += ST_NT) as[(c0 + c / NMOM) * VBS_AREA_SUMS + (c % NMOM)] = (i64)acc[c];
        __syncthreads();
    }
    if (misc[6]) {                                       // (set before the last barrier above by whoever saw it)
        if (tid == 0) slow_flag[n] = 9;
        return;
    }
    if (tid == 0) { ncomp_all[n * 2 + 1] = ncomp; fstat[n * 8 + 6] = ncomp; fstat[n * 8 + 4] = 0; }
}

// ---- host side ----------------------------------------------------------------------------------------------------
// rows per segment the kernel is compiled for; 0 = geometry outside the fused path (the round-2 kernels take it)
static int stage_rows(const vbs_handle* h, StageGeom* g, size_t* lds_bytes) {
    if (h->WW > 32) return 0;
    const int G = 64 / h->WW, NB = 16 * G;
    const int need = (h->H + NB - 1) / NB;
    const int R = need <= 6 ? 6 : need <= 12 ? 12 : need <= 22 ? 22 : 0;
    if (!R || h->W > 4096 || h->H > 2048 || h->maxm > 1024) return 0;
    g->H = h->H; g->W = h->W; g->WW = h->WW; g->G = G; g->NB = NB; g->maxm = h->maxm;
    g->stop = VBS_KNOB("VBS_STAGE_STOP");
    const size_t rowb = ((size_t)(NB * R + 2) * 2 + 15) / 16 * 16;
    const size_t acc_band = ((size_t)(8 * ((h->maxm + 1) / 2)) + 16 * (size_t)h->maxm + 15) / 16 * 16;
    const size_t acc_open = (size_t)CCL_MOM_COMPS * NMOM * 8 + (size_t)CCL_OPEN_COMPS * 4 + (size_t)CCL_ROOT_LIST * 8;
    const size_t acc = acc_band > acc_open ? acc_band : acc_open;
    const size_t mb = (size_t)ST_NT * 4 * (1 + ST_MB_CAP);
    const size_t misc = 32 * 4 + 16 * 4;
    const size_t fixed = rowb + acc + mb + misc;
    const size_t full = 160 * 1024;
    if (fixed + 4096 > full) return 0;
    size_t cap = (full - fixed) / 2 / 8 * 8;
    if (cap > CCL_NODE_MAX) cap = CCL_NODE_MAX / 8 * 8;
    const size_t want = (size_t)h->H * h->W / 24;        // far above a marker frame's runs (1280x1024: band 13.9 k)
    if (cap > want && want >= 4096) cap = want / 8 * 8;
    g->node_cap = (u32)cap;
    g->off_rowb = (u32)(2 * cap);
    g->off_acc = (u32)(g->off_rowb + rowb);
    g->off_mb = (u32)(g->off_acc + acc);
    g->off_tmp = (u32)(g->off_mb + mb);
    g->pair_cap = (u32)((size_t)CCL_MOM_COMPS * NMOM * 8 / 4);     // the list borrows the moment accumulators' area
    *lds_bytes = g->off_tmp + misc;
    return R;
}

bool stage_supported(const vbs_handle* h) {
    StageGeom g;
    size_t lds;
    return stage_rows(h, &g, &lds) != 0;
}

template <int R, int NS>
static bool stage_launch_t(vbs_handle* h, int nb, const StageGeom& g, size_t lds, hipStream_t s) {
    if (lds > h->stage_lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_stage<R, NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        h->stage_lds_set = lds;
    }
    VBS_LAUNCH(h, s, "k_stage", (k_stage<R, NS>), dim3(nb), dim3(ST_NT), lds, s, h->mask_bits, h->area_bits, h->ncomp,
               h->band_sums, h->area_first, h->area_sums, h->probe, h->fstat, h->slow_flag, g);
    return true;
}

// false: geometry outside the fused path (or the LDS it needs was refused): the caller runs the round-2 kernels
bool launch_stage(vbs_handle* h, int nb, hipStream_t s) {
    StageGeom g;
    size_t lds = 0;
    const int R = stage_rows(h, &g, &lds);
    if (!R) return false;
    const bool ns14 = h->bp.ns == 14;
    switch (R) {
        case 6: return ns14 ? stage_launch_t<6, 14>(h, nb, g, lds, s) : stage_launch_t<6, 8>(h, nb, g, lds, s);
        case 12: return ns14 ? stage_launch_t<12, 14>(h, nb, g, lds, s) : stage_launch_t<12, 8>(h, nb, g, lds, s);
        default: return ns14 ? stage_launch_t<22, 14>(h, nb, g, lds, s) : stage_launch_t<22, 8>(h, nb, g, lds, s);
    }
}
