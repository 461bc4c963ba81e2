// Device helpers shared by the two fused threshold+CCL kernels: k_stage.hip (one workgroup per frame, the batch path) and
// k_stage_lat.hip (a frame spread over several workgroups, the few-frames-per-call path).  See k_stage.hip for the method.
#pragma once
#include "ccl_common.h"

#define ST_MB_CAP 8                // probe requests a thread can hold (one per centroid, row and word)
#define SG_KB 4                    // slots of the band walk (a ring crosses a tile as two arcs)
#define SG_KO 3                    // slots of the opened-mask walk
#define SG_SEGMAX 8                // segments a thread can start; segment id = 8 tid + i
#define SG_REC 2048                // segment records per frame, at most (StageGeom::rec_cap)
#define SG_PQ 4096                 // segment pairs waiting to be united, at most (StageGeom::pq_cap)
#define NONE32 0xFFFFFFFFu
// why a frame was handed on (slow_flag value)
#define SLOW_SLOTS 1               // a tile needed more slots / segments / records than there are
#define SLOW_NCOMP 2               // more components than max_markers (band: 1024, open: 512)
#define SLOW_MAILBOX 3
#define SLOW_HOLES 4
#define SLOW_VERTEX 5              // a contour vertex of multiplicity > 2


__device__ __forceinline__ u64 mk64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
__device__ __forceinline__ u64 brev64(u64 x) {
    return mk64(__builtin_bitreverse32((u32)(x >> 32)), __builtin_bitreverse32((u32)x));
}

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

// the runs of B that hold a bit of S (S a subset of B); rB = brev64(B).  Adding S to B carries from the lowest seed of
// every run to the run's top; the same on the reversed words fills from the highest seed down.
__device__ __forceinline__ u64 fill_runs(u64 B, u64 rB, u64 S) {
    const u64 up = ((S + B) ^ B) & B;
    const u64 rS = brev64(S);
    const u64 dn = brev64(((rS + rB) ^ rB) & rB);
    return up | dn | S;
}

// sum of the positions of the set bits
__device__ __forceinline__ u32 sum_bitpos(u64 x) {
    const u32 lo = (u32)x, hi = (u32)(x >> 32);
    u32 s = (u32)__popc(lo & 0xAAAAAAAAu) + (u32)__popc(hi & 0xAAAAAAAAu);
    s += 2u * ((u32)__popc(lo & 0xCCCCCCCCu) + (u32)__popc(hi & 0xCCCCCCCCu));
    s += 4u * ((u32)__popc(lo & 0xF0F0F0F0u) + (u32)__popc(hi & 0xF0F0F0F0u));
    s += 8u * ((u32)__popc(lo & 0xFF00FF00u) + (u32)__popc(hi & 0xFF00FF00u));
    s += 16u * ((u32)__popc(lo & 0xFFFF0000u) + (u32)__popc(hi & 0xFFFF0000u));
    s += 32u * (u32)__popc(hi);
    return s;
}

// Segments found to belong together are only NOTED during a walk (a pair in an LDS queue, a handful of instructions where
// it happens); the unions run densely, one pair per thread, once the walk is over.
struct PairQ { u32* q; int* n; int cap; };
__device__ __forceinline__ void pq_push(const PairQ& Q, u32 a, u32 b) {
    const int i = atomicAdd(Q.n, 1);
    if (i < Q.cap) Q.q[i] = (a << 16) | b;               // (an overflowing queue hands the frame on: checked after the walk)
}

// One row of a labelling walk: Rn[k] = the runs of B that continue the segment in slot k (pm[k] = its pixels in the row
// above).  A run two segments reach stays with the lower slot and the two are united.  Returns the pixels given out.
template <int K, bool C8, bool UNITE>
__device__ __forceinline__ u64 seg_update(u64 B, u64 rB, const u64 (&pm)[K], const u32 (&sid)[K], u64 (&Rn)[K],
                                          const PairQ& Q) {
    u64 claimed = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        Rn[k] = 0;
        if (k >= 2 && !__any(pm[k] != 0ull)) continue;   // (wave-uniform; slots 0 and 1 always run: their chains interleave)
        u64 adj = pm[k];
        if (C8) adj |= (adj << 1) | (adj >> 1);
        u64 Rk = fill_runs(B, rB, B & adj);
        const u64 ov = Rk & claimed;
        if (ov) {
            if (UNITE) {
#pragma unroll
                for (int m = 0; m < K; ++m)
                    if (m < k && (Rn[m] & ov)) pq_push(Q, sid[k], sid[m]);
            }
            Rk &= ~claimed;
        }
        claimed |= Rk;
        Rn[k] = Rk;
    }
    return claimed;
}

// links across the right edge of the word: the segment holding bit 63 of this row with the one holding bit 0 of the word
// to the right (and, 8-connectivity, with bit 0 of its previous row; bit 63 of my previous row with its bit 0).
// Executed by every lane (DPP).  p63 / prs0: the values of the previous row; la / lb: the last pair united.
template <int K, bool C8>
__device__ __forceinline__ void seg_hlinks(const u64 (&pm)[K], const u32 (&sid)[K], bool hasr, u32& p63, u32& prs0, u32& la,
                                           u32& lb, const PairQ& Q) {
    u32 s63 = NONE16, s0 = NONE16;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (pm[k] >> 63) s63 = sid[k];
        if (pm[k] & 1ull) s0 = sid[k];
    }
    u32 rs0 = dpp_shl1(s0);
    if (!hasr) rs0 = NONE16;
    auto link = [&](u32 a, u32 b) {
        if (a != NONE16 && b != NONE16 && (a != la || b != lb)) { pq_push(Q, a, b); la = a; lb = b; }
    };
    link(s63, rs0);
    if (C8) { link(s63, prs0); link(p63, rs0); }
    p63 = s63; prs0 = rs0;
}

// exact shift of the moments m[a][b] (a + b <= 4, about the point o) to the point o - (dx, dy): sum (x + dx)^a (y + dy)^b
__device__ __forceinline__ void shift_moments_i64(const i64 (&m)[NMOM], i64 dx, i64 dy, i64 (&out)[NMOM]) {
    // index of (a, b): 0:(0,0) 1:(1,0) 2:(0,1) 3:(2,0) 4:(1,1) 5:(0,2) 6:(3,0) 7:(2,1) 8:(1,2) 9:(0,3) 10:(4,0) 11:(3,1) 12:(2,2) 13:(1,3) 14:(0,4)
    const i64 dx2 = dx * dx, dx3 = dx2 * dx, dx4 = dx2 * dx2, dy2 = dy * dy, dy3 = dy2 * dy, dy4 = dy2 * dy2;
    // x first: T[a][k] = sum_i C(a,i) dx^(a-i) m[i][k]
    const i64 T00 = m[0], T01 = m[2], T02 = m[5], T03 = m[9], T04 = m[14];
    const i64 T10 = m[1] + dx * m[0], T11 = m[4] + dx * m[2], T12 = m[8] + dx * m[5], T13 = m[13] + dx * m[9];
    const i64 T20 = m[3] + 2 * dx * m[1] + dx2 * m[0], T21 = m[7] + 2 * dx * m[4] + dx2 * m[2], T22 = m[12] + 2 * dx * m[8] + dx2 * m[5];
    const i64 T30 = m[6] + 3 * dx * m[3] + 3 * dx2 * m[1] + dx3 * m[0], T31 = m[11] + 3 * dx * m[7] + 3 * dx2 * m[4] + dx3 * m[2];
    const i64 T40 = m[10] + 4 * dx * m[6] + 6 * dx2 * m[3] + 4 * dx3 * m[1] + dx4 * m[0];
    // then y: out[a][b] = sum_k C(b,k) dy^(b-k) T[a][k]
    out[0] = T00;
    out[1] = T10;
    out[2] = T01 + dy * T00;
    out[3] = T20;
    out[4] = T11 + dy * T10;
    out[5] = T02 + 2 * dy * T01 + dy2 * T00;
    out[6] = T30;
    out[7] = T21 + dy * T20;
    out[8] = T12 + 2 * dy * T11 + dy2 * T10;
    out[9] = T03 + 3 * dy * T02 + 3 * dy2 * T01 + dy3 * T00;
    out[10] = T40;
    out[11] = T31 + dy * T30;
    out[12] = T22 + 2 * dy * T21 + dy2 * T20;
    out[13] = T13 + 3 * dy * T12 + 3 * dy2 * T11 + dy3 * T10;
    out[14] = T04 + 4 * dy * T03 + 6 * dy2 * T02 + 4 * dy3 * T01 + dy4 * T00;
}

