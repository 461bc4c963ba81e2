// Device helpers shared by the two fused threshold+CCL kernels: k_stage.hip (one workgroup per frame, the batch path) and
// k_stage_lat.hip (a frame spread over several workgroups, the few-frames-per-call path).  See k_stage.hip for the method.
#pragma once
#include "ccl_common.h"

#define ST_MB_CAP 8                // probe requests a thread can hold (one per centroid, row and word)
#define SG_KB 4                    // slots of the band walk (a ring crosses a tile as two arcs)
#define SG_KO 3                    // slots of the opened-mask walk
#define SG_SEGMAX 8                // segments a thread can start; segment id = 8 tid + i
// (SG_REC, the segment records per frame: common.h - vbs_create sizes a buffer by it)
#define SG_PQ 4096                 // segment pairs waiting to be united, at most (StageGeom::pq_cap)
#define NONE32 0xFFFFFFFFu
// why a frame was handed on (slow_flag value)
#define SLOW_SLOTS 1               // a tile needed more slots / segments / records than there are
#define SLOW_NCOMP 2               // more components than max_markers (band: 1024, open: 512)
#define SLOW_MAILBOX 3
#define SLOW_HOLES 4
#define SLOW_VERTEX 5              // a contour vertex of multiplicity > 2
#define SLOW_SEGS 6                // a tile started more than SG_SEGMAX segments
#define SLOW_RECS 7                // more segment / moment records than the frame's tables hold
#define SLOW_PAIRS 8               // more queued unions than the queue holds


__device__ __forceinline__ u64 mk64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
__device__ __forceinline__ u64 brev64(u64 x) {
    return mk64(__builtin_bitreverse32((u32)(x >> 32)), __builtin_bitreverse32((u32)x));
}

// a & ~b & c and a | ~b on 32-bit halves (one v_bitop3_b32 per half: see vertex_planes32)
__device__ __forceinline__ u64 and_not_and(u64 a, u64 b, u64 c) {
    return mk64((u32)a & ~(u32)b & (u32)c, (u32)(a >> 32) & ~(u32)(b >> 32) & (u32)(c >> 32));
}
__device__ __forceinline__ u64 or_not(u64 a, u64 b) { return mk64((u32)a | ~(u32)b, (u32)(a >> 32) | ~(u32)(b >> 32)); }

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
    // (the boolean parts on 32-bit halves: one v_bitop3_b32 each, see vertex_planes32)
    const u64 su = S + B;
    const u32 upl = (u32)B & ~(u32)su, uph = (u32)(B >> 32) & ~(u32)(su >> 32);          // ((S + B) ^ B) & B
    const u64 rS = brev64(S), sd = rS + rB;
    const u32 dl = (u32)rB & ~(u32)sd, dh = (u32)(rB >> 32) & ~(u32)(sd >> 32);           // reversed: lo half = bits 63 .. 32 of dn
    return mk64(upl | __builtin_bitreverse32(dh) | (u32)S, uph | __builtin_bitreverse32(dl) | (u32)(S >> 32));
}

// ---- the CHAIN_APPROX_SIMPLE vertex multiplicity of every pixel of a word, bit-parallel (see k_stage.hip) -------------------
// On 32-BIT HALVES on purpose: gfx950 has v_bitop3_b32 (any boolean function of three inputs in one instruction) and the
// compiler forms it from 32-bit expression trees only - 64-bit logic is split after instruction selection, when every
// and / or / not is already its own instruction (135 -> 96 vector instructions for this function, tools' vtx prototype).
// D0 .. D7: bit k = the neighbour of pixel k in chain direction d is foreground.  A vertex per maximal arc of background
// neighbours that starts at direction a (a background, a - 1 foreground), holds a 4-neighbour and is not exactly
// {a, a + 1, a + 2} (the border passes straight through); an isolated pixel is written once.  V1 / V2 / V3: at least one /
// two / three of the nine planes.
__device__ __forceinline__ void vertex_planes32(u32 B, u32 D0, u32 D1, u32 D2, u32 D3, u32 D4, u32 D5, u32 D6, u32 D7, u32& V1,
                                                u32& V2, u32& V3) {
#define KEPT_EVEN(Da, Dm1, Dp1, Dp2, Dp3) (~(Da) & (Dm1) & ((Dp1) | (Dp2) | ~(Dp3)))
#define KEPT_ODD(Da, Dm1, Dp1, Dp2, Dp3) (~(Da) & (Dm1) & ~(Dp1) & ((Dp2) | ~(Dp3)))
    const u32 k0 = KEPT_EVEN(D0, D7, D1, D2, D3), k1 = KEPT_ODD(D1, D0, D2, D3, D4);
    const u32 k2 = KEPT_EVEN(D2, D1, D3, D4, D5), k3 = KEPT_ODD(D3, D2, D4, D5, D6);
    const u32 k4 = KEPT_EVEN(D4, D3, D5, D6, D7), k5 = KEPT_ODD(D5, D4, D6, D7, D0);
    const u32 k6 = KEPT_EVEN(D6, D5, D7, D0, D1), k7 = KEPT_ODD(D7, D6, D0, D1, D2);
#undef KEPT_EVEN
#undef KEPT_ODD
    const u32 iso = ~(D0 | D1 | D2 | D3 | D4 | D5 | D6 | D7);
    V1 = k0; V2 = 0; V3 = 0;
#define ADDP(Kp) { V3 |= V2 & (Kp); V2 |= V1 & (Kp); V1 |= (Kp); }
    ADDP(k1) ADDP(k2) ADDP(k3) ADDP(k4) ADDP(k5) ADDP(k6) ADDP(k7) ADDP(iso)
#undef ADDP
    V1 &= B; V2 &= B; V3 &= B;
}
// B = row c of the opened plane, o2 / o0 = the rows above / below it, l? / r? = bit 63 of the word to the left / bit 0 of the
// word to the right in those rows (2: above, 1: this row, 0: below)
__device__ __forceinline__ void vertex_planes(u64 B, u64 o2, u64 o0, u32 l2, u32 l1, u32 l0, u32 r2, u32 r1, u32 r0, u64& V1, u64& V2,
                                              u64& V3) {
    const u32 Bl = (u32)B, Bh = (u32)(B >> 32), al = (u32)o2, ah = (u32)(o2 >> 32), bl = (u32)o0, bh = (u32)(o0 >> 32);
    // >> 1 with the right neighbour's bit 0 coming in at the top; << 1 with the left neighbour's bit 63 coming in at the bottom
    auto shr_lo = [](u32 lo, u32 hi) { return __builtin_amdgcn_alignbit(hi, lo, 1u); };
    auto shr_hi = [](u32 hi, u32 in) { return __builtin_amdgcn_alignbit(in, hi, 1u); };
    auto shl_lo = [](u32 lo, u32 in) { return (lo << 1) | in; };
    auto shl_hi = [](u32 lo, u32 hi) { return __builtin_amdgcn_alignbit(hi, lo, 31u); };
    u32 v1l, v2l, v3l, v1h, v2h, v3h;
    vertex_planes32(Bl, shr_lo(Bl, Bh), shr_lo(al, ah), al, shl_lo(al, l2), shl_lo(Bl, l1), shl_lo(bl, l0), bl, shr_lo(bl, bh), v1l, v2l,
                    v3l);
    vertex_planes32(Bh, shr_hi(Bh, r1), shr_hi(ah, r2), ah, shl_hi(al, ah), shl_hi(Bl, Bh), shl_hi(bl, bh), bh, shr_hi(bh, r0), v1h, v2h,
                    v3h);
    V1 = mk64(v1l, v1h); V2 = mk64(v2l, v2h); V3 = mk64(v3l, v3h);
}

// 4 x the Euler number's share of one word: the 2x2 windows whose top row is `a` and bottom row `bq` (an / bn: bit 0 of the
// words to their right), by bit quads - Q1 - Q3 - 2 QD (8-connected foreground).  On 32-bit halves (v_bitop3_b32, above).
__device__ __forceinline__ int euler_quads(u64 a, u64 bq, u32 an, u32 bn) {
    int e = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const u32 x = h ? (u32)(a >> 32) : (u32)a, y = h ? (u32)(bq >> 32) : (u32)bq;
        const u32 x1 = h ? __builtin_amdgcn_alignbit(an, x, 1u) : __builtin_amdgcn_alignbit((u32)(a >> 32), x, 1u);
        const u32 y1 = h ? __builtin_amdgcn_alignbit(bn, y, 1u) : __builtin_amdgcn_alignbit((u32)(bq >> 32), y, 1u);
        const u32 x2 = (x ^ x1) ^ (y ^ y1);
        const u32 pairs = (x & x1) | (x & y) | (x & y1) | (x1 & y) | (x1 & y1) | (y & y1);
        const u32 qd = (x & y1 & ~x1 & ~y) | (x1 & y & ~x & ~y1);
        e += __popc(x2 & ~pairs) - __popc(x2 & pairs) - 2 * __popc(qd);
    }
    return e;
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
        if (C8) {                                        // (halves: v_or3 / v_bitop3 per 32 bits)
            const u32 al = (u32)adj, ah = (u32)(adj >> 32);
            adj = mk64(al | (al << 1) | __builtin_amdgcn_alignbit(ah, al, 1u), ah | __builtin_amdgcn_alignbit(ah, al, 31u) | (ah >> 1));
        }
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

