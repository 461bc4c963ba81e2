// Union-find over runs of 1-bits in LDS (uint16 parents), shared by the labelling kernels k_ccl.hip (one work item per
// chunk of a row, words from memory) and k_stage.hip (one thread per column segment, words in registers): find / hook /
// union, the block prefix sum, and the links of one 64-px word to the row above.  See k_ccl.hip for the phases.
#pragma once
#include "common.h"

#define CCL_NT 1024
#define CCL_NODE_MAX 32767         // node indices are 15-bit (bit 15 of a resolved entry marks the root)
#define CCL_MOM_COMPS 256          // components per moment pass (15 x 8 B x 256 = 30 KB of accumulators)
#define CCL_OPEN_COMPS 512         // contour components (k_finalize's limit)
#define CCL_ROOT_LIST 1024         // runs without a run above them (root candidates) whose position is remembered
#define NMOM 15
#define NONE16 0xFFFFu

__device__ __forceinline__ u64 valid_mask(int j, int W) {
    int rem = W - 64 * j;
    if (rem >= 64) return ~0ull;
    if (rem <= 0) return 0ull;
    return (1ull << rem) - 1ull;
}

// ---- wave64 cross-lane primitives on DPP (gfx9: wave_shr/wave_shl/row_shr/row_bcast), a few cycles each;
//      __shfl_* would go through ds_bpermute and its LDS-crossbar latency on every step of a row ----------
__device__ __forceinline__ u32 dpp_shr1(u32 x) {        // lane i <- lane i-1, lane 0 <- 0
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ u32 dpp_shl1(u32 x) {        // lane i <- lane i+1, lane 63 <- 0
    return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ u64 dpp_shr1(u64 x) {
    return ((u64)dpp_shr1((u32)(x >> 32)) << 32) | dpp_shr1((u32)x);
}
__device__ __forceinline__ u64 dpp_shl1(u64 x) {
    return ((u64)dpp_shl1((u32)(x >> 32)) << 32) | dpp_shl1((u32)x);
}
__device__ __forceinline__ u32 ccl_find(volatile unsigned short* P, u32 x) {
    for (;;) {
        const u32 p = P[x];
        if (p == x) return x;
        const u32 gp = P[p];
        if (gp == p) return p;
        P[x] = (unsigned short)gp;                      // path halving (only ever towards a smaller member of the set)
        x = gp;
    }
}

// parent[a] = min(parent[a], b) on the packed table; returns the previous parent[a]
__device__ __forceinline__ u32 ccl_hook(unsigned short* P, u32 a, u32 b) {
    u32* wp = reinterpret_cast<u32*>(P) + (a >> 1);
    const u32 sh = (a & 1u) * 16u;
    u32 old = *(volatile u32*)wp;
    for (;;) {
        const u32 cur = (old >> sh) & 0xFFFFu;
        if (cur <= b) return cur;
        const u32 nw = (old & ~(0xFFFFu << sh)) | (b << sh);
        const u32 prev = atomicCAS(wp, old, nw);
        if (prev == old) return cur;
        old = prev;
    }
}

__device__ __forceinline__ void ccl_union(unsigned short* P, u32 a, u32 b) {
    for (;;) {
        a = ccl_find(P, a);
        b = ccl_find(P, b);
        if (a == b) return;
        if (a < b) { const u32 t = a; a = b; b = t; }
        const u32 old = ccl_hook(P, a, b);
        if (old == a) return;
        a = old;                                        // a had been hooked meanwhile: carry on from its parent
    }
}

// exclusive prefix sum over the CCL_NT threads; tmp holds >= 17 words
__device__ __forceinline__ u32 ccl_scan(u32 v, u32* tmp, u32* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const u32 t = lane < CCL_NT / 64 ? tmp[lane] : 0u;
        u32 ti = t;
#pragma unroll
        for (int d = 1; d < CCL_NT / 64; d <<= 1) {
            const u32 o = __shfl_up(ti, d);
            if (lane >= d) ti += o;
        }
        if (lane < CCL_NT / 64) tmp[lane] = ti - t;
        if (lane == CCL_NT / 64 - 1) tmp[16] = ti;
    }
    __syncthreads();
    const u32 ex = inc - v + tmp[wave];
    *total = tmp[16];
    __syncthreads();
    return ex;
}

// runs that START in word B: p = bit 63 of the word to its left (0 at the row start)
__device__ __forceinline__ u64 ccl_starts(u64 B, u32 p) { return B & ~((B << 1) | (u64)p); }


// Links of word B (row y, word j) to the row above.  pB / pA = bit 63 of the word to the left of B / of A (the word
// above B); aR = bit 0 of the word above-right.  bc / ba = nodes of row y / y - 1 before word j, so the node of the
// run of B that holds bit k is bc + (starts of B at or below k) - 1, which is bc - 1 for a run that came in from the
// left.  The links of a run, in ascending node order: above-left diagonal (8-connectivity), the runs of A it touches,
// above-right diagonal.  A run that came in from the left skips its links at bit 0 when the run above came in from the
// left too: the two touch one column earlier and were linked there.
//   PASS 0: the parent of every run that STARTS in this word = its first link (itself if it has none); every further
//           link goes to the pair list (node << 16 | other node) for the dense union pass; when the list is full the
//           word is flagged instead (returns true) and PASS 1 walks it again; runs without a link (root candidates)
//           go to the root list
//   PASS 1: the further links -> ccl_union (only for chunks flagged in pass 0)
struct CclLists {
    u32* pairs; int* npairs; int pair_cap;               // extra links
    u32* roots; int* nroots;                             // root candidates (node, position), opened mask only
};

template <int M8, int PASS>
__device__ __forceinline__ bool ccl_link_word(unsigned short* P, u64 B, u64 A, u32 pB, u32 pA, u32 aR, u32 bc, u32 ba,
                                              u32 pos0, const CclLists& L) {
    u64 adj = A;
    if (M8) adj |= (A << 1) | (A >> 1) | (u64)pA | ((u64)aR << 63);
    const u64 stB = ccl_starts(B, pB);
    if (PASS == 1 && !(B & adj)) return false;
    const u64 stA = ccl_starts(A, pA);
    bool overflow = false;
    auto further = [&](u32 node, u32 other) {            // a link beyond the first
        if (PASS == 1) { ccl_union(P, node, other); return; }
        const int k = atomicAdd(L.npairs, 1);
        if (k < L.pair_cap) L.pairs[k] = (node << 16) | other; else overflow = true;
    };
    u64 mB = B;
    while (mB) {
        const u64 lowbit = mB & (~mB + 1ull);
        const u64 t = mB + lowbit;
        const u64 g = mB & ~t;                          // one run of B (its part inside this word)
        mB &= t;
        const bool starts = (stB & lowbit) != 0;
        const u32 node = bc + (u32)__popcll(stB & ((lowbit << 1) - 1ull)) - 1u;
        bool have = !starts;                            // a run that came in from the left got its parent where it starts
        u32 par = node;
        if (g & adj) {
            u64 rm = g;
            if (M8) rm |= (g << 1) | (g >> 1);
            if (M8 && starts && (g & 1ull) && pA) { par = ba - 1u; have = true; }       // (a run from the left: linked earlier)
            u64 mA = A & rm;
            if (!starts && pA) mA &= ~(A & ~(A + 1ull));    // drop the run of A at bit 0: it came in from the left as well
            while (mA) {                                // the runs of A under it
                const u64 lb = mA & (~mA + 1ull);
                const u64 t2 = mA + lb;
                mA &= t2;
                const u32 na = ba + (u32)__popcll(stA & ((lb << 1) - 1ull)) - 1u;
                if (!have) { par = na; have = true; } else further(node, na);
            }
            if (M8 && (g >> 63) && aR) {
                const u32 na = ba + (u32)__popcll(stA) - ((A >> 63) ? 1u : 0u);
                if (!have) { par = na; have = true; } else further(node, na);
            }
        }
        if (PASS == 0 && starts) {
            P[node] = (unsigned short)par;
            if (par == node && L.roots) {               // no run above: a root candidate, remember where it starts
                const int k = atomicAdd(L.nroots, 1);
                if (k < CCL_ROOT_LIST) { L.roots[2 * k] = node; L.roots[2 * k + 1] = pos0 + (u32)(__ffsll((long long)g) - 1); }
            }
        }
    }
    return overflow;
}

