// a9-a13 labelling, fast path (marker_detection.py:170-196): connected components of the bit-packed band mask
// (4-connectivity, ndimage.label :176) and of the opened area mask (8-connectivity, cv2.findContours :196) with the
// per-component sums k_finalize needs.
//
// One workgroup per (frame, mask); nothing walks the image sequentially.  The image is cut into work items of CW
// consecutive 64-px words of one row ("chunks"; lane l of a wave takes chunk l, so a wave reads consecutive bytes):
//   A  runs of 1s per chunk (a run that crosses a chunk boundary is cut there) -> block prefix sum: the nodes are
//      numbered in raster order, 16-bit, cbase[item] = index of the chunk's first node
//   B  union-find over the nodes in LDS (uint16 parents, two per dword).  A component of a marker frame is a stack of
//      runs, one or two per row, so hooking every run to the run above it with atomics builds chains as deep as the
//      component is tall and every find walks them.  Instead: (1) every run takes as parent the FIRST run it touches
//      in the row above (else the run it continues from the chunk to its left, else itself) - plain stores, a forest
//      whose roots are smaller than their members; (2) pointer jumping, parent = parent[parent], until nothing moves
//      (log2 of the tallest chain rounds); (3) only the remaining links (a run that touches a second run above, the
//      continuation links not used in 1) go through the atomic union (compare-and-swap on the dword, towards the
//      smaller index, path halving) - on trees that are flat by then
//   C  flatten; roots ranked in raster order = ndimage.label's numbering (reversed: cv2's contour order); the parent
//      table becomes the component id of every node (bit 15 marks the root = the component's first run)
//   D  band: pixel count / sum x / sum y per component (center_of_mass :181)
//      open: contour-vertex moments about the component's first pixel (see k_label.hip), LDS atomics; bit-quad Euler
//            number (holes); and for every band centroid the component ids of the 2x2 pixel cell around it ("probes"):
//            k_finalize's pointPolygonTest needs nothing else, so no label image or run table goes to HBM.
//      The accumulating walks of D visit the chunks in an order that puts rows 64 apart on neighbouring lanes: in
//      raster order 16 lanes of a wave sit on one marker and their atomics serialise on one address (measured: the
//      moment pass 0.33 -> see profiles/README.md).
// LDS is laid out per handle (ccl_layout): <= 80 KB and <= 64 VGPRs put two workgroups (32 waves) on a CU at
// 1280x1024; larger frames take up to the whole 160 KB (one workgroup per CU).  Frames outside the fast path's limits
// (more runs than the node table holds - at most 32767 -, more than 512 contours, holes in the opened mask) set their
// slow flag and are redone by the general kernels of k_label.hip, which also own the capacity status.
#include "common.h"

#define CCL_NT 1024
#define CCL_NODE_MAX 32767         // node indices are 15-bit (bit 15 of a resolved entry marks the root)
#define CCL_MOM_COMPS 256          // components per moment pass (15 x 8 B x 256 = 30 KB of accumulators)
#define CCL_OPEN_COMPS 512         // contour components (k_finalize's limit)
#define NMOM 15
#define NONE16 0xFFFFu

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

// runs of word B that start inside it: p = 1 when the word to its left IN THE SAME CHUNK ends with a 1 (a run that
// continues from there is not a new node)
__device__ __forceinline__ u64 ccl_starts(u64 B, u32 p) { return B & ~((B << 1) | (u64)p); }

// links of word B (row y, word j of its chunk) to the word above (A) and its diagonal neighbours.  pB / pA: carry-in
// bits of B / A (see ccl_starts); aL = bit 63 of the word above-left (any chunk), aR = bit 0 of the word above-right,
// aR_same = that word belongs to the same chunk; left = B's bit-0 run continues the run that ends the chunk to its left.
// bc / ba = nodes of row y / y-1 before word j.  The links of a run, in ascending node order: above-left diagonal, the
// runs of A it touches, above-right diagonal, the run to its left.
//   PASS 0: the parent of every run that STARTS in this word = its first link (itself if it has none)
//   PASS 1: every other link (all links of a segment that continues a run from the previous word) -> ccl_union
//   PASS 0 returns true when the word has such links (pass 1 skips the chunks that have none)
template <int M8, int PASS>
__device__ __forceinline__ bool ccl_link_word(unsigned short* P, u64 B, u64 A, u32 pB, u32 pA, u32 aL, u32 aR,
                                              bool aR_same, bool left, u32 bc, u32 ba) {
    u64 adj = A;
    if (M8) adj |= (A << 1) | (A >> 1) | (u64)aL | ((u64)aR << 63);
    const u64 stB = ccl_starts(B, pB);
    if (PASS == 1 && !(B & adj) && !left) return false;
    bool extra = false;
    const u64 stA = ccl_starts(A, pA);
    u64 mB = B;
    while (mB) {
        const u64 lowbit = mB & (~mB + 1ull);
        const u64 t = mB + lowbit;
        const u64 g = mB & ~t;                          // one run of B (its part inside this word)
        mB &= t;
        const bool starts = (stB & lowbit) != 0;
        const u32 node = bc + (u32)__popcll(stB & ((lowbit << 1) - 1ull)) - 1u;
        bool have = !starts;                            // a continuing segment's parent was set by its first segment
        u32 par = node;
        if (g & adj) {
            u64 rm = g;
            if (M8) rm |= (g << 1) | (g >> 1);
            if (M8 && (g & 1ull) && aL) {
                if (!have) { par = ba - 1u; have = true; } else if (PASS == 1) ccl_union(P, node, ba - 1u); else extra = true;
            }
            u64 mA = A & rm;
            while (mA) {                                // the runs of A under it
                const u64 lb = mA & (~mA + 1ull);
                const u64 t2 = mA + lb;
                mA &= t2;
                const u32 na = ba + (u32)__popcll(stA & ((lb << 1) - 1ull)) - 1u;
                if (!have) { par = na; have = true; } else if (PASS == 1) ccl_union(P, node, na); else { extra = true; break; }
            }
            if (M8 && (g >> 63) && aR) {
                const u32 na = ba + (u32)__popcll(stA) - (((A >> 63) && aR_same) ? 1u : 0u);
                if (!have) { par = na; have = true; } else if (PASS == 1) ccl_union(P, node, na); else extra = true;
            }
        }
        if (left && (g & 1ull)) {                       // (only the chunk's first word passes left = true)
            if (!have) { par = node - 1u; have = true; } else if (PASS == 1) ccl_union(P, node, node - 1u); else extra = true;
        }
        if (PASS == 0 && starts) P[node] = (unsigned short)par;
    }
    return extra;
}

struct CclGeom {
    int H, W, WW, CW, NC, items;
    u32 inv_nc;                                          // ceil(2^32 / NC): item / NC = umulhi(item, inv_nc); 0 when NC = 1
    int R, vitems;                                       // spread order (CCL_ITEM_DECODE_SPREAD): R = ceil(H / 64) row groups
    u32 inv_r;
    u32 node_cap;                                        // entries of the parent table
    u32 off_cbase, off_acc, off_tmp;                     // byte offsets into the dynamic LDS
    int stop;                                            // debug builds: leave after phase `stop` (0 = run everything)
};

// component id (or NONE16) of pixel (x, y) from the resolved parent table
__device__ __forceinline__ u32 ccl_pixel_cid(const u64* __restrict__ bits, const unsigned short* P,
                                             const unsigned short* cbase, const CclGeom& g, int x, int y) {
    if (x < 0 || y < 0 || x >= g.W || y >= g.H) return NONE16;
    const u64* row = bits + (int64_t)y * g.WW;
    const int jw = x >> 6, k = x & 63;
    const u64 w = row[jw];
    if (!((w >> k) & 1ull)) return NONE16;
    const int c = jw / g.CW, j0 = c * g.CW;
    u32 base = cbase[y * g.NC + c], p = 0;
    for (int jj = j0; jj < jw; ++jj) { const u64 ww = row[jj]; base += (u32)__popcll(ccl_starts(ww, p)); p = (u32)(ww >> 63); }
    const u64 below = (k == 63) ? ~0ull : ((1ull << (k + 1)) - 1ull);
    return P[base + (u32)__popcll(ccl_starts(w, p) & below) - 1u] & 0x7FFFu;
}

#define CCL_ITEM_DECODE                                                                            \
    const int y = geo.inv_nc ? (int)__umulhi((u32)it, geo.inv_nc) : it, c = it - y * NC;           \
    const int j0 = c * CW, j1 = min(j0 + CW, WW);
// The same chunks in an order that puts rows 64 apart on neighbouring lanes: the lanes of a wave then work on
// different components (a marker is < 64 rows tall), so their LDS atomics on per-component accumulators do not
// collide on one address (raster order puts 16 rows of one marker in a wave: 16-way serialised atomics).
#define CCL_ITEM_DECODE_SPREAD                                                                     \
    const int yv = geo.inv_nc ? (int)__umulhi((u32)vt, geo.inv_nc) : vt, c = vt - yv * NC;         \
    const int grp = geo.inv_r ? (int)__umulhi((u32)yv, geo.inv_r) : yv, y = (yv - grp * geo.R) * 64 + grp; \
    if (y >= H) continue;                                                                          \
    const int it = y * NC + c, j0 = c * CW, j1 = min(j0 + CW, WW);

template <int MODE>                                      // 0: band mask, 4-connectivity; 1: opened mask, 8-connectivity
__global__ __launch_bounds__(CCL_NT, 8) void k_ccl(const u64* __restrict__ bits_all, u32* __restrict__ ncomp_all,
                                                   u32* __restrict__ first_all, u64* __restrict__ band_sums,
                                                   i64* __restrict__ area_sums, unsigned short* __restrict__ probe_all,
                                                   u32* __restrict__ fstat, u32* __restrict__ slow_flag,
                                                   const u8* __restrict__ lut_g, CclGeom geo, int maxm) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned short* P = reinterpret_cast<unsigned short*>(smem);                         // [node_cap]
    unsigned short* cbase = reinterpret_cast<unsigned short*>(smem + geo.off_cbase);     // [items + 1]
    unsigned char* accb = smem + geo.off_acc;                                            // band sums | open moments + anchors
    u32* tmp = reinterpret_cast<u32*>(smem + geo.off_tmp);                               // [32]
    int* misc = reinterpret_cast<int*>(tmp + 32);                                        // [4]
    u8* lut = reinterpret_cast<u8*>(misc + 4);                                           // [256] (open)
    const int n = blockIdx.x, tid = threadIdx.x;
    const int H = geo.H, W = geo.W, WW = geo.WW, CW = geo.CW, NC = geo.NC, items = geo.items;
    if (slow_flag[n]) return;                            // already handed to the general path
    const u64* bits = bits_all + (int64_t)n * H * WW;
    if (MODE == 1 && tid < 256) lut[tid] = lut_g[tid];
    if (tid < 4) misc[tid] = 0;                          // [0] Euler sum, [1..3] pointer-jumping flags

    // ---- A: runs per chunk, numbered in raster order ---------------------------------------------------------------
    for (int it = tid; it < items; it += CCL_NT) {
        CCL_ITEM_DECODE
        const u64* row = bits + (int64_t)y * WW;
        u32 cnt = 0, p = 0;
        for (int j = j0; j < j1; ++j) { const u64 w = row[j]; cnt += (u32)__popcll(ccl_starts(w, p)); p = (u32)(w >> 63); }
        cbase[it] = (unsigned short)cnt;
    }
    __syncthreads();
    if (geo.stop == 1) return;
    const int K = (items + CCL_NT - 1) / CCL_NT;
    u32 total;
    {
        const int i0 = min(tid * K, items), i1 = min(i0 + K, items);
        u32 s = 0;
        for (int i = i0; i < i1; ++i) s += cbase[i];
        u32 ex = ccl_scan(s, tmp, &total);
        if (total <= geo.node_cap)
            for (int i = i0; i < i1; ++i) { const u32 c = cbase[i]; cbase[i] = (unsigned short)ex; ex += c; }
    }
    if (total > geo.node_cap) {                          // block-uniform
        if (tid == 0) slow_flag[n] = 1;
        return;
    }
    if (tid == 0) cbase[items] = (unsigned short)total;
    __syncthreads();
    if (geo.stop == 2) return;

    // ---- B: parents, pointer jumping, the remaining links --------------------------------------------------------
    u64 extra_mask = 0;                                  // bit k: this thread's k-th chunk has links left for pass 1
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        int kk = -1;
        for (int it = tid; it < items; it += CCL_NT) {
            ++kk;
            u32 bc = cbase[it];
            if (cbase[it + 1] == bc) continue;           // no run starts in this chunk (and none enters: cut at chunks)
            if (pass == 1 && kk < 64 && !((extra_mask >> kk) & 1ull)) continue;
            bool extra = false;
            CCL_ITEM_DECODE
            const u64* row = bits + (int64_t)y * WW;
            const u64* up = row - WW;
            const bool hasu = y > 0;
            u32 ba = hasu ? (u32)cbase[it - NC] : 0u;
            u32 aL = (hasu && j0) ? (u32)(up[j0 - 1] >> 63) : 0u;
            u64 A = hasu ? up[j0] : 0ull;
            bool left = j0 && (row[j0 - 1] >> 63);      // the chunk's first run may continue the chunk to its left
            u32 pB = 0, pA = 0;
            for (int j = j0; j < j1; ++j) {
                const u64 B = row[j];
                const u64 An = (hasu && j + 1 < WW) ? up[j + 1] : 0ull;
                if (B) {
                    if (pass == 0) extra |= ccl_link_word<MODE, 0>(P, B, A, pB, pA, aL, (u32)(An & 1ull), j + 1 < j1, left, bc, ba);
                    else ccl_link_word<MODE, 1>(P, B, A, pB, pA, aL, (u32)(An & 1ull), j + 1 < j1, left, bc, ba);
                }
                left = false;
                bc += (u32)__popcll(ccl_starts(B, pB));
                ba += (u32)__popcll(ccl_starts(A, pA));
                pB = (u32)(B >> 63);
                pA = aL = (u32)(A >> 63);
                A = An;
            }
            if (pass == 0 && (extra || kk >= 64)) extra_mask |= 1ull << (kk & 63);
        }
        __syncthreads();
        if (pass == 0) {
            if (geo.stop == 7) return;
            // pointer jumping: every node ends on the root of its tree (parents only ever move to an ancestor, so the
            // unsynchronised reads inside a round are harmless)
            // (three flags in turn: the one cleared in round r was last read before the barrier of round r - 1)
            for (int f = 0;; f = f == 2 ? 0 : f + 1) {
                if (tid == 0) misc[1 + (f == 2 ? 0 : f + 1)] = 0;
                bool ch = false;
                for (u32 i = tid; i < total; i += CCL_NT) {
                    const u32 p = P[i], pp = P[p];
                    if (pp != p) { P[i] = (unsigned short)pp; ch = true; }
                }
                if (ch) misc[1 + f] = 1;
                __syncthreads();
                if (!misc[1 + f]) break;
            }
            if (geo.stop == 8) return;
        }
    }
    if (geo.stop == 3) return;

    // ---- C: flatten, rank the roots in raster order, resolve every node to its component id ---------------------
    for (u32 i = tid; i < total; i += CCL_NT) {
        u32 x = i, p;
        while ((p = ((volatile unsigned short*)P)[x]) != x) x = p;
        if (x != i) P[i] = (unsigned short)x;
    }
    __syncthreads();
    const u32 K2 = (total + CCL_NT - 1) / CCL_NT;
    const u32 r0 = min((u32)tid * K2, total), r1 = min(r0 + K2, total);
    u32 nroot = 0;
    for (u32 i = r0; i < r1; ++i) nroot += (P[i] == i);
    u32 ncomp;
    u32 cid0 = ccl_scan(nroot, tmp, &ncomp);
    if (ncomp > (u32)maxm || ncomp > (MODE == 0 ? 1024u : (u32)CCL_OPEN_COMPS)) {       // block-uniform
        if (tid == 0) slow_flag[n] = 1;
        return;
    }
    for (u32 i = r0; i < r1; ++i)
        if (P[i] == i) P[i] = (unsigned short)(0x8000u | cid0++);
    __syncthreads();
    for (u32 i = tid; i < total; i += CCL_NT) {
        const u32 v = P[i];
        if (!(v & 0x8000u)) P[i] = (unsigned short)(P[v] & 0x7FFFu);
    }
    if (geo.stop == 4) return;

    u32* first = first_all + (int64_t)n * maxm;
    if (MODE == 0) {
        // ---- D (band): count, sum x, sum y -------------------------------------------------------------------------
        u32* acnt = reinterpret_cast<u32*>(accb);                                        // [maxm]
        u64* asx = reinterpret_cast<u64*>(accb + 8 * ((maxm + 1) / 2));                   // [maxm]
        u64* asy = asx + maxm;                                                           // [maxm]
        for (u32 c = tid; c < ncomp; c += CCL_NT) { acnt[c] = 0; asx[c] = 0; asy[c] = 0; }
        __syncthreads();
        for (int vt = tid; vt < geo.vitems; vt += CCL_NT) {
            CCL_ITEM_DECODE_SPREAD
            u32 bc = cbase[it];
            if (cbase[it + 1] == bc) continue;
            const u64* row = bits + (int64_t)y * WW;
            u32 ccid = NONE16, cnt = 0, sx = 0, pB = 0;
            for (int j = j0; j < j1; ++j) {
                const u64 B = row[j], stB = ccl_starts(B, pB);
                u64 mB = B;
                while (mB) {
                    const u64 lowbit = mB & (~mB + 1ull);
                    const u64 t = mB + lowbit;
                    const u64 g = mB & ~t;
                    mB &= t;
                    const u32 v = P[bc + (u32)__popcll(stB & ((lowbit << 1) - 1ull)) - 1u], cid = v & 0x7FFFu;
                    const u32 len = (u32)__popcll(g), x0 = 64u * j + (u32)(__ffsll((long long)g) - 1);
                    if ((v & 0x8000u) && (stB & lowbit)) first[cid] = (u32)y * (u32)W + x0;
                    if (cid != ccid) {
                        if (cnt) { atomicAdd(&acnt[ccid], cnt); atomicAdd(&asx[ccid], (u64)sx); atomicAdd(&asy[ccid], (u64)cnt * (u64)y); }
                        ccid = cid; cnt = 0; sx = 0;
                    }
                    cnt += len;
                    sx += len * x0 + len * (len - 1) / 2;
                }
                bc += (u32)__popcll(stB);
                pB = (u32)(B >> 63);
            }
            if (cnt) { atomicAdd(&acnt[ccid], cnt); atomicAdd(&asx[ccid], (u64)sx); atomicAdd(&asy[ccid], (u64)cnt * (u64)y); }
        }
        __syncthreads();
        u64* bs = band_sums + (int64_t)n * maxm * 4;
        for (u32 c = tid; c < ncomp; c += CCL_NT) { bs[c * 4 + 0] = acnt[c]; bs[c * 4 + 1] = asx[c]; bs[c * 4 + 2] = asy[c]; }
        if (tid == 0) { ncomp_all[n * 2 + 0] = ncomp; fstat[n * 8 + 5] = ncomp; }
        return;
    }

    // ---- D (open) 0: the component's first pixel = start of its root run (the moments' origin) ---------------------
    u32* anchor = reinterpret_cast<u32*>(accb + CCL_MOM_COMPS * NMOM * 8);               // [CCL_OPEN_COMPS]
    __syncthreads();
    for (int it = tid; it < items; it += CCL_NT) {
        u32 bc = cbase[it];
        if (cbase[it + 1] == bc) continue;
        CCL_ITEM_DECODE
        const u64* row = bits + (int64_t)y * WW;
        u32 pB = 0;
        for (int j = j0; j < j1; ++j) {
            const u64 B = row[j];
            u64 st = ccl_starts(B, pB);
            while (st) {                                 // one new node per start bit, in order
                const int k = __ffsll((long long)st) - 1;
                st &= st - 1;
                const u32 v = P[bc++];
                if (v & 0x8000u) {
                    const u32 pos = (u32)y * (u32)W + 64u * j + (u32)k;
                    anchor[v & 0x7FFFu] = pos;
                    first[v & 0x7FFFu] = pos;
                }
            }
            pB = (u32)(B >> 63);
        }
    }
    // Euler number by bit quads (see k_label.hip): holes = components - E
    {
        const int NW = H * WW;
        int e4 = 0;
        for (int idx = tid; idx < NW; idx += CCL_NT) {
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
                    const u64 a1 = (a >> 1) | (an << 63), b1 = (bq >> 1) | (bn << 63);
                    const u64 x2 = (a ^ a1) ^ (bq ^ b1);
                    const u64 pairs = (a & a1) | (a & bq) | (a & b1) | (a1 & bq) | (a1 & b1) | (bq & b1);
                    const u64 qd = (a & b1 & ~a1 & ~bq) | (a1 & bq & ~a & ~b1);
                    e4 += __popcll(x2 & ~pairs) - __popcll(x2 & pairs) - 2 * __popcll(qd);
                    if (jc == 0) e4 += (int)((a ^ bq) & 1ull);
                }
            }
        }
        if (e4) atomicAdd(&misc[0], e4);
    }
    __syncthreads();
    if ((int)ncomp - misc[0] / 4 != 0) {                 // holes: RETR_EXTERNAL needs the fill passes of the general path
        if (tid == 0) slow_flag[n] = 1;
        return;
    }
    if (geo.stop == 5) return;

    // ---- D (open) 1: contour-vertex moments, CCL_MOM_COMPS components per pass, LDS atomics in the spread order ----
    u64* acc = reinterpret_cast<u64*>(accb);
    i64* as = area_sums + (int64_t)n * maxm * VBS_AREA_SUMS;
    for (u32 c0 = 0; c0 < ncomp; c0 += CCL_MOM_COMPS) {
        const u32 nc = min((u32)CCL_MOM_COMPS, ncomp - c0);
        for (u32 c = tid; c < nc * NMOM; c += CCL_NT) acc[c] = 0;
        __syncthreads();
        for (int vt = tid; vt < geo.vitems; vt += CCL_NT) {
            CCL_ITEM_DECODE_SPREAD
            u32 bc = cbase[it];
            if (cbase[it + 1] == bc) continue;
            const u64* rowm = bits + (int64_t)y * WW;
            const bool hasu = y > 0, hasd = y + 1 < H;
            u32 pB = 0;
            for (int j = j0; j < j1; ++j) {
                const u64 B = rowm[j];
                const u64 stB = ccl_starts(B, pB);
                const u32 bcw = bc;
                bc += (u32)__popcll(stB);
                pB = (u32)(B >> 63);
                if (!B) continue;
                const u64 up = hasu ? rowm[j - WW] : 0ull, dn = hasd ? rowm[j + WW] : 0ull;
                u64 bl = 0, br = 0, upL = 0, upR = 0, dnL = 0, dnR = 0;
                if ((B & 1ull) && j > 0) {
                    bl = rowm[j - 1] >> 63;
                    upL = hasu ? rowm[j - 1 - WW] >> 63 : 0ull; dnL = hasd ? rowm[j - 1 + WW] >> 63 : 0ull;
                }
                if ((B >> 63) && j + 1 < WW) {
                    br = rowm[j + 1] & 1ull;
                    upR = hasu ? rowm[j + 1 - WW] & 1ull : 0ull; dnR = hasd ? rowm[j + 1 + WW] & 1ull : 0ull;
                }
                const u64 NE = (up >> 1) | (upR << 63), NWd = (up << 1) | upL;
                const u64 SE = (dn >> 1) | (dnR << 63), SW = (dn << 1) | dnL;
                const u64 E = (B >> 1) | (br << 63), Wd = (B << 1) | bl;
                // border pixels that can be contour vertices: not 4-interior, not inside a straight horizontal edge
                // (patterns 241 / 31 of the vertex table: multiplicity 0)
                u64 bgw = B & ~(up & dn & E & Wd);
                bgw &= ~(E & Wd & ((~up & ~NE & ~NWd & dn & SE & SW) | (up & NE & NWd & ~dn & ~SE & ~SW)));
                u64 mB = B;
                while (mB) {
                    const u64 lowbit = mB & (~mB + 1ull);
                    const u64 t = mB + lowbit;
                    const u64 g = mB & ~t;
                    mB &= t;
                    u64 bg = bgw & g;
                    if (!bg) continue;
                    const u32 cid = (P[bcw + (u32)__popcll(stB & ((lowbit << 1) - 1ull)) - 1u] & 0x7FFFu) - c0;
                    if (cid >= nc) continue;
                    const u32 fp = anchor[cid + c0];
                    const int ay = (int)(fp / (u32)W), ax = (int)(fp - (u32)ay * (u32)W);
                    u64* a = acc + cid * NMOM;
                    while (bg) {
                        const int k = __ffsll((long long)bg) - 1;
                        bg &= bg - 1;
                        const u32 pat = (u32)((E >> k) & 1ull) | ((u32)((NE >> k) & 1ull) << 1) |
                                        ((u32)((up >> k) & 1ull) << 2) | ((u32)((NWd >> k) & 1ull) << 3) |
                                        ((u32)((Wd >> k) & 1ull) << 4) | ((u32)((SW >> k) & 1ull) << 5) |
                                        ((u32)((dn >> k) & 1ull) << 6) | ((u32)((SE >> k) & 1ull) << 7);
                        const int mult = lut[pat];
                        if (!mult) continue;
                        const int dx = 64 * j + k - ax, dy = y - ay;
                        atomicAdd(&a[0], (u64)mult);
                        if (max(abs(dx), abs(dy)) <= 150) {          // 4 * 150^4 < 2^31: products in 32 bits
                            const int x2 = dx * dx, y2 = dy * dy, mx = mult * dx, my = mult * dy;
                            if (dx) {
                                atomicAdd(&a[1], (u64)(i64)mx);
                                atomicAdd(&a[3], (u64)(i64)(mx * dx));
                                atomicAdd(&a[6], (u64)(i64)(mx * x2));
                                atomicAdd(&a[10], (u64)(i64)(mult * x2 * x2));
                            }
                            if (dy) {
                                atomicAdd(&a[2], (u64)(i64)my);
                                atomicAdd(&a[5], (u64)(i64)(my * dy));
                                atomicAdd(&a[9], (u64)(i64)(my * y2));
                                atomicAdd(&a[14], (u64)(i64)(mult * y2 * y2));
                            }
                            if (dx && dy) {
                                atomicAdd(&a[4], (u64)(i64)(mx * dy));
                                atomicAdd(&a[7], (u64)(i64)(my * x2));
                                atomicAdd(&a[8], (u64)(i64)(mx * y2));
                                atomicAdd(&a[11], (u64)(i64)(mx * x2 * dy));
                                atomicAdd(&a[12], (u64)(i64)(mult * x2 * y2));
                                atomicAdd(&a[13], (u64)(i64)(mx * dy * y2));
                            }
                        } else {
                            const i64 ml = mult, dl = dx, el = dy, x2 = dl * dl, y2 = el * el;
                            atomicAdd(&a[1], (u64)(ml * dl));            atomicAdd(&a[2], (u64)(ml * el));
                            atomicAdd(&a[3], (u64)(ml * x2));            atomicAdd(&a[4], (u64)(ml * dl * el));
                            atomicAdd(&a[5], (u64)(ml * y2));            atomicAdd(&a[6], (u64)(ml * x2 * dl));
                            atomicAdd(&a[7], (u64)(ml * x2 * el));       atomicAdd(&a[8], (u64)(ml * dl * y2));
                            atomicAdd(&a[9], (u64)(ml * y2 * el));       atomicAdd(&a[10], (u64)(ml * x2 * x2));
                            atomicAdd(&a[11], (u64)(ml * x2 * dl * el)); atomicAdd(&a[12], (u64)(ml * x2 * y2));
                            atomicAdd(&a[13], (u64)(ml * dl * el * y2)); atomicAdd(&a[14], (u64)(ml * y2 * y2));
                        }
                    }
                }
            }
        }
        __syncthreads();
        for (u32 c = tid; c < nc * NMOM; c += CCL_NT) as[(c0 + c / NMOM) * VBS_AREA_SUMS + (c % NMOM)] = (i64)acc[c];
        __syncthreads();
    }
    if (geo.stop == 6) return;

    // ---- D (open) 2: probes for pointPolygonTest: component ids of the 2x2 cell around every band centroid -----
    {
        const u32 nband = ncomp_all[n * 2 + 0];
        const u64* bs = band_sums + (int64_t)n * maxm * 4;
        unsigned short* pr = probe_all + (int64_t)n * maxm * 4;
        for (u32 i = tid; i < nband; i += CCL_NT) {
            const double cn = (double)bs[i * 4 + 0];
            const float xf = (float)((double)bs[i * 4 + 1] / cn), yf = (float)((double)bs[i * 4 + 2] / cn);
            const int ix = (int)floorf(xf), iy = (int)floorf(yf);
            ushort4 o;
            o.x = (unsigned short)ccl_pixel_cid(bits, P, cbase, geo, ix, iy);
            o.y = (unsigned short)ccl_pixel_cid(bits, P, cbase, geo, ix + 1, iy);
            o.z = (unsigned short)ccl_pixel_cid(bits, P, cbase, geo, ix, iy + 1);
            o.w = (unsigned short)ccl_pixel_cid(bits, P, cbase, geo, ix + 1, iy + 1);
            *reinterpret_cast<ushort4*>(pr + i * 4) = o;
        }
    }
    if (tid == 0) { ncomp_all[n * 2 + 1] = ncomp; fstat[n * 8 + 6] = ncomp; fstat[n * 8 + 4] = 0; }
}

// LDS layout of k_ccl<mode> for this handle: [parents u16 x node_cap][chunk bases u16 x (items + 1)][accumulators]
// [anchors][scan scratch + lut].  Two workgroups per CU (<= 80 KB each) when the expected number of runs fits the
// node table that leaves, else one workgroup with the whole 160 KB.
static bool ccl_layout(const vbs_handle* h, int mode, CclGeom* g, size_t* lds_bytes) {
    g->H = h->H; g->W = h->W; g->WW = h->WW;
    g->CW = h->WW < 5 ? h->WW : 5;
    g->NC = (h->WW + g->CW - 1) / g->CW;
    g->items = h->H * g->NC;
    g->inv_nc = g->NC == 1 ? 0u : (u32)((0x100000000ull + g->NC - 1) / g->NC);
    g->R = (h->H + 63) / 64;
    g->vitems = g->R * 64 * g->NC;
    g->inv_r = g->R == 1 ? 0u : (u32)((0x100000000ull + g->R - 1) / g->R);
    g->stop = VBS_KNOB("VBS_CCL_STOP");
    const size_t cb = ((size_t)(g->items + 1) * 2 + 15) / 16 * 16;
    const size_t acc = mode == 0 ? ((size_t)(8 * ((h->maxm + 1) / 2)) + 16 * (size_t)h->maxm + 15) / 16 * 16     // band sums
                                 : (size_t)CCL_MOM_COMPS * NMOM * 8 + (size_t)CCL_OPEN_COMPS * 4;                  // moments, anchors
    const size_t misc = 32 * 4 + 16 + 256;
    const size_t fixed = cb + acc + misc;
    const size_t half = 80 * 1024, full = 160 * 1024;
    if (fixed + 2 * 1024 > full) return false;
    // expected runs on marker frames (measured: 1280x1024 band 14.3 k / open 8 k; 1920x1200 band 29 k / open 16 k)
    const size_t expect = (size_t)h->H * h->W / (mode == 0 ? 72 : 130);
    size_t cap = fixed < half ? (half - fixed) / 2 : 0;
    if (cap < expect) cap = (full - fixed) / 2;
    cap = cap / 8 * 8;
    if (cap > CCL_NODE_MAX) cap = CCL_NODE_MAX / 8 * 8;
    g->node_cap = (u32)cap;
    g->off_cbase = (u32)(2 * cap);
    g->off_acc = (u32)(g->off_cbase + cb);
    g->off_tmp = (u32)(g->off_acc + acc);
    *lds_bytes = g->off_tmp + misc;
    return g->items < 65535 && cap >= 1024;
}

bool launch_ccl(vbs_handle* h, int nb, hipStream_t s) {
    CclGeom g0, g1;
    size_t l0 = 0, l1 = 0;
    const bool fast = ccl_layout(h, 0, &g0, &l0) && ccl_layout(h, 1, &g1, &l1);
    (void)hipMemsetAsync(h->slow_flag, 0, (size_t)nb * sizeof(u32), s);
    if (fast) {
        static size_t set0 = 0, set1 = 0;                // the largest dynamic-LDS sizes declared so far
        if (l0 > set0) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ccl<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l0); set0 = l0; }
        if (l1 > set1) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ccl<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1); set1 = l1; }
        VBS_LAUNCH(h, s, "k_ccl_band", k_ccl<0>, dim3(nb), dim3(CCL_NT), l0, s, h->band_bits, h->ncomp, h->band_first,
                   h->band_sums, h->area_sums, h->probe, h->fstat, h->slow_flag, h->lut, g0, h->maxm);
        VBS_LAUNCH(h, s, "k_ccl_open", k_ccl<1>, dim3(nb), dim3(CCL_NT), l1, s, h->open_bits, h->ncomp, h->area_first,
                   h->band_sums, h->area_sums, h->probe, h->fstat, h->slow_flag, h->lut, g1, h->maxm);
    }
    return fast;                                         // false: every frame takes the general kernel
}
