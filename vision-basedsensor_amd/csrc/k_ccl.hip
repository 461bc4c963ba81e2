// a9-a13 labelling, fast path (marker_detection.py:170-196): connected components of the bit-packed band mask
// (4-connectivity, ndimage.label :176) and of the opened area mask (8-connectivity, cv2.findContours :196) with the
// per-component sums k_finalize needs.
//
// One workgroup per (frame, mask); nothing walks the image sequentially.  The image is cut into work items of CW
// consecutive 64-px words of one row ("chunks"; lane l of a wave takes chunk l, so a wave reads consecutive bytes):
//   A  word-runs (maximal runs of 1s inside one word) per chunk -> block prefix sum: the nodes are numbered in
//      raster order, 16-bit, cbase[item] = index of the chunk's first node
//   B  union-find over the nodes in LDS (uint16 parents, two per dword; hooking = compare-and-swap on the dword,
//      always towards the smaller index, path halving): every word links its runs to the runs of the word above
//      (+ the two diagonal neighbours for 8-connectivity) and to the run that ends at bit 63 of the word to its left
//   C  flatten; roots ranked in raster order = ndimage.label's numbering (reversed: cv2's contour order); the parent
//      table becomes the component id of every node (bit 15 marks the root = the component's first run)
//   D  band: pixel count / sum x / sum y per component (center_of_mass :181)
//      open: contour-vertex moments about the component's first pixel (see k_label.hip), bit-quad Euler number
//            (holes), and for every band centroid the component ids of the 2x2 pixel cell around it ("probes"):
//            k_finalize's pointPolygonTest needs nothing else, so no label image or run table goes to HBM.
// 80 KB of LDS and <= 64 VGPRs: two workgroups (32 waves) per CU.  Frames outside the fast path's limits (more
// than CCL_NODE_CAP word-runs, more than 512 contours, holes in the opened mask, very wide rows) set their slow flag and are
// redone by the general kernels of k_label.hip, which also own the capacity status.
#include "common.h"

#define CCL_NT 1024
#define CCL_NODE_CAP 15872         // word-runs per mask (uint16 parents: 31 KB)
#define CCL_ITEM_CAP 7680          // chunks per mask (uint16 bases: 15 KB)
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

// links of word B (row y) to the word above (A), its diagonal neighbours (aL = bit 63 of the word above-left,
// aR = bit 0 of the word above-right) and the word to its left (bL = its bit 63).  bc / ba = node index of the
// first run of B / A.
template <int M8>
__device__ __forceinline__ void ccl_link_word(unsigned short* P, u64 B, u64 A, u32 aL, u32 aR, u32 bL, u32 bc, u32 ba) {
    if ((B & 1ull) && bL) ccl_union(P, bc, bc - 1);
    u64 adj = A;
    if (M8) adj |= (A << 1) | (A >> 1) | (u64)aL | ((u64)aR << 63);
    if (!(B & adj)) return;
    const u64 stA = A & ~(A << 1);
    u64 mB = B;
    u32 nb = bc;
    while (mB) {
        const u64 lowbit = mB & (~mB + 1ull);
        const u64 t = mB + lowbit;
        const u64 g = mB & ~t;                          // one run of B
        mB &= t;
        const u32 node = nb++;
        if (!(g & adj)) continue;
        u64 rm = g;
        if (M8) rm |= (g << 1) | (g >> 1);
        u64 mA = A & rm;
        while (mA) {                                    // the runs of A under it
            const u64 lb = mA & (~mA + 1ull);
            const u64 t2 = mA + lb;
            mA &= t2;
            ccl_union(P, node, ba + (u32)__popcll(stA & ((lb << 1) - 1ull)) - 1u);
        }
        if (M8) {
            if ((g & 1ull) && aL) ccl_union(P, node, ba - 1u);
            if ((g >> 63) && aR) ccl_union(P, node, ba + (u32)__popcll(stA));
        }
    }
}

struct CclGeom {
    int H, W, WW, CW, NC, items;
    u32 inv_nc;                                          // ceil(2^32 / NC): item / NC = umulhi(item, inv_nc); 0 when NC = 1
};

// component id (or NONE16) of pixel (x, y) from the resolved parent table
__device__ __forceinline__ u32 ccl_pixel_cid(const u64* __restrict__ bits, const unsigned short* P,
                                             const unsigned short* cbase, const CclGeom& g, int x, int y) {
    if (x < 0 || y < 0 || x >= g.W || y >= g.H) return NONE16;
    const u64* row = bits + (int64_t)y * g.WW;
    const int jw = x >> 6, k = x & 63;
    const u64 w = row[jw];
    if (!((w >> k) & 1ull)) return NONE16;
    const int c = jw / g.CW;
    u32 base = cbase[y * g.NC + c];
    for (int jj = c * g.CW; jj < jw; ++jj) { const u64 ww = row[jj]; base += (u32)__popcll(ww & ~(ww << 1)); }
    const u64 st = w & ~(w << 1);
    const u64 below = (k == 63) ? ~0ull : ((1ull << (k + 1)) - 1ull);
    return P[base + (u32)__popcll(st & below) - 1u] & 0x7FFFu;
}

template <int MODE>                                      // 0: band mask, 4-connectivity; 1: opened mask, 8-connectivity
__global__ __launch_bounds__(CCL_NT, 8) void k_ccl(const u64* __restrict__ bits_all, u32* __restrict__ ncomp_all,
                                                   u32* __restrict__ first_all, u64* __restrict__ band_sums,
                                                   i64* __restrict__ area_sums, unsigned short* __restrict__ probe_all,
                                                   u32* __restrict__ fstat, u32* __restrict__ slow_flag,
                                                   const u8* __restrict__ lut_g, CclGeom geo, int maxm) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned short* P = reinterpret_cast<unsigned short*>(smem);                         // [CCL_NODE_CAP]
    unsigned short* cbase = P + CCL_NODE_CAP;                                            // [CCL_ITEM_CAP]
    unsigned char* accb = smem + 2 * (CCL_NODE_CAP + CCL_ITEM_CAP);                      // 30 KB of accumulators
    u32* anchor = reinterpret_cast<u32*>(accb + CCL_MOM_COMPS * NMOM * 8);               // [CCL_OPEN_COMPS]
    u32* tmp = anchor + CCL_OPEN_COMPS;                                                  // [32]
    int* misc = reinterpret_cast<int*>(tmp + 32);                                        // [4]
    u8* lut = reinterpret_cast<u8*>(misc + 4);                                           // [256]
    const int n = blockIdx.x, tid = threadIdx.x;
    const int H = geo.H, W = geo.W, WW = geo.WW, CW = geo.CW, NC = geo.NC, items = geo.items;
    if (slow_flag[n]) return;                            // already handed to the general path
    const u64* bits = bits_all + (int64_t)n * H * WW;
    if (MODE == 1) {
        if (tid < 256) lut[tid] = lut_g[tid];
        if (tid == 0) misc[0] = 0;
    }

    // ---- A: word-runs per chunk, numbered in raster order ------------------------------------------------------
    for (int it = tid; it < items; it += CCL_NT) {
        const int y = geo.inv_nc ? (int)__umulhi((u32)it, geo.inv_nc) : it, c = it - y * NC;
        const int j0 = c * CW, j1 = min(j0 + CW, WW);
        const u64* row = bits + (int64_t)y * WW;
        u32 cnt = 0;
        for (int j = j0; j < j1; ++j) { const u64 w = row[j]; cnt += (u32)__popcll(w & ~(w << 1)); }
        cbase[it] = (unsigned short)cnt;
    }
    __syncthreads();
    const int K = (items + CCL_NT - 1) / CCL_NT;
    u32 total;
    {
        const int i0 = min(tid * K, items), i1 = min(i0 + K, items);
        u32 s = 0;
        for (int i = i0; i < i1; ++i) s += cbase[i];
        u32 ex = ccl_scan(s, tmp, &total);
        if (total <= CCL_NODE_CAP)
            for (int i = i0; i < i1; ++i) { const u32 c = cbase[i]; cbase[i] = (unsigned short)ex; ex += c; }
    }
    if (total > CCL_NODE_CAP) {                          // block-uniform
        if (tid == 0) slow_flag[n] = 1;
        return;
    }
    if (tid == 0) cbase[items] = (unsigned short)total;
    {
        u32* P32 = reinterpret_cast<u32*>(P);
        for (u32 i = tid; 2 * i < total; i += CCL_NT) P32[i] = (2 * i) | ((2 * i + 1) << 16);
    }
    __syncthreads();

    // ---- B: unions ------------------------------------------------------------------------------------------------
    for (int it = tid; it < items; it += CCL_NT) {
        u32 bc = cbase[it];
        if (cbase[it + 1] == bc) continue;               // no run in this chunk
        const int y = geo.inv_nc ? (int)__umulhi((u32)it, geo.inv_nc) : it, c = it - y * NC;
        const int j0 = c * CW, j1 = min(j0 + CW, WW);
        const u64* row = bits + (int64_t)y * WW;
        const u64* up = row - WW;
        const bool hasu = y > 0;
        u32 ba = hasu ? (u32)cbase[it - NC] : 0u;
        u32 bL = j0 ? (u32)(row[j0 - 1] >> 63) : 0u;
        u32 aL = (hasu && j0) ? (u32)(up[j0 - 1] >> 63) : 0u;
        u64 A = hasu ? up[j0] : 0ull;
        for (int j = j0; j < j1; ++j) {
            const u64 B = row[j];
            const u64 An = (hasu && j + 1 < WW) ? up[j + 1] : 0ull;
            if (B) ccl_link_word<MODE>(P, B, A, aL, (u32)(An & 1ull), bL, bc, ba);
            bc += (u32)__popcll(B & ~(B << 1));
            ba += (u32)__popcll(A & ~(A << 1));
            bL = (u32)(B >> 63);
            aL = (u32)(A >> 63);
            A = An;
        }
    }
    __syncthreads();

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

    u32* first = first_all + (int64_t)n * maxm;
    if (MODE == 0) {
        // ---- D (band): count, sum x, sum y -------------------------------------------------------------------------
        u32* acnt = reinterpret_cast<u32*>(accb);                                        // [1024]
        u64* asx = reinterpret_cast<u64*>(accb + 4096);                                  // [1024]
        u64* asy = asx + 1024;                                                           // [1024]
        for (u32 c = tid; c < ncomp; c += CCL_NT) { acnt[c] = 0; asx[c] = 0; asy[c] = 0; }
        __syncthreads();
        for (int it = tid; it < items; it += CCL_NT) {
            u32 node = cbase[it];
            if (cbase[it + 1] == node) continue;
            const int y = geo.inv_nc ? (int)__umulhi((u32)it, geo.inv_nc) : it, c = it - y * NC;
            const int j0 = c * CW, j1 = min(j0 + CW, WW);
            const u64* row = bits + (int64_t)y * WW;
            u32 ccid = NONE16, cnt = 0, sx = 0;
            for (int j = j0; j < j1; ++j) {
                u64 mB = row[j];
                while (mB) {
                    const u64 lowbit = mB & (~mB + 1ull);
                    const u64 t = mB + lowbit;
                    const u64 g = mB & ~t;
                    mB &= t;
                    const u32 v = P[node++], cid = v & 0x7FFFu;
                    const u32 len = (u32)__popcll(g), x0 = 64u * j + (u32)(__ffsll((long long)g) - 1);
                    if (v & 0x8000u) first[cid] = (u32)y * (u32)W + x0;
                    if (cid != ccid) {
                        if (cnt) { atomicAdd(&acnt[ccid], cnt); atomicAdd(&asx[ccid], (u64)sx); atomicAdd(&asy[ccid], (u64)cnt * (u64)y); }
                        ccid = cid; cnt = 0; sx = 0;
                    }
                    cnt += len;
                    sx += len * x0 + len * (len - 1) / 2;
                }
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
    __syncthreads();
    for (int it = tid; it < items; it += CCL_NT) {
        u32 node = cbase[it];
        if (cbase[it + 1] == node) continue;
        const int y = geo.inv_nc ? (int)__umulhi((u32)it, geo.inv_nc) : it, c = it - y * NC;
        const int j0 = c * CW, j1 = min(j0 + CW, WW);
        const u64* row = bits + (int64_t)y * WW;
        for (int j = j0; j < j1; ++j) {
            u64 mB = row[j];
            while (mB) {
                const u64 lowbit = mB & (~mB + 1ull);
                const u64 t = mB + lowbit;
                const u64 g = mB & ~t;
                mB &= t;
                const u32 v = P[node++];
                if (v & 0x8000u) {
                    const u32 pos = (u32)y * (u32)W + 64u * j + (u32)(__ffsll((long long)g) - 1);
                    anchor[v & 0x7FFFu] = pos;
                    first[v & 0x7FFFu] = pos;
                }
            }
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

    // ---- D (open) 1: contour-vertex moments, CCL_MOM_COMPS components per pass ---------------------------------
    u64* acc = reinterpret_cast<u64*>(accb);
    i64* as = area_sums + (int64_t)n * maxm * VBS_AREA_SUMS;
    for (u32 c0 = 0; c0 < ncomp; c0 += CCL_MOM_COMPS) {
        const u32 nc = min((u32)CCL_MOM_COMPS, ncomp - c0);
        for (u32 c = tid; c < nc * NMOM; c += CCL_NT) acc[c] = 0;
        __syncthreads();
        for (int it = tid; it < items; it += CCL_NT) {
            u32 node = cbase[it];
            if (cbase[it + 1] == node) continue;
            const int y = geo.inv_nc ? (int)__umulhi((u32)it, geo.inv_nc) : it, c = it - y * NC;
            const int j0 = c * CW, j1 = min(j0 + CW, WW);
            const u64* rowm = bits + (int64_t)y * WW;
            const bool hasu = y > 0, hasd = y + 1 < H;
            for (int j = j0; j < j1; ++j) {
                const u64 B = rowm[j];
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
                    const u32 cid = (P[node++] & 0x7FFFu) - c0;
                    u64 bg = bgw & g;
                    if (cid >= nc || !bg) continue;
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

// frames the fast path handed on -> list for the general kernels
__global__ __launch_bounds__(256) void k_slow_list(const u32* __restrict__ slow_flag, u32* __restrict__ list, int nb, int all) {
    __shared__ u32 cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int n = threadIdx.x; n < nb; n += 256)
        if (all || slow_flag[n]) list[1 + atomicAdd(&cnt, 1u)] = (u32)n;
    __syncthreads();
    if (threadIdx.x == 0) list[0] = cnt;
}

static size_t ccl_lds_bytes() {
    return 2 * (CCL_NODE_CAP + CCL_ITEM_CAP) + CCL_MOM_COMPS * NMOM * 8 + CCL_OPEN_COMPS * 4 + 32 * 4 + 16 + 256;
}

bool ccl_fast_geometry(const vbs_handle* h, CclGeom* g) {
    g->H = h->H; g->W = h->W; g->WW = h->WW;
    g->CW = h->WW < 5 ? h->WW : 5;
    g->NC = (h->WW + g->CW - 1) / g->CW;
    g->items = h->H * g->NC;
    g->inv_nc = g->NC == 1 ? 0u : (u32)((0x100000000ull + g->NC - 1) / g->NC);
    return g->items < CCL_ITEM_CAP;                     // (+ 1 entry for the total)
}

void launch_ccl(vbs_handle* h, int nb, hipStream_t s) {
    CclGeom g;
    const bool fast = ccl_fast_geometry(h, &g);
    (void)hipMemsetAsync(h->slow_flag, 0, (size_t)nb * sizeof(u32), s);
    if (fast) {
        const size_t lds = ccl_lds_bytes();
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ccl<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ccl<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        VBS_LAUNCH(h, s, "k_ccl_band", k_ccl<0>, dim3(nb), dim3(CCL_NT), lds, s, h->band_bits, h->ncomp, h->band_first,
                   h->band_sums, h->area_sums, h->probe, h->fstat, h->slow_flag, h->lut, g, h->maxm);
        VBS_LAUNCH(h, s, "k_ccl_open", k_ccl<1>, dim3(nb), dim3(CCL_NT), lds, s, h->open_bits, h->ncomp, h->area_first,
                   h->band_sums, h->area_sums, h->probe, h->fstat, h->slow_flag, h->lut, g, h->maxm);
    }
    // geometry outside the fast path (rows of more than CCL_ITEM_CAP chunks): every frame takes the general kernels
    VBS_LAUNCH(h, s, "k_slow_list", k_slow_list, dim3(1), dim3(256), 0, s, h->slow_flag, h->slow_list, nb, fast ? 0 : 1);
}
