// a9-a13 labelling, fast path (marker_detection.py:170-196): connected components of the bit-packed band mask
// (4-connectivity, ndimage.label :176) and of the opened area mask (8-connectivity, cv2.findContours :196) with the
// per-component sums k_finalize needs.
//
// One workgroup per (frame, mask); nothing walks the image sequentially.  The image is cut into work items of CW
// consecutive 64-px words of one row ("chunks"; lane l of a wave takes chunk l, so a wave reads consecutive bytes):
//   A  runs of 1s that START in the chunk -> block prefix sum: the runs (= nodes) are numbered in raster order, 15 bits,
//      cbase[item] = nodes before the chunk (bit 15: the chunk holds any pixel).  A run that enters a chunk from the
//      left is the node cbase - 1, so runs are never cut.  (Opened mask: the bit-quad Euler number rides along.)
//   B  union-find over the nodes in LDS (uint16 parents, two per dword).  A component of a marker frame is a stack of
//      runs, one or two per row, so hooking every run to the run above it with atomics builds chains as deep as the
//      component is tall and every find walks them.  Instead: (1) every run takes as parent the FIRST run it touches
//      in the row above (else itself) - plain stores, a forest whose roots are smaller than their members; (2) pointer
//      jumping, parent = parent[parent], until nothing moves (log2 of the tallest chain rounds); (3) only the
//      remaining links (a run that touches a second run above) go through the atomic union (compare-and-swap on the
//      dword, towards the smaller index, path halving) - on trees that are flat by then, and only in the chunks that
//      have such links
//   C  flatten; roots ranked in raster order = ndimage.label's numbering (reversed: cv2's contour order); the parent
//      table becomes the component id of every node (bit 15 marks the root = the component's first run)
//   D  band: pixel count / sum x / sum y per component (center_of_mass :181)
//      open: contour-vertex moments about the component's first pixel (see k_label.hip): the border pixels are listed
//            and then classified / accumulated one per thread (LDS atomics); and for every band centroid the
//            component ids of the 2x2 pixel cell around it ("probes"): k_finalize's pointPolygonTest needs nothing
//            else, so no label image or run table goes to HBM.
// LDS is laid out per handle (ccl_layout): <= 80 KB and <= 64 VGPRs put two workgroups (32 waves) on a CU at
// 1280x1024; larger frames take up to the whole 160 KB (one workgroup per CU).  Frames outside the fast path's limits
// (more runs than the node table holds - at most 32767 -, more than 512 contours, holes in the opened mask) set their
// slow flag and are redone by the general kernel of k_label.hip, which also owns the capacity status.
#include "ccl_common.h"

struct CclGeom {
    int H, W, WW, CW, NC, items;
    u32 inv_nc;                                          // ceil(2^32 / NC): item / NC = umulhi(item, inv_nc); 0 when NC = 1
    u32 node_cap;                                        // entries of the parent table
    u32 off_cbase, off_acc, off_tmp;                     // byte offsets into the dynamic LDS
    u32 pair_cap;                                        // entries of the extra-link list (it borrows the accumulator area)
    u32 cand_cap;                                        // border-pixel records per frame (global scratch)
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
    u32 base = cbase[y * g.NC + c] & 0x7FFFu, p = j0 ? (u32)(row[j0 - 1] >> 63) : 0u;
    for (int jj = j0; jj < jw; ++jj) { const u64 ww = row[jj]; base += (u32)__popcll(ccl_starts(ww, p)); p = (u32)(ww >> 63); }
    const u64 below = (k == 63) ? ~0ull : ((1ull << (k + 1)) - 1ull);
    return P[base + (u32)__popcll(ccl_starts(w, p) & below) - 1u] & 0x7FFFu;
}

// bits k-1, k, k+1 of the 66-bit string {l, X[0..63], r} (given as the dwords e0 = X[0..30] << 1 | l, e1 = X[31..62],
// e2 = X[63] | r << 1) as a 3-bit number, k = 0..63
__device__ __forceinline__ u32 ccl_win3(u32 e0, u32 e1, u32 e2, int k) {
    const bool hi = k >= 32;
    return __builtin_amdgcn_alignbit(hi ? e2 : e1, hi ? e1 : e0, (u32)k & 31u) & 7u;
}

#define CCL_ITEM_DECODE                                                                            \
    const int y = geo.inv_nc ? (int)__umulhi((u32)it, geo.inv_nc) : it, c = it - y * NC;           \
    const int j0 = c * CW, j1 = min(j0 + CW, WW);

template <int MODE>                                      // 0: band mask, 4-connectivity; 1: opened mask, 8-connectivity
__global__ __launch_bounds__(CCL_NT, 8) void k_ccl(const u64* __restrict__ bits_all, u32* __restrict__ ncomp_all,
                                                   u32* __restrict__ first_all, u64* __restrict__ band_sums,
                                                   i64* __restrict__ area_sums, unsigned short* __restrict__ probe_all,
                                                   u32* __restrict__ fstat, u32* __restrict__ slow_flag,
                                                   u32* __restrict__ cand_all, const u8* __restrict__ lut_g, CclGeom geo,
                                                   int maxm) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned short* P = reinterpret_cast<unsigned short*>(smem);                         // [node_cap]
    unsigned short* cbase = reinterpret_cast<unsigned short*>(smem + geo.off_cbase);     // [items + 1]
    unsigned char* accb = smem + geo.off_acc;                                            // band sums | open moments, anchors, roots
    u32* tmp = reinterpret_cast<u32*>(smem + geo.off_tmp);                               // [32]
    int* misc = reinterpret_cast<int*>(tmp + 32);                                        // [8]
    u8* lut = reinterpret_cast<u8*>(misc + 8);                                           // [512] (open)
    const int n = blockIdx.x, tid = threadIdx.x;
    const int H = geo.H, W = geo.W, WW = geo.WW, CW = geo.CW, NC = geo.NC, items = geo.items;
    if (slow_flag[n]) return;                            // already handed to the general path
    const u64* bits = bits_all + (int64_t)n * H * WW;
    if (MODE == 1 && tid < 512) {
        // vertex multiplicity by (3 bits above, 3 bits of the row, 3 bits below), bit 0 = left neighbour column:
        // index of the 8-neighbour table (bit d = neighbour in chain direction d: 0 E, 1 NE, 2 N, 3 NW, 4 W, 5 SW, 6 S, 7 SE)
        const u32 u = tid & 7, m = (tid >> 3) & 7, d = tid >> 6;
        lut[tid] = lut_g[((m >> 2) & 1) | (((u >> 2) & 1) << 1) | (((u >> 1) & 1) << 2) | ((u & 1) << 3) | ((m & 1) << 4) |
                         ((d & 1) << 5) | (((d >> 1) & 1) << 6) | (((d >> 2) & 1) << 7)];
    }
    if (tid < 8) misc[tid] = 0;                          // [0] Euler sum, [1..3] pointer-jumping flags, [4] roots, [5] pairs

    // ---- A: runs that start in each chunk, numbered in raster order (+ the Euler number of the opened mask) -------
    {
        int e4 = 0;
        for (int it = tid; it < items; it += CCL_NT) {
            CCL_ITEM_DECODE
            const u64* row = bits + (int64_t)y * WW;
            // the chunk's words (CW <= 5, and the first word of the next chunk for the windows that straddle it) are all
            // requested before any is used: one memory latency per chunk, not one per word
            const int nw = j1 - j0;
            const bool hasd = MODE == 1 && y + 1 < H;
            u64 wv[6], dv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const bool in = k <= nw && j0 + k < WW && (k < nw || MODE == 1);
                wv[k] = in ? row[j0 + k] : 0ull;
                dv[k] = (in && hasd) ? row[j0 + k + WW] : 0ull;
            }
            u32 cnt = 0, p = j0 ? (u32)(row[j0 - 1] >> 63) : 0u;
            u64 any = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                if (k >= nw) break;
                const int j = j0 + k;
                const u64 w = wv[k];
                cnt += (u32)__popcll(ccl_starts(w, p));
                p = (u32)(w >> 63);
                any |= w;
                if (MODE == 1) {
                    // bit quads (8-connected foreground): E = (Q1 - Q3 - 2 QD) / 4 over all 2x2 windows of the zero-padded
                    // image; this word counts the windows whose top row is its row (row 0 also the padding row above)
                    const u64 dn = dv[k], wn_ = wv[k + 1], dn_ = dv[k + 1];
                    if (!(w | dn) && y != 0 && !((wn_ | dn_) & 1ull)) continue;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q == 1 && y != 0) continue;
                        const u64 a = q ? 0ull : w, an = q ? 0ull : wn_, bq = q ? w : dn, bn = q ? wn_ : dn_;
                        if (a | bq | (an & 1ull) | (bn & 1ull)) {
                            const u64 a1 = (a >> 1) | (an << 63), b1 = (bq >> 1) | (bn << 63);
                            const u64 x2 = (a ^ a1) ^ (bq ^ b1);
                            const u64 pairs = (a & a1) | (a & bq) | (a & b1) | (a1 & bq) | (a1 & b1) | (bq & b1);
                            const u64 qd = (a & b1 & ~a1 & ~bq) | (a1 & bq & ~a & ~b1);
                            e4 += __popcll(x2 & ~pairs) - __popcll(x2 & pairs) - 2 * __popcll(qd);
                            if (j == 0) e4 += (int)((a ^ bq) & 1ull);               // window x = -1: only (0,y), (0,y+1)
                        }
                    }
                }
            }
            cbase[it] = (unsigned short)(cnt | (any ? 0x8000u : 0u));
        }
        if (MODE == 1 && e4) atomicAdd(&misc[0], e4);
    }
    __syncthreads();
    if (geo.stop == 1) return;
    const int K = (items + CCL_NT - 1) / CCL_NT;
    u32 total;
    {
        const int i0 = min(tid * K, items), i1 = min(i0 + K, items);
        u32 s = 0;
        for (int i = i0; i < i1; ++i) s += cbase[i] & 0x7FFFu;
        u32 ex = ccl_scan(s, tmp, &total);
        if (total <= geo.node_cap)
            for (int i = i0; i < i1; ++i) { const u32 c = cbase[i]; cbase[i] = (unsigned short)(ex | (c & 0x8000u)); ex += c & 0x7FFFu; }
    }
    if (total > geo.node_cap) {                          // block-uniform
        if (tid == 0) slow_flag[n] = 1;
        return;
    }
    __syncthreads();
    if (geo.stop == 2) return;

    // ---- B: parents, pointer jumping, the remaining links --------------------------------------------------------
    // (the pair list lives in the accumulator area, which nothing uses before phase D)
    CclLists L;
    L.pairs = reinterpret_cast<u32*>(accb);
    L.npairs = &misc[5];
    L.pair_cap = (int)geo.pair_cap;
    L.roots = MODE == 1 ? reinterpret_cast<u32*>(accb + CCL_MOM_COMPS * NMOM * 8 + CCL_OPEN_COMPS * 4) : nullptr;
    L.nroots = &misc[4];
    u64 extra_mask = 0;                                  // bit k: this thread's k-th chunk overflowed the pair list
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && misc[5] <= L.pair_cap) break;   // block-uniform: every further link is in the pair list
        int kk = -1;
        for (int it = tid; it < items; it += CCL_NT) {
            ++kk;
            const u32 cb = cbase[it];
            if (!(cb & 0x8000u)) continue;               // no pixel in this chunk
            if (pass == 1 && kk < 64 && !((extra_mask >> kk) & 1ull)) continue;
            bool extra = false;
            u32 bc = cb & 0x7FFFu;
            CCL_ITEM_DECODE
            const u64* row = bits + (int64_t)y * WW;
            const u64* up = row - WW;
            const bool hasu = y > 0;
            u32 ba = hasu ? (u32)(cbase[it - NC] & 0x7FFFu) : 0u;
            u32 pB = j0 ? (u32)(row[j0 - 1] >> 63) : 0u;
            u32 pA = (hasu && j0) ? (u32)(up[j0 - 1] >> 63) : 0u;
            u64 A = hasu ? up[j0] : 0ull;
            for (int j = j0; j < j1; ++j) {
                const u64 B = row[j];
                const u64 An = (hasu && j + 1 < WW) ? up[j + 1] : 0ull;
                if (B) {
                    const u32 pos0 = (u32)y * (u32)W + 64u * (u32)j;
                    if (pass == 0) extra |= ccl_link_word<MODE, 0>(P, B, A, pB, pA, (u32)(An & 1ull), bc, ba, pos0, L);
                    else ccl_link_word<MODE, 1>(P, B, A, pB, pA, (u32)(An & 1ull), bc, ba, pos0, L);
                }
                bc += (u32)__popcll(ccl_starts(B, pB));
                ba += (u32)__popcll(ccl_starts(A, pA));
                pB = (u32)(B >> 63);
                pA = (u32)(A >> 63);
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
            // the further links, densely: one pair per thread
            const int np = min(misc[5], L.pair_cap);
            for (int i = tid; i < np; i += CCL_NT) { const u32 pr = L.pairs[i]; ccl_union(P, pr >> 16, pr & 0xFFFFu); }
            __syncthreads();
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
    if (ncomp > (u32)maxm || ncomp > (MODE == 0 ? 1024u : (u32)CCL_OPEN_COMPS) ||
        (MODE == 1 && ((int)ncomp - misc[0] / 4 != 0 || misc[4] > CCL_ROOT_LIST))) {
        // block-uniform.  (Open: holes - RETR_EXTERNAL needs the fill passes of the general path -, or more runs without
        // a run above them than the root list remembers.)
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
        for (int it = tid; it < items; it += CCL_NT) {
            const u32 cb = cbase[it];
            if (!(cb & 0x8000u)) continue;
            u32 bc = cb & 0x7FFFu;
            CCL_ITEM_DECODE
            const u64* row = bits + (int64_t)y * WW;
            u32 ccid = NONE16, cnt = 0, sx = 0, pB = j0 ? (u32)(row[j0 - 1] >> 63) : 0u;
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

    // ---- D (open) 0: the component's first pixel = start of its root run (the moments' origin), from the root list ---
    u32* anchor = reinterpret_cast<u32*>(accb + CCL_MOM_COMPS * NMOM * 8);               // [CCL_OPEN_COMPS]  (y << 16) | x
    __syncthreads();
    for (int i = tid; i < misc[4]; i += CCL_NT) {
        const u32 v = P[L.roots[2 * i]];
        if (v & 0x8000u) {
            const u32 pos = L.roots[2 * i + 1], py = pos / (u32)W;
            anchor[v & 0x7FFFu] = (py << 16) | (pos - py * (u32)W);
            first[v & 0x7FFFu] = pos;
        }
    }
    if (geo.stop == 5) return;

    // ---- D (open) 1: contour-vertex moments ------------------------------------------------------------------------
    // (i) the chunk walk only LISTS the border pixels (component id, y, x in one dword, in this frame's slice of the
    //     general path's run table, idle on the fast path); (ii) the vertex classification, the moment terms and the
    //     LDS atomics run one listed pixel per thread.  Done inside the walk's three nested data-dependent loops
    //     (words, runs, pixels) every wave pays the longest trip count of its 64 lanes at every level: 0.31 us per
    //     frame against 0.1 this way.
    u32* cand = cand_all + (int64_t)n * geo.cand_cap;
    for (int itb = tid & ~63; itb < items; itb += CCL_NT) {          // (whole waves: the slot allocation scans the wave)
        const int it = itb + (tid & 63);
        const u32 cb = it < items ? (u32)cbase[it] : 0u;
        const bool occ = (cb & 0x8000u) != 0;
        u32 bc = cb & 0x7FFFu;
        const int yv = it < items ? it : 0;
        const int y = geo.inv_nc ? (int)__umulhi((u32)yv, geo.inv_nc) : yv, c = yv - y * NC;
        const int j0 = c * CW, j1 = min(j0 + CW, WW);
        const u64* rowm = bits + (int64_t)y * WW;
        const bool hasu = y > 0, hasd = y + 1 < H;
        // border pixels (not 4-interior) of the chunk's words, and how many
        u64 bgv[5] = {0, 0, 0, 0, 0};
        u32 cnt = 0;
        if (occ) {
            u32 bl = j0 ? (u32)(rowm[j0 - 1] >> 63) : 0u;
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int j = j0 + q;
                if (j < j1) {
                    const u64 B = rowm[j];
                    if (B) {
                        const u64 up = hasu ? rowm[j - WW] : 0ull, dn = hasd ? rowm[j + WW] : 0ull;
                        u32 br = 0;
                        if ((B >> 63) && j + 1 < WW) br = (u32)(rowm[j + 1] & 1ull);
                        bgv[q] = B & ~(up & dn & ((B >> 1) | ((u64)br << 63)) & ((B << 1) | (u64)bl));
                        cnt += (u32)__popcll(bgv[q]);
                    }
                    bl = (u32)(B >> 63);
                }
            }
        }
        // list slots for the wave's pixels: one atomic per wave
        u32 slot;
        {
            u32 inc = cnt;
            inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);   // row_shr:1
            inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);   // row_shr:2
            inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);   // row_shr:4
            inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);   // row_shr:8
            inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xa, 0xf, false);   // row_bcast:15
            inc += (u32)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xc, 0xf, false);   // row_bcast:31
            const u32 wtot = (u32)__builtin_amdgcn_readlane((int)inc, 63);
            u32 wbase_ = 0;
            if ((tid & 63) == 0 && wtot) wbase_ = (u32)atomicAdd(&misc[6], (int)wtot);
            slot = (u32)__builtin_amdgcn_readfirstlane((int)wbase_) + inc - cnt;
        }
        if (!cnt) continue;
        u32 pB = j0 ? (u32)(rowm[j0 - 1] >> 63) : 0u;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const int j = j0 + q;
            if (j >= j1) continue;
            const u64 B = rowm[j];
            const u64 stB = ccl_starts(B, pB);
            const u32 bcw = bc;
            bc += (u32)__popcll(stB);
            pB = (u32)(B >> 63);
            u64 mB = bgv[q] ? B : 0ull;
            while (mB) {
                const u64 lowbit = mB & (~mB + 1ull);
                const u64 t = mB + lowbit;
                const u64 g = mB & ~t;
                mB &= t;
                u64 bg = bgv[q] & g;
                if (!bg) continue;
                const u32 cid = P[bcw + (u32)__popcll(stB & ((lowbit << 1) - 1ull)) - 1u] & 0x7FFFu;
                const u32 rec = (cid << 23) | ((u32)y << 12) | (64u * (u32)j);
                while (bg) {
                    const u32 k = (u32)(__ffsll((long long)bg) - 1);
                    bg &= bg - 1;
                    if (slot < geo.cand_cap) cand[slot] = rec | k;
                    ++slot;
                }
            }
        }
    }
    __syncthreads();
    if ((u32)misc[6] > geo.cand_cap) {                   // block-uniform: more border pixels than the list holds
        if (tid == 0) slow_flag[n] = 1;
        return;
    }
    __threadfence_block();
    if (geo.stop == 9) return;
    u64* acc = reinterpret_cast<u64*>(accb);
    i64* as = area_sums + (int64_t)n * maxm * VBS_AREA_SUMS;
    const u32 nrec = (u32)misc[6];
    for (u32 c0 = 0; c0 < ncomp; c0 += CCL_MOM_COMPS) {
        const u32 nc = min((u32)CCL_MOM_COMPS, ncomp - c0);
        for (u32 c = tid; c < nc * NMOM; c += CCL_NT) acc[c] = 0;
        __syncthreads();
        // (neighbouring threads take neighbouring pixels of the list: their loads of the three rows coalesce.  Giving every
        //  thread a contiguous stretch of the list and keeping its sums in registers until the component changes saves
        //  nine atomics in ten and was measured 3.5 x slower: scattered loads, uneven stretches.)
        for (u32 i = tid; i < nrec; i += CCL_NT) {
            const u32 rec = cand[i];
            const u32 cid = (rec >> 23) - c0;
            if (cid >= nc) continue;
            const int y = (int)((rec >> 12) & 0x7FFu), x = (int)(rec & 0xFFFu), j = x >> 6, k = x & 63;
            const u64* rowm = bits + (int64_t)y * WW + j;
            const bool hasu = y > 0, hasd = y + 1 < H, el = k == 0 && j > 0, er = k == 63 && j + 1 < WW;
            const u64 md = rowm[0], up = hasu ? rowm[-WW] : 0ull, dn = hasd ? rowm[WW] : 0ull;
            u32 ml = 0, mr = 0, ul = 0, ur = 0, dl = 0, dr = 0;      // neighbour words matter only at bits 0 / 63
            if (el) { ml = (u32)(rowm[-1] >> 63); ul = hasu ? (u32)(rowm[-1 - WW] >> 63) : 0u; dl = hasd ? (u32)(rowm[-1 + WW] >> 63) : 0u; }
            if (er) { mr = (u32)(rowm[1] & 1ull); ur = hasu ? (u32)(rowm[1 - WW] & 1ull) : 0u; dr = hasd ? (u32)(rowm[1 + WW] & 1ull) : 0u; }
            auto win3 = [&](u64 w, u32 l, u32 r) {       // pixels k-1, k, k+1 of the row as bits 0, 1, 2
                u32 v = k ? (u32)(w >> (k - 1)) & 7u : (((u32)w << 1) & 6u) | l;
                return k == 63 ? (v & 3u) | (r << 2) : v;
            };
            const int mult = lut[win3(up, ul, ur) | (win3(md, ml, mr) << 3) | (win3(dn, dl, dr) << 6)];
            if (!mult) continue;
            const u32 fp = anchor[cid + c0];
            const int dy = y - (int)(fp >> 16), dx = x - (int)(fp & 0xFFFFu);
            u64* a = acc + cid * NMOM;
            atomicAdd(&a[0], (u64)mult);
            if (max(abs(dx), abs(dy)) <= 150) {          // 24-bit multiplies: every OPERAND below 2^23 (|mult d^2| <= 4 * 150^2,
                                                         // |d^2| <= 150^2), every product below 2^31 (4 * 150^4)
                const int x2 = __mul24(dx, dx), y2 = __mul24(dy, dy), mx = __mul24(mult, dx), my = __mul24(mult, dy);
                if (dx) {
                    atomicAdd(&a[1], (u64)(i64)mx);
                    atomicAdd(&a[3], (u64)(i64)__mul24(mx, dx));
                    atomicAdd(&a[6], (u64)(i64)__mul24(mx, x2));
                    atomicAdd(&a[10], (u64)(i64)__mul24(__mul24(mult, x2), x2));
                }
                if (dy) {
                    atomicAdd(&a[2], (u64)(i64)my);
                    atomicAdd(&a[5], (u64)(i64)__mul24(my, dy));
                    atomicAdd(&a[9], (u64)(i64)__mul24(my, y2));
                    atomicAdd(&a[14], (u64)(i64)__mul24(__mul24(mult, y2), y2));
                }
                if (dx && dy) {
                    atomicAdd(&a[4], (u64)(i64)__mul24(mx, dy));
                    atomicAdd(&a[7], (u64)(i64)__mul24(my, x2));
                    atomicAdd(&a[8], (u64)(i64)__mul24(mx, y2));
                    atomicAdd(&a[11], (u64)(i64)__mul24(__mul24(mx, dy), x2));       // (mx x2 alone would pass 2^23 for mult >= 3)
                    atomicAdd(&a[12], (u64)(i64)__mul24(__mul24(mult, x2), y2));
                    atomicAdd(&a[13], (u64)(i64)__mul24(__mul24(mx, dy), y2));
                }
            } else {
                const i64 ml_ = mult, dl_ = dx, el_ = dy, x2 = dl_ * dl_, y2 = el_ * el_;
                atomicAdd(&a[1], (u64)(ml_ * dl_));             atomicAdd(&a[2], (u64)(ml_ * el_));
                atomicAdd(&a[3], (u64)(ml_ * x2));              atomicAdd(&a[4], (u64)(ml_ * dl_ * el_));
                atomicAdd(&a[5], (u64)(ml_ * y2));              atomicAdd(&a[6], (u64)(ml_ * x2 * dl_));
                atomicAdd(&a[7], (u64)(ml_ * x2 * el_));        atomicAdd(&a[8], (u64)(ml_ * dl_ * y2));
                atomicAdd(&a[9], (u64)(ml_ * y2 * el_));        atomicAdd(&a[10], (u64)(ml_ * x2 * x2));
                atomicAdd(&a[11], (u64)(ml_ * x2 * dl_ * el_)); atomicAdd(&a[12], (u64)(ml_ * x2 * y2));
                atomicAdd(&a[13], (u64)(ml_ * dl_ * el_ * y2)); atomicAdd(&a[14], (u64)(ml_ * y2 * y2));
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

// LDS layout of k_ccl<mode> for this handle: [parents u16 x node_cap][chunk bases u16 x (items + 1)]
// [band sums | open moment accumulators + anchors + root list][scan scratch + lut].  Two workgroups per CU (<= 80 KB
// each) when the expected number of runs fits the node table that leaves, else one workgroup with the whole 160 KB.
static bool ccl_layout(const vbs_handle* h, int mode, CclGeom* g, size_t* lds_bytes) {
    g->H = h->H; g->W = h->W; g->WW = h->WW;
    g->CW = h->WW < 5 ? h->WW : 5;
    g->NC = (h->WW + g->CW - 1) / g->CW;
    g->items = h->H * g->NC;
    g->inv_nc = g->NC == 1 ? 0u : (u32)((0x100000000ull + g->NC - 1) / g->NC);
    g->stop = VBS_KNOB("VBS_CCL_STOP");
    const size_t cb = ((size_t)(g->items + 1) * 2 + 15) / 16 * 16;
    const size_t acc = mode == 0 ? ((size_t)(8 * ((h->maxm + 1) / 2)) + 16 * (size_t)h->maxm + 15) / 16 * 16     // band sums
                                 : (size_t)CCL_MOM_COMPS * NMOM * 8 + (size_t)CCL_OPEN_COMPS * 4 + (size_t)CCL_ROOT_LIST * 8;
    const size_t misc = 32 * 4 + 32 + 512;
    const size_t fixed = cb + acc + misc;
    const size_t half = 80 * 1024, full = 160 * 1024;
    if (fixed + 2 * 1024 > full) return false;
    // expected runs on marker frames (measured: 1280x1024 band 13.9 k / open 7.6 k; 1920x1200 band 27.9 k / open 15.1 k)
    const size_t expect = (size_t)h->H * h->W / (mode == 0 ? 72 : 130);
    size_t cap = fixed < half ? (half - fixed) / 2 : 0;
    if (cap < expect) cap = (full - fixed) / 2;
    cap = cap / 8 * 8;
    if (cap > CCL_NODE_MAX) cap = CCL_NODE_MAX / 8 * 8;
    g->node_cap = (u32)cap;
    g->off_cbase = (u32)(2 * cap);
    g->off_acc = (u32)(g->off_cbase + cb);
    g->off_tmp = (u32)(g->off_acc + acc);
    g->pair_cap = (u32)((mode == 0 ? acc : (size_t)CCL_MOM_COMPS * NMOM * 8) / 4);
    g->cand_cap = 2u * (u32)h->H * (u32)h->WW;           // the frame's slice of wbase (general path only)
    *lds_bytes = g->off_tmp + misc;
    return g->items < 65535 && h->W <= 4096 && h->H <= 2048 && cap >= 1024;     // (record: 9 + 11 + 12 bits)
}

bool launch_ccl(vbs_handle* h, int nb, hipStream_t s) {
    CclGeom g0, g1;
    size_t l0 = 0, l1 = 0;
    const bool fast = ccl_layout(h, 0, &g0, &l0) && ccl_layout(h, 1, &g1, &l1);
    if (fast) {
        // the dynamic-LDS sizes are declared per handle (a handle = one device): a function-local static would skip the call
        // for a second handle on another device
        if (l0 > h->ccl_lds_set[0]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_ccl<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l0) != hipSuccess) { (void)hipGetLastError(); return false; }
            h->ccl_lds_set[0] = l0;
        }
        if (l1 > h->ccl_lds_set[1]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_ccl<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1) != hipSuccess) { (void)hipGetLastError(); return false; }
            h->ccl_lds_set[1] = l1;
        }
        VBS_LAUNCH(h, s, "k_ccl_band", k_ccl<0>, dim3(nb), dim3(CCL_NT), l0, s, h->band_bits, h->ncomp, h->band_first,
                   h->band_sums, h->area_sums, h->probe, h->fstat, h->slow_flag, h->wbase, h->lut, g0, h->maxm);
        VBS_LAUNCH(h, s, "k_ccl_open", k_ccl<1>, dim3(nb), dim3(CCL_NT), l1, s, h->open_bits, h->ncomp, h->area_first,
                   h->band_sums, h->area_sums, h->probe, h->fstat, h->slow_flag, h->wbase, h->lut, g1, h->maxm);
    }
    return fast;                                         // false: every frame takes the general kernel
}
