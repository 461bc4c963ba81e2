// a9-a13 for a FEW frames per call (marker_detection.py:170-196 as MarkerTracker.process calls it, one frame at a time,
// :434-453): k_stage.hip's method with a frame spread over 2 C workgroups of four waves instead of one workgroup of twelve.
//
// k_stage labels a frame on ONE compute unit - the right shape for a batch (one frame per CU, 256 CUs), 190 us for a
// single frame.  Here a thread's tile is 64 px x ~8 rows instead of x 29, a plane has 256 C threads (C = 11 at
// 1280x1024), every wave has a SIMD to itself, and the band plane and the opened plane are walked by DIFFERENT workgroups at
// the same time (workgroups 0 .. C - 1 / C .. 2 C - 1 of a frame).  What the workgroups of a frame share goes through
// global memory - and a workgroup on another XCD sees it only through memory, at 1.5 - 3 us per dependent access, so
// everything shared is a flat list that is read with many independent loads in flight, never a table that is chased:
//   walk      as k_stage (same helpers, stage_common.h): rows through register delay lines, segments in slots, what
//             crosses a tile only NOTED as a pair (in LDS).  The link between a tile and the tile BELOW it is made by the
//             upper tile's thread, which computes the first row of the tile below itself (one more row step; the opened
//             walk has that row anyway) and knows that tile's segment ids - its first row's runs are its segments 0, 1, ..
//             The opened walk also writes its slots' pixels and segment ids for every row (32 B per thread and row): the
//             probes are answered from that table once both planes are resolved.
//   own part  every workgroup, all at once: the pairs between its own segments united in its LDS; its segments' sums (band:
//             count / sum x / sum y / first pixel; opened: first pixel, the 15 moments shifted to it) gathered per NODE =
//             root within the workgroup.  Out go ~ 25 nodes, ~ 20 pairs that reach into the next workgroup (own side
//             already a node) and every segment's node (lroot).
//   resolve   the LAST workgroup of a plane to arrive (an arrival counter; nobody waits for a workgroup that has not
//             started) stages the frame's ~ 260 nodes in its LDS, unites the pairs between workgroups, numbers the roots,
//             ranks them by first pixel and gathers the nodes' sums per component.  The opened plane's resolve then needs
//             the band centroids for the probes: the ONE wait of the kernel (the band resolve is the shorter one and has
//             a lower block index); it is bounded, and expiring reports VBS_EINTERNAL in the frame's status, never a
//             wrong table.
// Results are bit-identical to k_stage's (tests/test_gpu_parity.py::test_latency_stage_equals_the_batch_stage,
// tools/gpu_lat_stress.py), frames it cannot take are handed on to k_label with the same kind of slow flag.
// DESIGN 4.5 has the measurements that led here (356 -> 48 us per frame).
#include "stage_common.h"

#define LT_NT 256                  // threads per workgroup: four waves, one per SIMD
#define LT_CMAX 16                 // workgroups per frame, at most: segment ids 8 x 256 x 16 = 2^15 (bit 15 marks a root)
#define LT_REC 512                 // band segment records per workgroup
#define LT_PQ 1024                 // queued pairs per workgroup and walk (LDS)
#define LT_XPQ 512                 // of them, pairs that reach into another workgroup's segments (global)
#define LT_MREC 512                // moment records per workgroup
#define LT_SEG (LT_NT * SG_SEGMAX) // segment ids per workgroup
#define LT_NODE 256                // components WITHIN a workgroup's tiles ("nodes") it can hand to the frame's resolve
#define LT_NODES 1024              // nodes per frame the resolving workgroup stages in its LDS
#ifndef LT_ROWS
#define LT_ROWS 6                  // rows per thread aimed at (8: 50.8 us for a 1280x1024 frame, 6: 49.0, 5: 48.6 - with C at its cap)
#endif
// header words of a frame (VBS_LAT_HDR each, cleared by launch_labelling's fill together with the slow flags)
#define LH_ARRIVE1 0
#define LH_ARRIVE2 1
#define LH_FLAG 2                  // 1: the band components and sums are out; 2: the band plane handed the frame on
#define LH_WHY 3
#define LH_EULER 4
#define LH_WHYO 5                  // LH_WHY of the opened plane's walk
#define LH_NREC 16                 // [C] band nodes / pairs out of the band walk / pairs out of the opened walk / opened nodes
#define LH_NPQB 32
#define LH_NPQO 64
#define LH_NSEG 80

// tools/ builds: wall-clock stamps (100 MHz) of the phases in header words 96.., read with vbs_debug_lat_hdr (tools/gpu_lat_trace.py)
#ifdef VBS_DEBUG_KNOBS
#define LT_STAMP(slot) do { if (tid == 0) atomicMax(&hdr[96 + (slot)], (u32)wall_clock64()); } while (0)
#define LT_STAMP_MIN(slot) do { if (tid == 0) atomicMax(&hdr[96 + (slot)], ~(u32)wall_clock64()); } while (0)
#else
#define LT_STAMP(slot)
#define LT_STAMP_MIN(slot)
#endif

struct LatGeom {
    int H, W, WW, G, NB, R, C, FT, maxm;                  // FT = 256 C threads per frame, NB = 4 C G row blocks of R rows
    u32 mom_comps;
    u32 stride;                                           // bytes of scratch per frame
    u32 o_node, o_pq, o_lroot, o_rst;                     // byte offsets into it
    u32 l_node, l_comp, l_acc, l_tmp;                     // byte offsets into the dynamic LDS of the resolving workgroup (parents at 0)
    int dbg_drop;                                         // tools build: frame `dbg_drop - 1`'s band resolve "forgets" its flag
};

// exclusive prefix sum over the LT_NT threads; tmp holds >= 8 words
__device__ __forceinline__ u32 lt_scan(u32 v, u32* tmp, u32* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    if (lane == 63) tmp[wave] = inc;
    __syncthreads();
    u32 before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < LT_NT / 64; ++w) { const u32 t = tmp[w]; if (w < wave) before += t; tot += t; }
    __syncthreads();
    *total = tot;
    return inc - v + before;
}

// the tile of frame-thread ft
struct LatTile { int g, j, blk; bool act, hasl, hasr; u32 below; };   // below: frame-thread of the tile under this one
__device__ __forceinline__ LatTile lat_tile(u32 ft, int WW, int G) {
    LatTile t;
    const int lane = (int)(ft & 63u);
    t.g = lane / WW; t.j = lane - t.g * WW;
    t.act = t.g < G;
    t.blk = (int)(ft >> 6) * G + t.g;
    t.hasl = t.j > 0; t.hasr = t.act && t.j + 1 < WW;
    t.below = t.g + 1 < G ? ft + (u32)WW : (((ft >> 6) + 1u) << 6) + (u32)t.j;
    return t;
}

// The items of C per-workgroup regions (cnt[w] of them in region w, `cap` slots apart) as one flat list over the workgroup's
// threads, U items per thread in flight: every load of a round is issued before the first item is used (a region
// written on another XCD answers from memory).  pre (LDS, C + 1 words) = exclusive prefix of the counts.
template <int U, class LD, class USE>
__device__ __forceinline__ void lat_flat(const u32* pre, int C, u32 cap, LD ld, USE use) {
    const u32 total = pre[C];
    for (u32 i0 = threadIdx.x; i0 < total; i0 += LT_NT * U) {
        decltype(ld((size_t)0)) v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32 i = i0 + (u32)u * LT_NT;
            if (i < total) {
                u32 w = 0;
                for (int c = 1; c < C; ++c) w += i >= pre[c];
                v[u] = ld((size_t)w * cap + (i - pre[w]));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (i0 + (u32)u * LT_NT < total) use(v[u], i0 + (u32)u * LT_NT);
    }
}
// pre[0 .. C] <- exclusive prefix of cnt[0 .. C) (global), every count clipped to cap
__device__ __forceinline__ void lat_prefix(const u32* cnt, int C, u32 cap, u32* pre) {
    __syncthreads();
    if (threadIdx.x < (unsigned)C) pre[threadIdx.x + 1] = min(cnt[threadIdx.x], cap);
    __syncthreads();
    if (threadIdx.x == 0) { pre[0] = 0; for (int c = 1; c <= C; ++c) pre[c] += pre[c - 1]; }
    __syncthreads();
}

// End of a walk, every workgroup for itself, all workgroups at once: the queued pairs between two of ITS OWN segments are
// united in its LDS (a pair costs a chain of dependent LDS accesses: 2 600 of them per frame are ten per thread of one
// workgroup and one per thread when every workgroup takes its own), its segments' sums are gathered per root within the
// workgroup, and what goes out to the frame's resolve is one NODE per such root (~ 25 per workgroup instead of 210
// records), every segment's node (lroot, global ids: the few lookups the resolve needs) and the few pairs that reach
// into another workgroup - the links of its last row block with the first of the next - with its own side already a node.
//   here: Pl[s] <- root of local segment s, r[i] = that of the thread's own segment i (local numbering); pairs with a
//   foreign side -> xl.  Three barriers; the caller clears its accumulators BEFORE the call (first barrier).
__device__ __forceinline__ void lat_local_roots(unsigned short* Pl, const u32* lpq, int npairs, u32 base, u32 nseg, u32* xl, int* nx,
                                                u32 (&r)[SG_SEGMAX]) {
    const int tid = threadIdx.x;
    {
        const u32 b = 8u * (u32)tid;
        reinterpret_cast<uint4*>(Pl)[tid] = make_uint4(b | ((b + 1u) << 16), (b + 2u) | ((b + 3u) << 16), (b + 4u) | ((b + 5u) << 16),
                                                       (b + 6u) | ((b + 7u) << 16));
    }
    __syncthreads();
    for (int i = tid; i < npairs; i += LT_NT) {
        const u32 pr = lpq[i], a = (pr >> 16) - base, b = (pr & 0xFFFFu) - base;
        if (a < (u32)LT_SEG && b < (u32)LT_SEG) ccl_union(Pl, a, b);
        else { const int k = atomicAdd(nx, 1); if (k < LT_XPQ) xl[k] = pr; }
    }
    __syncthreads();
#pragma unroll
    for (u32 i = 0; i < SG_SEGMAX; ++i) {
        u32 x = 8u * (u32)tid + i;
        if (i < nseg) { u32 p; while ((p = Pl[x]) != x) x = p; }
        r[i] = x;
    }
    // (stored while other threads still follow chains: what they meet is the old parent or the root, both ancestors)
    reinterpret_cast<uint4*>(Pl)[tid] = make_uint4(r[0] | (r[1] << 16), r[2] | (r[3] << 16), r[4] | (r[5] << 16), r[6] | (r[7] << 16));
    __syncthreads();
}
// ... and out: the pairs between workgroups with this workgroup's side replaced by its node, every own segment's node
__device__ __forceinline__ void lat_local_out(const unsigned short* Pl, u32 base, const u32* xl, int nx, const u32 (&r)[SG_SEGMAX],
                                              u32* xpq, unsigned short* lroot_out) {
    const int tid = threadIdx.x;
    for (int k = tid; k < nx; k += LT_NT) {
        const u32 pr = xl[k];
        u32 a = pr >> 16, b = pr & 0xFFFFu;
        if (a - base < (u32)LT_SEG) a = base + Pl[a - base];
        if (b - base < (u32)LT_SEG) b = base + Pl[b - base];
        xpq[k] = (a << 16) | b;
    }
    reinterpret_cast<uint4*>(lroot_out)[tid] = make_uint4((base + r[0]) | ((base + r[1]) << 16), (base + r[2]) | ((base + r[3]) << 16),
                                                          (base + r[4]) | ((base + r[5]) << 16), (base + r[6]) | ((base + r[7]) << 16));
}

// The resolving workgroup's LDS copy of the frame's nodes
struct LatNodes { unsigned short* id; unsigned short* root; u32* pos; u32* cnt; u64* sx; u64* sy; };

// After a walk, by ONE workgroup: the components of the frame's nodes.  The nodes are staged in LDS (BAND: with their sums),
// P (LDS, indexed by segment id, only the nodes' entries are ever touched) starts as the identity on them, the unions are
// the pairs between workgroups (a foreign side looked up in lroot), then every node's root, the roots numbered (P[root] =
// bit 15 | number, comp_pos cleared).  Workgroup-uniform return: components; NONE32: more than `limit`; NONE32 - 1: more
// nodes than LT_NODES.  *nnodes = nodes.
template <bool BAND>
__device__ __forceinline__ u32 lat_resolve(const LatGeom& geo, unsigned short* P, const LatNodes& N, const uint4* nodes, const u32* nnode,
                                           const unsigned short* lroot, const u32* xpq, const u32* nxpq, u32* comp_pos, u32* pre,
                                           u32* tmp, u32 limit, u32* nnodes, u32* hdr, int st0) {
    const int tid = threadIdx.x;
    u32* prx = pre + 32;                                  // prefix of the pair counts (pre keeps the nodes': the moments use it again)
    __syncthreads();
    if (tid < geo.C) { pre[tid + 1] = min(nnode[tid], (u32)LT_NODE); prx[tid + 1] = min(nxpq[tid], (u32)LT_XPQ); }    // (one round of loads)
    __syncthreads();
    if (tid == 0) { pre[0] = 0; prx[0] = 0; for (int c = 1; c <= geo.C; ++c) { pre[c] += pre[c - 1]; prx[c] += prx[c - 1]; } }
    __syncthreads();
    const u32 T = pre[geo.C], TX = prx[geo.C];
    *nnodes = T;
    if (T > (u32)LT_NODES) return NONE32 - 1u;
    // the first 4 LT_NT pairs (all of them on marker frames) are fetched - pair, then both sides' nodes - while the nodes are staged
    uint2 pv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const u32 i = (u32)tid + (u32)u * LT_NT;
        if (i < TX) {
            u32 w = 0;
            for (int c = 1; c < geo.C; ++c) w += i >= prx[c];
            const u32 pr = xpq[(size_t)w * LT_XPQ + (i - prx[w])];
            pv[u] = make_uint2(lroot[pr >> 16], lroot[pr & 0xFFFFu]);
        }
    }
    struct NodeV { uint4 a, b; };
    lat_flat<4>(pre, geo.C, LT_NODE,
                [&](size_t o) { NodeV v; v.a = nodes[(BAND ? 2 : 8) * o]; if (BAND) v.b = nodes[2 * o + 1]; return v; },
                [&](const NodeV& v, u32 i) {
                    N.id[i] = (unsigned short)v.a.x; N.pos[i] = v.a.y; P[v.a.x] = (unsigned short)v.a.x;
                    if (BAND) { N.cnt[i] = v.a.z; N.sx[i] = mk64(v.b.x, v.b.y); N.sy[i] = mk64(v.b.z, v.b.w); }
                });
    __syncthreads();                                     // (every node's entry is in place)
    LT_STAMP(st0);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if ((u32)tid + (u32)u * LT_NT < TX) ccl_union(P, pv[u].x, pv[u].y);
    for (u32 i = (u32)tid + 4u * LT_NT; i < TX; i += LT_NT) {
        u32 w = 0;
        for (int c = 1; c < geo.C; ++c) w += i >= prx[c];
        const u32 pr = xpq[(size_t)w * LT_XPQ + (i - prx[w])];
        ccl_union(P, lroot[pr >> 16], lroot[pr & 0xFFFFu]);
    }
    __syncthreads();
    LT_STAMP(st0 + 1);
    u32 nroot = 0;
    for (u32 t = tid; t < T; t += LT_NT) {               // (read only)
        u32 x = N.id[t], p;
        while ((p = P[x]) != x) x = p;
        N.root[t] = (unsigned short)x;
        nroot += x == N.id[t];
    }
    LT_STAMP(st0 + 2);
    u32 ncomp;
    u32 c0 = lt_scan(nroot, tmp, &ncomp);                 // (its barriers: every root is found before the first number is written)
    if (ncomp > limit) return NONE32;
    for (u32 t = tid; t < T; t += LT_NT)
        if (N.root[t] == N.id[t]) { comp_pos[c0] = NONE32; P[N.id[t]] = (unsigned short)(0x8000u | c0++); }
    __syncthreads();
    return ncomp;
}
__device__ __forceinline__ void lat_rank(const u32* comp_pos, unsigned short* cidmap, u32 ncomp) {
    __syncthreads();
    for (u32 c = threadIdx.x; c < ncomp; c += LT_NT) {   // rank by first pixel (positions are distinct)
        const u32 p = comp_pos[c];
        u32 rank = 0;
        for (u32 q = 0; q < ncomp; ++q) rank += comp_pos[q] < p;
        cidmap[c] = (unsigned short)rank;
    }
    __syncthreads();
}

template <int NS>
__global__ __launch_bounds__(LT_NT, 1) void k_stage_lat(const u64* __restrict__ mask_all, const u64* __restrict__ area_all,
                                                        u32* __restrict__ ncomp_all, u64* __restrict__ band_sums,
                                                        u32* __restrict__ area_first, i64* __restrict__ area_sums,
                                                        unsigned short* __restrict__ probe_all, u32* __restrict__ fstat,
                                                        u32* __restrict__ slow_flag, u32* __restrict__ slow_total,
                                                        u32* __restrict__ hdr_all, unsigned char* __restrict__ scratch_all,
                                                        LatGeom geo) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int misc[16];         // [0] Euler sum, [4] records, [5] queued pairs, [6] why the frame is handed on, [7] moment records,
                                     // [8] this workgroup is the last of its plane, [9] the band plane's flag, [12] pairs between
                                     // workgroups, [13] nodes
    __shared__ u32 lpq[LT_PQ];                                   // the pairs this workgroup's walk queues
    __shared__ u32 xl[LT_XPQ];                                   // of them, the pairs with a side in another workgroup
    __shared__ __align__(16) unsigned short Pl[LT_SEG];          // parents over its own segments (lat_local)
    unsigned short* P = reinterpret_cast<unsigned short*>(smem);                          // [8 FT] parents, by segment id (resolving workgroup)
    LatNodes ND;                                                                         // the frame's nodes (resolving workgroup)
    ND.id = reinterpret_cast<unsigned short*>(smem + geo.l_node);
    ND.root = ND.id + LT_NODES;
    ND.pos = reinterpret_cast<u32*>(ND.root + LT_NODES);
    ND.cnt = ND.pos + LT_NODES;
    ND.sx = reinterpret_cast<u64*>(ND.cnt + LT_NODES);
    ND.sy = ND.sx + LT_NODES;
    // the dynamic LDS while a workgroup WALKS: band: its records [LT_REC] and sums per own segment id [LT_SEG]; opened: its moment
    // records [LT_MREC][16], first pixel per own segment and per node
    unsigned short* w_rsid = reinterpret_cast<unsigned short*>(smem);
    u32* w_rpos = reinterpret_cast<u32*>(smem + 2 * LT_REC);
    u32* w_rcnt = w_rpos + LT_REC;
    u32* w_rsx = w_rcnt + LT_REC;
    u32* w_rsy = w_rsx + LT_REC;
    u32* w_apos = w_rsy + LT_REC;                                                         // [LT_SEG]
    u32* w_acnt = w_apos + LT_SEG;
    u64* w_asx = reinterpret_cast<u64*>(w_acnt + LT_SEG);
    u64* w_asy = w_asx + LT_SEG;
    u32* w_mrec = reinterpret_cast<u32*>(smem);                                           // [LT_MREC][16]
    u32* w_spos = w_mrec + LT_MREC * 16;                                                  // [LT_SEG]
    u32* w_opos = w_spos + LT_SEG;                                                        // [LT_SEG]
    unsigned short* w_nidx = reinterpret_cast<unsigned short*>(w_opos + LT_SEG);          // [LT_SEG] a node's index in the workgroup's list
    u64* w_macc = reinterpret_cast<u64*>(w_nidx + LT_SEG);                                // [LT_NODE][NMOM] a node's moments about its first pixel
    u32* w_nid = reinterpret_cast<u32*>(w_macc + LT_NODE * NMOM);                         // [LT_NODE] its id
    u32* comp_pos = reinterpret_cast<u32*>(smem + geo.l_comp);                            // [1024] first pixel of a component
    unsigned short* cidmap = reinterpret_cast<unsigned short*>(smem + geo.l_comp + 4096); // [1024] its rank = component id
    unsigned char* accb = smem + geo.l_acc;                                              // band sums | anchors + moments
    u32* tmp = reinterpret_cast<u32*>(smem + geo.l_tmp);                                  // [8], then prefixes of the node / pair counts
    u32* pre = tmp + 8;
    const int H = geo.H, W = geo.W, WW = geo.WW, G = geo.G, R = geo.R, maxm = geo.maxm, C = geo.C;
    // workgroups 0 .. C - 1 of a frame take its band plane, C .. 2 C - 1 its opened plane, at the same time
    const int role = (int)blockIdx.x >= C, wg = (int)blockIdx.x - role * C, n = blockIdx.y, tid = threadIdx.x;
    const u32 FT = (u32)geo.FT, ftid = (u32)wg * LT_NT + (u32)tid;
    const LatTile T = lat_tile(ftid, WW, G);
    const int j = T.j, y0 = T.blk * R;
    const bool act = T.act, hasl = T.hasl, hasr = T.hasr;
    const u32 sbelow = T.below * SG_SEGMAX;              // segment ids of the tile below: its first row's runs are its segments 0, 1, ..
    const u64 vm = act ? valid_mask(j, W) : 0ull;
    const u32 sbase = ftid * SG_SEGMAX;
    // the frame's scratch
    u32* hdr = hdr_all + (size_t)n * VBS_LAT_HDR;
    unsigned char* sc = scratch_all + (size_t)n * geo.stride;
    // a workgroup's nodes: band [C][LT_NODE][2 x 16 B] id, first pixel, count, - | sum x, sum y; opened behind them [C][LT_NODE][8 x 16 B]
    // id, first pixel, 15 moments about it (int64)
    uint4* nodes = reinterpret_cast<uint4*>(sc + geo.o_node) + (size_t)role * C * LT_NODE * 2;
    u32* xpqg = reinterpret_cast<u32*>(sc + geo.o_pq) + (size_t)role * C * LT_XPQ;        // [2][C][LT_XPQ] pairs between workgroups
    unsigned short* lroot = reinterpret_cast<unsigned short*>(sc + geo.o_lroot) + (size_t)role * 8 * FT;   // [2][8 FT] a segment's node
    u64* rst = reinterpret_cast<u64*>(sc + geo.o_rst);   // [R][FT][4] opened walk: the slots' pixels in every row of every tile, their segments
    const u32 base = (u32)wg * LT_SEG;                                                    // this workgroup's first segment id
    PairQ Q;
    Q.q = lpq;
    Q.cap = LT_PQ;
    Q.n = &misc[5];
    if (tid < 16) misc[tid] = 0;
    const int64_t fo = (int64_t)n * H * WW;
    auto hand_on = [&](u32 why) {                        // (a resolving workgroup, uniformly; both may: the frame counts once)
        if (tid == 0 && atomicCAS(&slow_flag[n], 0u, why) == 0u) atomicAdd(slow_total, 1u);
    };
    // every store of this workgroup out, then one arrival; true in the workgroup that arrived last
    auto arrive = [&](int which) -> bool {
        __threadfence();
        __syncthreads();
        if (tid == 0) misc[8] = atomicAdd(&hdr[which], 1u) == (u32)(C - 1);
        __syncthreads();
        const bool last = misc[8] != 0;
        if (last) __threadfence();
        return last;
    };
    __syncthreads();
    LT_STAMP_MIN(0); LT_STAMP(1);

    // ================================ band plane ====================================================================
    if (role == 0) {
        constexpr int NA = NS / 2, NBL = NS / 2 - 1;     // rows of the window above / below its row
        const u64* M = mask_all + fo;
        auto ldraw = [&](int r) -> u64 {
            const int y = min(max(y0 + r, 0), H - 1);
            return M[(int64_t)y * WW + (act ? j : 0)];
        };
        auto rowval = [&](u64 raw, int r) -> u64 {       // source row y0 + r; outside the image: ones (ignored by the erosion)
            const int y = y0 + r;
            return (act && y >= 0 && y < H) ? or_not(raw, vm) : ~0ull;
        };
        u64 pf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pf[i] = ldraw(-NA + i);
        u64 cr[NBL + 1];
#pragma unroll
        for (int i = 0; i <= NBL; ++i) cr[i] = ~0ull;
        u64 h1 = ~0ull, a2[2] = {~0ull, ~0ull}, a4[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        u64 a8[6] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
        u64 pm[SG_KB];
        u32 sid[SG_KB], cnt[SG_KB], sy[SG_KB], sk[SG_KB], pos[SG_KB];
#pragma unroll
        for (int k = 0; k < SG_KB; ++k) { pm[k] = 0; sid[k] = 0; cnt[k] = 0; sy[k] = 0; sk[k] = 0; pos[k] = 0; }
        u32 nseg = 0, la = NONE16, lb = NONE16, p63 = NONE16, prs0 = NONE16;
        bool fail = false;
        auto emit = [&](u32 sid_, u32 pos_, u32 cnt_, u32 sk_, u32 sy_) {
            const int r = atomicAdd(&misc[4], 1);
            if (r < LT_REC) {
                w_rsid[r] = (unsigned short)sid_; w_rpos[r] = pos_; w_rcnt[r] = cnt_;
                w_rsx[r] = 64u * (u32)j * cnt_ + sk_; w_rsy[r] = sy_;
            } else fail = true;
        };
#pragma unroll 1
        for (int q = 0; q < R + NS; ++q) {               // (one step past the tile: the first row of the tile below)
            const int r = q - NA;                        // source row of this step
            const u64 v = rowval(pf[0], r);
            pf[0] = pf[1]; pf[1] = pf[2]; pf[2] = pf[3]; pf[3] = ldraw(r + 4);
#pragma unroll
            for (int i = NBL; i > 0; --i) cr[i] = cr[i - 1];
            cr[0] = v;
            const u64 n2 = h1 & v, n4 = a2[1] & n2;
            u64 e;
            if (NS == 14) {
                const u64 n8 = a4[3] & n4;
                e = a8[5] & n8;                          // rows r - 13 .. r
                a8[5] = a8[4]; a8[4] = a8[3]; a8[3] = a8[2]; a8[2] = a8[1]; a8[1] = a8[0]; a8[0] = n8;
            } else {
                e = a4[3] & n4;                          // ns = 8: rows r - 7 .. r
            }
            a4[3] = a4[2]; a4[2] = a4[1]; a4[1] = a4[0]; a4[0] = n4;
            a2[1] = a2[0]; a2[0] = n2;
            h1 = v;
            const int t = r - NBL;                       // the row whose window ends at r
            if (t < 0) continue;                         // (uniform)
            const u64 eh = hwin<NS, true>(e, hasl, hasr);
            const u64 B = (act && y0 + t < H) ? and_not_and(cr[NBL], eh, vm) : 0ull;      // :171-174  maxima = mask & (window holds a 0)
            if (t == R) {                                // (uniform) this tile's last row against the segments of the tile below
                u64 N = B;
                u32 i = 0;
                while (N) {
                    const u64 lowbit = N & (~N + 1ull), t2 = N + lowbit, gg = N & ~t2;
                    N &= t2;
#pragma unroll
                    for (int k = 0; k < SG_KB; ++k)
                        if (gg & pm[k]) pq_push(Q, sbelow + i, sid[k]);
                    ++i;
                }
                break;
            }
            bool live = false;
#pragma unroll
            for (int k = 0; k < SG_KB; ++k) live |= pm[k] != 0ull;
            if (!__any(B != 0ull || live)) continue;     // (wave-uniform)
            const u32 y = (u32)(y0 + t);
            const u64 rB = brev64(B);
            u64 Rn[SG_KB];
            const u64 claimed = seg_update<SG_KB, false, true>(B, rB, pm, sid, Rn, Q);
#pragma unroll
            for (int k = 0; k < SG_KB; ++k) {
                const u64 Rk = Rn[k];
                if (k < 2 || __any(Rk != 0ull)) {
                    const u32 c = (u32)__popcll(Rk);
                    cnt[k] += c; sy[k] += c * y; sk[k] += sum_bitpos(Rk);
                }
                pm[k] = Rk;
            }
            u64 N = B & ~claimed;
            while (N) {                                  // runs no segment reaches: new segments
                const u64 lowbit = N & (~N + 1ull), t2 = N + lowbit, gg = N & ~t2;
                N &= t2;
                bool done = false;
#pragma unroll
                for (int k = 0; k < SG_KB; ++k) {
                    if (!done && pm[k] == 0ull) {
                        if (cnt[k]) emit(sid[k], pos[k], cnt[k], sk[k], sy[k]);
                        if (nseg < SG_SEGMAX) sid[k] = sbase + nseg;
                        else fail = true;
                        ++nseg;
                        const u32 c = (u32)__popcll(gg);
                        pos[k] = y * (u32)W + 64u * (u32)j + (u32)(__ffsll((long long)gg) - 1);
                        pm[k] = gg; cnt[k] = c; sy[k] = c * y; sk[k] = sum_bitpos(gg);
                        done = true;
                    }
                }
                if (!done) fail = true;
            }
            if (__any(((B >> 63) | B) & 1ull)) seg_hlinks<SG_KB, false>(pm, sid, hasr, p63, prs0, la, lb, Q);    // (wave-uniform)
        }
#pragma unroll
        for (int k = 0; k < SG_KB; ++k)
            if (cnt[k]) emit(sid[k], pos[k], cnt[k], sk[k], sy[k]);
        if (fail) misc[6] = SLOW_SLOTS;
        LT_STAMP(2); LT_STAMP_MIN(12);
        // ---- this workgroup's own part of the resolve: its pairs, its segments' sums per node -----------------------------
        {
            const uint4 z = make_uint4(0, 0, 0, 0), o = make_uint4(NONE32, NONE32, NONE32, NONE32);
            reinterpret_cast<uint4*>(w_apos)[2 * tid] = o; reinterpret_cast<uint4*>(w_apos)[2 * tid + 1] = o;
            reinterpret_cast<uint4*>(w_acnt)[2 * tid] = z; reinterpret_cast<uint4*>(w_acnt)[2 * tid + 1] = z;
#pragma unroll
            for (int q = 0; q < 4; ++q) { reinterpret_cast<uint4*>(w_asx)[4 * tid + q] = z; reinterpret_cast<uint4*>(w_asy)[4 * tid + q] = z; }
            u32 r[SG_SEGMAX];
            __syncthreads();                             // (every wave's walk is over: the counts stand)
            lat_local_roots(Pl, lpq, min(misc[5], LT_PQ), base, min(nseg, (u32)SG_SEGMAX), xl, &misc[12], r);
            const int nrec = min(misc[4], LT_REC);
            for (int i = tid; i < nrec; i += LT_NT) {
                const u32 nd = Pl[w_rsid[i] - base];
                atomicMin(&w_apos[nd], w_rpos[i]); atomicAdd(&w_acnt[nd], w_rcnt[i]);
                atomicAdd(&w_asx[nd], (u64)w_rsx[i]); atomicAdd(&w_asy[nd], (u64)w_rsy[i]);
            }
            lat_local_out(Pl, base, xl, min(misc[12], LT_XPQ), r, xpqg + (size_t)wg * LT_XPQ, lroot + base);
            __syncthreads();
#pragma unroll
            for (u32 i = 0; i < SG_SEGMAX; ++i) {
                const u32 sl = 8u * (u32)tid + i;
                if (i < nseg && r[i] == sl) {            // a node: its sums go out
                    const int k = atomicAdd(&misc[13], 1);
                    if (k < LT_NODE) {
                        const u64 sx = w_asx[sl], sy_ = w_asy[sl];
                        nodes[2 * ((size_t)wg * LT_NODE + k)] = make_uint4(base + sl, w_apos[sl], w_acnt[sl], 0u);
                        nodes[2 * ((size_t)wg * LT_NODE + k) + 1] = make_uint4((u32)sx, (u32)(sx >> 32), (u32)sy_, (u32)(sy_ >> 32));
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (misc[5] > LT_PQ || misc[12] > LT_XPQ || misc[13] > LT_NODE) misc[6] = SLOW_SLOTS;
            hdr[LH_NREC + wg] = (u32)min(misc[13], LT_NODE);
            hdr[LH_NPQB + wg] = (u32)min(misc[12], LT_XPQ);
            if (misc[6]) atomicMax(&hdr[LH_WHY], (u32)misc[6]);
        }
        if (!arrive(LH_ARRIVE1)) return;
        {
            // ---- the frame's band components, by this workgroup alone ------------------------------------------------------
            LT_STAMP(3);
            u32 go = 1;
            const u32 whyw = __atomic_load_n(&hdr[LH_WHY], __ATOMIC_RELAXED);
            u32 ncomp = 0, nnodes = 0;
            if (whyw) { hand_on(whyw); go = 2; }
            else {
                u32 T = 0;
                ncomp = lat_resolve<true>(geo, P, ND, nodes, hdr + LH_NREC, lroot, xpqg, hdr + LH_NPQB, comp_pos, pre, tmp,
                                          min((u32)maxm, 1024u), &T, hdr, 14);
                if (ncomp >= NONE32 - 1u) { hand_on(ncomp == NONE32 ? SLOW_NCOMP : SLOW_SLOTS); go = 2; }
                nnodes = T;
            }
            LT_STAMP(4);
            if (go == 1) {
                // ---- first pixels, rank, component sums (center_of_mass :181) out of the nodes ---------------------------------
                u32* acnt = reinterpret_cast<u32*>(accb);                                // [maxm]
                u64* asx = reinterpret_cast<u64*>(accb + 8 * ((maxm + 1) / 2));           // [maxm]
                u64* asy = asx + maxm;                                                   // [maxm]
                for (u32 c = tid; c < ncomp; c += LT_NT) { acnt[c] = 0; asx[c] = 0; asy[c] = 0; }
                __syncthreads();
                for (u32 t = tid; t < nnodes; t += LT_NT) {
                    const u32 c = P[ND.root[t]] & 0x7FFFu;       // (numbered, not yet ranked: the sums move below)
                    atomicMin(&comp_pos[c], ND.pos[t]);
                    atomicAdd(&acnt[c], ND.cnt[t]); atomicAdd(&asx[c], ND.sx[t]); atomicAdd(&asy[c], ND.sy[t]);
                }
                lat_rank(comp_pos, cidmap, ncomp);
                // the sums go out (the opened plane's resolve reads the centroids back for its probes)
                u64* bs = band_sums + (int64_t)n * maxm * 4;
                for (u32 cu = tid; cu < ncomp; cu += LT_NT) {
                    const u32 c = cidmap[cu];                // (the sums were gathered by component NUMBER: the id is its rank)
                    bs[c * 4 + 0] = acnt[cu]; bs[c * 4 + 1] = asx[cu]; bs[c * 4 + 2] = asy[cu];
                }
                if (tid == 0) { ncomp_all[n * 2 + 0] = ncomp; fstat[n * 8 + 5] = ncomp; }
            }
            LT_STAMP(5);
            // the band sums are out (or the frame is handed on): the opened plane's resolve reads the centroids for its probes
            __threadfence();
            __syncthreads();
            if (tid == 0 && geo.dbg_drop != n + 1) atomicExch(&hdr[LH_FLAG], go);
        }
        return;
    }

    // ================================ opened area plane =============================================================
    unsigned short* pr = probe_all + (int64_t)n * maxm * 4;
    {
        const u64* A = area_all + fo;
        auto ldraw = [&](int r) -> u64 {
            const int y = min(max(y0 + r, 0), H - 1);
            return A[(int64_t)y * WW + (act ? j : 0)];
        };
        auto rowval = [&](u64 raw, int r) -> u64 {
            const int y = y0 + r;
            return (act && y >= 0 && y < H) ? or_not(raw, vm) : ~0ull;
        };
        u64 pf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pf[i] = ldraw(-5 + i);
        u64 e5[4] = {~0ull, ~0ull, ~0ull, ~0ull}, d5[4] = {0, 0, 0, 0};
        u64 o1 = 0, o2 = 0;                              // the opened rows before the newest one
        u32 l1 = 0, l2 = 0, r1 = 0, r2 = 0;              // bit 63 of the word to the left / bit 0 of the word to the right in those rows
        u64 pm[SG_KO];
        u32 sid[SG_KO];
        int mo[SG_KO][NMOM];                             // vertex moments about the tile's centre
#pragma unroll
        for (int k = 0; k < SG_KO; ++k) {
            pm[k] = 0; sid[k] = 0;
#pragma unroll
            for (int q = 0; q < NMOM; ++q) mo[k][q] = 0;
        }
        u32 nseg = 0, la = NONE16, lb = NONE16, p63 = NONE16, prs0 = NONE16;
        u32 why = 0;
        int e4 = 0;
        const int tch = R >> 1;
        const int mthr = R <= 64 ? 900 : 56;             // vertices (with multiplicity) an entry may hold: its sums stay below 2^31
        auto emit_mom = [&](u32 sid_, int (&m)[NMOM]) {
            const int r = atomicAdd(&misc[7], 1);
            if (r < LT_MREC) {
                uint4* dst = reinterpret_cast<uint4*>(w_mrec + (size_t)r * 16);
                dst[0] = make_uint4(sid_, (u32)m[0], (u32)m[1], (u32)m[2]);
                dst[1] = make_uint4((u32)m[3], (u32)m[4], (u32)m[5], (u32)m[6]);
                dst[2] = make_uint4((u32)m[7], (u32)m[8], (u32)m[9], (u32)m[10]);
                dst[3] = make_uint4((u32)m[11], (u32)m[12], (u32)m[13], (u32)m[14]);
            } else why = SLOW_RECS;
#pragma unroll
            for (int q = 0; q < NMOM; ++q) m[q] = 0;
        };
#pragma unroll 1
        for (int q = 0; q < R + 10; ++q) {
            const int r = q - 5;                         // source row of this step
            const u64 v = rowval(pf[0], r);
            pf[0] = pf[1]; pf[1] = pf[2]; pf[2] = pf[3]; pf[3] = ldraw(r + 4);
            // vertical erosion over 5 rows -> row r - 2, horizontal erosion; nothing outside the image
            u64 ve = v & e5[0] & e5[1] & e5[2] & e5[3];
            e5[3] = e5[2]; e5[2] = e5[1]; e5[1] = e5[0]; e5[0] = v;
            u64 er = 0;
            if (r >= -1) {                               // (uniform: the eroded rows the tile's output rows reach)
                const int ye = y0 + r - 2;
                er = hwin<5, true>(ve, hasl, hasr);
                er = (act && ye >= 0 && ye < H) ? (er & vm) : 0ull;
            }
            // vertical dilation over 5 rows -> row r - 4, horizontal dilation
            const u64 vd = er | d5[0] | d5[1] | d5[2] | d5[3];
            d5[3] = d5[2]; d5[2] = d5[1]; d5[1] = d5[0]; d5[0] = er;
            const int rho = r - 4;                       // the opened row this step completes (relative to y0)
            if (rho < -1) continue;                      // (uniform)
            u64 o0 = hwin<5, false>(vd, hasl, hasr);
            {
                const int y = y0 + rho;
                o0 = (act && y >= 0 && y < H) ? (o0 & vm) : 0ull;               // :195  morphologyEx(MORPH_OPEN, 5x5)
            }
            u32 l0 = dpp_shr1((u32)(o0 >> 63)), r0 = dpp_shl1((u32)o0 & 1u);
            if (!hasl) l0 = 0;
            if (!hasr) r0 = 0;
            const int c = rho - 1;                       // the row to label now: its neighbours above and below are known
            bool live = false;
#pragma unroll
            for (int k = 0; k < SG_KO; ++k) live |= pm[k] != 0ull;
            // ---- Euler number by bit quads (8-connected foreground): E = (Q1 - Q3 - 2 QD) / 4 over all 2x2 windows of the
            //      zero-padded image; a word counts the windows whose top row is its row (image row 0 also the padding row)
            if (c >= 0) {
                const bool top = (y0 + c) == 0;
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    if (qq == 1 && !top) continue;
                    const u64 a = qq ? 0ull : o1, bq = qq ? o1 : o0;
                    const u32 an = qq ? 0u : r1, bn = qq ? r1 : r0;
                    if (a | bq | an | bn) {
                        e4 += euler_quads(a, bq, an, bn);
                        if (j == 0) e4 += (int)((a ^ bq) & 1ull);           // window x = -1: only (0,y), (0,y+1)
                    }
                }
            }
            if (c >= 0 && __any(o1 != 0ull || live)) {
                const u64 B = o1;
                // ---- segments ---------------------------------------------------------------------------------------------
                const u64 rB = brev64(B);
                u64 Rn[SG_KO];
                const u64 claimed = seg_update<SG_KO, true, true>(B, rB, pm, sid, Rn, Q);
#pragma unroll
                for (int k = 0; k < SG_KO; ++k) pm[k] = Rn[k];
                u64 N = B & ~claimed;
                while (N) {
                    const u64 lowbit = N & (~N + 1ull), t2 = N + lowbit, gg = N & ~t2;
                    N &= t2;
                    bool done = false;
#pragma unroll
                    for (int k = 0; k < SG_KO; ++k) {
                        if (!done && pm[k] == 0ull) {
                            if (mo[k][0]) emit_mom(sid[k], mo[k]);
                            if (nseg < SG_SEGMAX) {
                                sid[k] = sbase + nseg;
                                w_spos[sbase + nseg - base] = (u32)(y0 + c) * (u32)W + 64u * (u32)j + (u32)(__ffsll((long long)gg) - 1);
                            } else why = SLOW_SEGS;
                            ++nseg;
                            pm[k] = gg;
                            done = true;
                        }
                    }
                    if (!done) why = SLOW_SLOTS;
                }
                if (__any(((B >> 63) | B) & 1ull)) seg_hlinks<SG_KO, true>(pm, sid, hasr, p63, prs0, la, lb, Q);
                else { p63 = NONE16; prs0 = NONE16; }    // (no pixel at a word edge in this row: nothing for the next row to meet)
                // ---- contour vertices -------------------------------------------------------------------------------------
                if (__any(B != 0ull)) {
                    u64 V1, V2, V3;                        // at least one / two / three vertices at a pixel (stage_common.h)
                    vertex_planes(B, o2, o0, l2, l1, l0, r2, r1, r0, V1, V2, V3);
                    if (V3) why = SLOW_VERTEX;            // multiplicity > 2: impossible after a 5x5 opening; general path
                    const int tc = c - tch, tc2 = __mul24(tc, tc), tc3 = __mul24(tc2, tc), tc4 = tc2 * tc2;
#pragma unroll
                    for (int k = 0; k < SG_KO; ++k) {
                        u64 vg = V1 & pm[k];
                        if (!__any(vg != 0ull)) continue;
                        if (vg) {
                            int s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
                            auto half = [&](u32 bits, int base) {
                                while (bits) {
                                    const int dx = __ffs((int)bits) - 1 + base;
                                    bits &= bits - 1u;
                                    const int x2 = __mul24(dx, dx);
                                    s0 += 1; s1 += dx; s2 += x2; s3 += __mul24(x2, dx); s4 += __mul24(x2, x2);
                                }
                            };
                            half((u32)vg, -32);
                            half((u32)(vg >> 32), 0);
                            const u64 v2 = vg & V2;
                            if (v2) { half((u32)v2, -32); half((u32)(v2 >> 32), 0); }
                            int (&m)[NMOM] = mo[k];
                            if (m[0] > mthr) emit_mom(sid[k], m);
                            m[0] += s0;                    m[1] += s1;                    m[2] += __mul24(s0, tc);
                            m[3] += s2;                    m[4] += __mul24(s1, tc);       m[5] += __mul24(s0, tc2);
                            m[6] += s3;                    m[7] += __mul24(s2, tc);       m[8] += __mul24(s1, tc2);
                            m[9] += __mul24(s0, tc3);      m[10] += s4;                   m[11] += s3 * tc;
                            m[12] += __mul24(s2, tc2);     m[13] += __mul24(s1, tc3);     m[14] += s0 * tc4;
                        }
                    }
                }
            } else if (c >= 0) {
                p63 = NONE16; prs0 = NONE16;             // an empty row: nothing to link the next one with
            }
            // the slots' pixels in row c and their segments: what a probe (the component at a pixel next to a band centroid,
            // asked once the band plane is resolved - by then this walk is over) is answered from
            if (c >= 0) {
                uint4* dst = reinterpret_cast<uint4*>(rst + ((size_t)c * FT + ftid) * 4);
                dst[0] = make_uint4((u32)pm[0], (u32)(pm[0] >> 32), (u32)pm[1], (u32)(pm[1] >> 32));
                dst[1] = make_uint4((u32)pm[2], (u32)(pm[2] >> 32), sid[0] | (sid[1] << 16), sid[2]);
            }
            o2 = o1; o1 = o0; l2 = l1; l1 = l0; r2 = r1; r1 = r0;
        }
#pragma unroll
        for (int k = 0; k < SG_KO; ++k)
            if (mo[k][0]) emit_mom(sid[k], mo[k]);
        // this tile's last row against the segments of the tile below: o1 is that tile's first row by now (l1 / r1: bit 63 of
        // the word to its left / bit 0 of the word to its right); 8-connectivity: also the tiles below-left and below-right
        {
            const u64 Bn = o1;
            const u32 nrun = (u32)__popcll(Bn & ~(Bn << 1));
            u32 nrl = dpp_shr1(nrun);                    // runs in the first row of the tile below-left (every lane executes this)
            u64 N = Bn;
            u32 i = 0;
            while (N) {
                const u64 lowbit = N & (~N + 1ull), t2 = N + lowbit, gg = N & ~t2;
                N &= t2;
                const u64 ga = gg | (gg << 1) | (gg >> 1);
#pragma unroll
                for (int k = 0; k < SG_KO; ++k)
                    if (ga & pm[k]) pq_push(Q, sbelow + i, sid[k]);
                ++i;
            }
#pragma unroll
            for (int k = 0; k < SG_KO; ++k) {
                if ((pm[k] >> 63) && r1) pq_push(Q, sbelow + SG_SEGMAX, sid[k]);                  // its run at bit 0 is its segment 0
                if ((pm[k] & 1ull) && l1) pq_push(Q, sbelow - SG_SEGMAX + nrl - 1u, sid[k]);      // its run at bit 63 is its last
            }
        }
        if (e4) atomicAdd(&misc[0], e4);
        if (why) misc[6] = (int)why;
        LT_STAMP(7); LT_STAMP_MIN(13);
        // ---- this workgroup's own part of the resolve: its pairs, its segments' first pixels and moment records per node ----
        {
            const uint4 o = make_uint4(NONE32, NONE32, NONE32, NONE32), z = make_uint4(0, 0, 0, 0);
            reinterpret_cast<uint4*>(w_opos)[2 * tid] = o; reinterpret_cast<uint4*>(w_opos)[2 * tid + 1] = o;
            for (int q = tid; q < LT_NODE * NMOM / 2; q += LT_NT) reinterpret_cast<uint4*>(w_macc)[q] = z;
            u32 r[SG_SEGMAX];
            __syncthreads();                             // (every wave's walk is over: the counts stand)
            lat_local_roots(Pl, lpq, min(misc[5], LT_PQ), base, min(nseg, (u32)SG_SEGMAX), xl, &misc[12], r);
#pragma unroll
            for (u32 i = 0; i < SG_SEGMAX; ++i) {
                const u32 sl = 8u * (u32)tid + i;
                if (i < nseg) {
                    atomicMin(&w_opos[r[i]], w_spos[sl]);
                    if (r[i] == sl) {                    // a node: its place in the workgroup's list
                        const int k = atomicAdd(&misc[13], 1);
                        w_nidx[sl] = (unsigned short)min(k, LT_NODE);
                        if (k < LT_NODE) w_nid[k] = base + sl;
                    }
                }
            }
            lat_local_out(Pl, base, xl, min(misc[12], LT_XPQ), r, xpqg + (size_t)wg * LT_XPQ, lroot + base);
            __syncthreads();                             // (every node's first pixel and index stand)
            // the moment records (about their tile's centre) -> their node's moments about ITS first pixel: exact in int64, so
            // the frame's resolve shifts ~ 260 nodes to their component's first pixel instead of 1 500 records
            const int nm = min(misc[7], LT_MREC);
            for (int q = tid; q < nm; q += LT_NT) {
                const uint4* src = reinterpret_cast<const uint4*>(w_mrec + (size_t)q * 16);
                const uint4 w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
                const u32 sg = w0.x, nl = Pl[sg - base], k = w_nidx[nl];
                if (k >= (u32)LT_NODE) continue;         // (more nodes than the list holds: the frame is handed on)
                const u32 ot = sg / SG_SEGMAX, ol = ot & 63u, og = ol / (u32)WW;        // the thread that wrote it: its tile
                const int ox = 64 * (int)(ol - og * (u32)WW) + 32, oy = (int)((ot >> 6) * (u32)G + og) * R + (R >> 1);
                const u32 fp = w_opos[nl], fy = fp / (u32)W, fx = fp - fy * (u32)W;
                const i64 m[NMOM] = {(int)w0.y, (int)w0.z, (int)w0.w, (int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w, (int)w2.x,
                                     (int)w2.y, (int)w2.z, (int)w2.w, (int)w3.x, (int)w3.y, (int)w3.z, (int)w3.w};
                i64 ms[NMOM];
                shift_moments_i64(m, (i64)(ox - (int)fx), (i64)(oy - (int)fy), ms);
                u64* a = w_macc + (size_t)k * NMOM;
#pragma unroll
                for (int e = 0; e < NMOM; ++e)
                    if (ms[e]) atomicAdd(&a[e], (u64)ms[e]);
            }
            __syncthreads();
            const int nn = min(misc[13], LT_NODE);
            if (tid < nn) {                              // node tid goes out: id, first pixel, its 15 moments
                const u32 id = w_nid[tid], fp = w_opos[id - base];
                const u64* a = w_macc + (size_t)tid * NMOM;
                uint4* dst = nodes + ((size_t)wg * LT_NODE + tid) * 8;
                dst[0] = make_uint4(id, fp, (u32)a[0], (u32)(a[0] >> 32));
#pragma unroll
                for (int e = 0; e < 7; ++e)
                    dst[1 + e] = make_uint4((u32)a[1 + 2 * e], (u32)(a[1 + 2 * e] >> 32), (u32)a[2 + 2 * e], (u32)(a[2 + 2 * e] >> 32));
            }
        }
        __syncthreads();
        if (tid == 0) {
            if (misc[5] > LT_PQ || misc[12] > LT_XPQ || misc[13] > LT_NODE) misc[6] = SLOW_SLOTS;
            hdr[LH_NPQO + wg] = (u32)min(misc[12], LT_XPQ);
            hdr[LH_NSEG + wg] = (u32)min(misc[13], LT_NODE);
            if (misc[0]) atomicAdd(&hdr[LH_EULER], (u32)misc[0]);
            if (misc[6]) atomicMax(&hdr[LH_WHYO], (u32)misc[6]);
        }
    }
    if (!arrive(LH_ARRIVE2)) return;
    LT_STAMP(8);
    // ---- the frame's opened components, by this workgroup alone --------------------------------------------------------
    {
        const u32 whyw = __atomic_load_n(&hdr[LH_WHYO], __ATOMIC_RELAXED);
        if (whyw) { hand_on(16u + whyw); return; }
    }
    u32 nnodes = 0;
    const u32 ncomp = lat_resolve<false>(geo, P, ND, nodes, hdr + LH_NSEG, lroot, xpqg, hdr + LH_NPQO, comp_pos, pre, tmp,
                                         min((u32)maxm, (u32)CCL_OPEN_COMPS), &nnodes, hdr, 17);
    if (ncomp >= NONE32 - 1u) { hand_on(16u + (ncomp == NONE32 ? SLOW_NCOMP : SLOW_SLOTS)); return; }
    for (u32 t = tid; t < nnodes; t += LT_NT) {
        const u32 c = P[ND.root[t]] & 0x7FFFu;
        atomicMin(&comp_pos[c], ND.pos[t]);
        if (ND.root[t] != ND.id[t]) P[ND.id[t]] = (unsigned short)c;      // (every node's entry: its component's number)
    }
    lat_rank(comp_pos, cidmap, ncomp);
    LT_STAMP(9);
    if ((int)ncomp - (int)__atomic_load_n(&hdr[LH_EULER], __ATOMIC_RELAXED) / 4 != 0) {      // holes: RETR_EXTERNAL needs the fill passes of the general path
        hand_on(16u + SLOW_HOLES);
        return;
    }
    // ---- the component's first pixel (the moments' origin) -------------------------------------------------------------
    u32* anchor = reinterpret_cast<u32*>(accb);                                          // [CCL_OPEN_COMPS]  (y << 16) | x
    u64* acc = reinterpret_cast<u64*>(accb + 4 * CCL_OPEN_COMPS);                         // [mom_comps][NMOM]
    {
        u32* first = area_first + (int64_t)n * maxm;
        for (u32 c = tid; c < ncomp; c += LT_NT) {
            const u32 pos = comp_pos[c], cid = cidmap[c], py = pos / (u32)W;
            anchor[cid] = (py << 16) | (pos - py * (u32)W);
            first[cid] = pos;
        }
    }
    // ---- segment moments -> component moments about its first pixel, `mom_comps` components per pass -----------------------------
    i64* as = area_sums + (int64_t)n * maxm * VBS_AREA_SUMS;
    for (u32 c0 = 0; c0 < ncomp; c0 += geo.mom_comps) {
        const u32 nc = min(geo.mom_comps, ncomp - c0);
        for (u32 c = tid; c < nc * NMOM; c += LT_NT) acc[c] = 0;
        __syncthreads();
        struct MomV { uint4 w[8]; };
        lat_flat<2>(pre, C, LT_NODE,
                    [&](size_t o) { MomV v; const uint4* src = nodes + o * 8;
#pragma unroll
                                    for (int e = 0; e < 8; ++e) v.w[e] = src[e];
                                    return v; },
                    [&](const MomV& v, u32 t) {
                const u32 cid = (u32)cidmap[P[ND.root[t]] & 0x7FFFu] - c0;
                if (cid >= nc) return;                   // another pass's component
                const u32 np = v.w[0].y, ny = np / (u32)W, nx = np - ny * (u32)W, fp = anchor[cid + c0];
                i64 m[NMOM], o[NMOM];
                m[0] = (i64)mk64(v.w[0].z, v.w[0].w);
#pragma unroll
                for (int e = 0; e < 7; ++e) { m[1 + 2 * e] = (i64)mk64(v.w[1 + e].x, v.w[1 + e].y); m[2 + 2 * e] = (i64)mk64(v.w[1 + e].z, v.w[1 + e].w); }
                shift_moments_i64(m, (i64)((int)nx - (int)(fp & 0xFFFFu)), (i64)((int)ny - (int)(fp >> 16)), o);
                u64* a = acc + cid * NMOM;
#pragma unroll
                for (int q = 0; q < NMOM; ++q)
                    if (o[q]) atomicAdd(&a[q], (u64)o[q]);
            });
        __syncthreads();
        for (u32 c = tid; c < nc * NMOM; c += LT_NT) as[(c0 + c / NMOM) * VBS_AREA_SUMS + (c % NMOM)] = (i64)acc[c];
        __syncthreads();
    }
    LT_STAMP(10);
    // ---- probes: the opened component at the four pixels around every band centroid -------------------------------------
    // the band plane's resolve runs beside this one and is shorter; its flag says the centroids are out
    if (tid == 0) {
        u32 f = 0;
        for (int it = 0; it < (1 << 19); ++it) {
            f = __atomic_load_n(&hdr[LH_FLAG], __ATOMIC_RELAXED);
            if (f) break;
            __builtin_amdgcn_s_sleep(8);
        }
        misc[9] = (int)f;
    }
    __syncthreads();
    __threadfence();
    {
        const int f = misc[9];
        if (f != 1) {
            // 2: the band plane handed the frame on.  0: the wait ran out (a workgroup of the frame never arrived): say so
            if (f == 0 && tid == 0) atomicMin((int*)&fstat[n * 8 + 2], VBS_EINTERNAL);
            return;
        }
    }
    const u32 nband = ncomp_all[n * 2 + 0];
    const u64* bs = band_sums + (int64_t)n * maxm * 4;
    for (u32 e0 = tid; e0 < nband * 4; e0 += LT_NT * 2) {   // (two probes per thread in flight: centroid -> row slots -> node)
        u32 sg[2];
        uint4 t0[2], t1[2];
        int kb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const u32 e = e0 + (u32)u * LT_NT;
            kb[u] = -1;
            if (e >= nband * 4) continue;
            const u32 c = e >> 2, q = e & 3u;
            const double cn = (double)(u32)bs[c * 4 + 0];
            const float xf = (float)((double)bs[c * 4 + 1] / cn), yf = (float)((double)bs[c * 4 + 2] / cn);
            const int px = (int)floorf(xf) + (int)(q & 1u), py = (int)floorf(yf) + (int)(q >> 1);
            if (px < 0 || py < 0 || px >= W || py >= H) continue;
            const int ob = py / R, oi = py - ob * R, ow = ob / G, og = ob - ow * G;
            const u32 owner = (u32)(ow * 64 + og * WW + (px >> 6));
            const uint4* src = reinterpret_cast<const uint4*>(rst + ((size_t)oi * FT + owner) * 4);
            t0[u] = src[0]; t1[u] = src[1];
            kb[u] = px & 63;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            sg[u] = NONE16;
            if (kb[u] < 0) continue;
            const u64 m0 = mk64(t0[u].x, t0[u].y), m1 = mk64(t0[u].z, t0[u].w), m2 = mk64(t1[u].x, t1[u].y);
            if ((m0 >> kb[u]) & 1ull) sg[u] = t1[u].z & 0xFFFFu;
            if ((m1 >> kb[u]) & 1ull) sg[u] = t1[u].z >> 16;
            if ((m2 >> kb[u]) & 1ull) sg[u] = t1[u].w & 0xFFFFu;
            if (sg[u] != NONE16) sg[u] = lroot[sg[u]];   // its node
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const u32 e = e0 + (u32)u * LT_NT;
            if (e < nband * 4) pr[e] = sg[u] != NONE16 ? cidmap[P[sg[u]] & 0x7FFFu] : (unsigned short)NONE16;
        }
    }
    if (tid == 0) { ncomp_all[n * 2 + 1] = ncomp; fstat[n * 8 + 6] = ncomp; fstat[n * 8 + 4] = 0; }
    LT_STAMP(11);
}

// ---- host side ----------------------------------------------------------------------------------------------------
// false = geometry outside this path
static bool lat_geom(const vbs_handle* h, LatGeom* g, size_t* lds_bytes) {
    if (h->W > 4096 || h->H > 2048 || h->maxm > 1024) return false;
    const int G = 64 / h->WW;                            // (WW <= 64: vbs_create)
    const int NW = (h->H + LT_ROWS * G - 1) / (LT_ROWS * G);
    int C = (NW + 3) / 4;
    if (C > LT_CMAX) C = LT_CMAX;
    if (C < 1) C = 1;
    const int NB = 4 * C * G, R = (h->H + NB - 1) / NB;
    if (R > 128 || R < 1) return false;
    g->H = h->H; g->W = h->W; g->WW = h->WW; g->G = G; g->NB = NB; g->R = R; g->C = C; g->FT = LT_NT * C; g->maxm = h->maxm;
    g->mom_comps = (u32)CCL_MOM_COMPS;
    g->dbg_drop = VBS_KNOB("VBS_LAT_DROP");
    const size_t FT = (size_t)g->FT;
    auto up16 = [](size_t x) { return (x + 15) / 16 * 16; };
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += up16(bytes); return (u32)at; };
    g->o_node = take((size_t)C * LT_NODE * (32 + 128));
    g->o_pq = take((size_t)2 * C * LT_XPQ * 4);
    g->o_lroot = take(2 * 8 * FT * 2);
    g->o_rst = take((size_t)R * FT * 32);
    g->stride = (u32)((o + 255) / 256 * 256);
    // LDS of the resolving workgroup | of a walking one (the kernel's prologue has the layouts)
    const size_t par = up16(8 * FT * 2);
    const size_t acc_band = up16((size_t)(8 * ((h->maxm + 1) / 2)) + 16 * (size_t)h->maxm);
    const size_t acc_open = up16((size_t)4 * CCL_OPEN_COMPS + (size_t)g->mom_comps * NMOM * 8);
    g->l_node = (u32)par;
    g->l_comp = (u32)(par + (size_t)LT_NODES * 28);
    g->l_acc = g->l_comp + 4096 + 2048;
    g->l_tmp = (u32)(g->l_acc + (acc_band > acc_open ? acc_band : acc_open));
    const size_t res = g->l_tmp + 256;
    const size_t walk_band = (size_t)LT_REC * 18 + (size_t)LT_SEG * 24;
    const size_t walk_open = (size_t)LT_MREC * 64 + (size_t)LT_SEG * 10 + (size_t)LT_NODE * (NMOM * 8 + 8);
    *lds_bytes = std::max(res, std::max(walk_band, walk_open));
    return *lds_bytes <= 160 * 1024;
}

// bytes of per-frame scratch this geometry needs (vbs_create allocates VBS_LAT_MAXN of them); 0 = outside the path
size_t stage_lat_scratch(const vbs_handle* h) {
    LatGeom g;
    size_t lds;
    return lat_geom(h, &g, &lds) ? (size_t)g.stride : 0;
}

template <int NS>
static bool stage_lat_launch_t(vbs_handle* h, int nb, const LatGeom& g, size_t lds, hipStream_t s) {
    if (lds > h->lat_lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_stage_lat<NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        h->lat_lds_set = lds;
    }
    VBS_LAUNCH(h, s, "k_stage_lat", (k_stage_lat<NS>), dim3(2 * g.C, nb), dim3(LT_NT), lds, s, h->mask_bits, h->area_bits, h->ncomp,
               h->band_sums, h->area_first, h->area_sums, h->probe, h->fstat, h->slow_flag, h->slow_total, h->lat_hdr,
               h->lat_scratch, g);
    return true;
}

// The few-frames form of launch_stage.  The caller has cleared the frames' headers (h->lat_hdr) on `s`.
// false: not for this pass (more than VBS_LAT_MAXN frames, geometry outside the path, no scratch)
bool launch_stage_lat(vbs_handle* h, int nb, hipStream_t s) {
    LatGeom g;
    size_t lds = 0;
    if (nb > h->lat_slots || !h->lat_scratch || !lat_geom(h, &g, &lds)) return false;
    return h->bp.ns == 14 ? stage_lat_launch_t<14>(h, nb, g, lds, s) : stage_lat_launch_t<8>(h, nb, g, lds, s);
}
