// a9-a13, fused fast path (marker_detection.py:170-196): band = mask & ~erode(mask) and the 5x5 opening of the area
// mask, the connected components of both (4- / 8-connectivity) and the per-component sums k_finalize needs, in ONE
// kernel per pass, one workgroup of 768 threads per frame.  Nothing but the two input bit planes is read from memory
// and neither the band / opened planes nor any list of pixels or runs is written.
//
// A thread owns a SEGMENT OF A WORD COLUMN: word column j (64 px) x R consecutive rows (R = 29 at 1280x1024), and streams
// down it: a row is loaded (a few rows ahead), passes through register delay lines and is labelled and summed on the spot;
// no plane is ever held.  A wave holds G = 64 / WW row blocks side by side (lane = g * WW + j): the words left and right
// of a lane's word are in the neighbouring lanes (DPP moves).  One rolled loop per plane.
//   morph   vertical 14- / 5-row AND / OR by doubling (2, 4, 8, 14 rows) in the delay lines; a thread simply starts a
//           window's reach above its first row and runs that far past its last one.  The horizontal windows work on 64-bit
//           words by doubling with the neighbour lane's word (v_alignbit funnel shifts).
//   label   NOT run by run: a thread labels its own 64 x R tile.  It keeps up to K "slots"; a slot is a tile-local piece
//           of a component ("segment") = the mask of its pixels in the previous row.  The pixels of a segment in the next
//           row are the runs that touch that mask, found for all runs at once with two carry chains (add the seeds to the
//           row: the carry fills each run upwards from its lowest seed; the same on the reversed word fills downwards).
//           Runs no slot reaches start new segments.  Only what crosses a tile goes through the union-find in LDS (uint16
//           parents over <= 8 segments per thread), and only as a queued pair: two segments that meet in a run, the
//           segment holding bit 63 of a row with the one holding bit 0 of the word to the right (ids by DPP;
//           8-connectivity: also the rows above / below), and after the walk the first row of a tile with the last row of
//           the tile above.  The unions run densely after the walk.
//   sums    per slot in registers, one record per segment, records -> components once the union-find is resolved.
//           band: count / sum x / sum y (sum of bit positions by six masked popcounts).
//           open: CHAIN_APPROX_SIMPLE vertex multiplicity BIT-PARALLEL (the 256-entry table as boolean functions of the
//           eight shifted neighbour planes: an arc of background neighbours that starts at direction a counts unless it has
//           no 4-neighbour or is exactly {a, a+1, a+2}), so only real vertices are visited; the 15 moments up to order 4
//           about the TILE'S CENTRE (small numbers: int32), as s_a t^b from the row sums s_a = sum mult dx^a; a dense pass
//           shifts every record to its component's first pixel (exact, int64) and adds it to the component.
//   order   components are ranked by their first pixel (ndimage.label's order; reversed: cv2.findContours')
//   probes  component ids of the 2x2 pixel cell around every band centroid: the thread that owns the pixel notes the
//           segment that holds it (requests posted through a small LDS mailbox); ids follow once the components are known
// Frames the fast path cannot take (more than K segments alive in a tile or 8 in all, too many records / pairs /
// components, holes in the opened mask, a vertex of multiplicity > 2, a crowded mailbox) set their slow flag (the value
// says why): k_morph and k_label (k_label.hip) redo them.
#include "stage_common.h"

#define ST_NT_SMALL 256           // small frames: four waves, THREE workgroups per CU (see stage_threads)
#define ST_NT 768                  // threads per frame: 12 waves, three per SIMD, so that a thread may use 168 registers

struct StageGeom {
    int H, W, WW, G, NB, R, maxm;   // G row blocks per wave, NB = 12 G row blocks of R rows
    u32 off_rec, off_bot, off_pq, off_mb, off_tmp;       // byte offsets into the dynamic LDS (segment parents at 0)
    u32 mrec_cap, mrec_stride;      // moment records per frame (global scratch), dwords between two frames' records
    u32 rec_cap, pq_cap, mom_comps; // LDS table sizes: band records, queued pairs, components per moment pass
    int stop;                       // debug builds: leave after phase `stop`
    int retry;                      // 1: the second chance of the frames the 256-thread instance could not hold (launch_stage)
};

// exclusive prefix sum over the NT threads; tmp holds >= 17 words
template <int NT>
__device__ __forceinline__ u32 st_scan(u32 v, u32* tmp, u32* total) {
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
        const u32 t = lane < NT / 64 ? tmp[lane] : 0u;
        u32 ti = t;
#pragma unroll
        for (int d = 1; d < NT / 64; d <<= 1) {
            const u32 o = __shfl_up(ti, d);
            if (lane >= d) ti += o;
        }
        if (lane < NT / 64) tmp[lane] = ti - t;
        if (lane == NT / 64 - 1) tmp[16] = ti;
    }
    __syncthreads();
    const u32 ex = inc - v + tmp[wave];
    *total = tmp[16];
    __syncthreads();
    return ex;
}

// After a walk: the queued unions, then the components of the segments.  The roots (P[s] == s after flattening) are numbered, every segment's
// entry becomes the number of its root (bit 15 marks the root), comp_pos[c] = first pixel of component c in raster order
// (minimum over its segments' first pixels), cidmap[c] = rank of that pixel = the component's id.  Workgroup-uniform
// return: components, or NONE32 when there are more than `limit`.
template <int NT>
__device__ __forceinline__ u32 seg_resolve(unsigned short* P, u32 sbase, u32 nseg, const unsigned short* rec_sid,
                                           const u32* rec_pos, u32 nrec, const u32* seg_pos, u32* comp_pos,
                                           unsigned short* cidmap, u32* tmp, u32 limit, const PairQ& Q) {
    const int tid = threadIdx.x;
    {
        const int np = min(*Q.n, Q.cap);
        for (int i = tid; i < np; i += NT) { const u32 pr = Q.q[i]; ccl_union(P, pr >> 16, pr & 0xFFFFu); }
    }
    __syncthreads();
    for (u32 i = 0; i < nseg; ++i) {                     // flatten (no halving: a late store must be a root)
        u32 x = sbase + i, p;
        while ((p = ((volatile unsigned short*)P)[x]) != x) x = p;
        if (x != sbase + i) P[sbase + i] = (unsigned short)x;
    }
    __syncthreads();
    u32 nroot = 0;
    for (u32 i = 0; i < nseg; ++i) nroot += (P[sbase + i] == sbase + i);
    u32 ncomp;
    u32 c0 = st_scan<NT>(nroot, tmp, &ncomp);
    if (ncomp > limit) return NONE32;
    for (u32 i = 0; i < nseg; ++i)
        if (P[sbase + i] == sbase + i) { comp_pos[c0] = NONE32; P[sbase + i] = (unsigned short)(0x8000u | c0++); }
    __syncthreads();
    for (u32 i = 0; i < nseg; ++i) {
        const u32 v = P[sbase + i];
        if (!(v & 0x8000u)) P[sbase + i] = (unsigned short)(P[v] & 0x7FFFu);
    }
    __syncthreads();
    if (seg_pos) { for (u32 i = 0; i < nseg; ++i) atomicMin(&comp_pos[P[sbase + i] & 0x7FFFu], seg_pos[sbase + i]); }
    else { for (u32 r = tid; r < nrec; r += NT) atomicMin(&comp_pos[P[rec_sid[r]] & 0x7FFFu], rec_pos[r]); }
    __syncthreads();
    for (u32 c = tid; c < ncomp; c += NT) {           // rank by first pixel (positions are distinct)
        const u32 p = comp_pos[c];
        u32 rank = 0;
        for (u32 q = 0; q < ncomp; ++q) rank += comp_pos[q] < p;
        cidmap[c] = (unsigned short)rank;
    }
    __syncthreads();
    return ncomp;
}

template <int NS, int NT>
__global__ __launch_bounds__(NT, 3) void k_stage(const u64* __restrict__ mask_all, const u64* __restrict__ area_all,
                                                    u32* __restrict__ ncomp_all, u64* __restrict__ band_sums,
                                                    u32* __restrict__ area_first, i64* __restrict__ area_sums,
                                                    unsigned short* __restrict__ probe_all, u32* __restrict__ fstat,
                                                    u32* __restrict__ slow_flag, u32* __restrict__ slow_total,
                                                    u32* __restrict__ mrec_all, StageGeom geo) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned short* P = reinterpret_cast<unsigned short*>(smem);                          // [8 NT] segment parents
    unsigned char* recb = smem + geo.off_rec;                                            // segment records
    unsigned char* botb = smem + geo.off_bot;                                            // last-row slots | component tables
    PairQ Q;
    Q.q = reinterpret_cast<u32*>(smem + geo.off_pq);                                      // [pq_cap]
    Q.cap = (int)geo.pq_cap;
    u32* mb_cnt = reinterpret_cast<u32*>(smem + geo.off_mb);                              // [NT] probe requests
    u32* mb_req = mb_cnt + NT;                                                         // [NT][ST_MB_CAP]
    u32* tmp = reinterpret_cast<u32*>(smem + geo.off_tmp);                                // [32]
    int* misc = reinterpret_cast<int*>(tmp + 32);                                         // [16]
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = geo.H, W = geo.W, WW = geo.WW, G = geo.G, R = geo.R, maxm = geo.maxm;
    const int g = lane / WW, j = lane - g * WW;
    const bool act = g < G;
    const int blk = wave * G + g, y0 = blk * R;
    const bool hasl = j > 0, hasr = act && j + 1 < WW;
    const u64 vm = act ? valid_mask(j, W) : 0ull;
    const int NBW = geo.NB * WW, bj = act ? blk * WW + j : 0;
    const bool hasu = act && blk > 0;
    const u32 sbase = (u32)tid * SG_SEGMAX;
    Q.n = &misc[5];
    if (tid < 16) misc[tid] = 0;     // [0] Euler sum, [4] records, [5] queued pairs, [6] hand the frame on (why), [7] moment records
    mb_cnt[tid] = 0;
    const int64_t fo = (int64_t)n * H * WW;
    const u32 rec_cap = geo.rec_cap;
    unsigned short* rec_sid = reinterpret_cast<unsigned short*>(recb);                   // [rec_cap] records, by column
    u32* rec_pos = reinterpret_cast<u32*>(recb + 2 * rec_cap);
    u32* rec_cnt = rec_pos + rec_cap;
    u32* rec_sx = rec_cnt + rec_cap;
    u32* rec_sy = rec_sx + rec_cap;
    u32* seg_pos = reinterpret_cast<u32*>(recb);                                          // [8 NT] opened mask: first pixel by segment id
    // last-row slots of every tile; the component tables take their place once the tiles are linked
    u64* bot_mask = reinterpret_cast<u64*>(botb);                                         // [K][NBW]
    unsigned short* bot_sid = reinterpret_cast<unsigned short*>(botb + (size_t)8 * SG_KB * NBW);   // [K][NBW]
    u32* comp_pos = reinterpret_cast<u32*>(botb);                                         // [1024] first pixel of a component
    unsigned short* cidmap = reinterpret_cast<unsigned short*>(botb + 4096);              // [1024] its rank = component id
    unsigned char* accb = botb + 4096 + 2048;                                            // band sums | anchors + moments
    auto hand_on = [&](u32 why) {                        // (called by every thread, workgroup-uniformly)
        if (tid == 0) { slow_flag[n] = why; atomicAdd(slow_total, 1u); }
    };
    if (geo.retry) {
        // The launch behind the 256-thread instance: only the frames that one handed on because ITS tables or its taller
        // tiles could not hold them (slots, segments, records, pairs, mailbox) - this instance has smaller tiles and tables
        // 2.7 x as large; it takes the frame's flag back and labels it, or hands it on itself.  Nothing to do on marker
        // frames of the usual density: the workgroups leave at once.
        if (*slow_total == 0u) return;
        const u32 f = slow_flag[n];
        if (f != SLOW_SLOTS && f != SLOW_SEGS && f != SLOW_RECS && f != SLOW_PAIRS && f != SLOW_MAILBOX) return;
        __syncthreads();                                 // (every thread has read the flag)
        if (tid == 0) { slow_flag[n] = 0u; atomicSub(slow_total, 1u); }
    }
    __syncthreads();

    // ================================ band plane ====================================================================
    {
        constexpr int NA = NS / 2, NBL = NS / 2 - 1;     // rows of the window above / below its row
        const u64* M = mask_all + fo;
        // rows in flight: RAW loads (nothing touches a loaded value before its step, so the wait for it sits there, four
        // steps after the load was issued); rows outside the image are loaded from a clamped address and replaced at use
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
        u64 cr[NBL + 1];                                 // the last NBL + 1 source rows (the window's own row is NBL back)
#pragma unroll
        for (int i = 0; i <= NBL; ++i) cr[i] = ~0ull;
        // delay lines: h1 = the row before, a2[i] = AND of the 2 rows ending i + 1 rows back, a4 / a8 likewise
        u64 h1 = ~0ull, a2[2] = {~0ull, ~0ull}, a4[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        u64 a8[6] = {~0ull, ~0ull, ~0ull, ~0ull, ~0ull, ~0ull};
        u64 pm[SG_KB];
        u32 sid[SG_KB], cnt[SG_KB], sy[SG_KB], sk[SG_KB], pos[SG_KB];
#pragma unroll
        for (int k = 0; k < SG_KB; ++k) { pm[k] = 0; sid[k] = 0; cnt[k] = 0; sy[k] = 0; sk[k] = 0; pos[k] = 0; }
        u32 nseg = 0, la = NONE16, lb = NONE16, p63 = NONE16, prs0 = NONE16;
        bool fail = false;
        u64 firstB = 0;
        auto emit = [&](u32 sid_, u32 pos_, u32 cnt_, u32 sk_, u32 sy_) {
            const int r = atomicAdd(&misc[4], 1);
            if ((u32)r < rec_cap) {
                rec_sid[r] = (unsigned short)sid_; rec_pos[r] = pos_; rec_cnt[r] = cnt_;
                rec_sx[r] = 64u * (u32)j * cnt_ + sk_; rec_sy[r] = sy_;
            } else fail = true;
        };
        // Two steps per trip: what the delay lines shift between two steps is register renaming then, not moves (0.755 ->
        // 0.74 us per frame).  As a lambda called twice: the same body as an unrolled inner loop keeps the moves, four
        // steps per trip need the 168th register and 100 KB of code for nothing more.
        auto band_step = [&](const int q) __attribute__((always_inline)) {
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
            if (t < 0) return;                         // (uniform)
            const u64 eh = hwin<NS, true>(e, hasl, hasr);
            const u64 B = (act && y0 + t < H) ? and_not_and(cr[NBL], eh, vm) : 0ull;      // :171-174  maxima = mask & (window holds a 0)
            if (t == 0) firstB = B;
            bool live = false;
#pragma unroll
            for (int k = 0; k < SG_KB; ++k) live |= pm[k] != 0ull;
            if (!__any(B != 0ull || live)) return;     // (wave-uniform)
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
                        if (nseg < SG_SEGMAX) { sid[k] = sbase + nseg; P[sbase + nseg] = (unsigned short)(sbase + nseg); }
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
                };
#pragma unroll 1
        for (int q = 0; q < R + NS - 1; q += 2) {
            band_step(q);
            if (q + 1 < R + NS - 1) band_step(q + 1);
        }
#pragma unroll
        for (int k = 0; k < SG_KB; ++k) {
            if (cnt[k]) emit(sid[k], pos[k], cnt[k], sk[k], sy[k]);
            if (act) { bot_mask[(size_t)k * NBW + bj] = pm[k]; bot_sid[(size_t)k * NBW + bj] = (unsigned short)sid[k]; }
        }
        if (fail) misc[6] = SLOW_SLOTS;
        __syncthreads();
        if (misc[6]) { hand_on((u32)misc[6]); return; }
        if (geo.stop == 2) return;
        // the first row of the tile against the last row of the tile above (its runs are segments sbase + 0, 1, .. in order)
        if (hasu) {
            u64 N = firstB;
            u32 i = 0;
            while (N) {
                const u64 lowbit = N & (~N + 1ull), t2 = N + lowbit, gg = N & ~t2;
                N &= t2;
#pragma unroll
                for (int k = 0; k < SG_KB; ++k)
                    if (gg & bot_mask[(size_t)k * NBW + bj - WW]) pq_push(Q, sbase + i, bot_sid[(size_t)k * NBW + bj - WW]);
                ++i;
            }
        }
        __syncthreads();
        if (geo.stop == 3) return;
        const u32 nrec = (u32)misc[4];
        const u32 ncomp = seg_resolve<NT>(P, sbase, min(nseg, (u32)SG_SEGMAX), rec_sid, rec_pos, nrec, nullptr, comp_pos, cidmap, tmp,
                                      min((u32)maxm, 1024u), Q);
        if (misc[5] > Q.cap) { hand_on(SLOW_PAIRS); return; }
        if (ncomp == NONE32) { hand_on(SLOW_NCOMP); return; }
        if (geo.stop == 4) return;
        // ---- component sums (center_of_mass :181) ----------------------------------------------------------------------
        u32* acnt = reinterpret_cast<u32*>(accb);                                        // [maxm]
        u64* asx = reinterpret_cast<u64*>(accb + 8 * ((maxm + 1) / 2));                   // [maxm]
        u64* asy = asx + maxm;                                                           // [maxm]
        for (u32 c = tid; c < ncomp; c += NT) { acnt[c] = 0; asx[c] = 0; asy[c] = 0; }
        __syncthreads();
        for (u32 r = tid; r < nrec; r += NT) {
            const u32 cid = cidmap[P[rec_sid[r]] & 0x7FFFu];
            atomicAdd(&acnt[cid], rec_cnt[r]); atomicAdd(&asx[cid], (u64)rec_sx[r]); atomicAdd(&asy[cid], (u64)rec_sy[r]);
        }
        __syncthreads();
        // the sums go out; the probe requests (2x2 pixel cell around every centroid) go to the threads that own the pixels
        u64* bs = band_sums + (int64_t)n * maxm * 4;
        unsigned short* pr = probe_all + (int64_t)n * maxm * 4;
        for (u32 c = tid; c < ncomp; c += NT) {
            const u32 cn_ = acnt[c];
            const u64 sx = asx[c], sy_ = asy[c];
            bs[c * 4 + 0] = cn_; bs[c * 4 + 1] = sx; bs[c * 4 + 2] = sy_;
            const double cn = (double)cn_;
            const float xf = (float)((double)sx / cn), yf = (float)((double)sy_ / cn);
            const int ix = (int)floorf(xf), iy = (int)floorf(yf);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int px = ix + (q & 1), py = iy + (q >> 1);
                if (px < 0 || py < 0 || px >= W || py >= H) { pr[c * 4 + q] = (unsigned short)NONE16; continue; }
                // one request per row and word: the pixel (ix + 1, py) rides along when it lies in the same word
                const bool pair = (q & 1) == 0 && px + 1 < W && (px & 63) != 63;
                if ((q & 1) && px > 0 && (px & 63) != 0) continue;                   // rode along with (ix, py)
                const int ob = py / R, oi = py - ob * R, ow = ob / G, og = ob - ow * G;
                const int owner = ow * 64 + og * WW + (px >> 6);
                const u32 slot = atomicAdd(&mb_cnt[owner], 1u);
                if (slot < ST_MB_CAP)
                    mb_req[owner * ST_MB_CAP + slot] = c | ((u32)q << 10) | ((u32)oi << 12) | ((u32)(px & 63) << 19) | ((u32)pair << 25);
                else misc[6] = SLOW_MAILBOX;
            }
        }
        if (tid == 0) { ncomp_all[n * 2 + 0] = ncomp; fstat[n * 8 + 5] = ncomp; misc[0] = 0; misc[4] = 0; misc[5] = 0; }
        __syncthreads();
        if (misc[6]) { hand_on((u32)misc[6]); return; }      // a crowded mailbox
        if (geo.stop == 10) return;
    }

    // ================================ opened area plane =============================================================
    const u32 nband = ncomp_all[n * 2 + 0];
    u32 ncomp;
    u32* mrec = mrec_all + (size_t)n * geo.mrec_stride;                                  // [mrec_cap][16]  segment id, 15 moments
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
        // this thread's probe requests: which of its rows have any
        u64 rowm[2] = {0, 0};                            // (R <= 128)
        const u32 nreq = min(mb_cnt[tid], (u32)ST_MB_CAP);
        for (u32 q = 0; q < nreq; ++q) {
            const u32 rr = (mb_req[tid * ST_MB_CAP + q] >> 12) & 127u;
            if (rr < 64) rowm[0] |= 1ull << rr; else rowm[1] |= 1ull << (rr - 64);
        }
        auto row_asked = [&](int c) -> bool { return ((c < 64 ? rowm[0] >> c : rowm[1] >> (c - 64)) & 1ull) != 0; };
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
        u64 firstB = 0;
        int e4 = 0;
        const int tch = R >> 1;
        const int mthr = R <= 64 ? 900 : 56;             // vertices (with multiplicity) an entry may hold: its sums stay below 2^31
        auto emit_mom = [&](u32 sid_, int (&m)[NMOM]) {
            const int r = atomicAdd(&misc[7], 1);
            if ((u32)r < geo.mrec_cap) {
                uint4* dst = reinterpret_cast<uint4*>(mrec + (size_t)r * 16);
                dst[0] = make_uint4(sid_, (u32)m[0], (u32)m[1], (u32)m[2]);
                dst[1] = make_uint4((u32)m[3], (u32)m[4], (u32)m[5], (u32)m[6]);
                dst[2] = make_uint4((u32)m[7], (u32)m[8], (u32)m[9], (u32)m[10]);
                dst[3] = make_uint4((u32)m[11], (u32)m[12], (u32)m[13], (u32)m[14]);
            } else why = SLOW_RECS;
#pragma unroll
            for (int q = 0; q < NMOM; ++q) m[q] = 0;
        };
        // (two steps per trip, as in the band loop)
        auto open_step = [&](const int q) __attribute__((always_inline)) {
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
            if (rho < -1) return;                      // (uniform)
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
            if (c >= 0 && __any(o1 != 0ull || live || row_asked(c))) {
                const u64 B = o1;
                if (c == 0) firstB = B;
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
                                sid[k] = sbase + nseg; P[sbase + nseg] = (unsigned short)(sbase + nseg);
                                seg_pos[sbase + nseg] = (u32)(y0 + c) * (u32)W + 64u * (u32)j + (u32)(__ffsll((long long)gg) - 1);
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
                // ---- probes: the segment that holds a pixel of this row ---------------------------------------------------
                if (row_asked(c)) {
                    for (u32 qq = 0; qq < nreq; ++qq) {
                        const u32 rq = mb_req[tid * ST_MB_CAP + qq];
                        if (((rq >> 12) & 127u) != (u32)c) continue;
                        const u32 q0 = (rq >> 10) & 3u;
                        for (u32 d = 0; d <= ((rq >> 25) & 1u); ++d) {
                            const u32 kb = ((rq >> 19) & 63u) + d;
                            u32 s = NONE16;
#pragma unroll
                            for (int k = 0; k < SG_KO; ++k)
                                if ((pm[k] >> kb) & 1ull) s = sid[k];
                            pr[(rq & 1023u) * 4 + q0 + d] = (unsigned short)s;
                        }
                    }
                }
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
                            // row sums s_a = sum mult dx^a about the word's centre (24-bit multiplies), the two halves of the
                            // word in turn (32-bit bit scans); a vertex of multiplicity 2 is simply visited twice
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
            o2 = o1; o1 = o0; l2 = l1; l1 = l0; r2 = r1; r1 = r0;
                };
#pragma unroll 1
        for (int q = 0; q < R + 10; q += 2) {
            open_step(q);
            if (q + 1 < R + 10) open_step(q + 1);
        }
#pragma unroll
        for (int k = 0; k < SG_KO; ++k) {
            if (mo[k][0]) emit_mom(sid[k], mo[k]);
            if (act) { bot_mask[(size_t)k * NBW + bj] = pm[k]; bot_sid[(size_t)k * NBW + bj] = (unsigned short)sid[k]; }
        }
        if (e4) atomicAdd(&misc[0], e4);
        if (why) misc[6] = (int)why;
        __syncthreads();
        if (misc[6]) { hand_on(16u + (u32)misc[6]); return; }
        if (geo.stop == 12) return;
        // first row of the tile against the last row of the tile above, with the diagonal neighbours across the word edges
        if (hasu && firstB) {
            u64 N = firstB;
            u32 i = 0;
            while (N) {
                const u64 lowbit = N & (~N + 1ull), t2 = N + lowbit, gg = N & ~t2;
                N &= t2;
                const u64 ga = gg | (gg << 1) | (gg >> 1);
#pragma unroll
                for (int k = 0; k < SG_KO; ++k) {
                    if (ga & bot_mask[(size_t)k * NBW + bj - WW]) pq_push(Q, sbase + i, bot_sid[(size_t)k * NBW + bj - WW]);
                    if ((gg & 1ull) && hasl && (bot_mask[(size_t)k * NBW + bj - WW - 1] >> 63))
                        pq_push(Q, sbase + i, bot_sid[(size_t)k * NBW + bj - WW - 1]);
                    if ((gg >> 63) && hasr && (bot_mask[(size_t)k * NBW + bj - WW + 1] & 1ull))
                        pq_push(Q, sbase + i, bot_sid[(size_t)k * NBW + bj - WW + 1]);
                }
                ++i;
            }
        }
        __syncthreads();
        if (geo.stop == 13) return;
        ncomp = seg_resolve<NT>(P, sbase, min(nseg, (u32)SG_SEGMAX), nullptr, nullptr, 0, seg_pos, comp_pos, cidmap, tmp,
                            min((u32)maxm, (u32)CCL_OPEN_COMPS), Q);
        if (misc[5] > Q.cap) { hand_on(16u + SLOW_PAIRS); return; }
        if (ncomp == NONE32) { hand_on(16u + SLOW_NCOMP); return; }
        if ((int)ncomp - misc[0] / 4 != 0) {             // holes: RETR_EXTERNAL needs the fill passes of the general path
            hand_on(16u + SLOW_HOLES);
            return;
        }
    }
    if (geo.stop == 14) return;
    // ---- the component's first pixel (the moments' origin) -------------------------------------------------------------
    u32* anchor = reinterpret_cast<u32*>(accb);                                          // [CCL_OPEN_COMPS]  (y << 16) | x
    u64* acc = reinterpret_cast<u64*>(accb + 4 * CCL_OPEN_COMPS);                         // [mom_comps][NMOM]
    {
        u32* first = area_first + (int64_t)n * maxm;
        for (u32 c = tid; c < ncomp; c += NT) {
            const u32 pos = comp_pos[c], cid = cidmap[c], py = pos / (u32)W;
            anchor[cid] = (py << 16) | (pos - py * (u32)W);
            first[cid] = pos;
        }
    }
    // ---- segment moments -> component moments about its first pixel, `mom_comps` components per pass -----------------------------
    i64* as = area_sums + (int64_t)n * maxm * VBS_AREA_SUMS;
    const u32 nmrec = (u32)misc[7];
    for (u32 c0 = 0; c0 < ncomp; c0 += geo.mom_comps) {
        const u32 nc = min(geo.mom_comps, ncomp - c0);
        for (u32 c = tid; c < nc * NMOM; c += NT) acc[c] = 0;
        __syncthreads();
        for (u32 r = tid; r < nmrec; r += NT) {
            const uint4* src = reinterpret_cast<const uint4*>(mrec + (size_t)r * 16);
            const uint4 w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
            const u32 s = w0.x, cid = (u32)cidmap[P[s] & 0x7FFFu] - c0;
            if (cid >= nc) continue;                     // another pass's component
            const u32 ot = s / SG_SEGMAX, ol = ot & 63u, og = ol / (u32)WW;             // the thread that wrote it: its tile
            const int ox = 64 * (int)(ol - og * (u32)WW) + 32, oy = (int)((ot >> 6) * (u32)G + og) * R + (R >> 1);
            const u32 fp = anchor[cid + c0];
            const i64 m[NMOM] = {(int)w0.y, (int)w0.z, (int)w0.w, (int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w, (int)w2.x,
                                 (int)w2.y, (int)w2.z, (int)w2.w, (int)w3.x, (int)w3.y, (int)w3.z, (int)w3.w};
            i64 o[NMOM];
            shift_moments_i64(m, (i64)(ox - (int)(fp & 0xFFFFu)), (i64)(oy - (int)(fp >> 16)), o);
            u64* a = acc + cid * NMOM;
#pragma unroll
            for (int q = 0; q < NMOM; ++q)
                if (o[q]) atomicAdd(&a[q], (u64)o[q]);
        }
        __syncthreads();
        for (u32 c = tid; c < nc * NMOM; c += NT) as[(c0 + c / NMOM) * VBS_AREA_SUMS + (c % NMOM)] = (i64)acc[c];
        __syncthreads();
    }
    // ---- probes: segment -> component ----------------------------------------------------------------------------------
    for (u32 e = tid; e < nband * 4; e += NT) {
        const u32 v = pr[e];
        if (v != NONE16) pr[e] = cidmap[P[v] & 0x7FFFu];
    }
    if (tid == 0) { ncomp_all[n * 2 + 1] = ncomp; fstat[n * 8 + 6] = ncomp; fstat[n * 8 + 4] = 0; }
}

// ---- host side ----------------------------------------------------------------------------------------------------
// Threads per frame.  768 (12 waves: one workgroup per CU at three waves per SIMD) is the shape of a large frame.  A small
// one (the reference's 480x450 crop: 8 words wide) gives a 768-thread workgroup tiles of 5 rows that each walk 13 / 10 rows of
// halo as well: 18 row steps for 5 rows.  256 threads (four waves, one per SIMD; THREE workgroups per CU, so still three
// waves per SIMD, with the tables of a small frame cut to fit three in the LDS) have tiles of 15 rows: 28 steps for 15,
// 0.52 x the thread-steps per frame: k_stage 0.236 -> 0.183 us per 480x450 frame.  (384 threads, two workgroups per CU, were
// measured SLOWER than 768, 0.279: six waves do not spread evenly over four SIMDs.)  A frame that needs more than the
// smaller tables hold takes the general kernels, like any frame beyond the large ones; a handle whose max_markers makes
// the tables too large for three workgroups (1024) keeps 768.  VBS_OPT_STAGE_IMPL = 3 keeps 768 everywhere (test hook: the
// two shapes must agree bit for bit).
static int stage_threads(const vbs_handle* h, int nb) {
    if (h->stage_impl == 3) return ST_NT;
    const int G = 64 / h->WW, R768 = (h->H + (ST_NT / 64) * G - 1) / ((ST_NT / 64) * G);
    if (h->bp.ns == 8) return R768 <= 8 ? ST_NT_SMALL : ST_NT;
    // large frames: 256 threads have tiles of 86 rows at 1280x1024 - 99 row steps for 86 rows against 42 for 29, 0.78 x the
    // thread-steps: k_stage 0.76 -> 0.685 us per frame once a pass brings three frames per CU (768 frames).  At 512 frames
    // the kernel alone is slower (0.79: two workgroups = two waves per SIMD) and the call still faster (286 k against 280 k
    // frames/s): the slots it leaves free take the other pass stream's kernels.  Below that a frame on 768 threads is done
    // sooner (42 steps against 99).  VBS_OPT_STAGE_IMPL = 4: 256 for any pass (test hook).
    return (h->stage_impl == 4 || nb >= 512) ? ST_NT_SMALL : ST_NT;
}

// false = geometry outside the fused path (the round-2 kernels take it)
static bool stage_geom(const vbs_handle* h, StageGeom* g, size_t* lds_bytes, int nt) {
    if (h->W > 4096 || h->H > 2048 || h->maxm > 1024) return false;
    const int G = 64 / h->WW, NB = (nt / 64) * G;        // (WW <= 64: vbs_create)
    const int R = (h->H + NB - 1) / NB;
    if (R > 128 || R < 1) return false;                  // (row bit masks; int32 moments about the tile's centre)
    g->H = h->H; g->W = h->W; g->WW = h->WW; g->G = G; g->NB = NB; g->R = R; g->maxm = h->maxm;
    g->stop = VBS_KNOB("VBS_STAGE_STOP");
    g->retry = 0;
    const size_t NBW = (size_t)NB * h->WW;
    auto up16 = [](size_t x) { return (x + 15) / 16 * 16; };
    // table sizes.  768 threads: one workgroup per CU, the whole LDS is there to be used.  256: three per CU
    const bool half = nt < ST_NT;
    g->rec_cap = (u32)(half ? 768 : SG_REC);
    g->pq_cap = (u32)(half ? 1536 : SG_PQ);
    g->mom_comps = (u32)(half ? CCL_MOM_COMPS / 4 : CCL_MOM_COMPS);
    const size_t par = up16((size_t)nt * SG_SEGMAX * 2);
    size_t rec = up16((size_t)g->rec_cap * 18);
    if (rec < (size_t)nt * SG_SEGMAX * 4) rec = (size_t)nt * SG_SEGMAX * 4;           // (the opened mask's first-pixel table)
    const size_t bot = up16((size_t)SG_KB * NBW * 10);
    const size_t acc_band = up16((size_t)(8 * ((h->maxm + 1) / 2)) + 16 * (size_t)h->maxm);
    const size_t acc_open = up16((size_t)4 * CCL_OPEN_COMPS + (size_t)g->mom_comps * NMOM * 8);
    const size_t comp = 4096 + 2048 + (acc_band > acc_open ? acc_band : acc_open);
    const size_t botc = bot > comp ? bot : comp;
    const size_t pq = (size_t)g->pq_cap * 4, mb = (size_t)nt * 4 * (1 + ST_MB_CAP), misc = 32 * 4 + 16 * 4;
    g->off_rec = (u32)par;
    g->off_bot = (u32)(par + rec);
    g->off_pq = (u32)(par + rec + botc);
    g->off_mb = (u32)(g->off_pq + pq);
    g->off_tmp = (u32)(g->off_mb + mb);
    *lds_bytes = g->off_tmp + misc;
    // moment records live in the frame's slice of the general path's word table (idle on the fast path) - unless that
    // slice holds fewer than SG_REC of them: small frames with many blobs (the reference's real layout: 65 dots in 467x437,
    // tiles of 5 rows, ~650 records against the 437 its slice holds) have their own buffer (vbs_create: stage_mrec)
    g->mrec_stride = h->stage_mrec ? 16u * (u32)SG_REC : 2u * (u32)h->H * (u32)h->WW;
    g->mrec_cap = g->mrec_stride / 16u < (u32)SG_REC ? g->mrec_stride / 16u : (u32)SG_REC;
    return *lds_bytes <= (size_t)(half ? 160 * 1024 / 3 - 256 : 160 * 1024);
}

bool stage_supported(const vbs_handle* h) {
    StageGeom g;
    size_t lds;
    return stage_geom(h, &g, &lds, ST_NT);
}

template <int NS, int NT>
static bool stage_launch_t(vbs_handle* h, int nb, const StageGeom& g, size_t lds, hipStream_t s) {
    size_t& set = h->stage_lds_set[NT == ST_NT ? 0 : 1];     // (of the NS instances a handle runs only one)
    if (lds > set) {
        // (per thread count; of the NS instances a handle runs only one)
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_stage<NS, NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        set = lds;
    }
    VBS_LAUNCH(h, s, g.retry ? "k_stage_retry" : "k_stage", (k_stage<NS, NT>), dim3(nb), dim3(NT), lds, s, h->mask_bits, h->area_bits, h->ncomp,
               h->band_sums, h->area_first, h->area_sums, h->probe, h->fstat, h->slow_flag, h->slow_total,
               h->stage_mrec ? h->stage_mrec : h->wbase, g);
    return true;
}

// false: geometry outside the fused path (or the LDS it needs was refused): the caller runs the round-2 kernels
bool launch_stage(vbs_handle* h, int nb, hipStream_t s) {
    StageGeom g;
    size_t lds = 0;
    if (stage_threads(h, nb) < ST_NT && stage_geom(h, &g, &lds, ST_NT_SMALL) &&
        (h->bp.ns == 14 ? stage_launch_t<14, ST_NT_SMALL>(h, nb, g, lds, s) : stage_launch_t<8, ST_NT_SMALL>(h, nb, g, lds, s))) {
        // Its taller tiles and smaller tables give out earlier on dense layouts (17 x 17 dots at a pitch of 56 px in 1280x1024:
        // every frame); those frames get a second chance on 768 threads before the general kernels (11.1 -> 4.7 us per frame
        // there; a launch of workgroups that leave at once otherwise).
        if (stage_geom(h, &g, &lds, ST_NT)) {
            g.retry = 1;
            (void)(h->bp.ns == 14 ? stage_launch_t<14, ST_NT>(h, nb, g, lds, s) : stage_launch_t<8, ST_NT>(h, nb, g, lds, s));
        }
        return true;
    }
    if (!stage_geom(h, &g, &lds, ST_NT)) return false;
    return h->bp.ns == 14 ? stage_launch_t<14, ST_NT>(h, nb, g, lds, s) : stage_launch_t<8, ST_NT>(h, nb, g, lds, s);
}
