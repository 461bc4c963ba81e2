// a14 / f4: first-frame identity assignment on the device — `MarkerTracker._process_first_frame`
// (marker_detection.py:275-347), restated exactly as vision-basedsensor_amd/ids.py does on the host (id_mode
// "as_written" | "full", kmeans = "optimal"), so the reference table never leaves the GPU.
// One workgroup.  Every float64 step follows NumPy's operation order without contraction (mean by sequential row
// accumulation, norm = sqrt(x*x + y*y), cumulative sums left to right, sse = (c2_j - c2_i) - (c1_j - c1_i)^2 / cnt,
// first minimum wins), so layers and orders are those of the host path.
#include "common.h"

#define IDS_MAXN 1024
#define IDS_MAXK 16

__device__ __forceinline__ double dsq(double a) { return __dmul_rn(a, a); }

__global__ __launch_bounds__(256) void k_assign_ids(const double* __restrict__ det, const int32_t* __restrict__ count_p,
                                                    int num_layers, int full_mode, int32_t* __restrict__ ids_out,
                                                    double* __restrict__ xy_out, int cap, int32_t* __restrict__ m_out) {
    __shared__ double px[IDS_MAXN], py[IDS_MAXN], rad[IDS_MAXN], th[IDS_MAXN], srt[IDS_MAXN];
    __shared__ double c1[IDS_MAXN + 1], c2[IDS_MAXN + 1], cost[2][IDS_MAXN + 1];
    __shared__ unsigned short back[IDS_MAXK + 1][IDS_MAXN + 1], ord[IDS_MAXN], lay[IDS_MAXN], pos[IDS_MAXN];
    __shared__ int lcount[IDS_MAXK + 2], lstart[IDS_MAXK + 2], lbase[IDS_MAXK + 2], lslot[IDS_MAXK + 2];
    __shared__ double mean[2];
    __shared__ int ci_s;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int cnt_in = *count_p;
    if (cnt_in < 0) { if (tid == 0) *m_out = 1000 * cnt_in; return; }      // frame 0 carries a device status
    const int n = min(cnt_in, IDS_MAXN);
    if (n == 0) { if (tid == 0) *m_out = -1; return; }      // "No markers detected in first frame!"
    for (int i = tid; i < n; i += nthr) { px[i] = det[i * 6 + 0]; py[i] = det[i * 6 + 1]; }
    __syncthreads();
    if (tid < 2) {                                      // pts.mean(axis=0): rows added one after the other
        const double* v = tid ? py : px;
        double s = 0.0;
        for (int i = 0; i < n; ++i) s = __dadd_rn(s, v[i]);
        mean[tid] = __ddiv_rn(s, (double)n);
    }
    __syncthreads();
    for (int i = tid; i < n; i += nthr)
        rad[i] = __dsqrt_rn(__dadd_rn(dsq(__dsub_rn(px[i], mean[0])), dsq(__dsub_rn(py[i], mean[1]))));
    __syncthreads();
    if (tid == 0) {                                     // np.argmin: first minimum
        int b = 0;
        for (int i = 1; i < n; ++i) if (rad[i] < rad[b]) b = i;
        ci_s = b;
    }
    __syncthreads();
    const int ci = ci_s, nr = n - 1;                   // rest = every marker but ci, in detection order
    const double cx = px[ci], cy = py[ci];
    __syncthreads();
    for (int i = tid; i < nr; i += nthr) {
        const int src = i < ci ? i : i + 1;
        const double vx = __dsub_rn(px[src], cx), vy = __dsub_rn(py[src], cy);
        srt[i] = __dsqrt_rn(__dadd_rn(dsq(vx), dsq(vy)));          // radius (moved to rad[] below)
        th[i] = atan2(vy, vx);
    }
    __syncthreads();
    for (int i = tid; i < nr; i += nthr) rad[i] = srt[i];
    __syncthreads();
    const int k = max(1, min(min(num_layers, IDS_MAXK), nr));
    if (nr > 0) {
        // ---- kmeans_1d(radius, k): stable argsort, prefix sums, exact DP over contiguous partitions ----
        for (int i = tid; i < nr; i += nthr) {
            const double r = rad[i];
            int rank = 0;
            for (int j = 0; j < nr; ++j) rank += (rad[j] < r) || (rad[j] == r && j < i);
            ord[rank] = (unsigned short)i;
            srt[rank] = r;
        }
        __syncthreads();
        if (tid == 0) {
            c1[0] = 0.0; c2[0] = 0.0;
            for (int i = 0; i < nr; ++i) { c1[i + 1] = __dadd_rn(c1[i], srt[i]); c2[i + 1] = __dadd_rn(c2[i], dsq(srt[i])); }
        }
        for (int j = tid; j <= nr; j += nthr) cost[0][j] = j == 0 ? 0.0 : (double)INFINITY;
        __syncthreads();
        for (int c = 1; c <= k; ++c) {
            const double* prev = cost[(c - 1) & 1];
            double* cur = cost[c & 1];
            for (int j = tid; j <= nr; j += nthr) {
                double best = (double)INFINITY;
                int bi = 0;
                bool have = false;
                for (int i = 0; i <= nr; ++i) {
                    double sse = (double)INFINITY;
                    if (j > i) {
                        const double d1 = __dsub_rn(c1[j], c1[i]);
                        sse = __dsub_rn(__dsub_rn(c2[j], c2[i]), __ddiv_rn(dsq(d1), (double)(j - i)));
                    }
                    const double cand = __dadd_rn(prev[i], sse);
                    if (!have || cand < best) { best = cand; bi = i; have = true; }     // first minimum
                }
                cur[j] = best;
                back[c][j] = (unsigned short)bi;
            }
            __syncthreads();
        }
        if (tid == 0) {
            int j = nr;
            for (int c = k; c >= 1; --c) {
                const int i = back[c][j];
                for (int q = i; q < j; ++q) lay[ord[q]] = (unsigned short)c;          // layer = label + 1
                j = i;
            }
        }
        __syncthreads();
    }
    // ---- table in dict order: (0,0), then layer-major ----
    if (tid <= IDS_MAXK + 1) { lcount[tid] = 0; lslot[tid] = -1; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 0; i < nr; ++i) { lcount[lay[i]]++; lslot[lay[i]] = i; }           // slot = last member in detection order
        int base = 1;
        for (int l = 1; l <= k; ++l) {
            lbase[l] = base;
            base += full_mode ? lcount[l] : (lcount[l] > 0 ? 1 : 0);
        }
        lbase[k + 1] = base;
    }
    __syncthreads();
    const int M = lbase[k + 1];
    if (M > cap) { if (tid == 0) *m_out = -2; return; }
    if (tid == 0) {
        ids_out[0] = 0; ids_out[1] = 0; xy_out[0] = cx; xy_out[1] = cy;
        *m_out = M;
    }
    if (!full_mode) {
        for (int l = 1 + tid; l <= k; l += nthr)
            if (lslot[l] >= 0) {
                const int i = lslot[l], src = i < ci ? i : i + 1, o = lbase[l];
                ids_out[2 * o] = l; ids_out[2 * o + 1] = 0;
                xy_out[2 * o] = px[src]; xy_out[2 * o + 1] = py[src];
            }
        return;
    }
    // full: members of a layer sorted by angle (stable), index 0 at the smallest |angle| (first in sorted order)
    for (int i = tid; i < nr; i += nthr) {
        const double t = th[i];
        const int l = lay[i];
        int rank = 0;
        for (int j = 0; j < nr; ++j) rank += (lay[j] == l) && ((th[j] < t) || (th[j] == t && j < i));
        pos[i] = (unsigned short)rank;
    }
    if (tid <= IDS_MAXK + 1) lstart[tid] = 0;
    __syncthreads();
    if (tid >= 1 && tid <= k && lcount[tid] > 0) {      // argmin |theta| over the sorted members: first minimum
        double best = (double)INFINITY;
        int bp = 0;
        for (int p = 0; p < lcount[tid]; ++p)
            for (int i = 0; i < nr; ++i)
                if (lay[i] == tid && pos[i] == p) { if (fabs(th[i]) < best) { best = fabs(th[i]); bp = p; } break; }
        lstart[tid] = bp;
    }
    __syncthreads();
    for (int i = tid; i < nr; i += nthr) {
        const int l = lay[i], cnt = lcount[l], p = pos[i], src = i < ci ? i : i + 1, o = lbase[l] + p;
        int idx = (p - lstart[l]) % cnt;
        if (idx < 0) idx += cnt;
        ids_out[2 * o] = l; ids_out[2 * o + 1] = idx;
        xy_out[2 * o] = px[src]; xy_out[2 * o + 1] = py[src];
    }
}

void launch_assign_ids(vbs_handle* h, const double* det, const int32_t* count, int num_layers, int full_mode,
                       int32_t* ids_out, double* xy_out, int cap, int32_t* m_out, hipStream_t s) {
    VBS_LAUNCH(h, s, "k_assign_ids", k_assign_ids, dim3(1), dim3(256), 0, s, det, count, num_layers, full_mode, ids_out,
               xy_out, cap, m_out);
}
