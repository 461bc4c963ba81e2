// a15 (+ a19 / a20 fused): the tracking of one frame against the reference IDs, shared by k_track (k_solve.hip) and the
// one-launch form of the few-frames path, k_finalize_track (k_label.hip).  Float64 without contraction, as the reference
// computes; the contraction mode of the including file is restored to the compiler's default at the end.
#pragma once
#include "common.h"

struct CamD {
    float fx, fy, cx, cy;
    double k[5];
    double R[9], T[3];
    float dmm;
    int has_dist;
};

static CamD make_cam(const vbs_camera& c) {
    CamD d;
    d.fx = c.K[0]; d.fy = c.K[4]; d.cx = c.K[2]; d.cy = c.K[5];
    d.has_dist = 0;
    for (int i = 0; i < 5; ++i) { d.k[i] = (double)c.dist[i]; if (c.dist[i] != 0.f) d.has_dist = 1; }
    for (int i = 0; i < 9; ++i) d.R[i] = (double)c.R[i];
    for (int i = 0; i < 3; ++i) d.T[i] = (double)c.T[i];
    d.dmm = c.marker_diameter_mm;
    return d;
}

#pragma clang fp contract(off)
// cv2.undistortPoints(pts, K, dist, None, K): 5 fixed-point iterations of the inverse Brown-Conrady model
static __device__ void undistort_point(const CamD& c, double u, double v, double* uo, double* vo) {
    double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy;
    double x0 = (u - cx) / fx, y0 = (v - cy) / fy;
    double x = x0, y = y0;
    if (c.has_dist) {
        for (int it = 0; it < 5; ++it) {
            double r2 = x * x + y * y;
            double icd = 1.0 / (1.0 + ((c.k[4] * r2 + c.k[1]) * r2 + c.k[0]) * r2);
            double dxx = 2.0 * c.k[2] * x * y + c.k[3] * (r2 + 2.0 * x * x);
            double dyy = c.k[2] * (r2 + 2.0 * y * y) + 2.0 * c.k[3] * x * y;
            x = (x0 - dxx) * icd;
            y = (y0 - dyy) * icd;
        }
    }
    *uo = x * fx + cx;
    *vo = y * fy + cy;
}

// _calculate_3d_position with NumPy's promotion rules: f_avg, 2.0/f_avg and f_avg**2 are float32
// (float32 scalars with Python numbers), everything that touches u, v or d is float64.
static __device__ bool solve_marker(const CamD& c, double u, double v, double d, double* X) {
    float f_avg = (c.fx + c.fy) / 2.0f;
    double du = u - (double)c.cx, dv = v - (double)c.cy;
    double Rr = sqrt(du * du + dv * dv);
    if (Rr < 1e-6) return false;
    float k32 = c.dmm / f_avg;
    float f2 = f_avg * f_avg;
    double d_eff = (double)k32 * sqrt(Rr * Rr + (double)f2);
    double h = (double)f_avg * (d_eff / d);
    double pc[3] = {h * du / (double)c.fx - c.T[0], h * dv / (double)c.fy - c.T[1], h - c.T[2]};
    for (int i = 0; i < 3; ++i) X[i] = c.R[0 * 3 + i] * pc[0] + c.R[1 * 3 + i] * pc[1] + c.R[2 * 3 + i] * pc[2];
    return isfinite(X[0]) && isfinite(X[1]) && isfinite(X[2]);
}

// marker_detection.py:349-396 (_track_markers) [+ the fused 3-D solve] for frame n: one workgroup of 256 threads, thread per reference ID.
// The reference takes the nearest of ALL detections and drops it when it is farther than min_dist; only a detection within
// min_dist can therefore be reported, and those lie in the 3 x 3 cells around the reference position of a grid whose pitch is
// at least min_dist + 1.  The detections are hung into such a grid in LDS (32 x 32 buckets, cell coordinates taken modulo the
// grid: an aliased cell only adds candidates that the distance then rejects), and a reference ID looks at ~ 9 short lists
// instead of every detection - one frame per call spent 10 of its 23 us here.  Coordinates beyond +-2^30 cells, a min_dist that
// is not in [0, 4095] or NaN: the plain scan.
#define TRK_GX 32
#define TRK_GY 32
__device__ __forceinline__ void track_frame(int n, const double* __restrict__ det64, const int32_t* __restrict__ counts, int maxm,
                                            const double* __restrict__ ref_xy, int m_ref, double min_dist,
                                            float* __restrict__ table, int do3d, const CamD& cam, double min_size) {
    __shared__ double mx[1024], my[1024];
    __shared__ unsigned int cell_head[TRK_GX * TRK_GY];          // (32-bit: LDS atomics)
    __shared__ unsigned short cell_next[1024];
    __shared__ int grid_off;
    const int tid = threadIdx.x;
    const int cnt = min(max(counts[n], 0), min(maxm, 1024));     // a status (< 0) tracks nothing; never past the tables
    const bool want_grid = min_dist >= 0.0 && min_dist <= 4095.0 && cnt > 16;      // (false for NaN)
    int sh = 5;
    while (want_grid && (double)(1 << sh) < min_dist + 1.0) ++sh;
    const double inv_pitch = 1.0 / (double)(1 << sh);
    if (want_grid) {
        for (int i = tid; i < TRK_GX * TRK_GY; i += blockDim.x) cell_head[i] = 0xFFFFu;
        if (tid == 0) grid_off = 0;
    }
    for (int i = tid; i < cnt; i += blockDim.x) {
        mx[i] = det64[((int64_t)n * maxm + i) * 6 + 0];
        my[i] = det64[((int64_t)n * maxm + i) * 6 + 1];
    }
    __syncthreads();
    if (want_grid) {
        // lists in descending index order would need a sort; the order inside a list does not matter (a tie of the two
        // smallest distances goes to the reference's own loop below)
        for (int i = tid; i < cnt; i += blockDim.x) {
            const double fx = floor(mx[i] * inv_pitch), fy = floor(my[i] * inv_pitch);
            if (!(fabs(fx) < 0x1p30 && fabs(fy) < 0x1p30)) { grid_off = 1; continue; }
            const int b = ((int)fy & (TRK_GY - 1)) * TRK_GX + ((int)fx & (TRK_GX - 1));
            cell_next[i] = (unsigned short)atomicExch(&cell_head[b], (unsigned int)i);
        }
    }
    __syncthreads();
    for (int r = tid; r < m_ref; r += blockDim.x) {
        double ox = ref_xy[2 * r], oy = ref_xy[2 * r + 1];
        // cdist 'euclidean' + argmin (the first minimum of the rounded distances).  The square root is monotone, so the
        // minimum distance is the root of the minimum squared distance, taken once; only a detection whose squared distance
        // lies within a few units in the last place of that minimum could round to the same root and win on its index, so
        // the runner-up is tracked too and the reference's loop over rounded roots runs only for such a near tie (never on
        // marker frames) - instead of 169 float64 square roots per reference ID
        double m2 = 1e300, m2b = 1e300;
        int bi = -1;
        const double fx = floor(ox * inv_pitch), fy = floor(oy * inv_pitch);
        if (want_grid && !grid_off && fabs(fx) < 0x1p30 && fabs(fy) < 0x1p30) {
            const int cx0 = (int)fx, cy0 = (int)fy;
#pragma unroll 1
            for (int k9 = 0; k9 < 9; ++k9) {
                const int b = ((cy0 + k9 / 3 - 1) & (TRK_GY - 1)) * TRK_GX + ((cx0 + k9 % 3 - 1) & (TRK_GX - 1));
                for (int i = (int)cell_head[b]; i != 0xFFFF; i = cell_next[i]) {
                    const double dx = ox - mx[i], dy = oy - my[i], d2 = dx * dx + dy * dy;
                    const bool lt = d2 < m2;
                    m2b = lt ? m2 : (d2 < m2b ? d2 : m2b);
                    bi = lt ? i : bi;
                    m2 = lt ? d2 : m2;
                }
            }
        } else {
            for (int i = 0; i < cnt; ++i) {
                const double dx = ox - mx[i], dy = oy - my[i], d2 = dx * dx + dy * dy;
                const bool lt = d2 < m2;
                m2b = lt ? m2 : (d2 < m2b ? d2 : m2b);
                bi = lt ? i : bi;
                m2 = lt ? d2 : m2;
            }
        }
        double best = bi >= 0 ? sqrt(m2) : 1e300;
        if (bi >= 0 && m2b <= m2 * (1.0 + 0x1p-48)) {
            best = 1e300; bi = -1;
            for (int i = 0; i < cnt; ++i) {
                const double dx = ox - mx[i], dy = oy - my[i];
                const double ds = sqrt(dx * dx + dy * dy);
                if (ds < best) { best = ds; bi = i; }
            }
        }
        float* row = table + ((int64_t)n * m_ref + r) * VBS_TABLE_COLS;
        float o[VBS_TABLE_COLS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (bi >= 0 && !(best > min_dist)) {
            const double* d = det64 + ((int64_t)n * maxm + bi) * 6;
            double major = d[2], minor = d[3], ang = d[4];
            int flags = VBS_FLAG_TRACKED;
            o[1] = (float)mx[bi]; o[2] = (float)my[bi]; o[3] = (float)major; o[4] = (float)minor;
            o[5] = (float)ang; o[9] = (float)bi;
            if (do3d && major >= min_size) {
                double u, v, X[3];
                undistort_point(cam, mx[bi], my[bi], &u, &v);
                if (solve_marker(cam, u, v, major, X)) {
                    flags |= VBS_FLAG_XYZ;
                    o[6] = (float)X[0]; o[7] = (float)X[1]; o[8] = (float)X[2];
                }
            }
            o[0] = (float)flags;
        }
#pragma unroll
        for (int c = 0; c < VBS_TABLE_COLS; ++c) row[c] = o[c];
    }
}

#pragma clang fp contract(fast)
