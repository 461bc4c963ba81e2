// a15, a19-a21, f1: identity tracking, undistort + closed-form 3-D solve, last-seen displacement,
// plane-fit pose.  All arithmetic in float64 (the reference is float64 fed by float32 camera
// parameters); tables are stored as float32 (the all-gather format).
//   k_track        marker_detection.py:349-396 (_track_markers) [+ fused 3-D solve]
//   k_solve3d      3d_reconstruction.py:185-238 (_undistort_points, _calculate_3d_position)
//   k_displacement 3d_reconstruction.py:240-316 (_track_markers)
//   k_plane_fit    ForceDistribution.py:138-162 (fit_plane_least_squares)
//   k_deviation_plane  ForceDistribution.py:168-208 (deviation field), :218-243 (end points + plane), :262-268, :274
#include <algorithm>

#include "track_common.h"
#pragma clang fp contract(off)

// one workgroup per frame; thread per reference ID
__global__ __launch_bounds__(256) void k_track(const double* __restrict__ det64,
                                               const int32_t* __restrict__ counts, int maxm,
                                               const double* __restrict__ ref_xy, int m_ref, double min_dist,
                                               float* __restrict__ table, int do3d, CamD cam,
                                               double min_size) {
    track_frame(blockIdx.x, det64, counts, maxm, ref_xy, m_ref, min_dist, table, do3d, cam, min_size);
}

__global__ __launch_bounds__(256) void k_solve3d(float* __restrict__ table, int64_t rows, CamD cam,
                                                 double min_size) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    float* row = table + i * VBS_TABLE_COLS;
    int flags = (int)row[0];
    row[6] = row[7] = row[8] = 0.f;
    flags &= ~VBS_FLAG_XYZ;
    if ((flags & VBS_FLAG_TRACKED) && (double)row[3] >= min_size) {
        double u, v, X[3];
        undistort_point(cam, (double)row[1], (double)row[2], &u, &v);
        if (solve_marker(cam, u, v, (double)row[3], X)) {
            flags |= VBS_FLAG_XYZ;
            row[6] = (float)X[0]; row[7] = (float)X[1]; row[8] = (float)X[2];
        }
    }
    row[0] = (float)flags;
}

// first frame that holds any row surviving load_marker_data's size filter (:172-176) -> *fmin
template <typename TT>
__global__ __launch_bounds__(256) void k_disp_first(const TT* __restrict__ table, int n, int m_ref, double min_size,
                                                    int* __restrict__ fmin) {
    int f = blockIdx.x;
    bool any = false;
    for (int r = threadIdx.x; r < m_ref && !any; r += 256) {
        const TT* row = table + ((int64_t)f * m_ref + r) * VBS_TABLE_COLS;
        any = ((int)row[0] & VBS_FLAG_TRACKED) && (double)row[3] >= min_size;
    }
    if (__syncthreads_or(any) && threadIdx.x == 0) atomicMin(fmin, f);
}

// Last-seen displacement (:263-314), chunk-parallel: block = (chunk of CH frames) x (256 reference IDs).  A lane
// first looks BACK from its chunk for the frame in which its ID was last seen (the reference's marker_dict entry),
// then walks its chunk forward.  Emits frames [f0, f1) of a table that holds frames [0, n).
template <typename TT>
__global__ __launch_bounds__(256) void k_displacement(const TT* __restrict__ table, int n, int m_ref, int warmup,
                                                      double min_size, double max_disp, int f0, int f1,
                                                      const int* __restrict__ fmin_p, TT* __restrict__ disp) {
    constexpr int CH = 32;
    const int r = blockIdx.y * 256 + threadIdx.x;
    if (r >= m_ref) return;
    const int fmin = *fmin_p;
    const int64_t fstart = (fmin >= 0x7f7f7f7f) ? (int64_t)n : (int64_t)fmin + max(warmup, 0);
    const int c0 = f0 + blockIdx.x * CH, c1 = min(c0 + CH, f1);
    bool have = false, last_ok = false;
    double L[3] = {0, 0, 0};
    for (int f = c0 - 1; f >= 0 && f >= fstart; --f) {             // carry-in
        const TT* row = table + ((int64_t)f * m_ref + r) * VBS_TABLE_COLS;
        int flags = (int)row[0];
        if ((flags & VBS_FLAG_TRACKED) && (double)row[3] >= min_size) {
            have = true; last_ok = flags & VBS_FLAG_XYZ;
            L[0] = (double)row[6]; L[1] = (double)row[7]; L[2] = (double)row[8];
            break;
        }
    }
    for (int f = c0; f < c1; ++f) {
        const TT* row = table + ((int64_t)f * m_ref + r) * VBS_TABLE_COLS;
        TT* o = disp + ((int64_t)(f - f0) * m_ref + r) * VBS_DISP_COLS;
        TT out[VBS_DISP_COLS] = {0, 0, 0, 0, 0};
        int flags = (int)row[0];
        bool present = (flags & VBS_FLAG_TRACKED) && (double)row[3] >= min_size && f >= fstart;
        if (present) {
            bool ok = flags & VBS_FLAG_XYZ;
            double C[3] = {(double)row[6], (double)row[7], (double)row[8]};
            if (have && last_ok && ok) {
                double dx = C[0] - L[0], dy = C[1] - L[1], dz = C[2] - L[2];
                double mm = sqrt(dx * dx + dy * dy + dz * dz);
                if (!(mm > max_disp)) {
                    out[0] = (TT)1; out[1] = (TT)dx; out[2] = (TT)dy; out[3] = (TT)dz; out[4] = (TT)mm;
                }
            }
            have = true; last_ok = ok;
            L[0] = C[0]; L[1] = C[1]; L[2] = C[2];
        }
        for (int c = 0; c < VBS_DISP_COLS; ++c) o[c] = out[c];
    }
}

// one wave per frame: normal equations of Z = aX + bY + c about the centroid
__global__ __launch_bounds__(64) void k_plane_fit(const float* __restrict__ table, int m_ref,
                                                  float* __restrict__ plane) {
    const int n = blockIdx.x, lane = threadIdx.x;
    const float* t = table + (int64_t)n * m_ref * VBS_TABLE_COLS;
    double s[4] = {0, 0, 0, 0};                          // n, X, Y, Z
    for (int r = lane; r < m_ref; r += 64) {
        const float* row = t + r * VBS_TABLE_COLS;
        if ((int)row[0] & VBS_FLAG_XYZ) { s[0] += 1; s[1] += row[6]; s[2] += row[7]; s[3] += row[8]; }
    }
    for (int q = 0; q < 4; ++q)
        for (int off = 32; off >= 1; off >>= 1) s[q] += __shfl_xor(s[q], off);
    double cnt = s[0];
    double mxv = cnt > 0 ? s[1] / cnt : 0, myv = cnt > 0 ? s[2] / cnt : 0, mzv = cnt > 0 ? s[3] / cnt : 0;
    double c[5] = {0, 0, 0, 0, 0};                       // xx, xy, yy, xz, yz (centred)
    for (int r = lane; r < m_ref; r += 64) {
        const float* row = t + r * VBS_TABLE_COLS;
        if ((int)row[0] & VBS_FLAG_XYZ) {
            double x = row[6] - mxv, y = row[7] - myv, z = row[8] - mzv;
            c[0] += x * x; c[1] += x * y; c[2] += y * y; c[3] += x * z; c[4] += y * z;
        }
    }
    for (int q = 0; q < 5; ++q)
        for (int off = 32; off >= 1; off >>= 1) c[q] += __shfl_xor(c[q], off);
    if (lane == 0) {
        float* o = plane + (int64_t)n * VBS_PLANE_COLS;
        double det = c[0] * c[2] - c[1] * c[1];
        o[0] = (float)cnt;
        if (cnt >= 3 && fabs(det) > 1e-300) {
            double a = (c[3] * c[2] - c[4] * c[1]) / det;
            double b = (c[4] * c[0] - c[3] * c[1]) / det;
            double cc = mzv - a * mxv - b * myv;
            o[1] = (float)a; o[2] = (float)b; o[3] = (float)cc;
            o[4] = (float)(atan(sqrt(a * a + b * b)) * 57.29577951308232);
        } else {
            o[1] = o[2] = o[3] = o[4] = 0.f;
        }
    }
}

// The deviation field between a tilted and a vertical loading (process_marker_data :196-204) and the plane through its end
// points (visualize_deviations :218-243): per marker present in all four rows, deviation = (tilt_end - tilt_start) -
// (vert_end - vert_start); end point = reference position (Z = 0 in 'plane' mode :222) + scale * deviation; plane and tilt
// over the end points as fit_plane_least_squares; mean scaled deviation (:263) and mean magnitude (:274).  One wave.
__global__ __launch_bounds__(64) void k_deviation_plane(const float* __restrict__ vs, const float* __restrict__ ve,
                                                        const float* __restrict__ ts, const float* __restrict__ te,
                                                        const float* __restrict__ ref, int m_ref, int shell, double scale,
                                                        float* __restrict__ dev, float* __restrict__ out) {
    const int lane = threadIdx.x;
    auto row_ok = [](const float* r) { return ((int)r[0] & VBS_FLAG_XYZ) != 0; };
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};              // n, end X, end Y, end Z, dX, dY, dZ, |d|
    for (int r = lane; r < m_ref; r += 64) {
        const float *a = vs + r * VBS_TABLE_COLS, *b = ve + r * VBS_TABLE_COLS, *c = ts + r * VBS_TABLE_COLS, *d = te + r * VBS_TABLE_COLS;
        float* o = dev + r * 4;
        if (row_ok(a) && row_ok(b) && row_ok(c) && row_ok(d)) {
            const double dx = ((double)d[6] - c[6]) - ((double)b[6] - a[6]);
            const double dy = ((double)d[7] - c[7]) - ((double)b[7] - a[7]);
            const double dz = ((double)d[8] - c[8]) - ((double)b[8] - a[8]);
            o[0] = 1.f; o[1] = (float)dx; o[2] = (float)dy; o[3] = (float)dz;
            s[0] += 1;
            s[1] += ref[r * 3 + 0] + scale * dx; s[2] += ref[r * 3 + 1] + scale * dy;
            s[3] += (shell ? (double)ref[r * 3 + 2] : 0.0) + scale * dz;
            s[4] += scale * dx; s[5] += scale * dy; s[6] += scale * dz;
            s[7] += sqrt(dx * dx + dy * dy + dz * dz);
        } else {
            o[0] = o[1] = o[2] = o[3] = 0.f;
        }
    }
    for (int q = 0; q < 8; ++q)
        for (int off = 32; off >= 1; off >>= 1) s[q] += __shfl_xor(s[q], off);
    const double cnt = s[0];
    const double mxv = cnt > 0 ? s[1] / cnt : 0, myv = cnt > 0 ? s[2] / cnt : 0, mzv = cnt > 0 ? s[3] / cnt : 0;
    __syncthreads();                                     // (dev[] rows written above are read again below)
    double c[5] = {0, 0, 0, 0, 0};                       // xx, xy, yy, xz, yz of the end points (centred)
    for (int r = lane; r < m_ref; r += 64) {
        const float* o = dev + r * 4;
        if (o[0] != 0.f) {
            const double x = ref[r * 3 + 0] + scale * o[1] - mxv, y = ref[r * 3 + 1] + scale * o[2] - myv;
            const double z = (shell ? (double)ref[r * 3 + 2] : 0.0) + scale * o[3] - mzv;
            c[0] += x * x; c[1] += x * y; c[2] += y * y; c[3] += x * z; c[4] += y * z;
        }
    }
    for (int q = 0; q < 5; ++q)
        for (int off = 32; off >= 1; off >>= 1) c[q] += __shfl_xor(c[q], off);
    if (lane == 0) {
        const double det = c[0] * c[2] - c[1] * c[1];
        out[0] = (float)cnt;
        if (cnt >= 3 && fabs(det) > 1e-300) {
            const double a = (c[3] * c[2] - c[4] * c[1]) / det, b = (c[4] * c[0] - c[3] * c[1]) / det;
            out[1] = (float)a; out[2] = (float)b; out[3] = (float)(mzv - a * mxv - b * myv);
            out[4] = (float)(atan(sqrt(a * a + b * b)) * 57.29577951308232);
        } else {
            out[1] = out[2] = out[3] = out[4] = 0.f;
        }
        out[5] = cnt > 0 ? (float)(s[4] / cnt) : 0.f; out[6] = cnt > 0 ? (float)(s[5] / cnt) : 0.f;
        out[7] = cnt > 0 ? (float)(s[6] / cnt) : 0.f; out[8] = cnt > 0 ? (float)(s[7] / cnt) : 0.f;
    }
}

void launch_deviation_plane(vbs_handle* h, const float* vs, const float* ve, const float* ts, const float* te, const float* ref,
                            int m_ref, int shell, double scale, float* dev, float* out, hipStream_t s) {
    VBS_LAUNCH(h, s, "k_deviation_plane", k_deviation_plane, dim3(1), dim3(64), 0, s, vs, ve, ts, te, ref, m_ref, shell, scale,
               dev, out);
}

// float64 point interfaces: which = 0 undistort [n,2] -> [n,2]; which = 1 (u,v,d) [n,3] -> xyz [n,3], ok [n]
__global__ __launch_bounds__(256) void k_points(int which, const double* __restrict__ in, int n, CamD cam,
                                                double* __restrict__ out, int32_t* __restrict__ ok) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (which == 0) {
        undistort_point(cam, in[2 * i], in[2 * i + 1], &out[2 * i], &out[2 * i + 1]);
    } else {
        double X[3] = {0, 0, 0};
        bool good = solve_marker(cam, in[3 * i], in[3 * i + 1], in[3 * i + 2], X);
        out[3 * i] = X[0]; out[3 * i + 1] = X[1]; out[3 * i + 2] = X[2];
        ok[i] = good ? 1 : 0;
    }
}

void launch_points(int which, const double* in, int n, const vbs_camera& cam, double* out, int32_t* ok,
                   hipStream_t s) {
    hipLaunchKernelGGL(k_points, dim3((n + 255) / 256), dim3(256), 0, s, which, in, n, make_cam(cam), out, ok);
}

void launch_track(vbs_handle* h, const double* det, const int32_t* counts32, int nb,
                  const double* ref_xy, int m_ref, double min_dist, float* table, hipStream_t s) {
    CamD cam{};
    VBS_LAUNCH(h, s, "k_track", k_track, dim3(nb), dim3(256), 0, s, det, counts32, h->maxm,
                       ref_xy, m_ref, min_dist, table, 0, cam, 0.0);
}

void launch_track_fused(vbs_handle* h, int nb, const double* ref_xy, int m_ref, double min_dist,
                        float* table, const vbs_camera* cam, double min_size, hipStream_t s) {
    CamD c{};
    if (cam) c = make_cam(*cam);
    VBS_LAUNCH(h, s, "k_track", k_track, dim3(nb), dim3(256), 0, s, (const double*)h->det64,
                       (const int32_t*)h->cnt, h->maxm, ref_xy, m_ref, min_dist, table, cam ? 1 : 0, c, min_size);
}

void launch_solve3d(vbs_handle* h, float* table, int n, int m_ref, const vbs_camera& cam, double min_size,
                    hipStream_t s) {
    int64_t rows = (int64_t)n * m_ref;
    VBS_LAUNCH(h, s, "k_solve3d", k_solve3d, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, table, rows,
                       make_cam(cam), min_size);
}

__global__ __launch_bounds__(256) void k_fill_u32(u32* __restrict__ p, u32 value, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = value;
}

void launch_fill(u32* p, u32 value, size_t n, hipStream_t s) {
    if (!n) return;
    hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)), dim3(256), 0, s, p, value, n);
}

void launch_displacement(vbs_handle* h, const float* table, int n, int m_ref, int warmup, double min_size,
                         double max_disp, int f0, int f1, float* disp, hipStream_t s) {
    int* fmin = reinterpret_cast<int*>(h->fstat + (size_t)h->maxb * 8);      // one spare word behind the counters
    launch_fill(reinterpret_cast<u32*>(fmin), 0x7f7f7f7fu, 1, s);             // "no frame"
    VBS_LAUNCH(h, s, "k_disp_first", k_disp_first<float>, dim3(n), dim3(256), 0, s, table, n, m_ref, min_size, fmin);
    dim3 grid((f1 - f0 + 31) / 32, (m_ref + 255) / 256);
    VBS_LAUNCH(h, s, "k_displacement", k_displacement<float>, grid, dim3(256), 0, s, table, n, m_ref, warmup, min_size,
               max_disp, f0, f1, fmin, disp);
}

void launch_displacement64(const double* table, int n, int m_ref, int warmup, double min_size, double max_disp,
                           double* disp, int* fmin_scratch, hipStream_t s) {
    launch_fill(reinterpret_cast<u32*>(fmin_scratch), 0x7f7f7f7fu, 1, s);
    hipLaunchKernelGGL(k_disp_first<double>, dim3(n), dim3(256), 0, s, table, n, m_ref, min_size, fmin_scratch);
    dim3 grid((n + 31) / 32, (m_ref + 255) / 256);
    hipLaunchKernelGGL(k_displacement<double>, grid, dim3(256), 0, s, table, n, m_ref, warmup, min_size, max_disp, 0, n,
                       fmin_scratch, disp);
}

void launch_plane_fit(vbs_handle* h, const float* table, int n, int m_ref, float* plane, hipStream_t s) {
    VBS_LAUNCH(h, s, "k_plane_fit", k_plane_fit, dim3(n), dim3(64), 0, s, table, m_ref, plane);
}
