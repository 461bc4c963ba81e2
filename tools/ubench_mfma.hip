// Can the vector ALU of a SIMD issue while its matrix pipe runs?  (gfx950; what bounds k_ncc_mfma / k_blur16)
// A workgroup of 8 waves = 2 waves per SIMD; each wave runs a chain of matrix instructions, a chain of vector
// instructions, or both interleaved, per the mode of its half (waves 0-3 = one wave per SIMD, waves 4-7 the other).
//   cycles per iteration per SIMD for: M alone, V alone, M (wave A) beside V (wave B), M and V interleaved in ONE wave,
//   independent against dependent matrix chains, float16 16x16x32 and int8 16x16x64.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o tools/ubench_mfma   (binary not tracked)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

// what a wave does per iteration: NM matrix instructions (DEP: one accumulator chain, else 4 independent accumulators)
// and NV vector fmas (4 independent chains)
template <int NM, int NV, bool DEP, bool I8>
__device__ __forceinline__ float body(int iters, float seed) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(1.0f / (seed + i + 1)); }
    i4 ai = {1, 2, 3, 4}, bi = {5, 6, 7, 8};
    f4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    i4 iacc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[4] = {seed, seed + 1, seed + 2, seed + 3};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < (NM > NV ? NM : NV); ++k) {
            if (k < NM) {
                if (I8) iacc[DEP ? 0 : k & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ai, bi, iacc[DEP ? 0 : k & 3], 0, 0, 0);
                else acc[DEP ? 0 : k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[DEP ? 0 : k & 3], 0, 0, 0);
            }
            if (k < NV) v[k & 3] = __builtin_fmaf(v[k & 3], 1.0001f, 0.5f);
        }
    }
    float s = v[0] + v[1] + v[2] + v[3];
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3] + (float)iacc[j][1];
    return s;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    const int half = threadIdx.x >> 8;                   // waves 0-3 / 4-7: one wave per SIMD each
    float r = 0;
    if (MODE == 0) r = half == 0 ? body<16, 0, false, false>(iters, seed) : 0.0f;                        // M alone (independent)
    if (MODE == 1) r = half == 0 ? body<16, 0, true, false>(iters, seed) : 0.0f;                         // M alone (one chain)
    if (MODE == 2) r = half == 0 ? body<0, 64, false, false>(iters, seed) : 0.0f;                        // V alone: 64 fmas
    if (MODE == 3) r = half == 0 ? body<16, 0, false, false>(iters, seed) : body<0, 64, false, false>(iters, seed);   // M beside V
    if (MODE == 4) r = half == 0 ? body<16, 64, false, false>(iters, seed) : 0.0f;                       // M and V in one wave
    if (MODE == 5) r = half == 0 ? body<16, 0, true, false>(iters, seed) : body<0, 64, false, false>(iters, seed);    // chain beside V
    if (MODE == 6) r = body<16, 0, false, false>(iters, seed);                                           // M in both waves
    if (MODE == 7) r = body<16, 0, true, false>(iters, seed);                                            // a chain in both waves
    if (MODE == 8) r = half == 0 ? body<16, 0, false, true>(iters, seed) : 0.0f;                         // int8 M alone
    if (MODE == 9) r = half == 0 ? body<16, 0, false, true>(iters, seed) : body<0, 64, false, false>(iters, seed);    // int8 M beside V
    if (MODE == 10) r = body<16, 64, false, false>(iters, seed);                                         // M + V in both waves
    if (MODE == 11) r = body<16, 64, true, false>(iters, seed);                                          // chain + V in both waves
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE> void run(const char* name) {
    float* d; (void)hipMalloc(&d, 256 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, 10, 1.5f);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters, 1.5f); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms  = %.1f ns per iteration\n", name, ms, ms * 1e6 / iters);
    (void)hipFree(d);
}
int main() {
    printf("one workgroup of 8 waves per CU (2 waves per SIMD); an iteration = 16 matrix instructions and / or 64 v_fma_f32\n");
    run<0>("16 MFMA f16 16x16x32, independent, alone");
    run<1>("16 MFMA f16, one dependent chain, alone");
    run<2>("64 v_fma alone");
    run<3>("16 MFMA (wave A) beside 64 v_fma (wave B)");
    run<4>("16 MFMA + 64 v_fma interleaved in one wave");
    run<5>("dependent chain (A) beside 64 v_fma (B)");
    run<6>("16 MFMA independent in both waves");
    run<7>("a dependent chain in both waves");
    run<8>("16 MFMA i8 16x16x64 independent, alone");
    run<9>("16 MFMA i8 (A) beside 64 v_fma (B)");
    run<10>("16 MFMA + 64 v_fma in both waves");
    run<11>("dependent chain + 64 v_fma in both waves");
    return 0;
}
