#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(unsigned* out, unsigned seed, int iters) {
    unsigned a[8], acc[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + i + 1); acc[i] = i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_udot4(a[i], 0x01020304u + r, acc[i], false);
                if (MODE == 1) acc[i] = __umul24(a[i] & 0xFFFFFFu, (unsigned)(r + 3)) + acc[i];
                if (MODE == 2) { us2 x = __builtin_bit_cast(us2, a[i]); us2 w = {(unsigned short)(r+1),(unsigned short)(r+2)}; us2 c = __builtin_bit_cast(us2, acc[i]); c = x * w + c; acc[i] = __builtin_bit_cast(unsigned, c); }
                if (MODE == 3) acc[i] = a[i] * (unsigned)(r + 3) + acc[i];
                if (MODE == 4) acc[i] = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a[i]), us2{(unsigned short)(r+1),(unsigned short)(r+2)}, acc[i], false);
            }
        }
    }
    unsigned s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name) {
    unsigned* d; hipMalloc(&d, 1024 * 256 * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 3u, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 3u, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)256 * 8 * 4 * iters * 64;   // wave-instructions
    // 8 blocks x 4 waves per CU = 8 waves per SIMD
    printf("%-14s %.3f ms  cycles/wave-instr/SIMD @2.4GHz: %.2f\n", name, ms, ms * 1e-3 * 2.4e9 / (instr / 1024));
}
int main() { run<0>("dot4_u32_u8"); run<1>("mad_u32_u24"); run<2>("pk_mad_u16"); run<3>("mul_lo+add"); run<4>("dot2_u32_u16"); return 0; }
