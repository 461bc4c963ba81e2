// wave-instruction issue rates of the integer / bit instructions the labelling kernel is made of (gfx950).
// 8 independent chains per lane, 8 waves per SIMD: throughput, not latency.  usage: ./tools/ubench2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
template <int MODE>
__global__ void k(unsigned* out, unsigned seed, int iters) {
    unsigned a[8], acc[8];
    u64 b[8], bacc[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + i + 1); acc[i] = i; b[i] = (u64)a[i] * 0x9E3779B97F4A7C15ull; bacc[i] = i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) acc[i] = (acc[i] & a[i]) ^ (unsigned)(r + 1);                 // 1 bitop3 / and+xor
                if (MODE == 1) bacc[i] = (bacc[i] << 1) ^ b[i];                              // 64-bit shift by 1 + xor64
                if (MODE == 2) bacc[i] = bacc[i] + b[i];                                     // 64-bit add
                if (MODE == 3) acc[i] = __popc(acc[i] ^ a[i]) + acc[i];                      // bcnt (+ xor)
                if (MODE == 4) acc[i] = acc[i] * a[i] + (unsigned)r;                         // mul_lo_u32
                if (MODE == 5) acc[i] = __builtin_bitreverse32(acc[i]) + a[i];               // bfrev + add
                if (MODE == 6) acc[i] = __builtin_amdgcn_alignbit(acc[i], a[i], 7u) + 1u;    // alignbit + add
                if (MODE == 7) acc[i] = (acc[i] > a[i] ? acc[i] - a[i] : acc[i] + a[i]);     // cmp + cndmask-ish
                if (MODE == 8) acc[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)acc[i], 0x138, 0xf, 0xf, false) + a[i];   // dpp mov + add
                if (MODE == 9) acc[i] = (unsigned)__ffs(acc[i] | 1u) + acc[i] * 3u;          // ffbl + ...
                if (MODE == 10) bacc[i] = (bacc[i] >> 1) | (b[i] << 63);                     // 64-bit funnel by shifts
                if (MODE == 11) acc[i] = __umul24(acc[i], a[i]) + (unsigned)r;               // mad_u32_u24
                if (MODE == 12) bacc[i] = (bacc[i] & b[i]) | (~bacc[i] & (b[i] >> 3));       // 64-bit logic + shift by 3
                if (MODE == 13) { unsigned lo = (unsigned)bacc[i], hi = (unsigned)(bacc[i] >> 32); bacc[i] = (((u64)__builtin_amdgcn_alignbit(hi, lo, 31u)) << 32 | (lo << 1)) ^ b[i]; }  // shift by 1 by hand
            }
        }
    }
    unsigned s = 0; for (int i = 0; i < 8; ++i) s += acc[i] + (unsigned)bacc[i] + (unsigned)(bacc[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, double ops) {
    unsigned* d; hipMalloc(&d, 1024 * 256 * 4 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 1000;
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 3u, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256 * 8), dim3(256), 0, 0, d, 3u, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double groups = (double)256 * 8 * 4 * iters * 64;   // wave-level statement executions
    printf("%-28s %.3f ms  cycles per statement per SIMD @2.4GHz: %.2f\n", name, ms, ms * 1e-3 * 2.4e9 / (groups / 1024));
}
int main() {
    run<0>("and+xor (u32)", 1); run<1>("shl64 by 1 + xor64", 1); run<2>("add64", 1); run<3>("xor+bcnt(+acc)", 1); run<4>("mul_lo_u32+add", 1);
    run<5>("bfrev+add", 1); run<6>("alignbit+add", 1); run<7>("cmp+sub+add+cndmask", 1); run<8>("dpp mov+add", 1); run<9>("ffbl+or+mul3", 1);
    run<10>("shr64 1 | shl64 63", 1); run<11>("mad_u32_u24", 1); run<12>("sel64 + shr64 3", 1); run<13>("shl64 by 1 by hand + xor64", 1);
    return 0;
}
