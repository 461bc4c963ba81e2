// Issue cost of the vector instructions the three big kernels are made of (gfx950), as inline assembly so that the
// instruction measured is the instruction named.  Every lane runs 8 independent chains; 4 waves per SIMD (1024 threads per CU);
// result: SIMD cycles per wave-instruction at the clock the run held (taken from v_fma_f32 = the reference line).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu   (binary not tracked)
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed) {
    unsigned a[8], b[8];
    float f[8], g[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + 3 * i + 1); b[i] = a[i] ^ 0x5bd1e995u; f[i] = 1.0f + 0.001f * (a[i] & 255); g[i] = 0.5f + f[i]; }
    unsigned long long sacc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g[i]));
#define ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define LSHL(i) asm volatile("v_lshlrev_b32 %0, 4, %0" : "+v"(a[i]));
#define PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
#define ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b[i]));
#define BFE(i) asm volatile("v_bfe_u32 %0, %0, %1, 8" : "+v"(a[i]) : "v"(b[i]));
#define CVTI(i) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(a[i])); asm volatile("" : "+v"(a[i]));
#define CVTU8(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(a[i]) : "v"(f[i]));
#define PKRTZ(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(a[i]) : "v"(f[i]), "v"(g[i])); asm volatile("" : "+v"(f[i]));
#define MIXLO(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(a[i]) : "v"(b[i]), "v"(f[i]));
#define FMAMK(i) asm volatile("v_fmamk_f32 %0, %0, 0xc323d70a, %1" : "+v"(f[i]) : "v"(g[i]));
#define MULABS(i) asm volatile("v_mul_f32_e64 %0, %0, |%0|" : "+v"(f[i]));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(g[i]));
#define CMPS(i) { unsigned long long m; asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(f[i]), "v"(g[i])); sacc += m; }
#define CMPVCC(i) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n\tv_cndmask_b32_e32 %2, %2, %3, vcc" : : "v"(f[i]), "v"(g[i]), "v"(a[i]), "v"(b[i]) : "vcc");
#define WLANE(i) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(a[i]) : "s"((unsigned)it));
#define RLANE(i) { unsigned s_; asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s_) : "v"(a[i])); sacc += s_; }
#define XOR(i) asm volatile("v_xor_b32 %0, 0x80808080, %0" : "+v"(a[i]));
#define SDWASUB(i) asm volatile("v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "+v"(a[i]) : "v"(b[i]));
#define LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(a[i]) : "v"(b[i]));
#define BCNT(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
#define BFREV(i) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[i]));
#define CNDM(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(0x5555555555555555ull));
#define DPPMOV(i) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
#define ADD64(i) { unsigned long long t = ((unsigned long long)a[i] << 32) | b[i]; asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(t) : "v"(t)); a[i] = (unsigned)(t >> 32); b[i] = (unsigned)t; }
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
#define FMA64(i) { double d = (double)f[i]; asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d)); f[i] = (float)d; }
            if (MODE == 0) { REP8(FMA) }
            if (MODE == 1) { REP8(ADDU) }
            if (MODE == 2) { REP8(LSHL) }
            if (MODE == 3) { REP8(PERM) }
            if (MODE == 4) { REP8(ALIGN) }
            if (MODE == 5) { REP8(BFE) }
            if (MODE == 6) { REP8(CVTI) }
            if (MODE == 7) { REP8(CVTU8) }
            if (MODE == 8) { REP8(PKRTZ) }
            if (MODE == 9) { REP8(MIXLO) }
            if (MODE == 10) { REP8(FMAMK) }
            if (MODE == 11) { REP8(MULABS) }
            if (MODE == 12) { REP8(MAX3) }
            if (MODE == 13) { REP8(CMPS) }
            if (MODE == 14) { REP8(CMPVCC) }
            if (MODE == 15) { REP8(WLANE) }
            if (MODE == 16) { REP8(RLANE) }
            if (MODE == 17) { REP8(XOR) }
            if (MODE == 18) { REP8(SDWASUB) }
            if (MODE == 19) { REP8(LSHLADD) }
            if (MODE == 20) { REP8(BCNT) }
            if (MODE == 21) { REP8(BFREV) }
            if (MODE == 22) { REP8(CNDM) }
            if (MODE == 23) { REP8(DPPMOV) }
            if (MODE == 24) { REP8(ADD64) }
            if (MODE == 25) { REP8(MULLO) }
            if (MODE == 26) { REP8(MUL24) }
            if (MODE == 27) { REP8(SQRT) }
        }
    }
    unsigned s = (unsigned)sacc ^ (unsigned)(sacc >> 32);
    for (int i = 0; i < 8; ++i) s += a[i] + b[i] + __float_as_uint(f[i]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double g_ref = 0;
template <int MODE> void run(const char* name, int per_stmt) {
    unsigned* d; (void)hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, d, 10, 3u);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, d, iters, 3u); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // 1024 workgroups of 4 waves on 256 CUs = 16 waves per CU = 4 per SIMD; wave-instructions per SIMD = 4 waves x iters x 4 x 8 x per_stmt
    const double inst = 4.0 * iters * 4 * 8 * per_stmt;
    const double ns = ms * 1e6 / inst;
    if (MODE == 0) g_ref = ns;
    printf("%-46s %.3f ms  %.2f ns per wave-instruction per SIMD = %.2f x v_fma_f32\n", name, ms, ns, g_ref > 0 ? ns / g_ref : 1.0);
    (void)hipFree(d);
}
int main() {
    printf("4 waves per SIMD, 8 independent chains per lane; v_fma_f32 is the unit\n");
    run<0>("v_fma_f32", 1); run<1>("v_add_u32", 1); run<2>("v_lshlrev_b32", 1); run<3>("v_perm_b32", 1); run<4>("v_alignbit_b32", 1);
    run<5>("v_bfe_u32", 1); run<6>("v_cvt_f32_i32", 1); run<7>("v_cvt_pk_u8_f32", 1); run<8>("v_cvt_pkrtz_f16_f32", 1);
    run<9>("v_fma_mixlo_f16", 1); run<10>("v_fmamk_f32 (32-bit literal)", 1); run<11>("v_mul_f32 x, |x| (VOP3 modifiers)", 1);
    run<12>("v_max3_f32", 1); run<13>("v_cmp_gt_f32 -> scalar pair (+ s_add_u64)", 1); run<14>("v_cmp_gt_f32 vcc + v_cndmask (2 instr)", 2);
    run<15>("v_writelane_b32", 1); run<16>("v_readlane_b32 (+ s_add)", 1); run<17>("v_xor_b32 literal", 1); run<18>("v_sub_u32_sdwa", 1);
    run<19>("v_lshl_add_u32", 1); run<20>("v_bcnt_u32_b32", 1); run<21>("v_bfrev_b32", 1); run<22>("v_cndmask_b32 (scalar mask)", 1);
    run<23>("v_mov_b32 dpp wave_shr:1", 1); run<24>("v_lshl_add_u64", 1); run<25>("v_mul_lo_u32", 1); run<26>("v_mul_u32_u24", 1); run<27>("v_sqrt_f32", 1);
    return 0;
}
