// Micro-benchmark: cycles per wave64 VALU instruction per SIMD for the instruction kinds the sine kernel uses.
// 8 independent dependency chains per lane (ILP 8), 1..8 waves per SIMD.   hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3f + i;
    float b = seed * 0.5f, c = seed * 0.25f;
    double d[8], db = seed * 0.5, dc = seed * 0.25;      // VGPR pairs for the packed-f32 instructions
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = seed + threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 1) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f9e0419" : "+v"(a[i]) : "v"(b));
                if (KIND == 2) asm volatile("v_fmamk_f32 %0, %0, 0x3f9e0419, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 3) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 6) asm volatile("v_lshl_add_u32 %0, %0, 31, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                if (KIND == 8) asm volatile("v_add_f32 %0, 0xcb400000, %0" : "+v"(a[i]));          // literal operand
                if (KIND == 9) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 10) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 11) asm volatile("v_sub_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 12) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 13) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 14) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
                if (KIND == 15) asm volatile("v_fma_f32 %0, %0, s4, %1" : "+v"(a[i]) : "v"(b));    // SGPR operand
                if (KIND == 16) asm volatile("v_sub_f32_e32 %0, s4, %0" : "+v"(a[i]));              // VOP2, SGPR src0
                if (KIND == 17) asm volatile("v_mul_f32_e32 %0, s4, %0" : "+v"(a[i]));
                if (KIND == 18) asm volatile("v_fmac_f32_e32 %0, s4, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 19) asm volatile("v_sub_f32_e64 %0, |%0|, s4" : "+v"(a[i]));           // VOP3, abs + SGPR
                if (KIND == 20) asm volatile("v_sub_f32_e64 %0, |%0|, %1" : "+v"(a[i]) : "v"(b));  // VOP3, abs, VGPRs
                if (KIND == 21) asm volatile("v_max_f32_e32 %0, 0, %0" : "+v"(a[i]));               // inline constant
                if (KIND == 22) asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (KIND == 23) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b));    // VOP3, 2 distinct VGPRs
                if (KIND == 24) asm volatile("v_cmp_lt_f32_e64 s[6:7], %0, %1" :: "v"(a[i]), "v"(b) : "s6", "s7");
                if (KIND == 25) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[8:9]" : "+v"(a[i]) : "v"(b));
                if (KIND == 26) asm volatile("v_mov_b32 %0, s4" : "=v"(a[i]));
                if (KIND == 27) asm volatile("v_mul_f32_e32 %0, %0, %0" : "+v"(a[i]));
                if (KIND == 28) asm volatile("v_fma_f32 %0, s4, s4, %0" : "+v"(a[i]));             // one SGPR read twice
                if (KIND == 29) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));             // one VGPR
                if (KIND == 30) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));   // 3 distinct VGPRs
                if (KIND == 31) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 32) asm volatile("v_subrev_f32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (KIND == 33) asm volatile("v_mul_f32_dpp %0, %1, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (KIND == 34) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
                if (KIND == 35) asm volatile("v_subrev_f32_dpp %0, %1, |%0| row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (KIND == 36) asm volatile("v_add_f32_e64 %0, %0, |%0|" : "+v"(a[i]));
                if (KIND == 37) asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                if (KIND == 38) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b));
                if (KIND == 39) asm volatile("v_subrev_f32_dpp %0, %1, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b));
                if (KIND == 40) asm volatile("v_sub_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(a[i]) : "v"(b));
                if (KIND == 41) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 42) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 43) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(db));
                if (KIND == 46) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));
                if (KIND == 47) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(db));
                if (KIND == 44) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                if (KIND == 45) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
int run(const char* name, int waves_per_simd) {
    int dev = 0; hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, dev));
    int blocks = p.multiProcessorCount * waves_per_simd;   // 256 threads = 4 waves = one per SIMD
    float* out; CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    int iters = 4000;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    double instr_per_simd = (double)iters * 64 * waves_per_simd;
    double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
    printf("%-28s waves/SIMD %d  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, cyc);
    CHK(hipFree(out));
    return 0;
}
int main() {
    for (int w : {6}) {
        run<0>("v_fma_f32", w); run<1>("v_fmaak_f32 (literal)", w); run<2>("v_fmamk_f32 (literal)", w); run<3>("v_fmac_f32", w);
        run<4>("v_add_f32", w); run<5>("v_mul_f32", w); run<6>("v_lshl_add_u32", w); run<7>("v_cndmask_b32", w);
        run<8>("v_add_f32 literal", w); run<9>("v_max3_f32", w); run<10>("v_med3_f32", w); run<11>("v_sub_f32 |abs|", w);
        run<12>("v_sqrt_f32", w); run<13>("v_rcp_f32", w); run<14>("v_cmp_lt_f32", w); run<15>("v_fma_f32 sgpr", w);
        run<16>("v_sub_f32_e32 s,v", w); run<17>("v_mul_f32_e32 s,v", w); run<18>("v_fmac_f32_e32 s,v", w); run<19>("v_sub_f32_e64 |v|,s", w);
        run<20>("v_sub_f32_e64 |v|,v", w); run<21>("v_max_f32 0,v", w); run<22>("v_max_f32 v,v", w); run<23>("v_fma_f32 a,b,a", w);
        run<24>("v_cmp_lt_f32_e64 sgpr dst", w); run<25>("v_cndmask_e64 sgpr mask", w); run<26>("v_mov_b32 v,s", w); run<27>("v_mul_f32 a,a", w);
        run<28>("v_fma_f32 s,s,v", w); run<29>("v_fma_f32 a,a,a", w); run<30>("v_fma_f32 b,c,a", w); run<31>("v_mad_u32_u24", w);
        run<32>("v_subrev_f32_dpp newbcast", w); run<33>("v_mul_f32_dpp newbcast", w); run<34>("v_fmac_f32_dpp newbcast", w);
        run<35>("v_subrev_dpp newbcast |abs|", w); run<36>("v_add_f32_e64 v,|v|", w); run<37>("v_min_f32", w); run<38>("v_mov_b32_dpp newbcast", w);
        run<39>("v_subrev_dpp quad_perm", w); run<40>("v_sub_f32_sdwa", w); run<41>("v_and_b32", w); run<42>("v_add_u32", w); run<44>("v_xor_b32", w); run<45>("v_lshlrev_b32", w);
        run<43>("v_pk_mul_f32 (2 x f32)", w); run<46>("v_pk_fma_f32 (2 x f32)", w); run<47>("v_pk_add_f32 (2 x f32)", w);
    }
    return 0;
}
