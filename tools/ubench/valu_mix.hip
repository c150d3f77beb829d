// Micro-benchmark 2: do "half-rate" VALU instructions (min/max/cmp/cndmask/SGPR operands) overlap with full-rate ones
// (fma/add/mul) when they are MIXED in one instruction stream?   hipcc --offload-arch=gfx950 -O2 valu_mix.hip -o valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    float a[8], m[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 1e-3f + i; m[i] = a[i] * 0.5f; }
    float b = seed * 0.5f, c = seed * 0.25f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) {      // 1 fast + 1 slow (no SGPR)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(m[i]) : "v"(b));
                }
                if (KIND == 1) {      // 2 slow of different kinds
                    asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                    asm volatile("v_subrev_f32_e32 %0, s4, %0" : "+v"(m[i]));
                }
                if (KIND == 2) {      // 3 fast + 1 slow
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                    asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                    asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(c));
                    asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(m[i]) : "v"(b));
                }
                if (KIND == 3) {      // the box mix: 3 subrev(s) 2 mul(s) 2 fmac(s) 3 sub|.|(s) max3 3 max 1 mul 2 fmac(v) ...
                    asm volatile("v_subrev_f32_e32 %0, s4, %0" : "+v"(a[i]));
                    asm volatile("v_mul_f32_e32 %0, s5, %0" : "+v"(m[i]));
                    asm volatile("v_fmac_f32_e32 %0, s6, %1" : "+v"(m[i]) : "v"(a[i]));
                    asm volatile("v_sub_f32_e64 %0, |%0|, s7" : "+v"(a[i]));
                    asm volatile("v_max_f32_e32 %0, 0, %0" : "+v"(m[i]));
                    asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(a[i]) : "v"(m[i]));
                }
                if (KIND == 4) {      // same with VGPR operands instead of SGPRs
                    asm volatile("v_subrev_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                    asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(m[i]) : "v"(c));
                    asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(m[i]) : "v"(b), "v"(a[i]));
                    asm volatile("v_sub_f32_e64 %0, |%0|, %1" : "+v"(a[i]) : "v"(c));
                    asm volatile("v_max_f32_e32 %0, 0, %0" : "+v"(m[i]));
                    asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(a[i]) : "v"(m[i]));
                }
                if (KIND == 5) {      // KIND 4 with the max replaced by add |.|
                    asm volatile("v_subrev_f32_e32 %0, %1, %0" : "+v"(a[i]) : "v"(b));
                    asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(m[i]) : "v"(c));
                    asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(m[i]) : "v"(b), "v"(a[i]));
                    asm volatile("v_sub_f32_e64 %0, |%0|, %1" : "+v"(a[i]) : "v"(c));
                    asm volatile("v_add_f32_e64 %0, %0, |%0|" : "+v"(m[i]));
                    asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(a[i]) : "v"(m[i]));
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + m[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
int run(const char* name, int per_unit, int waves_per_simd) {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int blocks = p.multiProcessorCount * waves_per_simd;
    float* out; CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    int iters = 2000;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    double instr_per_simd = (double)iters * 64 * per_unit * waves_per_simd;
    printf("%-44s waves/SIMD %d  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms * 1e-3 * 2.4e9 / instr_per_simd);
    CHK(hipFree(out));
    return 0;
}
int main() {
    for (int w : {4, 6}) {
        run<0>("1 fma + 1 max", 2, w);
        run<1>("1 max + 1 subrev(sgpr)", 2, w);
        run<2>("fma+mul+add + 1 max", 4, w);
        run<3>("box mix, SGPR operands (6 instr)", 6, w);
        run<4>("box mix, VGPR operands", 6, w);
        run<5>("box mix, VGPR operands, add|.| for max", 6, w);
    }
    return 0;
}
