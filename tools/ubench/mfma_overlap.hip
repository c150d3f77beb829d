// Micro-benchmark 3: does the f32 matrix pipe (v_mfma_f32_16x16x4_f32, 32 cycles per SIMD) overlap with VALU work,
// (a) from OTHER waves of the same SIMD, (b) inside one wave's instruction stream?  Per iteration every wave issues
// 8 MFMAs (two chains of 4 dependent ones = one hidden layer for two 16-ray blocks) and 96 VALU fmas (the sines).
//   hipcc --offload-arch=gfx950 -O2 mfma_overlap.hip -o mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

#define VALU12(x0, x1, x2, x3) \
    _Pragma("unroll") for (int q = 0; q < 3; q++) { \
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(b), "v"(c)); \
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(b), "v"(c)); \
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(b), "v"(c)); \
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(b), "v"(c)); }

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    f4 accA = {seed, seed, seed, seed}, accB = accA;
    f4 vA = accA * 0.5f, vB = accA * 0.25f;
    float wa = seed + threadIdx.x * 1e-3f, wb = seed * 0.5f;
    float b = seed * 0.5f, c = seed * 0.25f;
    for (int it = 0; it < iters; it++) {
        if (KIND == 0) {                        // VALU only
#pragma unroll
            for (int r = 0; r < 4; r++) { VALU12(vA.x, vA.y, vA.z, vA.w) VALU12(vB.x, vB.y, vB.z, vB.w) }
        }
        if (KIND == 1) {                        // MFMA only: two chains of four dependent MFMAs
#pragma unroll
            for (int r = 0; r < 4; r++) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, wb, accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, wb, accB, 0, 0, 0);
            }
        }
        if (KIND == 2) {                        // serial inside the wave: chain A, sines of A, chain B, sines of B
#pragma unroll
            for (int r = 0; r < 4; r++) accA = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, wb, accA, 0, 0, 0);
            vA = accA;
#pragma unroll
            for (int r = 0; r < 4; r++) { VALU12(vA.x, vA.y, vA.z, vA.w) }
            wa = vA.x;
#pragma unroll
            for (int r = 0; r < 4; r++) accB = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, wb, accB, 0, 0, 0);
            vB = accB;
#pragma unroll
            for (int r = 0; r < 4; r++) { VALU12(vB.x, vB.y, vB.z, vB.w) }
            wb = vB.x;
        }
        if (KIND == 3) {                        // software-pipelined: chain A interleaved with the sines of B and vice versa
#pragma unroll
            for (int r = 0; r < 4; r++) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, wb, accA, 0, 0, 0);
                VALU12(vB.x, vB.y, vB.z, vB.w)
            }
            wb = vB.x;
            vA = accA;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, wb, accB, 0, 0, 0);
                VALU12(vA.x, vA.y, vA.z, vA.w)
            }
            wa = vA.x;
            vB = accB;
        }
        if (KIND == 4) {                        // both chains first (A0 B0 A1 B1 ...), then all 96 fma
#pragma unroll
            for (int r = 0; r < 4; r++) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, wb, accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, wb, accB, 0, 0, 0);
            }
            vA = accA; vB = accB;
#pragma unroll
            for (int r = 0; r < 4; r++) { VALU12(vA.x, vA.y, vA.z, vA.w) VALU12(vB.x, vB.y, vB.z, vB.w) }
            wa = vA.x; wb = vB.x;
        }
        if (KIND == 5) {                        // chain A, chain B, then all 96 fma
#pragma unroll
            for (int r = 0; r < 4; r++) accA = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, wb, accA, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) accB = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, wb, accB, 0, 0, 0);
            vA = accA; vB = accB;
#pragma unroll
            for (int r = 0; r < 4; r++) { VALU12(vA.x, vA.y, vA.z, vA.w) VALU12(vB.x, vB.y, vB.z, vB.w) }
            wa = vA.x; wb = vB.x;
        }
    }
    f4 s = accA + accB + vA + vB;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w + wa + wb;
}

template <int KIND>
int run(const char* name, int waves_per_simd) {
    hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
    int blocks = p.multiProcessorCount * waves_per_simd;
    float* out; CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    int iters = 4000;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    printf("%-64s waves/SIMD %d  %.0f SIMD cycles per wave-iteration (at 2.4 GHz)\n", name, waves_per_simd, ms * 1e-3 * 2.4e9 / ((double)iters * waves_per_simd));
    CHK(hipFree(out));
    return 0;
}
int main() {
    printf("per wave-iteration: 8 MFMA 16x16x4 f32 (256 cycles of the matrix pipe) and/or 96 VALU fma (235 cycles of issue)\n");
    for (int w : {1, 2, 4}) {
        run<0>("VALU only (96 fma)", w);
        run<1>("MFMA only (2 chains of 4 dependent)", w);
        run<2>("serial in the wave: chain A, 48 fma on A, chain B, 48 fma on B", w);
        run<3>("pipelined in the wave: chain A interleaved with 48 fma on B, ...", w);
        run<4>("8 MFMA (A0 B0 A1 B1 ...) then 96 fma", w);
        run<5>("8 MFMA (chain A, chain B) then 96 fma", w);
    }
    return 0;
}
