// Micro-benchmark: what ONE wave per SIMD issues per cycle as a function of its instruction-level parallelism.
// ILP independent v_fma_f32 chains per lane (1, 2, 3, 4, 8), one block of 256 threads per CU (= one wave per SIMD), 1 or 2 blocks per CU.
//   hipcc --offload-arch=gfx950 -O2 lone_wave_ilp.hip -o lone_wave_ilp && ./lone_wave_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ILP, int MIX>
__global__ void __launch_bounds__(256) k(float* out, long long* cyc, int iters, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 1e-3f + i;
    float b = seed * 0.5f, c = seed * 0.25f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / ILP; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (MIX == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MIX == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (MIX == 2) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ILP, int MIX>
int run(const char* name, int blocks) {
    float* out; long long* cyc;
    CHK(hipMalloc(&out, blocks * 256 * 4)); CHK(hipMalloc(&cyc, blocks * 8));
    const int iters = 2000;
    hipLaunchKernelGGL((k<ILP, MIX>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
    hipLaunchKernelGGL((k<ILP, MIX>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
    CHK(hipDeviceSynchronize());
    static long long h[4096];
    CHK(hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost));
    double m = 0; for (int i = 0; i < blocks; i++) m += h[i]; m /= blocks;
    printf("{\"kind\": \"%s\", \"ilp\": %d, \"waves_per_simd\": %d, \"cycles_per_wave_instruction\": %.2f}\n", name, ILP, blocks / 256, m / (iters * 64.0 / ILP * ILP));
    CHK(hipFree(out)); CHK(hipFree(cyc));
    return 0;
}

int main() {
    for (int w = 1; w <= 2; w++) {
        run<1, 0>("fma", 256 * w); run<2, 0>("fma", 256 * w); run<4, 0>("fma", 256 * w); run<8, 0>("fma", 256 * w);
        run<1, 1>("max3", 256 * w); run<2, 1>("max3", 256 * w); run<4, 1>("max3", 256 * w);
        run<1, 2>("sqrt", 256 * w); run<2, 2>("sqrt", 256 * w); run<4, 2>("sqrt", 256 * w);
    }
    return 0;
}
