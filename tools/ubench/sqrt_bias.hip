// How does v_sqrt_f32 err?  For every non-negative bit pattern below 2^95 (as sqrt_ sees them: scaled by 2^32), compare
// the raw instruction with the correctly rounded square root: exact / 1 ulp low / 1 ulp high / worse.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/sqrt_bias.hip -o /tmp/sqrt_bias && /tmp/sqrt_bias
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long* out) {
    const uint32_t limit = 0x6F000000u;
    unsigned long long exact = 0, low = 0, high = 0, worse = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < limit; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __builtin_bit_cast(float, (uint32_t)b);
        const float xs = x * 4294967296.0f;
        const float y = __builtin_amdgcn_sqrtf(xs);
        const float r = __builtin_sqrtf(xs);
        const int d = __builtin_bit_cast(int, y) - __builtin_bit_cast(int, r);
        if (d == 0) exact++; else if (d == -1) low++; else if (d == 1) high++; else worse++;
    }
    atomicAdd(&out[0], exact); atomicAdd(&out[1], low); atomicAdd(&out[2], high); atomicAdd(&out[3], worse);
}
int main() {
    unsigned long long* d; unsigned long long h[4] = {0, 0, 0, 0};
    hipMalloc(&d, sizeof h); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("v_sqrt_f32 vs correctly rounded: exact %llu, 1 ulp low %llu, 1 ulp high %llu, worse %llu\n", h[0], h[1], h[2], h[3]);
    return 0;
}
