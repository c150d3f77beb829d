// rt_kernels.hip — gfx950 kernels of the per-pixel Monte Carlo sample path.
//
//   trace_paths<KIND,NOBJ>   complete-path form (F20/F24: cornell_box_v3/pathtracer.py:81-106,
//                            renderer.py:27-36).  One wavefront lane per pixel-sample.  Each
//                            wave is persistent and runs a two-phase state machine:
//                              phase A  march: all marching lanes take sphere-tracing steps
//                                       until >= wait_lanes lanes have finished their raycast
//                                       (ballot + popcount, wave-uniform branch);
//                              phase B  shade: finished lanes run the surface interaction /
//                                       miss handling, russian roulette, write finished
//                                       samples, and idle lanes are REFILLED with fresh
//                                       pixel-samples (ballot + mbcnt prefix rank into the
//                                       wave's work range, claimed in chunks from one global
//                                       atomic) -> live-ray compaction across the bounce loop.
//                            Per-sample radiance goes to the staging buffer; results do not
//                            depend on the schedule.
//   accumulate_samples       adds the staged samples of each pixel IN SAMPLE ORDER into
//                            image_buffer (T7), exactly like "image_buffer[i,j] += vec4(color,1)".
//   persistent_steps<KIND>   src/ form (F1-F4,F6,F7: src/pathtracer.py:16-103, src/scene.py:59-84):
//                            one lane per pixel, ray state in ray_buffer between launches.
//   refresh, post_process, pack/unpack tiles, math_probe (test hook).
#include <hip/hip_runtime.h>

#include "rt_trace.hpp"

namespace rt {

// -------------------------------------------------------------------------------------------
// image_buffer[i,j] += vec4(color, 1) for k = 0..K-1 in order (renderer.py:36)
__global__ void __launch_bounds__(256) accumulate_samples(const Params P) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    int x = 0, y = 0;
    const bool valid = q < (uint32_t)P.np && pixel_of(P, q, x, y);
    // the `deposits` work counter is counted where the deposits happen (one per staged sample added to T7), per wave
    {
        const uint32_t n = wave_sum(valid ? (uint32_t)P.K : 0u);
        if ((threadIdx.x & 63) == 0 && n) atomicAdd(&P.counters->shard[blockIdx.x & 63u][5], (unsigned long long)n);
    }
    if (!valid) return;
    float4* dst = P.image_buffer + ((size_t)x * P.cfg.height + y);
    float4 acc = *dst;
    // a pixel's K records (12 bytes each) are contiguous (item-linear staging): fetch them 8 at a time (96 B per lane as
    // six 16-byte loads when the pixel's run starts on a 16-byte boundary, i.e. K % 4 == 0) and add them strictly in
    // sample order
    const float* src = P.stage + (size_t)q * (size_t)P.K * 3u;
    int k = 0;
    if ((P.K & 3) == 0) {
        for (; k + 8 <= P.K; k += 8) {
            float4 v[6];
            const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)k * 3u);
#pragma unroll
            for (int i = 0; i < 6; i++) v[i] = s4[i];
            const float f[24] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w,
                                 v[3].x, v[3].y, v[3].z, v[3].w, v[4].x, v[4].y, v[4].z, v[4].w, v[5].x, v[5].y, v[5].z, v[5].w};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                acc.x += f[3 * i];
                acc.y += f[3 * i + 1];
                acc.z += f[3 * i + 2];
                acc.w += 1.0f;
            }
        }
    }
    for (; k < P.K; k++) {
        acc.x += src[3 * k];
        acc.y += src[3 * k + 1];
        acc.z += src[3 * k + 2];
        acc.w += 1.0f;
    }
    *dst = acc;
}

// The same sum over the DENSE staging (rt_trace.hpp stage_sample): a block takes `acc_batch` consecutive items at a time (a multiple
// of K and of the chunk: whole pixels, whole chunks), reads the chunks' records as they lie — in completion order, coalesced — and
// drops each at its sample's place in LDS (the byte beside the record says which sample of the chunk it is); then one thread per
// pixel adds the K records of its pixel in sample order, as above.  The pixel stride in LDS is odd (in words): the threads' reads
// fall into different banks.
__global__ void __launch_bounds__(256) accumulate_dense(const Params P) {
    extern __shared__ float lds_rec[];
    __shared__ uint32_t fill_s[128];
    const uint32_t B = P.acc_batch, chunk = P.chunk, K = (uint32_t)P.K;
    const uint32_t stride = 3u * K + ((3u * K) & 1u ? 0u : 1u);
    const uint32_t n_batch = (P.total_items + B - 1u) / B;
    const StageRec* stage = reinterpret_cast<const StageRec*>(P.stage);
    uint32_t n_dep = 0;
    for (uint32_t b = blockIdx.x; b < n_batch; b += gridDim.x) {
        const uint32_t b0 = b * B;
        const uint32_t nb = P.total_items - b0 < B ? P.total_items - b0 : B;
        const uint32_t c0 = b0 / chunk, nc = (nb + chunk - 1u) / chunk;
        if (threadIdx.x < nc) fill_s[threadIdx.x] = P.stage_fill[c0 + threadIdx.x];
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < nb; s += 256u) {
            const uint32_t cl = chunk == 1u ? s : __umulhi(s, P.acc_magic_chunk);
            const uint32_t sl = s - cl * chunk;
            if (sl < fill_s[cl]) {
                const StageRec r = stage[b0 + s];
                const uint32_t i = cl * chunk + (uint32_t)P.stage_idx[b0 + s];     // the record's item, relative to the batch
                const uint32_t pl = K == 1u ? i : __umulhi(i, P.acc_magic_k);     // its pixel ...
                float* d = lds_rec + pl * stride + (i - pl * K) * 3u;              // ... and sample
                d[0] = r.r, d[1] = r.g, d[2] = r.b;
            }
        }
        __syncthreads();
        const uint32_t n_pix = nb / K, q0 = b0 / K;
        for (uint32_t pl = threadIdx.x; pl < n_pix; pl += 256u) {
            int x = 0, y = 0;
            if (!pixel_of(P, q0 + pl, x, y)) continue;
            float4* dst = P.image_buffer + ((size_t)x * P.cfg.height + y);
            float4 acc = *dst;
            const float* src = lds_rec + pl * stride;
            for (uint32_t k = 0; k < K; k++) {
                acc.x += src[3u * k];
                acc.y += src[3u * k + 1u];
                acc.z += src[3u * k + 2u];
                acc.w += 1.0f;
            }
            *dst = acc;
            n_dep += K;
        }
        __syncthreads();
    }
    n_dep = wave_sum(n_dep);
    if ((threadIdx.x & 63) == 0 && n_dep) atomicAdd(&P.counters->shard[blockIdx.x & 63u][5], (unsigned long long)n_dep);
}

// -------------------------------------------------------------------------------------------
// refresh() src/renderer.py:12-22
__global__ void refresh_kernel(float4* image_buffer, rtpbr_ray* ray_buffer, float2* diff_buffer, float* diff_pixels,
                               int adaptive, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    image_buffer[i] = make_float4(0, 0, 0, 0);
    ray_buffer[i].depth = 0;
    if (adaptive) {  // src/renderer.py:19-21
        diff_buffer[i] = make_float2(1.0f, 1.0f);
        diff_pixels[i] = 1e32f;
    }
}

// post_process() src/postprocessor.py:24-43
__global__ void post_process_kernel(const Params P) {
    size_t n = (size_t)P.cfg.width * P.cfg.height;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    vec3 last = mk(P.image_pixels[i * 3 + 0], P.image_pixels[i * 3 + 1], P.image_pixels[i * 3 + 2]);
    vec3 c = tone_map(P.cfg, P.image_buffer[i]);
    P.image_pixels[i * 3 + 0] = c.x;
    P.image_pixels[i * 3 + 1] = c.y;
    P.image_pixels[i * 3 + 2] = c.z;
    if (P.cfg.adaptive_sampling) {  // src/postprocessor.py:40-43
        vec3 dc = mk(fabs_(c.x - last.x), fabs_(c.y - last.y), fabs_(c.z - last.z));
        float2 d = P.diff_buffer[i];
        d.x += brightness(dc);
        d.y += 1.0f;
        P.diff_buffer[i] = d;
        P.diff_pixels[i] = d.x / d.y;
    }
}

// tile pack / unpack for the multi-GPU gather
__global__ void pack_tiles_kernel(const Params P, float4* dst) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (uint32_t)P.np) return;
    int x, y;
    float4 v = make_float4(0, 0, 0, 0);
    if (pixel_of(P, q, x, y)) v = P.image_buffer[(size_t)x * P.cfg.height + y];
    dst[q] = v;
}
__global__ void unpack_tiles_kernel(const Params P, const float4* src) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (uint32_t)P.np) return;
    int x, y;
    if (pixel_of(P, q, x, y)) P.image_buffer[(size_t)x * P.cfg.height + y] = src[q];
}

// test hook: evaluate one of the exact math functions on the device (tests/test_gpu_parity.py::test_exact_math_functions_match_oracle_bitwise)
__global__ void math_probe(int op, const float* a, const float* b, float* out, float* out2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f;
    float r = 0.0f, r2 = 0.0f;
    switch (op) {
        case 0: sincos_(x, &r, &r2); break;
        case 1: r = exp_(x); break;
        case 2: r = atan2_(x, y); break;
        case 3: r = asin_(x); break;
        case 4: r = sqrt_(x); break;
        case 5: r = x / y; break;
        case 7: r = sin_pi_(x); break;
        case 8: r = log_(x); break;
        case 9: r = pow_(x, y); break;
        case 6: { uint32_t n0 = __builtin_bit_cast(uint32_t, y); r = rng_next(__builtin_bit_cast(uint32_t, x), n0); } break;
        default: break;
    }
    out[i] = r;
    if (out2) out2[i] = r2;
}

// test hook: custom correctly rounded sqrt_ vs the compiler's IEEE sqrt for every bit pattern
// in [0, 2^95) (0x6F000000 patterns); counts mismatches.
__global__ void sqrt_exhaustive(unsigned long long* mismatches) {
    const uint32_t limit = 0x6F000000u;
    unsigned long long bad = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < limit; b += (uint64_t)gridDim.x * blockDim.x) {
        float x = __builtin_bit_cast(float, (uint32_t)b);
        float a = sqrt_(x), r = sqrt_ieee_(x);
        if (__builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, r)) bad++;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// -------------------------------------------------------------------------------------------
// Cost-ordered ownership of the src/ pool kernel (rt_persistent.hpp): a counting sort of the local pixels by the march
// steps they took since the last plan (log-scale buckets, 8 per octave; heaviest bucket first; the order inside a bucket
// is whatever the atomics give — it steers the schedule only, results do not depend on who owns a pixel), and the choice
// of the pixels that get waves of their own.
RT_D uint32_t cost_bucket(uint32_t c) {
    if (c == 0u) return 0u;
    const int e = 31 - __builtin_clz(c);
    const uint32_t m = e >= 3 ? (c >> (e - 3)) & 7u : (c << (3 - e)) & 7u;
    const uint32_t b = (uint32_t)e * 8u + m + 1u;
    return b < 255u ? b : 255u;
}
// smallest cost that falls into bucket b (b >= 1)
RT_D uint32_t bucket_floor(uint32_t b) {
    const uint32_t e = (b - 1u) >> 3, m = (b - 1u) & 7u;
    return e >= 3u ? (8u + m) << (e - 3u) : (8u + m) >> (3u - e);
}
__global__ void __launch_bounds__(256) plan_hist(const uint32_t* __restrict__ cost, uint32_t np, PlanBuf* plan) {
    __shared__ uint32_t h[256];
    __shared__ unsigned long long tot;
    h[threadIdx.x] = 0;
    if (threadIdx.x == 0) tot = 0;
    __syncthreads();
    unsigned long long mine = 0;
    for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < np; q += gridDim.x * 256u) {
        const uint32_t c = cost[q];
        mine += c;
        atomicAdd(&h[cost_bucket(c)], 1u);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&tot, mine);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&plan->hist[threadIdx.x], h[threadIdx.x]);
    if (threadIdx.x == 0 && tot) atomicAdd(&plan->total, tot);
}
// one block: where every bucket starts in `order` (heaviest first) and how many pixels the heavy waves take.
// A pixel is heavy when its own chain of march steps is long against BOTH the frame's mean pixel (mean_x16 / 16 times)
// and a wave's share of the whole frame in march iterations, total / (64 lanes x waves) (bulk_x16 / 16 times).
__global__ void __launch_bounds__(256) plan_scan(PlanBuf* plan, uint32_t np, uint32_t n_waves, uint32_t heavy_own, uint32_t mean_x16, uint32_t bulk_x16,
                                                 uint32_t tiny_waves, uint32_t n_cls, uint32_t chain_on, uint32_t chain_ref_waves) {      // chain_on: 0 = no chain set, else the most waves it may take; chain_ref_waves: waves of the grid the pool kernel has ALONE
    __shared__ uint32_t h[256];
    const uint32_t b = threadIdx.x;
    h[b] = plan->hist[b];
    __syncthreads();
    uint32_t above = 0;                                  // pixels in heavier buckets
    for (uint32_t j = b + 1u; j < 256u; j++) above += h[j];
    plan->cursor[b] = above;
    if (b == 0) {
        const double total = (double)plan->total;
        const double t_mean = total / (double)(np ? np : 1u) * (double)mean_x16 / 16.0;
        const double t_bulk = total / (64.0 * (double)(n_waves ? n_waves : 1u)) * (double)bulk_x16 / 16.0;
        const double thr = t_mean > t_bulk ? t_mean : t_bulk;
        uint32_t n_heavy = 0;
        if (heavy_own > 0u && total > 0.0)
            for (uint32_t j = 255u; j >= 1u; j--) {
                if ((double)bucket_floor(j) <= thr) break;
                n_heavy += h[j];
            }
        const unsigned long long cap = (unsigned long long)heavy_own * (unsigned long long)(n_waves / 4u);
        if (n_heavy > cap) n_heavy = (uint32_t)cap;
        plan->n_heavy = n_heavy;
        // Small heavy waves (2..8 pixels: a context never waits for a lane, the lean tracked loop runs most of the time) cost
        // wave slots.  When the longest chain is several times a wave's share of the frame the launch is as long as that
        // chain and most of the chip idles anyway (small frames): up to a quarter of the waves; otherwise `tiny_waves`.
        uint32_t top = 255u;
        while (top > 0u && h[top] == 0u) top--;
        const double chain = top ? (double)bucket_floor(top) : 0.0;
        const double bulk = total / (64.0 * (double)(n_waves ? n_waves : 1u));
        plan->tiny_waves = chain > 3.0 * bulk ? n_waves / 4u : (tiny_waves < n_waves / 8u ? tiny_waves : n_waves / 8u);
        // ---- the chain set (rt_chain.hpp): a launch that is as long as its longest chain hands the head of the list to the chain
        // kernel.  Waves are packed greedily, heaviest pixel first, by the SUM of their pixels' steps — measured: a chain wave takes
        // ~800 cycles per lane-step whether it holds one pixel or eight (the raycasts of different pixels hardly overlap: a wave
        // iterates as long as its longest ray and every iteration pays for the most expensive form any lane needs) — every wave's
        // sum below the heaviest pixel's own steps (x 1.15); at most 8 pixels per wave, at most 2048 waves / half the grid's;
        // pixels lighter than an eighth of the heaviest (or not heavy at all) stay in the pool kernel.
        plan->n_chain = 0;
        plan->n_chain_waves = 0;
        plan->chain_start[0] = 0;
        // ("chain-bound" is decided against a wave's share under the grid the pool kernel has BESIDE the chain kernel — the host
        // passes it whether or not room has been made yet: it makes room only AFTER a plan has said so, and the answer must not
        // change because room was made)
        const double bulk_ref = total / (64.0 * (double)(chain_ref_waves ? chain_ref_waves : (n_waves ? n_waves : 1u)));
        if (chain_on && chain > 3.0 * bulk_ref && n_heavy > 0u) {
            const double T = chain * 1.15, thr_c = thr > chain / 8.0 ? thr : chain / 8.0;
            const uint32_t max_w = chain_on < 2048u ? chain_on : 2048u;
            // (round 5, after the lean loops lost a third of their instructions: a wave of ONE pixel runs them nearly all the time,
            // 610-720 cycles per step; a wave of two or three iterates in whatever form its lanes can share — per-wave records at
            // 768x432: 820-1000 cycles per lane-step, the three-pixel waves ended at 45-56 Mcycles, the heaviest pixel alone at 36 —
            // so a wave of several pixels is loaded to 1 / 1.45 of the limit)
            const double multi_x = 1.45;
            uint32_t wv = 0, cnt = 0, pos = 0;
            double load = 0.0;
            bool full = false;
            for (uint32_t j = 255u; j >= 1u && !full; j--) {
                const double c = (double)bucket_floor(j) * 1.045;        // (a bucket spans 9 %)
                if (c <= thr_c) break;
                for (uint32_t i = 0; i < h[j]; i++) {
                    double add = c;
                    if (cnt > 0u && ((load + add) * multi_x > T || cnt >= 8u)) {
                        wv++;
                        plan->chain_start[wv] = pos;          // (closes the previous wave)
                        if (wv >= max_w) { full = true; break; }
                        cnt = 0u;
                        load = 0.0;
                        add = c;
                    }
                    load += add;
                    cnt++;
                    pos++;
                }
            }
            if (!full && cnt > 0u) {
                wv++;
                plan->chain_start[wv] = pos;
            }
            plan->n_chain_waves = wv;
            plan->n_chain = plan->chain_start[wv];
        }
        // ---- age-weighted shares: move every residency slot's weight by (mean lifetime of all light waves / its own)
        // — with equal shares the five waves of a SIMD end at 75 ... 160 Mcycles because the arbiter favours the older
        // wave; the weights that make them end together are close to 1 / lifetime after ONE measurement and settle in two or
        // three.  Weights are 6-bit integers around 16 (entries of `order` per round), at most 4 : 1 apart.
        if (n_cls >= 2u && n_cls <= 8u) {
            if (!plan->age_valid)
                for (uint32_t c = 0; c < 8u; c++) plan->age_wf[c] = 16.0f;
            else if (plan->age_cls != n_cls)      // the grid changed (room made for the chain kernel): keep the tuned weights of the
                for (uint32_t c = plan->age_cls; c < 8u; c++) plan->age_wf[c] = plan->age_wf[plan->age_cls ? plan->age_cls - 1u : 0u];      // classes that remain, the last one's for new ones
            double tot = 0.0, cnt = 0.0;
            bool have = true;
            for (uint32_t c = 0; c < n_cls; c++) {
                if (plan->life_cnt[c] == 0u) have = false;
                tot += (double)plan->life_sum[c];
                cnt += (double)plan->life_cnt[c];
            }
            if (have && tot > 0.0) {
                const double mean = tot / cnt;
                double wsum = 0.0;
                for (uint32_t c = 0; c < n_cls; c++) {
                    const double m = (double)plan->life_sum[c] / (double)plan->life_cnt[c];
                    double f = mean / m;
                    f = f < 0.6 ? 0.6 : (f > 1.6 ? 1.6 : f);                 // one step moves a weight by at most -40 % / +60 %
                    plan->age_wf[c] = (float)((double)plan->age_wf[c] * f);
                    wsum += (double)plan->age_wf[c];
                }
                double wmax = 0.0;
                for (uint32_t c = 0; c < n_cls; c++) {                       // mean weight 16
                    plan->age_wf[c] = (float)((double)plan->age_wf[c] * 16.0 * (double)n_cls / wsum);
                    wmax = plan->age_wf[c] > wmax ? plan->age_wf[c] : wmax;
                }
                for (uint32_t c = 0; c < n_cls; c++) {
                    float w = plan->age_wf[c];
                    w = w < (float)wmax * 0.25f ? (float)wmax * 0.25f : w;
                    w = w > 60.0f ? 60.0f : w;
                    plan->age_wf[c] = w;
                    const uint32_t wi = (uint32_t)(w + 0.5f);
                    plan->age_w[c] = wi < 1u ? 1u : wi;
                }
                plan->age_valid = 1u;
                plan->age_cls = n_cls;
            }
            for (uint32_t c = 0; c < 8u; c++) {
                plan->life_sum[c] = 0ull;
                plan->life_cnt[c] = 0u;
            }
        }
    }
}
// order[cursor[bucket]++] = q, one global atomic per (block, bucket); consumes the costs
__global__ void __launch_bounds__(256) plan_scatter(uint32_t* __restrict__ cost, uint32_t np, PlanBuf* plan, uint32_t* __restrict__ order) {
    __shared__ uint32_t cnt[256], base[256];
    for (uint32_t q0 = blockIdx.x * 256u; q0 < np; q0 += gridDim.x * 256u) {
        cnt[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t q = q0 + threadIdx.x;
        uint32_t b = 0, slot = 0;
        if (q < np) {
            b = cost_bucket(cost[q]);
            cost[q] = 0;
            slot = atomicAdd(&cnt[b], 1u);
        }
        __syncthreads();
        if (cnt[threadIdx.x]) base[threadIdx.x] = atomicAdd(&plan->cursor[threadIdx.x], cnt[threadIdx.x]);
        __syncthreads();
        if (q < np) order[base[b] + slot] = q;
        __syncthreads();
    }
}
void launch_plan(uint32_t* cost, uint32_t* order, PlanBuf* plan, uint32_t np, uint32_t n_waves, int heavy_own, int mean_x16, int bulk_x16,
                 int tiny_waves, int n_cu, int n_cls, int chain_on, uint32_t chain_ref_waves, hipStream_t st) {
    (void)hipMemsetAsync(plan, 0, offsetof(PlanBuf, age_valid), st);      // (the self-tuned age weights and the lifetimes survive)
    long long need = ((long long)np + 255) / 256, grid = (long long)n_cu * 8;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(plan_hist, dim3((unsigned)grid), dim3(256), 0, st, cost, np, plan);
    hipLaunchKernelGGL(plan_scan, dim3(1), dim3(256), 0, st, plan, np, n_waves, (uint32_t)heavy_own, (uint32_t)mean_x16, (uint32_t)bulk_x16,
                       (uint32_t)tiny_waves, (uint32_t)n_cls, (uint32_t)chain_on, chain_ref_waves);
    hipLaunchKernelGGL(plan_scatter, dim3((unsigned)grid), dim3(256), 0, st, cost, np, plan, order);
}

// ---- launchers used by rt_capi.hip -------------------------------------------------------
#define RT_DISPATCH_SIG(sig, KERNEL, ...) \
        else if (kind == KIND_BOXES && P.n_obj == 8 && P.box_sig == sig) { auto k = KERNEL<KIND_BOXES, 8, sig>; __VA_ARGS__; }
#define RT_DISPATCH_KIND(KERNEL, ...)                                                       \
    do {                                                                                    \
        if (false) {}                                                                       \
        RT_BOX_SIGNATURES(RT_DISPATCH_SIG, KERNEL, __VA_ARGS__)                             \
        else if (kind == KIND_BOXES && P.n_obj == 8) { auto k = KERNEL<KIND_BOXES, 8>; __VA_ARGS__; }  \
        else if (kind == KIND_BOXES) { auto k = KERNEL<KIND_BOXES, 0>; __VA_ARGS__; }        \
        else if (kind == KIND_BUNNY) { auto k = KERNEL<KIND_BUNNY, 0>; __VA_ARGS__; }        \
        else if (kind == KIND_MIXED) { auto k = KERNEL<KIND_MIXED, 0>; __VA_ARGS__; }        \
        else { auto k = KERNEL<KIND_GENERIC, 0>; __VA_ARGS__; }                              \
    } while (0)

void launch_trace(const Params& P, int kind, int grid, hipStream_t st) {
    if (P.scheduler == 1) RT_DISPATCH_KIND(trace_paths_pool, hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, st, P));
    else RT_DISPATCH_KIND(trace_paths, hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, st, P));
}
void launch_primary(const Params& P, int kind, int n_cu, hipStream_t st) {
    long long need = ((long long)P.total_items + 255) / 256;
    long long grid = (long long)n_cu * 8;            // 32 waves per CU: the kernel needs only ~40 VGPRs
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    // scenes of up to 8 one-Lipschitz shapes use the culling instance (8-way unrolled, runtime object count)
    if (P.cull_ok && P.box_sig == 0 && kind == KIND_GENERIC) {
        hipLaunchKernelGGL((primary_rays<KIND_GENERIC, 8>), dim3((unsigned)grid), dim3(256), 0, st, P);
        return;
    }
    if (P.cull_ok && P.box_sig == 0 && kind == KIND_BOXES && P.n_obj < 8) {
        hipLaunchKernelGGL((primary_rays<KIND_BOXES, 8>), dim3((unsigned)grid), dim3(256), 0, st, P);
        return;
    }
    if (!P.cull_ok) {
        // the host could not bound the scene (non-finite / huge extent, steep cone, > 8 shapes): lock-step march
        // without culling; the NOBJ = 0 instances have no nearest_culled branch
        if (kind == KIND_BOXES) hipLaunchKernelGGL((primary_rays<KIND_BOXES, 0>), dim3((unsigned)grid), dim3(256), 0, st, P);
        else if (kind == KIND_BUNNY) hipLaunchKernelGGL((primary_rays<KIND_BUNNY, 0>), dim3((unsigned)grid), dim3(256), 0, st, P);
        else if (kind == KIND_MIXED) hipLaunchKernelGGL((primary_rays<KIND_MIXED, 0>), dim3((unsigned)grid), dim3(256), 0, st, P);
        else hipLaunchKernelGGL((primary_rays<KIND_GENERIC, 0>), dim3((unsigned)grid), dim3(256), 0, st, P);
        return;
    }
    RT_DISPATCH_KIND(primary_rays, hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), 0, st, P));
}
int trace_blocks_per_cu(int kind, int n_obj, uint32_t box_sig, int scheduler) {
    int per_cu = 0;
    hipError_t e = hipSuccess;
    Params P;
    P.n_obj = n_obj;
    P.box_sig = box_sig;
    if (scheduler == 1) RT_DISPATCH_KIND(trace_paths_pool, e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 256, 0));
    else RT_DISPATCH_KIND(trace_paths, e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 256, 0));
    return e == hipSuccess ? per_cu : 0;
}
// Small fills as a kernel of our own: a hipMemsetAsync of a few KB is a blit dispatch with ~10 us of idle queue on either side
// (rocprofv3 kernel trace of the one-step src/ launches: shade -> 11 us -> fill 3.7 us -> 10.5 us -> gen, while gen -> march -> shade
// follow each other without a gap): 25 us of a 160 us launch.
__global__ void __launch_bounds__(256) zero_words_kernel(uint4* p, int n16) {
    for (int i = threadIdx.x; i < n16; i += 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
void launch_zero(void* p, size_t bytes, hipStream_t st) {      // bytes: a multiple of 16, p 16-byte aligned
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<uint4*>(p), (int)(bytes / 16));
}
void launch_accumulate(const Params& P, int n_cu, hipStream_t st) {
    if (P.stage_dense) {
        const uint32_t K = (uint32_t)P.K;
        const size_t lds = (size_t)(P.acc_batch / K) * (3u * K + ((3u * K) & 1u ? 0u : 1u)) * sizeof(float);
        const long long n_batch = ((long long)P.total_items + P.acc_batch - 1) / P.acc_batch;
        const long long room = (long long)(n_cu > 0 ? n_cu : 256) * 8;
        hipLaunchKernelGGL(accumulate_dense, dim3((unsigned)(n_batch < room ? n_batch : room)), dim3(256), lds, st, P);
        return;
    }
    int grid = (P.np + 255) / 256;
    hipLaunchKernelGGL(accumulate_samples, dim3(grid), dim3(256), 0, st, P);
}
void launch_persistent_pool(const Params& P, int kind, int steps, int grid, hipStream_t st) {
    if (kind == KIND_BOXES) hipLaunchKernelGGL((persistent_pool<KIND_BOXES>), dim3(grid), dim3(256), 0, st, P, steps);
    else if (kind == KIND_BUNNY) hipLaunchKernelGGL((persistent_pool<KIND_BUNNY>), dim3(grid), dim3(256), 0, st, P, steps);
    else if (kind == KIND_MIXED) hipLaunchKernelGGL((persistent_pool<KIND_MIXED>), dim3(grid), dim3(256), 0, st, P, steps);
    else hipLaunchKernelGGL((persistent_pool<KIND_GENERIC>), dim3(grid), dim3(256), 0, st, P, steps);
}
void launch_chain_steps(const Params& P, int kind, int steps, int grid, hipStream_t st) {
    if (kind == KIND_BOXES) hipLaunchKernelGGL((chain_steps<KIND_BOXES>), dim3(grid), dim3(256), 0, st, P, steps);
    else if (kind == KIND_BUNNY) hipLaunchKernelGGL((chain_steps<KIND_BUNNY>), dim3(grid), dim3(256), 0, st, P, steps);
    else if (kind == KIND_MIXED) hipLaunchKernelGGL((chain_steps<KIND_MIXED>), dim3(grid), dim3(256), 0, st, P, steps);
    else hipLaunchKernelGGL((chain_steps<KIND_GENERIC>), dim3(grid), dim3(256), 0, st, P, steps);
}
int persistent_pool_blocks_per_cu(int kind) {
    int per_cu = 0;
    hipError_t e;
    if (kind == KIND_BOXES) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persistent_pool<KIND_BOXES>, 256, 0);
    else if (kind == KIND_BUNNY) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persistent_pool<KIND_BUNNY>, 256, 0);
    else if (kind == KIND_MIXED) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persistent_pool<KIND_MIXED>, 256, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persistent_pool<KIND_GENERIC>, 256, 0);
    return e == hipSuccess ? per_cu : 0;
}
void launch_persistent(const Params& P, int kind, int steps, hipStream_t st) {
    int grid = (P.np + 255) / 256;
    if (kind == KIND_BOXES) hipLaunchKernelGGL((persistent_steps<KIND_BOXES>), dim3(grid), dim3(256), 0, st, P, steps);
    else if (kind == KIND_BUNNY) hipLaunchKernelGGL((persistent_steps<KIND_BUNNY>), dim3(grid), dim3(256), 0, st, P, steps);
    else if (kind == KIND_MIXED) hipLaunchKernelGGL((persistent_steps<KIND_MIXED>), dim3(grid), dim3(256), 0, st, P, steps);
    else hipLaunchKernelGGL((persistent_steps<KIND_GENERIC>), dim3(grid), dim3(256), 0, st, P, steps);
}
// the wavefront split of one src/ bounce-step (rt_split.hpp): gen and shade one lane per local pixel, march on persistent waves
void launch_src_gen(const Params& P, int kind, hipStream_t st) {
    int grid = (P.np + 255) / 256;
    if (kind == KIND_BOXES) hipLaunchKernelGGL((src_gen<KIND_BOXES>), dim3(grid), dim3(256), 0, st, P);
    else if (kind == KIND_BUNNY) hipLaunchKernelGGL((src_gen<KIND_BUNNY>), dim3(grid), dim3(256), 0, st, P);
    else if (kind == KIND_MIXED) hipLaunchKernelGGL((src_gen<KIND_MIXED>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((src_gen<KIND_GENERIC>), dim3(grid), dim3(256), 0, st, P);
}
void launch_src_march(const Params& P, int kind, int grid, hipStream_t st) {
    if (kind == KIND_BOXES) hipLaunchKernelGGL((src_march<KIND_BOXES>), dim3(grid), dim3(256), 0, st, P);
    else if (kind == KIND_BUNNY) hipLaunchKernelGGL((src_march<KIND_BUNNY>), dim3(grid), dim3(256), 0, st, P);
    else if (kind == KIND_MIXED) hipLaunchKernelGGL((src_march<KIND_MIXED>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((src_march<KIND_GENERIC>), dim3(grid), dim3(256), 0, st, P);
}
int src_march_blocks_per_cu(int kind) {
    int per_cu = 0;
    hipError_t e;
    if (kind == KIND_BOXES) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, src_march<KIND_BOXES>, 256, 0);
    else if (kind == KIND_BUNNY) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, src_march<KIND_BUNNY>, 256, 0);
    else if (kind == KIND_MIXED) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, src_march<KIND_MIXED>, 256, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, src_march<KIND_GENERIC>, 256, 0);
    return e == hipSuccess ? per_cu : 0;
}
template <bool COUNT>
static void launch_src_shade_gen_t(const Params& P, int kind, hipStream_t st) {
    int grid = (P.np + 255) / 256;
    if (kind == KIND_BOXES) hipLaunchKernelGGL((src_shade_gen<KIND_BOXES, COUNT>), dim3(grid), dim3(256), 0, st, P);
    else if (kind == KIND_BUNNY) hipLaunchKernelGGL((src_shade_gen<KIND_BUNNY, COUNT>), dim3(grid), dim3(256), 0, st, P);
    else if (kind == KIND_MIXED) hipLaunchKernelGGL((src_shade_gen<KIND_MIXED, COUNT>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((src_shade_gen<KIND_GENERIC, COUNT>), dim3(grid), dim3(256), 0, st, P);
}
void launch_src_shade_gen(const Params& P, int kind, bool count, hipStream_t st) {
    if (count) launch_src_shade_gen_t<true>(P, kind, st);
    else launch_src_shade_gen_t<false>(P, kind, st);
}
void launch_src_shade(const Params& P, int kind, hipStream_t st) {
    int grid = (P.np + 255) / 256;
    if (kind == KIND_BOXES) hipLaunchKernelGGL((src_shade<KIND_BOXES>), dim3(grid), dim3(256), 0, st, P);
    else if (kind == KIND_BUNNY) hipLaunchKernelGGL((src_shade<KIND_BUNNY>), dim3(grid), dim3(256), 0, st, P);
    else if (kind == KIND_MIXED) hipLaunchKernelGGL((src_shade<KIND_MIXED>), dim3(grid), dim3(256), 0, st, P);
    else hipLaunchKernelGGL((src_shade<KIND_GENERIC>), dim3(grid), dim3(256), 0, st, P);
}
void launch_refresh(float4* ib, rtpbr_ray* rb, float2* db, float* dp, int adaptive, size_t n, hipStream_t st) {
    int grid = (int)((n + 255) / 256);
    hipLaunchKernelGGL(refresh_kernel, dim3(grid), dim3(256), 0, st, ib, rb, db, dp, adaptive, n);
}
void launch_post_process(const Params& P, hipStream_t st) {
    size_t n = (size_t)P.cfg.width * P.cfg.height;
    int grid = (int)((n + 255) / 256);
    hipLaunchKernelGGL(post_process_kernel, dim3(grid), dim3(256), 0, st, P);
}
void launch_pack(const Params& P, float4* dst, hipStream_t st) {
    int grid = (P.np + 255) / 256;
    hipLaunchKernelGGL(pack_tiles_kernel, dim3(grid), dim3(256), 0, st, P, dst);
}
void launch_unpack(const Params& P, const float4* src, hipStream_t st) {
    int grid = (P.np + 255) / 256;
    hipLaunchKernelGGL(unpack_tiles_kernel, dim3(grid), dim3(256), 0, st, P, src);
}
void launch_sqrt_exhaustive(unsigned long long* mismatches, hipStream_t st) {
    hipLaunchKernelGGL(sqrt_exhaustive, dim3(256 * 16), dim3(256), 0, st, mismatches);
}
void launch_math_probe(int op, const float* a, const float* b, float* out, float* out2, int n, hipStream_t st) {
    int grid = (n + 255) / 256;
    hipLaunchKernelGGL(math_probe, dim3(grid), dim3(256), 0, st, op, a, b, out, out2, n);
}

}  // namespace rt
