// rt_chain.hpp — the CHAIN KERNEL of the src/ persistent-ray form: the heaviest pixels of a chain-bound launch, beside the pool
// kernel.  A fused launch of `steps` bounce-steps cannot end before its heaviest pixel has walked its dependency chain — at
// 768x432 (the reference's own window, src/config.py:7) 48 000 sequential march steps of a pixel that looks into the wedge
// where a sphere rests on the ground — and in the pool kernel that pixel shares a wave, the pool bookkeeping and the one-bound
// tracked march with others: measured 1 250 cycles per step of the critical lane against ~600 for the steps themselves.  So
// when the plan finds the launch chain-bound (rt_kernels.hip plan_scan: longest chain > 3x a wave's share of the frame) the
// head of the cost-ordered list — the CHAIN SET — leaves the pool kernel and runs here, concurrently, on a second stream:
//   * wave w walks the pixels order[chain_start[w] .. chain_start[w + 1]): the very heaviest ALONE in their wave (a lone
//     lane never waits for anybody: every step is as cheap as its own arithmetic allows), lighter ones in twos, fours, eights,
//     packed by the plan so that every wave's predicted time stays below the heaviest pixel's;
//   * one lane per pixel, lock step over the bounce-steps (src/pathtracer.py:65-91 as written: roulette, track_once, raycast,
//     raytrace), no pool, no swap, no passes;
//   * the raycast runs the two-bound tracked march (tracked_iteration<TWO>: one- and two-object lean loops) — this kernel is
//     small enough for the 21 pair instances that do not fit beside the pool kernel's code.
// Same device functions, same RNG positions, one context per pixel: ray_buffer, image_buffer and the counters are bit for bit
// the pool kernel's (tests/test_gpu_parity.py).
#pragma once
#include "rt_persistent.hpp"

namespace rt {

template <int KIND, int NOBJ = 0, uint32_t SIG = 0>
RT_D void chain_steps_impl(const Params& P, int steps) {
    __shared__ ObjFull lds_obj[MAX_OBJ];
    __shared__ float4 xch_all[4][8];      // per wave: the exchange buffer of the object-parallel evaluation (nearest_op3)
    stage_objects(P, lds_obj);
    const OpView OV = {lds_obj, ((P.src_op & 1) != 0 && P.n_obj <= 8) ? xch_all[threadIdx.x >> 6] : nullptr};
    const int lane = threadIdx.x & 63;
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
    uint32_t n_steps = 0, n_raycasts = 0, n_hits = 0, n_sky = 0, n_samples = 0, n_dep = 0;
    const uint32_t n_cw = (P.order && P.plan) ? P.plan->n_chain_waves : 0u;
    if (w < n_cw) {
        // the launch ends when the slowest of these waves does: they issue ahead of the pool kernel's waves on their SIMD
        __builtin_amdgcn_s_setprio((short)3);
        const uint32_t base = P.plan->chain_start[w];
        const uint32_t k = P.plan->chain_start[w + 1u] - base;
        const rtpbr_config& g = P.cfg;
        uint32_t q = 0;
        int px = 0, py = 0;
        bool valid = (uint32_t)lane < k;
        if (valid) {
            q = P.order[base + (uint32_t)lane];
            valid = q < (uint32_t)P.np && pixel_of(P, q, px, py);
        }
        const size_t pi = (size_t)px * g.height + py;
        // self-adaptive sampling mask (src/pathtracer.py:97-101)
        if (valid && g.adaptive_sampling && !(P.diff_pixels[pi] > g.noise_threshold)) valid = false;
        Lane L;
        L.state = ST_IDLE;
        L.n_steps = L.n_raycasts = L.n_hits = L.n_sky = 0;
        L.o = L.d = mk(0, 0, 0);
        L.t = L.w = L.s = L.dist = L.t_eval = 0.0f;
        L.idx = 0;
        L.steps_left = 0;
        vec3 col = mk(0, 0, 0);
        int depth = 0;
        if (valid) {
            const rtpbr_ray rb = P.ray_buffer[pi];
            L.o = mk(rb.origin[0], rb.origin[1], rb.origin[2]);
            L.d = mk(rb.direction[0], rb.direction[1], rb.direction[2]);
            col = mk(rb.color[0], rb.color[1], rb.color[2]);
            depth = rb.depth;
        }
        constexpr bool TRK = KIND == KIND_BOXES || KIND == KIND_GENERIC;
        const bool trk_ok = TRK && P.cull_ok != 0 && P.src_track != 0;
        Trk Tk = {-1.0f, -1.0f, 0};
#ifdef RT_DEBUG_PHASE
        const unsigned long long t_w0 = __builtin_readcyclecounter();
        unsigned long long t_march = 0;
        unsigned dbg_form[7] = {0, 0, 0, 0, 0, 0, 0}, dbg_steps[7] = {0, 0, 0, 0, 0, 0, 0};
#endif
        for (int s = 0; s < steps; s++) {
            uint32_t key = 0, cnt = 0;
            bool need = false;
            if (valid) {
                key = rng_key(g.seed, (uint32_t)px, (uint32_t)py, P.sample_base + (uint32_t)s);
                // russian_roulette :65-77
                float p = (depth == 0) ? 1.0f : g.quality_per_sample;
                p -= (float)depth * (1.0f / (float)g.max_raytrace);
                if (rng_next(key, cnt) > p) {
                    col = mk(0, 0, 0);
                    depth = -depth;
                    n_samples++;
                } else {
                    col = col * (1.0f / p);
                    // track_once :53-62
                    if (depth < 1 || depth > g.max_raytrace) {
                        float4 acc = P.image_buffer[pi];
                        acc.x += col.x;
                        acc.y += col.y;
                        acc.z += col.z;
                        acc.w += 1.0f;
                        P.image_buffer[pi] = acc;
                        n_dep++;
                        gen_ray(P, px, py, key, cnt, L.o, L.d);
                        col = mk(1, 1, 1);
                        depth = 0;
                    }
                    need = true;
                    // start of raycast() src/scene.py:60-63
                    L.t = 0.0f;
                    L.w = g.omega0;
                    L.s = 0.0f;
                    L.dist = g.max_dis;
                    L.steps_left = g.max_raymarch;
                    L.state = ST_MARCH;
                    n_raycasts++;
                    Tk.lb2 = Tk.lb3 = -1.0f;
                }
            }
            // ---- raycast (wave-uniform loop: all lanes of the wave march their rays together)
            int n_march = __popcll(__ballot(L.state == ST_MARCH));
#ifdef RT_DEBUG_PHASE
            const unsigned long long t_m0 = __builtin_readcyclecounter();
#endif
            while (n_march > 0) {
                if constexpr (TRK) {
                    if (trk_ok) {
                        int it = 1;
#ifdef RT_DEBUG_PHASE
                        const int form = tracked_iteration<KIND, NOBJ, SIG, true>(P, L, Tk, n_march, 0, it, nullptr, nullptr, OV);
                        dbg_form[form]++;
                        dbg_steps[form] += (unsigned)it;
#else
                        tracked_iteration<KIND, NOBJ, SIG, true>(P, L, Tk, n_march, 0, it, nullptr, nullptr, OV);
#endif
                    } else if (L.state == ST_MARCH) {
                        march_step_src<KIND, NOBJ, SIG>(P, L);
                    }
                } else {
                    if (L.state == ST_MARCH) march_step_src<KIND, NOBJ, SIG>(P, L);
                }
                n_march = __popcll(__ballot(L.state == ST_MARCH));
            }
#ifdef RT_DEBUG_PHASE
            t_march += __builtin_readcyclecounter() - t_m0;
#endif
            // ---- raytrace() src/pathtracer.py:16-36 after raycast(); depth += 1 (scene.py:83)
            if (need) {
                depth += 1;
                if (L.state == ST_HIT) {
                    const ObjFull ob = lds_obj[L.idx];
                    surface_interaction<KIND>(P, ob, L.o, L.o, L.d, col, key, cnt);
                    n_hits++;
                    float intensity = brightness(col);
                    col = col * mk(ob.emission[0], ob.emission[1], ob.emission[2]);
                    float visible = brightness(col);
                    bool stop = intensity < visible || visible < g.vis_lo || visible > g.vis_hi;
                    if (stop) depth = -depth;
                } else {
                    depth = -depth;
                    col = col * sky_color(P, L.d);
                    n_sky++;
                    if (g.primary_miss == RTPBR_PRIMARY_BLACK) col = col * (depth < -1 ? 1.0f : 0.0f);
                }
                n_samples++;
                L.state = ST_IDLE;
            }
        }
        if (valid) {
            rtpbr_ray rb;
            rb.origin[0] = L.o.x; rb.origin[1] = L.o.y; rb.origin[2] = L.o.z;
            rb.direction[0] = L.d.x; rb.direction[1] = L.d.y; rb.direction[2] = L.d.z;
            rb.color[0] = col.x; rb.color[1] = col.y; rb.color[2] = col.z;
            rb.depth = depth;
            P.ray_buffer[pi] = rb;
            if (P.cost_buffer) P.cost_buffer[q] += L.n_steps;
        }
        n_steps = L.n_steps;
#ifdef RT_DEBUG_PHASE
        {   // per-wave record over diff_buffer (unused without adaptive sampling): pixels, lifetime, march cycles, lane 0's steps, iterations / steps by form
            const uint32_t tot = wave_sum(L.n_steps);
            if (lane == 0) {
                unsigned long long* d = reinterpret_cast<unsigned long long*>(P.diff_buffer) + (size_t)w * 8u;
                d[0] = (unsigned long long)k | ((unsigned long long)L.n_steps << 32);
                d[1] = __builtin_readcyclecounter() - t_w0;
                d[2] = t_march;
                d[3] = tot;
                d[4] = (unsigned long long)dbg_form[1] | ((unsigned long long)dbg_steps[1] << 32);
                d[5] = (unsigned long long)dbg_form[2] | ((unsigned long long)dbg_steps[2] << 32);
                d[6] = (unsigned long long)dbg_form[3] | ((unsigned long long)dbg_form[4] << 32);
                d[7] = 0x1234567ull;
            }
        }
#endif
    }
    flush_counters(P, n_steps, n_raycasts, n_hits, n_sky, n_samples, n_dep);
}

template <int KIND>
__global__ void __launch_bounds__(256) chain_steps(const Params P, int steps) { chain_steps_impl<KIND>(P, steps); }

}  // namespace rt
