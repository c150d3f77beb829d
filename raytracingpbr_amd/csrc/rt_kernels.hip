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
    if (q >= (uint32_t)P.np) return;
    int x, y;
    if (!pixel_of(P, q, x, y)) return;
    float4* dst = P.image_buffer + ((size_t)x * P.cfg.height + y);
    float4 acc = *dst;
    // a pixel's K records are contiguous (item-linear staging): fetch them 8 at a time (128 B per
    // lane, every cache line is touched once) and add them strictly in sample order
    const float4* src = P.stage + (size_t)q * (size_t)P.K;
    int k = 0;
    for (; k + 8 <= P.K; k += 8) {
        float4 c[8];
#pragma unroll
        for (int i = 0; i < 8; i++) c[i] = src[k + i];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            acc.x += c[i].x;
            acc.y += c[i].y;
            acc.z += c[i].z;
            acc.w += 1.0f;
        }
    }
    for (; k < P.K; k++) {
        float4 c = src[k];
        acc.x += c.x;
        acc.y += c.y;
        acc.z += c.z;
        acc.w += 1.0f;
    }
    *dst = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { /* deposits are counted on the host: pixels*K */ }
}

// -------------------------------------------------------------------------------------------
// src/ persistent-ray form: russian_roulette -> track_once -> raytrace (src/pathtracer.py:16-91),
// raycast src/scene.py:59-84.  One lane per owned pixel; `steps` bounce-steps per launch.
template <int KIND>
__global__ void __launch_bounds__(256) persistent_steps(const Params P, int steps) {
    __shared__ ObjFull lds_obj[MAX_OBJ];
    stage_objects(P, lds_obj);
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    int px = 0, py = 0;
    bool valid = q < (uint32_t)P.np && pixel_of(P, q, px, py);
    // self-adaptive sampling mask (src/pathtracer.py:97-101)
    if (valid && P.cfg.adaptive_sampling && !(P.diff_pixels[(size_t)px * P.cfg.height + py] > P.cfg.noise_threshold)) valid = false;
    uint32_t n_steps = 0, n_raycasts = 0, n_hits = 0, n_sky = 0, n_samples = 0, n_dep = 0;
    if (valid) {
        const rtpbr_config& g = P.cfg;
        size_t pi = (size_t)px * g.height + py;
        rtpbr_ray rb = P.ray_buffer[pi];
        vec3 o = mk(rb.origin[0], rb.origin[1], rb.origin[2]);
        vec3 d = mk(rb.direction[0], rb.direction[1], rb.direction[2]);
        vec3 col = mk(rb.color[0], rb.color[1], rb.color[2]);
        int depth = rb.depth;
        float4 acc = P.image_buffer[pi];
        for (int s = 0; s < steps; s++) {
            uint32_t key = rng_key(g.seed, (uint32_t)px, (uint32_t)py, P.sample_base + (uint32_t)s), cnt = 0;
            // russian_roulette :65-77
            float p = (depth == 0) ? 1.0f : g.quality_per_sample;
            p -= (float)depth * (1.0f / (float)g.max_raytrace);
            if (rng_next(key, cnt) > p) {
                col = mk(0, 0, 0);
                depth = -depth;
            } else {
                col = col * (1.0f / p);
                // track_once :53-62
                if (depth < 1 || depth > g.max_raytrace) {
                    acc.x += col.x;
                    acc.y += col.y;
                    acc.z += col.z;
                    acc.w += 1.0f;
                    n_dep++;
                    gen_ray(P, px, py, key, cnt, o, d);
                    col = mk(1, 1, 1);
                    depth = 0;
                }
                // raycast src/scene.py:59-84
                float t = 0.0f, w = g.omega0, sstep = 0.0f, dist = g.max_dis;
                int idx = 0;
                bool hit = false;
                for (int it = 0; it < g.max_raymarch; it++) {
                    float ld = dist;
                    nearest<KIND, 0>(P, o, idx, dist);
                    n_steps++;
                    if (w > 1.0f && ld + dist < sstep) {
                        sstep -= w * sstep;
                        w = 1.0f;
                        t += sstep;
                        o = fma3(sstep, d, o);
                        continue;
                    }
                    sstep = w * dist;
                    t += sstep;
                    o = fma3(sstep, d, o);
                    hit = dist < t * g.hit_eps;
                    if (hit || t >= g.max_dis) break;
                }
                depth += 1;
                n_raycasts++;
                // raytrace :16-36
                if (hit) {
                    const ObjFull ob = lds_obj[idx];
                    surface_interaction<KIND>(P, ob, o, o, d, col, key, cnt);
                    n_hits++;
                    float intensity = brightness(col);
                    col = col * mk(ob.emission[0], ob.emission[1], ob.emission[2]);
                    float visible = brightness(col);
                    bool stop = intensity < visible || visible < g.vis_lo || visible > g.vis_hi;
                    if (stop) depth = -depth;
                } else {
                    depth = -depth;
                    col = col * sky_color(P, d);
                    n_sky++;
                    if (g.primary_miss == RTPBR_PRIMARY_BLACK) col = col * (depth < -1 ? 1.0f : 0.0f);
                }
            }
            n_samples++;
        }
        rb.origin[0] = o.x; rb.origin[1] = o.y; rb.origin[2] = o.z;
        rb.direction[0] = d.x; rb.direction[1] = d.y; rb.direction[2] = d.z;
        rb.color[0] = col.x; rb.color[1] = col.y; rb.color[2] = col.z;
        rb.depth = depth;
        P.ray_buffer[pi] = rb;
        P.image_buffer[pi] = acc;
    }
    flush_counters(P, n_steps, n_raycasts, n_hits, n_sky, n_samples, n_dep);
}

// -------------------------------------------------------------------------------------------
// src/ persistent-ray form on the LDS ray pool ("persistent-lane
// scheduler").  Same arithmetic as persistent_steps, but a CONTEXT (one pixel advancing through
// its `steps` bounce-steps) is decoupled from a lane: contexts whose raycast finished are parked
// in the wave's LDS slots for shading while the lane takes over a parked context that is ready
// to march, exactly like trace_paths_pool.  A pixel is owned by one context for the whole launch,
// so its deposits into image_buffer happen in step order (bit-exact with the sequential form).
enum { G_OX = 0, G_OY, G_OZ, G_DX, G_DY, G_DZ, G_CR, G_CG, G_CB, G_DEPTH, G_IDX, G_Q, G_S, G_KEY, G_CNT, G_COUNT };
static_assert(G_COUNT == POOL_WORDS, "pixel-context record must fill the pool record");

struct PixCtx {
    vec3 o, d, col;
    int depth, idx;
    uint32_t q;
    int s;
    uint32_t key, cnt;
};

// one iteration of raycast() src/scene.py:59-84 (the ray origin itself moves)
template <int KIND>
RT_D void march_step_src(const Params& P, Lane& L) {
    float ld = L.dist;
    int idx;
    float dist;
    nearest<KIND, 0>(P, L.o, idx, dist);
    L.idx = idx;
    L.dist = dist;
    L.n_steps++;
    L.steps_left--;
    bool fb = (L.w > 1.0f) && (ld + dist < L.s);
    float s_fb = L.s - L.w * L.s;
    float s_nm = L.w * dist;
    float s_new = fb ? s_fb : s_nm;
    L.w = fb ? 1.0f : L.w;
    L.s = s_new;
    L.t += s_new;
    L.o = fma3(s_new, L.d, L.o);
    bool hit = !fb && (dist < L.t * P.cfg.hit_eps);
    bool done = (!fb && (hit || L.t >= P.cfg.max_dis)) || L.steps_left == 0;
    if (done) L.state = hit ? ST_HIT : ST_MISS;
}

// Advance a context through the part of its step sequence that needs no marching: roulette,
// deposit + camera-ray regeneration (src/pathtracer.py:53-77).  Returns true when the context
// is ready to march its next raycast, false when all `steps` are done (state written back).
template <int KIND>
RT_D bool pix_advance(const Params& P, PixCtx& X, int steps, uint32_t& n_samples, uint32_t& n_dep) {
    const rtpbr_config& g = P.cfg;
    int px, py;
    pixel_of(P, X.q, px, py);
    const size_t pi = (size_t)px * g.height + py;
    while (X.s < steps) {
        X.key = rng_key(g.seed, (uint32_t)px, (uint32_t)py, P.sample_base + (uint32_t)X.s);
        X.cnt = 0;
        float p = (X.depth == 0) ? 1.0f : g.quality_per_sample;
        p -= (float)X.depth * (1.0f / (float)g.max_raytrace);
        if (rng_next(X.key, X.cnt) > p) {
            X.col = mk(0, 0, 0);
            X.depth = -X.depth;
            X.s++;
            n_samples++;
            continue;
        }
        X.col = X.col * (1.0f / p);
        if (X.depth < 1 || X.depth > g.max_raytrace) {
            float4 acc = P.image_buffer[pi];
            acc.x += X.col.x;
            acc.y += X.col.y;
            acc.z += X.col.z;
            acc.w += 1.0f;
            P.image_buffer[pi] = acc;
            n_dep++;
            gen_ray(P, px, py, X.key, X.cnt, X.o, X.d);
            X.col = mk(1, 1, 1);
            X.depth = 0;
        }
        return true;
    }
    rtpbr_ray rb;
    rb.origin[0] = X.o.x; rb.origin[1] = X.o.y; rb.origin[2] = X.o.z;
    rb.direction[0] = X.d.x; rb.direction[1] = X.d.y; rb.direction[2] = X.d.z;
    rb.color[0] = X.col.x; rb.color[1] = X.col.y; rb.color[2] = X.col.z;
    rb.depth = X.depth;
    P.ray_buffer[pi] = rb;
    return false;
}

template <int KIND>
__global__ void __launch_bounds__(256) persistent_pool(const Params P, int steps) {
    __shared__ ObjFull lds_obj[MAX_OBJ];
    __shared__ uint32_t pool_all[4][G_COUNT][64];
    __shared__ uint32_t sstate_all[4][64];
    __shared__ uint32_t tbl_all[4][64];
    stage_objects(P, lds_obj);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t (*pool)[64] = pool_all[wave];
    uint32_t* sstate = sstate_all[wave];
    const PoolView V = {pool_all[wave], sstate_all[wave], tbl_all[wave]};
    sstate[lane] = SL_EMPTY;

    Lane L;
    L.state = ST_IDLE;
    L.n_steps = L.n_raycasts = L.n_hits = L.n_sky = 0;
    L.o = L.d = mk(0, 0, 0);
    L.t = L.w = L.s = L.dist = L.t_eval = 0.0f;
    L.idx = 0;
    L.steps_left = 0;
    // bookkeeping of the context being marched
    vec3 a_col = mk(0, 0, 0);
    int a_depth = 0, a_s = 0;
    uint32_t a_q = 0, a_key = 0, a_cnt = 0;
    uint32_t n_samples = 0, n_dep = 0;
    WorkRange wr = {0, 0, false};
    unsigned long long m_ready = 0, m_shade = 0;
    const int T = P.shade_lanes;
    const int m_swap = P.swap_lanes;

    auto f2u = [](float x) { return __builtin_bit_cast(uint32_t, x); };
    auto u2f = [](uint32_t x) { return __builtin_bit_cast(float, x); };
    auto src_march_init = [&](Lane& l) {
        l.t = 0.0f;
        l.w = P.cfg.omega0;
        l.s = 0.0f;
        l.dist = P.cfg.max_dis;
        l.steps_left = P.cfg.max_raymarch;
        l.state = ST_MARCH;
        l.n_raycasts++;
    };

    for (;;) {
        // ================================================================ phase B on the slots
        {
            const int n_shade = __popcll(m_shade);
            const int n_ready = __popcll(m_ready);
            const int n_free = 64 - n_shade - n_ready;
            const bool run_b = n_shade >= T || (n_ready == 0 && (n_shade > 0 || (n_free > 0 && !wr.drained)));
            if (run_b) {
                uint32_t st = sstate[lane];
                PixCtx X;
                X.o = X.d = X.col = mk(0, 0, 0);
                X.depth = X.idx = X.s = 0;
                X.q = X.key = X.cnt = 0;
                bool have = false;
                if (st == SL_HIT || st == SL_MISS) {
                    X.o = mk(u2f(pool[G_OX][lane]), u2f(pool[G_OY][lane]), u2f(pool[G_OZ][lane]));
                    X.d = mk(u2f(pool[G_DX][lane]), u2f(pool[G_DY][lane]), u2f(pool[G_DZ][lane]));
                    X.col = mk(u2f(pool[G_CR][lane]), u2f(pool[G_CG][lane]), u2f(pool[G_CB][lane]));
                    X.depth = (int)pool[G_DEPTH][lane];
                    X.idx = (int)pool[G_IDX][lane];
                    X.q = pool[G_Q][lane];
                    X.s = (int)pool[G_S][lane];
                    X.key = pool[G_KEY][lane];
                    X.cnt = pool[G_CNT][lane];
                    // raytrace() src/pathtracer.py:16-36 after raycast(); depth += 1 (scene.py:83)
                    X.depth += 1;
                    if (st == SL_HIT) {
                        const ObjFull ob = lds_obj[X.idx];
                        surface_interaction<KIND>(P, ob, X.o, X.o, X.d, X.col, X.key, X.cnt);
                        L.n_hits++;
                        float intensity = brightness(X.col);
                        X.col = X.col * mk(ob.emission[0], ob.emission[1], ob.emission[2]);
                        float visible = brightness(X.col);
                        bool stop = intensity < visible || visible < P.cfg.vis_lo || visible > P.cfg.vis_hi;
                        if (stop) X.depth = -X.depth;
                    } else {
                        X.depth = -X.depth;
                        X.col = X.col * sky_color(P, X.d);
                        L.n_sky++;
                        if (P.cfg.primary_miss == RTPBR_PRIMARY_BLACK) X.col = X.col * (X.depth < -1 ? 1.0f : 0.0f);
                    }
                    X.s++;
                    n_samples++;
                    have = true;
                    st = SL_EMPTY;
                }
                // free slots take the next pixel of this wave's range
                uint32_t newq = 0;
                bool got = claim_items(P, wr, st == SL_EMPTY && !have, lane, newq);
                if (got) {
                    int px, py;
                    if (pixel_of(P, newq, px, py)) {
                        const size_t pi = (size_t)px * P.cfg.height + py;
                        bool masked = P.cfg.adaptive_sampling && !(P.diff_pixels[pi] > P.cfg.noise_threshold);
                        if (!masked) {
                            rtpbr_ray rb = P.ray_buffer[pi];
                            X.o = mk(rb.origin[0], rb.origin[1], rb.origin[2]);
                            X.d = mk(rb.direction[0], rb.direction[1], rb.direction[2]);
                            X.col = mk(rb.color[0], rb.color[1], rb.color[2]);
                            X.depth = rb.depth;
                            X.q = newq;
                            X.s = 0;
                            have = true;
                        }
                    }
                }
                bool ready = false;
                if (have) ready = pix_advance<KIND>(P, X, steps, n_samples, n_dep);
                if (ready) {
                    pool[G_OX][lane] = f2u(X.o.x); pool[G_OY][lane] = f2u(X.o.y); pool[G_OZ][lane] = f2u(X.o.z);
                    pool[G_DX][lane] = f2u(X.d.x); pool[G_DY][lane] = f2u(X.d.y); pool[G_DZ][lane] = f2u(X.d.z);
                    pool[G_CR][lane] = f2u(X.col.x); pool[G_CG][lane] = f2u(X.col.y); pool[G_CB][lane] = f2u(X.col.z);
                    pool[G_DEPTH][lane] = (uint32_t)X.depth;
                    pool[G_Q][lane] = X.q;
                    pool[G_S][lane] = (uint32_t)X.s;
                    pool[G_KEY][lane] = X.key;
                    pool[G_CNT][lane] = X.cnt;
                    st = SL_READY;
                }
                sstate[lane] = st;
                m_ready = __ballot(st == SL_READY);
                m_shade = 0;
            }
        }

        // ================================================================ dispatch (pool_swap, as in trace_paths_pool)
        {
            const bool is_done = L.state == ST_HIT || L.state == ST_MISS;
            uint32_t rec[POOL_WORDS];
            rec[G_OX] = f2u(L.o.x); rec[G_OY] = f2u(L.o.y); rec[G_OZ] = f2u(L.o.z);
            rec[G_DX] = f2u(L.d.x); rec[G_DY] = f2u(L.d.y); rec[G_DZ] = f2u(L.d.z);
            rec[G_CR] = f2u(a_col.x); rec[G_CG] = f2u(a_col.y); rec[G_CB] = f2u(a_col.z);
            rec[G_DEPTH] = (uint32_t)a_depth;
            rec[G_IDX] = (uint32_t)L.idx;
            rec[G_Q] = a_q;
            rec[G_S] = (uint32_t)a_s;
            rec[G_KEY] = a_key; rec[G_CNT] = a_cnt;
            const int r = pool_swap(V, lane, is_done, L.state == ST_IDLE, L.state == ST_HIT ? SL_HIT : SL_MISS, rec, m_ready, m_shade);
            if (r & 2) L.state = ST_IDLE;
            if (r & 1) {
                L.o = mk(u2f(rec[G_OX]), u2f(rec[G_OY]), u2f(rec[G_OZ]));
                L.d = mk(u2f(rec[G_DX]), u2f(rec[G_DY]), u2f(rec[G_DZ]));
                a_col = mk(u2f(rec[G_CR]), u2f(rec[G_CG]), u2f(rec[G_CB]));
                a_depth = (int)rec[G_DEPTH];
                a_q = rec[G_Q];
                a_s = (int)rec[G_S];
                a_key = rec[G_KEY]; a_cnt = rec[G_CNT];
                src_march_init(L);
            }
        }

        // ================================================================ march
        {
            int n_march = __popcll(__ballot(L.state == ST_MARCH));
            if (n_march == 0) {
                const bool any_ray = __ballot(L.state != ST_IDLE) != 0;
                if (!any_ray && m_ready == 0 && m_shade == 0 && wr.drained) break;
                continue;
            }
            const int n_ready = __popcll(m_ready);
            int n_done;
            do {
                if (L.state == ST_MARCH) march_step_src<KIND>(P, L);
                n_march = __popcll(__ballot(L.state == ST_MARCH));
                n_done = __popcll(__ballot(L.state == ST_HIT || L.state == ST_MISS));
            } while (n_march > 0 && n_done < (n_ready > 0 ? m_swap : 2 * m_swap));
        }
    }
    flush_counters(P, L.n_steps, L.n_raycasts, L.n_hits, L.n_sky, n_samples, n_dep);
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
void launch_accumulate(const Params& P, hipStream_t st) {
    int grid = (P.np + 255) / 256;
    hipLaunchKernelGGL(accumulate_samples, dim3(grid), dim3(256), 0, st, P);
}
void launch_persistent_pool(const Params& P, int kind, int steps, int grid, hipStream_t st) {
    if (kind == KIND_BOXES) hipLaunchKernelGGL((persistent_pool<KIND_BOXES>), dim3(grid), dim3(256), 0, st, P, steps);
    else if (kind == KIND_BUNNY) hipLaunchKernelGGL((persistent_pool<KIND_BUNNY>), dim3(grid), dim3(256), 0, st, P, steps);
    else if (kind == KIND_MIXED) hipLaunchKernelGGL((persistent_pool<KIND_MIXED>), dim3(grid), dim3(256), 0, st, P, steps);
    else hipLaunchKernelGGL((persistent_pool<KIND_GENERIC>), dim3(grid), dim3(256), 0, st, P, steps);
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
