// rt_persistent.hpp — the src/ persistent-ray form (src/pathtracer.py:16-103, src/scene.py:59-84): the lock-step kernel
// (one lane per pixel) and the pool kernel (contexts decoupled from lanes: static strided ownership, pass-major
// residencies, sparse-wave culling).  Included at the end of rt_trace.hpp, whose LDS ray pool (PoolView, pool_swap),
// counters and object staging it uses; compiled ahead of time (rt_kernels.hip) and per scene at run time (rt_jit_tu.hip).
#pragma once
#include "rt_trace.hpp"

namespace rt {

// -------------------------------------------------------------------------------------------
// src/ persistent-ray form: russian_roulette -> track_once -> raytrace (src/pathtracer.py:16-91),
// raycast src/scene.py:59-84.  One lane per owned pixel; `steps` bounce-steps per launch.
template <int KIND, int NOBJ = 0, uint32_t SIG = 0>
RT_D void persistent_steps_impl(const Params& P, int steps) {
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
                    nearest<KIND, NOBJ, SIG>(P, o, idx, dist);
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
#ifdef RT_DEBUG_PHASE
    {   // the launch's critical path: the pixel with the most (sequential) march steps; and the per-wave maximum, summed
        atomicMax(&P.counters->dbg[0], (unsigned long long)n_steps);
        uint32_t m = n_steps;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)m, off, 64); m = o > m ? o : m; }
        if ((threadIdx.x & 63) == 0) atomicAdd(&P.counters->dbg[1], (unsigned long long)m);
    }
#endif
    flush_counters(P, n_steps, n_raycasts, n_hits, n_sky, n_samples, n_dep);
}

// -------------------------------------------------------------------------------------------
// src/ persistent-ray form on the LDS ray pool ("persistent-lane
// scheduler").  Same arithmetic as persistent_steps, but a CONTEXT (one pixel advancing through
// bounce-steps) is decoupled from a lane: contexts whose raycast finished are parked in the
// wave's LDS slots for shading while the lane takes over a parked context that is ready to
// march, exactly like trace_paths_pool.
//
// Ownership is STATIC and STRIDED (round 3; the round-2 kernel claimed chunks of pixels from a global counter and
// walked each pixel through all `steps` of the launch): wave g of the NW resident waves owns the pixels q = g + k NW,
// k < n_own.  A context lives as long as `steps` bounce-steps (~25 ms at 256 steps), so with dynamic claiming the
// launch ended with every wave draining 128 contexts of uniformly staggered progress — waves were resident for only
// 45 % (768x432) / 61 % (1080p) of the kernel (profiles/r03a_src_*).  Now
//   * strided ownership gives every wave a statistically identical sample of the frame (sky and object pixels alike):
//     balanced without a shared queue;
//   * a wave that owns no more pixels than it has contexts (64 lanes + 64 slots) keeps them all resident for the whole
//     launch: they advance together and finish within the spread of a 256-term sum;
//   * a wave that owns more walks them PASS-MAJOR in residencies of S bounce-steps (item i = pass * n_own + k): the
//     state goes back to ray_buffer after S steps and the slot takes the wave's next item, so all pixels advance
//     together and the launch drains for the length of ONE residency, not of a whole context.  Item i needs item
//     i - n_own (the same pixel's previous residency) finished: items are handed out in order and only below
//     low_water + n_own, low_water = the smallest item still in flight in this wave (recomputed when the hand-out
//     reaches the bound).  Same wave, same CU: the write-back is visible to the later read without any fence.
// A pixel is advanced by ONE context at a time and its steps run in order, so its deposits into image_buffer happen
// in step order (bit-exact with the sequential form) under every ownership / residency choice.
enum { G_OX = 0, G_OY, G_OZ, G_DX, G_DY, G_DZ, G_CR, G_CG, G_CB, G_DEPTH, G_IDX, G_K, G_S, G_KEY, G_CNT, G_COUNT };
static_assert(G_COUNT == POOL_WORDS, "pixel-context record must fill the pool record");

struct PixCtx {
    vec3 o, d, col;
    int depth, idx;
    uint32_t k;        // the owner wave's k-th pixel: q = wave + k * n_waves
    int s;             // bounce-step of this launch the context is at
    uint32_t key, cnt;
};

// wave-uniform constants of the ownership / residency scheme
struct SrcWave {
    uint32_t g, nw;        // this wave, resident waves
    uint32_t n_own;        // pixels owned
    uint32_t n_items;      // n_own * passes
    uint32_t s_mask;       // residency length - 1 (a power of two), ~0u when the wave keeps its pixels for the whole launch
    int lg_s;              // log2(residency length); 31 when single-pass (s >> 31 == 0)
};

// one iteration of raycast() src/scene.py:59-84 (the ray origin itself moves)
template <int KIND, int NOBJ = 0, uint32_t SIG = 0>
RT_D void march_step_src(const Params& P, Lane& L) {
    float ld = L.dist;
    int idx;
    float dist;
    nearest<KIND, NOBJ, SIG>(P, L.o, idx, dist);
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

// The same iteration with the object loop culled at wave level (nearest_culled: exact Lipschitz bounds, lb / ub kept by
// the caller).  Pays when FEW lanes march — a launch ends with every wave marching the handful of pixels whose raycasts
// graze the ground for hundreds of steps, and those are near ONE object: the other six are skipped for the whole wave.
template <int KIND, int NOBJ, uint32_t SIG>
RT_D void march_step_src_culled(const Params& P, Lane& L, float& ub, float (&lb)[NOBJ > 0 ? NOBJ : 1], uint32_t* dbg_evaluated = nullptr) {
    const bool active = L.state == ST_MARCH;
    float ld = L.dist;
    int idx;
    float dist;
    nearest_culled<KIND, NOBJ, SIG>(P, L.o, L.t, active, ub, lb, idx, dist, dbg_evaluated);
    float moved = 0.0f;
    if (active) {
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
        // the next evaluation point is |s_new| * |d| away; |d| <= 1 + 2^-20
        moved = fabs_(s_new) * 1.000001f;
        ub = dist + moved;
    }
#pragma unroll
    for (int i = 0; i < (NOBJ > 0 ? NOBJ : 1); i++) lb[i] -= moved;
}

// Advance a context through the part of its step sequence that needs no marching: roulette,
// deposit + camera-ray regeneration (src/pathtracer.py:53-77).  Returns true when the context
// is ready to march its next raycast, false when its residency (or the launch) is over and the
// state has been written back.  `fresh`: the context was just loaded (X.s is the first step of
// its residency); otherwise the caller has just completed step X.s - 1.
template <int KIND>
RT_D bool pix_advance(const Params& P, const SrcWave& Wv, PixCtx& X, int steps, bool fresh, uint32_t& n_samples, uint32_t& n_dep) {
    const rtpbr_config& g = P.cfg;
    int px, py;
    pixel_of(P, Wv.g + X.k * Wv.nw, px, py);
    const size_t pi = (size_t)px * g.height + py;
    for (;;) {
        if (!fresh && (((uint32_t)X.s & Wv.s_mask) == 0u || X.s >= steps)) break;
        fresh = false;
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

RT_D uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

template <int KIND, int NOBJ = 0, uint32_t SIG = 0>
RT_D void persistent_pool_impl(const Params& P, int steps) {
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

    // ---- what this wave owns and how it walks it (all wave-uniform)
    SrcWave Wv;
    Wv.g = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (uint32_t)wave));
    Wv.nw = gridDim.x * 4u;
    const uint32_t np = (uint32_t)P.np;
    Wv.n_own = np > Wv.g ? (np - Wv.g - 1u) / Wv.nw + 1u : 0u;
    const uint32_t S = P.chunk;                                   // residency length when the wave owns more than it can hold
    const bool multi = Wv.n_own > 128u && S < (uint32_t)steps;
    Wv.s_mask = multi ? S - 1u : 0xffffffffu;
    Wv.lg_s = multi ? 31 - __builtin_clz(S) : 31;
    const uint32_t n_pass = multi ? ((uint32_t)steps + S - 1u) >> Wv.lg_s : 1u;
    Wv.n_items = Wv.n_own * n_pass;
    uint32_t next_item = 0;                    // items are handed out in order ...
    uint32_t safe_until = Wv.n_own;            // ... and only below this bound: low_water + n_own (pass 0 needs nothing)

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
    uint32_t a_k = 0, a_key = 0, a_cnt = 0;
    uint32_t n_samples = 0, n_dep = 0;
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
#ifdef RT_DEBUG_PHASE
    unsigned long long tB = 0, tD = 0, tA = 0, tc = __builtin_readcyclecounter(), t_start = tc;
    unsigned long long dbg_passes = 0, dbg_shaded = 0, dbg_march_iters = 0, dbg_march_lanes = 0, dbg_sparse_iters = 0;
    uint32_t dbg_evaluated = 0;     // objects evaluated by the culled steps (wave level)
#endif

    for (;;) {
        // ================================================================ phase B on the slots
        {
            const int n_shade = __popcll(m_shade);
            const int n_ready = __popcll(m_ready);
            const int n_free = 64 - n_shade - n_ready;
            // the hand-out has (nearly) reached its bound: find the oldest item still in flight in this wave (done here,
            // not inside the pass, so that a wave whose hand-out is blocked by a straggler learns when it has finished)
            if (multi && safe_until < Wv.n_items && next_item + 64u > safe_until) {
                uint32_t mine = 0xffffffffu;
                if (L.state != ST_IDLE) mine = ((uint32_t)a_s >> Wv.lg_s) * Wv.n_own + a_k;
                if (sstate[lane] != SL_EMPTY) {
                    const uint32_t it = (pool[G_S][lane] >> Wv.lg_s) * Wv.n_own + pool[G_K][lane];
                    mine = it < mine ? it : mine;
                }
                uint32_t low = wave_min_u32(mine);
                low = low == 0xffffffffu ? next_item : low;
                safe_until = (uint32_t)__builtin_amdgcn_readfirstlane((int)(low + Wv.n_own));
            }
            const uint32_t limit = Wv.n_items < safe_until ? Wv.n_items : safe_until;
            const bool run_b = n_shade >= T || (n_ready == 0 && (n_shade > 0 || (n_free > 0 && next_item < limit)));
            if (run_b) {
                uint32_t st = sstate[lane];
#ifdef RT_DEBUG_PHASE
                dbg_passes++;
                dbg_shaded += (unsigned)n_shade;
#endif
                PixCtx X;
                X.o = X.d = X.col = mk(0, 0, 0);
                X.depth = X.idx = X.s = 0;
                X.k = X.key = X.cnt = 0;
                bool have = false, fresh = false;
                if (st == SL_HIT || st == SL_MISS) {
                    X.o = mk(u2f(pool[G_OX][lane]), u2f(pool[G_OY][lane]), u2f(pool[G_OZ][lane]));
                    X.d = mk(u2f(pool[G_DX][lane]), u2f(pool[G_DY][lane]), u2f(pool[G_DZ][lane]));
                    X.col = mk(u2f(pool[G_CR][lane]), u2f(pool[G_CG][lane]), u2f(pool[G_CB][lane]));
                    X.depth = (int)pool[G_DEPTH][lane];
                    X.idx = (int)pool[G_IDX][lane];
                    X.k = pool[G_K][lane];
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
                // free slots take the wave's next items, in order: the j-th free slot gets item next_item + j
                {
                    const bool want = st == SL_EMPTY && !have;
                    const unsigned long long wm = __ballot(want);
                    const uint32_t need = (uint32_t)__popcll(wm);
                    const uint32_t avail = limit - next_item;          // next_item <= limit always
                    const uint32_t take = need < avail ? need : avail;
                    const uint32_t rank = (uint32_t)wave_rank(wm);
                    if (want && rank < take) {
                        const uint32_t item = next_item + rank;
                        const uint32_t pass = multi ? item / Wv.n_own : 0u;
                        const uint32_t k = item - pass * Wv.n_own;
                        int px, py;
                        if (pixel_of(P, Wv.g + k * Wv.nw, px, py)) {
                            const size_t pi = (size_t)px * P.cfg.height + py;
                            bool masked = P.cfg.adaptive_sampling && !(P.diff_pixels[pi] > P.cfg.noise_threshold);
                            if (!masked) {
                                rtpbr_ray rb = P.ray_buffer[pi];
                                X.o = mk(rb.origin[0], rb.origin[1], rb.origin[2]);
                                X.d = mk(rb.direction[0], rb.direction[1], rb.direction[2]);
                                X.col = mk(rb.color[0], rb.color[1], rb.color[2]);
                                X.depth = rb.depth;
                                X.k = k;
                                X.s = (int)(pass << Wv.lg_s);
                                have = true;
                                fresh = true;
                            }
                        }
                    }
                    next_item += take;
                }
                bool ready = false;
                if (have) ready = pix_advance<KIND>(P, Wv, X, steps, fresh, n_samples, n_dep);
                if (ready) {
                    pool[G_OX][lane] = f2u(X.o.x); pool[G_OY][lane] = f2u(X.o.y); pool[G_OZ][lane] = f2u(X.o.z);
                    pool[G_DX][lane] = f2u(X.d.x); pool[G_DY][lane] = f2u(X.d.y); pool[G_DZ][lane] = f2u(X.d.z);
                    pool[G_CR][lane] = f2u(X.col.x); pool[G_CG][lane] = f2u(X.col.y); pool[G_CB][lane] = f2u(X.col.z);
                    pool[G_DEPTH][lane] = (uint32_t)X.depth;
                    pool[G_K][lane] = X.k;
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

        RT_PHASE(tB)
        // ================================================================ dispatch (pool_swap, as in trace_paths_pool)
        {
            const bool is_done = L.state == ST_HIT || L.state == ST_MISS;
            uint32_t rec[POOL_WORDS];
            rec[G_OX] = f2u(L.o.x); rec[G_OY] = f2u(L.o.y); rec[G_OZ] = f2u(L.o.z);
            rec[G_DX] = f2u(L.d.x); rec[G_DY] = f2u(L.d.y); rec[G_DZ] = f2u(L.d.z);
            rec[G_CR] = f2u(a_col.x); rec[G_CG] = f2u(a_col.y); rec[G_CB] = f2u(a_col.z);
            rec[G_DEPTH] = (uint32_t)a_depth;
            rec[G_IDX] = (uint32_t)L.idx;
            rec[G_K] = a_k;
            rec[G_S] = (uint32_t)a_s;
            rec[G_KEY] = a_key; rec[G_CNT] = a_cnt;
            const int r = pool_swap(V, lane, is_done, L.state == ST_IDLE, L.state == ST_HIT ? SL_HIT : SL_MISS, rec, m_ready, m_shade);
            if (r & 2) L.state = ST_IDLE;
            if (r & 1) {
                L.o = mk(u2f(rec[G_OX]), u2f(rec[G_OY]), u2f(rec[G_OZ]));
                L.d = mk(u2f(rec[G_DX]), u2f(rec[G_DY]), u2f(rec[G_DZ]));
                a_col = mk(u2f(rec[G_CR]), u2f(rec[G_CG]), u2f(rec[G_CB]));
                a_depth = (int)rec[G_DEPTH];
                a_k = rec[G_K];
                a_s = (int)rec[G_S];
                a_key = rec[G_KEY]; a_cnt = rec[G_CNT];
                src_march_init(L);
            }
        }

        RT_PHASE(tB)
        // ================================================================ march
        {
            int n_march = __popcll(__ballot(L.state == ST_MARCH));
            if (n_march == 0) {
                const bool any_ray = __ballot(L.state != ST_IDLE) != 0;
                if (!any_ray && m_ready == 0 && m_shade == 0 && next_item >= Wv.n_items) break;
                continue;
            }
            const int n_ready = __popcll(m_ready);
            int n_done;
            bool sparse = false;
            if constexpr (NOBJ > 0 && (KIND == KIND_BOXES || KIND == KIND_GENERIC)) sparse = P.cull_ok && n_march <= P.sparse_lanes;
            if (sparse) {
                if constexpr (NOBJ > 0 && (KIND == KIND_BOXES || KIND == KIND_GENERIC)) {
                    // few lanes march: cull the object loop for the wave.  The bounds start from "nothing known" (the first
                    // step evaluates every object, as the plain step does) and live only for this march phase.
                    float lb[NOBJ];
#pragma unroll
                    for (int i = 0; i < NOBJ; i++) lb[i] = -1.0f;
                    float ub = 3.0e38f;
#ifdef RT_DEBUG_PHASE
                    const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
                    do {
#ifdef RT_DEBUG_PHASE
                        dbg_march_iters++;
                        dbg_march_lanes += (unsigned)n_march;
                        dbg_sparse_iters++;
#endif
#ifdef RT_DEBUG_PHASE
                        march_step_src_culled<KIND, NOBJ, SIG>(P, L, ub, lb, &dbg_evaluated);
#else
                        march_step_src_culled<KIND, NOBJ, SIG>(P, L, ub, lb);
#endif
                        n_march = __popcll(__ballot(L.state == ST_MARCH));
                        n_done = __popcll(__ballot(L.state == ST_HIT || L.state == ST_MISS));
                    } while (n_march > 0 && n_done < (n_ready > 0 ? m_swap : 2 * m_swap));
#ifdef RT_DEBUG_PHASE
                    tD += __builtin_readcyclecounter() - ts0;      // (cycles of the sparse march loops, reported in place of the dispatch phase)
#endif
                }
            } else {
                do {
#ifdef RT_DEBUG_PHASE
                    dbg_march_iters++;
                    dbg_march_lanes += (unsigned)n_march;
#endif
                    if (L.state == ST_MARCH) march_step_src<KIND, NOBJ, SIG>(P, L);
                    n_march = __popcll(__ballot(L.state == ST_MARCH));
                    n_done = __popcll(__ballot(L.state == ST_HIT || L.state == ST_MISS));
                } while (n_march > 0 && n_done < (n_ready > 0 ? m_swap : 2 * m_swap));
            }
        }
        RT_PHASE(tA)
    }
#ifdef RT_DEBUG_PHASE
    if (lane == 0) {   // cycles per phase and wave lifetime (>> 10), passes, slots shaded, march iterations, lanes marching
        atomicAdd(&P.counters->dbg[0], tB >> 10);
        atomicAdd(&P.counters->dbg[1], dbg_sparse_iters | ((tD >> 10) << 32));      // march iterations that ran the culled step; their cycles >> 10 in the high word
        atomicAdd(&P.counters->dbg[2], tA >> 10);
        atomicAdd(&P.counters->dbg[3], (__builtin_readcyclecounter() - t_start) >> 10);
        atomicAdd(&P.counters->dbg[4], dbg_passes);
        atomicAdd(&P.counters->dbg[5], dbg_shaded | ((unsigned long long)dbg_evaluated << 40));
        atomicAdd(&P.counters->dbg[6], dbg_march_iters);
        atomicAdd(&P.counters->dbg[7], dbg_march_lanes);
    }
#endif
    flush_counters(P, L.n_steps, L.n_raycasts, L.n_hits, L.n_sky, n_samples, n_dep);
}

template <int KIND>
__global__ void __launch_bounds__(256) persistent_steps(const Params P, int steps) { persistent_steps_impl<KIND>(P, steps); }
template <int KIND>
__global__ void __launch_bounds__(256) persistent_pool(const Params P, int steps) { persistent_pool_impl<KIND>(P, steps); }

}  // namespace rt
