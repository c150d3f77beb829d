// rt_persistent.hpp — the src/ persistent-ray form (src/pathtracer.py:16-103, src/scene.py:59-84): the lock-step kernel
// (one lane per pixel) and the pool kernel (contexts decoupled from lanes: cost-ordered ownership with heavy waves and
// age-weighted shares, pass-major residencies, exact tracked-object march, ski-rental exits).  Included at the end of rt_trace.hpp, whose LDS ray pool (PoolView, pool_swap),
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
    zero_next_counters(P);
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
// Ownership is STATIC (round 3; the round-2 kernel claimed chunks of pixels from a global counter and walked each pixel
// through all `steps` of the launch): every resident wave owns a fixed set of pixels, n_own of them — strided over the
// frame in round 3 (q = g + k NW), over the cost-ordered list since round 4 (struct SrcWave below).  A context lives as long as `steps` bounce-steps (~25 ms at 256 steps), so with dynamic claiming the
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
enum { G_OX = 0, G_OY, G_OZ, G_DX, G_DY, G_DZ, G_CR, G_CG, G_CB, G_DEPTH, G_META, G_K, G_Q, G_KEY, G_CNT, G_COUNT };
static_assert(G_COUNT == POOL_WORDS, "pixel-context record must fill the pool record");
// G_META: nearest-object index (5 bits) | bounce-step of this launch the context is at (9 bits: <= 256 steps per kernel) |
// march steps the context has taken since it was loaded (18 bits, wraps: it only steers the NEXT launch's schedule)
RT_D uint32_t gmeta_pack(int idx, int s, uint32_t cost) { return (uint32_t)idx | ((uint32_t)s << 5) | (cost << 14); }
RT_D int gmeta_idx(uint32_t m) { return (int)(m & 31u); }
RT_D int gmeta_s(uint32_t m) { return (int)((m >> 5) & 511u); }
RT_D uint32_t gmeta_cost(uint32_t m) { return m >> 14; }

struct PixCtx {
    vec3 o, d, col;
    int depth, idx;
    uint32_t k;        // the owner wave's k-th pixel
    uint32_t q;        // local pixel (pixel_of)
    int s;             // bounce-step of this launch the context is at
    uint32_t cost;     // march steps since the context was loaded
    uint32_t key, cnt;
};

// Wave-uniform constants of the ownership / residency scheme.  A wave owns the pixels order[base + k * stride], k < n_own
// (order = identity when there is no plan yet).  Cost-ordered ownership (round 4): `order` lists the local pixels by the
// march steps they took over the last launches, heaviest first (plan kernels, rt_kernels.hip).  The first n_heavy of
// them — pixels whose own dependency chain is comparable with a whole wave's share of the frame: the horizon of the
// ground sphere, camera rays of 200-512 steps each, 256 times per launch — are dealt in CONTIGUOUS runs of heavy_own to
// the first waves ("heavy waves": rays of similar length march together, keep all their pixels resident and run the
// tracked-object march below); the rest is dealt round-robin to the other waves, so every light wave gets the same cost
// profile and, because the order is kept, starts its own heaviest pixels first in every pass.
struct SrcWave {
    uint32_t skip;         // entries at the head of `order` that are not the pool kernel's (the chain set)
    uint32_t base, stride; // this wave's pixels in `order` (after `skip`): base + (k / run) * stride + k % run
    uint32_t run_inv;      // consecutive entries per round (low 6 bits; 1 = plain strided dealing) | ceil(2^20 / run) << 6
    uint32_t n_own;        // pixels owned
    uint32_t n_items;      // n_own * passes
    uint32_t s_mask;       // residency length - 1 (a power of two), ~0u when the wave keeps its pixels for the whole launch
    int lg_s;              // log2(residency length); 31 when single-pass (s >> 31 == 0)
};
RT_D uint32_t chain_skip(const Params& P) {      // entries of `order` that belong to the chain kernel (wave-uniform)
    return (P.chain_on && P.order && P.plan) ? P.plan->n_chain : 0u;
}
RT_D uint32_t own_pixel(const Params& P, const SrcWave& Wv, uint32_t k) {
    // round r = k / run (k < 2^14: (k * ceil(2^20 / run)) >> 20, run <= 63; run = 1: r = k)
    const uint32_t run = Wv.run_inv & 63u, inv = Wv.run_inv >> 6;
    const uint32_t r = run > 1u ? (uint32_t)(((unsigned long long)k * inv) >> 20) : k;      // (64-bit: k * inv reaches 2^34)
    const uint32_t i = Wv.skip + Wv.base + r * Wv.stride + (k - r * run);
    return P.order ? P.order[i] : i;
}

// one iteration of raycast() src/scene.py:59-84 after nearest() (the ray origin itself moves); returns the step length
RT_D float march_update_src(const Params& P, Lane& L, int idx, float dist) {
    // (bitwise on purpose, as in march_update: `&&` / `||` would be lowered to divergent branches with exec-mask
    // juggling — two dozen scalar instructions in a loop whose length in instructions is what matters)
    const float ld = L.dist;
    L.idx = idx;
    L.dist = dist;
    L.n_steps++;
    L.steps_left--;
    const bool fb = (L.w > 1.0f) & (ld + dist < L.s);
    const float s_fb = L.s - L.w * L.s;
    const float s_nm = L.w * dist;
    const float s_new = fb ? s_fb : s_nm;
    L.w = fb ? 1.0f : L.w;
    L.s = s_new;
    L.t += s_new;
    L.o = fma3(s_new, L.d, L.o);
    const bool nfb = !fb;
    const bool hit = nfb & (dist < L.t * P.cfg.hit_eps);
    const bool done = (nfb & (hit | (L.t >= P.cfg.max_dis))) | (L.steps_left == 0);
    L.state = done ? (hit ? ST_HIT : ST_MISS) : L.state;
    return s_new;
}
// The same iteration inside a lean loop: the lane's L.idx already is the object (IDX = false: no write), and when the
// over-relaxation of the raycast is over — W1: every marching lane has w == 1, the caller checked — the fallback test cannot
// fire and s = 1.0f * dist IS dist: nine instructions of bookkeeping less per step, same values (src/scene.py:67-78).
template <bool W1, bool IDX>
RT_D float march_update_src_lean(const Params& P, Lane& L, int idx, float dist) {
    if constexpr (!W1) {
        const int keep = L.idx;
        const float s_new = march_update_src(P, L, idx, dist);
        if constexpr (!IDX) L.idx = keep;
        return s_new;
    } else {
        if constexpr (IDX) L.idx = idx;
        L.dist = dist;
        L.n_steps++;
        L.steps_left--;
        // (L.s is only read by the fallback test, which needs w > 1: dead for the rest of this raycast)
        L.t += dist;
        L.o = fma3(dist, L.d, L.o);
        const bool hit = dist < L.t * P.cfg.hit_eps;
        const bool done = (hit | (L.t >= P.cfg.max_dis)) | (L.steps_left == 0);
        L.state = done ? (hit ? ST_HIT : ST_MISS) : L.state;
        return dist;
    }
}
template <int KIND, int NOBJ = 0, uint32_t SIG = 0>
RT_D void march_step_src(const Params& P, Lane& L) {
    int idx;
    float dist;
    nearest<KIND, NOBJ, SIG>(P, L.o, idx, dist);
    march_update_src(P, L, idx, dist);
}

// ---- tracked-object march (round 4).  Sphere tracing is a dependency chain, and the launch of the fused src/ kernel is
// as long as the longest chain of ONE pixel (73 206 sequential steps at 1080p, ~230 dependent instructions
// each).  Those chains belong to rays that graze ONE object for hundreds of steps, so for them the other objects can be
// skipped — exactly.  |sdf_j| is 1-Lipschitz: after a FULL evaluation at p0 (distances d_j, nearest k, second smallest
// d2) every other object satisfies, at a later position p,
//     computed |sdf_j|(p) >= |sdf_j|(p) - eps >= |sdf_j|(p0) - |p - p0| - eps >= d2 - 2 eps - (path marched since),
// eps = the rounding of one computed distance.  The lane keeps lb = d2 - eps - (marched) - (rounding of the positions and
// of lb itself); while lb > |sdf_k|(p) + eps, object k is STRICTLY the nearest and (k, |sdf_k|(p)) is bit for bit what
// nearest() returns.  When the test fails the lane waits (no step is taken, nothing is counted) for the wave's next
// full evaluation.  eps = 2^-19 (|p|_1 + cull_extent): 16 ulps of everything that enters a distance (positions reach
// MAX_DIS after an escape, so |p| is taken from the ray, not from the scene).
// One VGPR of state per lane (lb; <= 0 = "needs a full evaluation"); the tracked object is L.idx.
RT_D float track_eps(const Params& P, vec3 o) {
    return 1.9073486328125e-06f * (((fabs_(o.x) + fabs_(o.y)) + fabs_(o.z)) + P.cull_extent);
}
// the allowance of a whole lean loop entered at o with the bound lb (see march_fast_src_obj): every position of the loop lies
// within lb of o, |p|_1 within 2 lb of |o|_1
RT_D float track_eps_loop(const Params& P, vec3 o, float lb) {
    // (a scene of one or two objects has second / third = 3e38: every position a distance is EVALUATED at has t < MAX_DIS — the
    // raycast ends there — so the path marched in the loop is below min(lb, MAX_DIS); without the clamp 2 lb overflowed, the
    // allowance was +inf and no lean loop ever took a step in such scenes: exact, but slow)
    return 1.9073486328125e-06f * ((((fabs_(o.x) + fabs_(o.y)) + fabs_(o.z)) + P.cull_extent) + 2.0f * fmin_(fmax_(lb, 0.0f), P.cfg.max_dis));
}
// after a step of length s: the next position is |s| |d| away (|d| <= 1 + 2^-20); a quarter of eps covers the rounding
// of the three coordinates (<= ulp(|p|) = eps / 16) and of the two roundings in this update (<= eps / 32 each)
RT_D float track_decay(float lb, float s_new, float eps) { return fma_(fabs_(s_new), -1.000001f, lb) - 0.25f * eps; }

// What a lane knows from its last full evaluation (round 5: TWO bounds).  lb2 bounds every object but the nearest one
// (L.idx) from below, lb3 every object but the two nearest (L.idx and k2); <= 0 = not valid.  A ray that grazes ONE surface
// keeps lb2 valid for hundreds of steps; a ray in the WEDGE between two surfaces (a sphere resting on the ground) has
// second ~ nearest, so lb2 fails at once — measured on the launch-critical raycasts: every lean attempt failed on its first
// step and each step cost a full evaluation plus a wasted attempt — while lb3 holds: the two-object loop below.
// RT_TRK_W1 = 0 drops the second instance (no relaxation bookkeeping) of every lean loop of the kernels that keep both bounds.
// (Round 6 measured whether those kernels' CODE SIZE — 47 KB for the split march, 53 KB for the chain kernel, most of it the 21 x 2
// pair instances — costs instruction-cache misses: one generic pair / one-object loop instance with its constants from the LDS
// table took the march kernel to 23 / 18 KB and one-step launches from 0.239 to 0.247 / 0.250 ms at 768x432, 0.467 to 0.484 /
// 0.488 at 1080p: the code is not the problem, the few more instructions per step are.  Removed again.)
#ifndef RT_TRK_W1
#define RT_TRK_W1 1
#endif
#ifndef RT_POOL_OP
#define RT_POOL_OP 0       // 1: the fused pool kernel's sparse phases use the object-parallel evaluation (nearest_op3 below) too
#endif
#ifndef RT_POOL_TWO
#define RT_POOL_TWO 0      // 1: the fused pool kernel keeps both bounds too (experiment / small-frame instance)
#endif
struct Trk {
    float lb2, lb3;
    int k2;
};
template <int KIND, int NOBJ, uint32_t SIG>
RT_D void march_step_src_full2(const Params& P, Lane& L, Trk& T) {      // one bound only (src_track = 1)
    const float eps = track_eps(P, L.o);
    int idx;
    float dist, second;
    nearest_exact2<KIND, NOBJ, SIG>(P, L.o, idx, dist, second);
    const float s_new = march_update_src(P, L, idx, dist);
    T.lb2 = track_decay(second - eps, s_new, eps);
    T.lb3 = -1.0f;
    T.k2 = idx;
}
template <int KIND, int NOBJ, uint32_t SIG>
RT_D void march_step_src_full3(const Params& P, Lane& L, Trk& T) {
    const float eps = track_eps(P, L.o);
    int idx, idx2;
    float dist, second, third;
    nearest_exact3<KIND, NOBJ, SIG>(P, L.o, idx, dist, idx2, second, third);
    const float s_new = march_update_src(P, L, idx, dist);
    T.lb2 = track_decay(second - eps, s_new, eps);
    T.lb3 = idx2 >= 0 ? track_decay(third - eps, s_new, eps) : -1.0f;
    T.k2 = idx2 >= 0 ? idx2 : idx;
}
// ---- OBJECT-PARALLEL nearest() for sparse waves (round 6).  A wave whose list has run out, a chain wave of a few pixels, a
// pool wave in its drain: a handful of UNRELATED rays march, each tracking its own object, and every iteration is a full
// evaluation (~230-330 instructions: the unrolled object loop with three-smallest tracking) executed for six lanes of 64, at a
// lone wave's dependent-chain pace.  With at most 8 rays marching and at most 8 objects the wave evaluates them the other way
// round: lane (r, j) = 8 r + j evaluates object j — ONE object, from the LDS table (the general-matrix form of to_local, the
// run-time shape switch: the arithmetic of the generic ahead-of-time instance, which the run-time instances' sparse rotations
// and literals equal bit for bit, rt_device.hpp to_local) — for the r-th marching ray, whose position it gets through an
// 8-entry exchange buffer in LDS; a three-step DPP butterfly inside each group of eight lanes (quad_perm, quad_perm,
// row_half_mirror: no LDS, no permute unit) yields the smallest distance, the LOWEST index that attains it (nearest() visits
// the objects in index order with a strict `<`: src/scene.py:48-54), and — the same again with the winner masked out — the
// second and third smallest with the second's index: everything nearest_exact3 returns.  The group's first lane writes the four
// words back to the exchange buffer, the ray's home lane reads them.  ~200 instructions for the seven-object scene (the shape
// switch serialises the types present: sphere + box + cylinder ~ 80; five reductions of three DPP steps), independent of the
// number of marching rays; measured in the tails of one-step launches (instrumented build): 2.04 kcycles per call against 2.67 for
// the unrolled three-smallest evaluation and 1.73 for the tracked rounds it replaces (two LDS round trips and the dependent DPP
// chain are a third of it).  Bit-identical by construction: the same |sdf_j|(p) per object, the same winner, valid bounds.
struct OpView {
    const ObjFull* lds_obj;     // the block's object table (stage_objects)
    float4* xch;                // 8 entries of this wave: ray positions in, results out
};
#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL>
RT_D uint32_t dpp_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
#else
template <int CTRL>
RT_D uint32_t dpp_u32(uint32_t v) { return v; }
#endif
// minimum over the 8 lanes of a group (all lanes of the wave active): xor 1, xor 2 inside the quad, then the other quad of the
// half row.  On UNSIGNED words: a distance is |sdf| >= +0, and non-negative floats order like their bit patterns (a NaN above
// every number: it never wins, as in the serial search's `d < best`) — v_min_u32 takes the DPP operand directly, v_min_f32 would
// be preceded by two canonicalising v_max_f32 per step.
RT_D uint32_t seg8_min(uint32_t v) {
    uint32_t o = dpp_u32<0xB1>(v);        // quad_perm [1,0,3,2]
    v = o < v ? o : v;
    o = dpp_u32<0x4E>(v);                 // quad_perm [2,3,0,1]
    v = o < v ? o : v;
    o = dpp_u32<0x141>(v);                // row_half_mirror: lane i <-> 7 - i of the eight
    v = o < v ? o : v;
    return v;
}
// Called by ALL lanes of the wave from wave-uniform control flow; at most 8 lanes `marching`, P.n_obj <= 8.  Marching lanes get
// what nearest() returns for their position p (idx, best) and what nearest_exact3 adds: the second and third smallest distance
// (3e38: there is none) and an object that attains the second (-1: none); `none` = nearest_init and no object within MAX_DIS
// (nearest() then returns (0, MAX_DIS); the bounds are not to be used).
template <int KIND>
RT_D void nearest_op3(const Params& P, const OpView& V, bool marching, unsigned long long mm, vec3 p, int& idx, float& best, int& idx2,
                      float& second, float& third, bool& none) {
    const int lane = (int)(threadIdx.x & 63u);
    const int seg = lane >> 3, j = lane & 7;
    const int r = wave_rank(mm);
    if (marching) V.xch[r] = make_float4(p.x, p.y, p.z, 0.0f);
    lds_wave_fence();
    // (groups beyond the marching rays compute on whatever the buffer holds: nobody fetches their result)
    const float4 q = V.xch[seg];
    const bool real = j < P.n_obj;
    const ObjM ob = *reinterpret_cast<const ObjM*>(V.lds_obj + (real ? j : 0));
    constexpr uint32_t BIG = 0x7f61b1e6u;      // 3.0e38f
    const uint32_t d = real ? __builtin_bit_cast(uint32_t, fabs_(signed_distance<KIND>(P, ob, mk(q.x, q.y, q.z)))) : BIG;
    const uint32_t m1 = seg8_min(d);
    const uint32_t i1 = seg8_min(d == m1 ? (uint32_t)j : 8u) & 7u;      // (8 = every distance is a NaN: object 0, as the serial search)
    const uint32_t d2 = (uint32_t)j == i1 ? BIG : d;
    const uint32_t m2 = seg8_min(d2);
    const uint32_t i2 = seg8_min(d2 == m2 ? (uint32_t)j : 8u) & 7u;
    const uint32_t d3 = (((uint32_t)j == i2) & (m2 < BIG)) ? BIG : d2;
    const uint32_t m3 = seg8_min(d3);
    lds_wave_fence();      // every group has read its position: the buffer now takes the results
    if (j == 0) V.xch[seg] = make_float4(__builtin_bit_cast(float, m1), __builtin_bit_cast(float, m2), __builtin_bit_cast(float, m3), __builtin_bit_cast(float, i1 | (i2 << 8)));
    lds_wave_fence();
    if (marching) {
        const float4 res = V.xch[r];
        const uint32_t ii = __builtin_bit_cast(uint32_t, res.w);
        none = P.cfg.nearest_init && !(res.x < P.cfg.max_dis);
        best = none ? P.cfg.max_dis : res.x;
        idx = none ? 0 : (int)(ii & 255u);
        second = res.y;
        third = res.z;
        idx2 = res.y < 3.0e38f ? (int)(ii >> 8) : -1;
    }
}
// one step of raycast() for the wave's (<= 8) marching lanes on the object-parallel evaluation, both bounds fresh (what
// march_step_src_full3 does on the unrolled one); all lanes call it
template <int KIND>
RT_D void march_step_src_op(const Params& P, const OpView& V, Lane& L, Trk& T) {
    const bool marching = L.state == ST_MARCH;
    const unsigned long long mm = __ballot(marching);
    const float eps = track_eps(P, L.o);
    int idx = 0, idx2 = -1;
    float dist = 0.0f, second = 3.0e38f, third = 3.0e38f;
    bool none = false;
    nearest_op3<KIND>(P, V, marching, mm, L.o, idx, dist, idx2, second, third, none);
    if (marching) {
        const float s_new = march_update_src(P, L, idx, dist);
        T.lb2 = none ? -1.0f : track_decay(second - eps, s_new, eps);
        T.lb3 = ((idx2 >= 0) & !none) ? track_decay(third - eps, s_new, eps) : -1.0f;
        T.k2 = idx2 >= 0 ? idx2 : idx;
    }
}

// `can`: marching lanes whose lb is valid.  Lanes track different objects: one round per distinct object (grazing rays
// share theirs), each a wave-uniform jump into that object's unrolled code.
template <int KIND, int NOBJ, uint32_t SIG, bool TWO = false>
RT_D void march_step_src_tracked(const Params& P, Lane& L, Trk& T, bool can, unsigned long long* dbg_rounds = nullptr) {
    float& lb = T.lb2;
    const float eps = track_eps(P, L.o);
    float dk = 0.0f;
    unsigned long long todo = __ballot(can);
    while (todo) {
        // (evaluated by every lane, kept by the lanes that track this object: were the call inside `if (L.idx == kw)` the
        // compiler would substitute the per-lane L.idx for the wave-uniform kw and turn the scalar jump into a
        // divergent evaluation of ALL objects)
#ifdef RT_DEBUG_PHASE
        if (dbg_rounds) (*dbg_rounds)++;
#endif
        const int kw = __builtin_amdgcn_readlane(L.idx, (int)__builtin_ctzll(todo));
        const float v = sdf_object<KIND, NOBJ, SIG>(P, kw, L.o);
        const bool mine = can && L.idx == kw;
        dk = mine ? v : dk;
        todo &= ~__ballot(mine);
    }
    const bool ok = can & (lb > dk + eps) & (!P.cfg.nearest_init | (dk < P.cfg.max_dis));
    if (ok) {
        const float s_new = march_update_src(P, L, L.idx, dk);
        lb = track_decay(lb, s_new, eps);
        if constexpr (TWO) T.lb3 = track_decay(T.lb3, s_new, eps);
    } else if (can) {
        lb = -1.0f;
    }
}

// The lean loop of the tracked march: EVERY marching lane tracks the same object I and holds a valid bound.  One
// object evaluation, the bound test, the raycast bookkeeping and ONE vote per step — a third of the instructions of
// the general tracked iteration (no per-object rounds, no policy).  A lone wave issues an instruction every ~6 cycles
// whatever it is, so the length of the critical pixel's chain in time is (steps) x (instructions per iteration) x 6
// cycles: this loop is what shortens it.  Runs until a lane fails its bound (it then waits for the wave's next full
// evaluation, lb <= 0), a lane finishes its raycast, or max_it steps were taken; returns the steps taken.
template <int KIND, int NOBJ, uint32_t SIG, int I, bool TWO = false, bool W1 = false>      // I >= 0: object I of the unrolled table; I < 0: object k of the run-time table
RT_D int march_fast_src_obj(const Params& P, Lane& L, Trk& T, int k, int max_it, int* why = nullptr) {
    float& lb = T.lb2;
    ObjTab tab = obj_table();
    const bool marching = L.state == ST_MARCH;
    const unsigned long long mm = __ballot(marching);
    int it = 0;
    unsigned long long fail = 0ull;
    // ONE rounding allowance for the whole loop (a lone wave pays ~6.5 cycles for every instruction of its chain, and the allowance
    // took five of them per step): the loop only goes on while lb > 0 and every step takes at least |s| off lb, so the path marched
    // in here stays below the lb the lane came with and |p|_1 grows by less than sqrt(3) (1 + 2^-20) times that: 2 lb is on the safe
    // side.  A larger allowance only ends the loop earlier (the full evaluation takes over): results are unchanged.
    const float eps = track_eps_loop(P, L.o, lb);
    for (;;) {
        asm volatile("" : "+s"(tab));
        float dk;
        if constexpr (I >= 0) {
            const ObjM o = load_obj<SIG, (I >= 0 ? I : 0)>(tab);
            dk = fabs_(signed_distance<KIND>(P, o, L.o, RT_SIG_CLS((I >= 0 ? I : 0)), jit_type(I)));
        } else {
            const ObjM o = tab[k];
            dk = fabs_(signed_distance<KIND>(P, o, L.o));
        }
        const bool ok = marching & (lb > dk + eps) & (!P.cfg.nearest_init | (dk < P.cfg.max_dis));
        // (the loop's exit as lane masks in scalar registers: a lane goes on iff it stepped and still marches; written with bool
        // selects the compiler materialised four 0/1 VGPRs and a dozen scalar instructions per step of this loop — and the vote
        // on `ok` is put together from the votes on its comparisons: a vote on the combined flag goes through a 0/1 VGPR)
        const unsigned long long okm = mm & __builtin_amdgcn_ballot_w64(lb > dk + eps) & (P.cfg.nearest_init ? __builtin_amdgcn_ballot_w64(dk < P.cfg.max_dis) : ~0ull);
        fail = mm & ~okm;
        if (ok) {
            const float s_new = march_update_src_lean<W1, false>(P, L, k, dk);      // (every marching lane's L.idx is k already)
            lb = track_decay(lb, s_new, eps);
            if constexpr (TWO) T.lb3 = track_decay(T.lb3, s_new, eps);
        }
        it++;
        const unsigned long long live = okm & __builtin_amdgcn_ballot_w64(L.state == ST_MARCH);
        if ((live != mm) | (max_it > 0 && it >= max_it)) {      // (max_it <= 0: no cap — a literal at the call, the test folds away)
#if RT_DEBUG_PHASE == 4
            if (why) *why = fail ? 2 : (live != mm ? 1 : 0);      // bound failed / a raycast ended / max_it
#endif
            break;
        }
    }
    // a lane whose bound failed in the last step waits for the wave's next full evaluation (from the scalar mask, after the loop:
    // written as a select on `ok` the compiler re-evaluates it in every iteration)
    if ((fail >> (threadIdx.x & 63u)) & 1ull) lb = -1.0f;
    return it;
}
template <int KIND, int NOBJ, uint32_t SIG, bool TWO = false, bool W1 = false>
RT_D int march_fast_src(const Params& P, Lane& L, Trk& lb, int k, int max_it, int* why = nullptr) {
    int it = 0;
    if constexpr (NOBJ > 0) {
        static_for<NOBJ, 1>([&](auto Ic) {
            constexpr int i = decltype(Ic)::value;
            if (k == i) it = march_fast_src_obj<KIND, NOBJ, SIG, i, TWO, W1>(P, L, lb, i, max_it, why);
        });
    } else {
        it = march_fast_src_obj<KIND, NOBJ, SIG, -1, TWO, W1>(P, L, lb, k, max_it, why);
    }
    return it;
}

// The lean loop for lanes that track DIFFERENT objects (round 6): every marching lane holds a valid bound for its own nearest
// object.  Each lane gathers ITS object's constants from the block's LDS table once (16 registers) and the loop evaluates that one
// object per lane — general rotation, per-lane shape switch: the object-parallel evaluation's arithmetic, i.e. bit for bit the
// |sdf_k|(p) of the unrolled code — with the bound test, the raycast bookkeeping and one vote per step: ~100 instructions for ANY
// number of rays on ANY objects, where the tracked rounds cost one round per distinct object (1.7 kcycles measured), the
// object-parallel step 2.0 k and the unrolled full evaluation 2.7 k.  The rays that make a launch long graze one surface each for
// hundreds of steps — different surfaces in the same wave: this is their loop.  Ends like the one-object loop: a lane fails its
// bound (it then waits for the wave's next full / object-parallel evaluation), a raycast ends, or max_it steps were taken.
template <int KIND, bool TWO>
RT_D int march_fast_src_lanes(const Params& P, const ObjFull* lds_obj, Lane& L, Trk& T, int max_it) {
    float& lb = T.lb2;
    const bool marching = L.state == ST_MARCH;
    const unsigned long long mm = __ballot(marching);
    const ObjM ob = *reinterpret_cast<const ObjM*>(lds_obj + (marching ? L.idx : 0));
    const float eps = track_eps_loop(P, L.o, lb);
    int it = 0;
    unsigned long long fail = 0ull;
    for (;;) {
        const float dk = fabs_(signed_distance<KIND>(P, ob, L.o));
        const bool ok = marching & (lb > dk + eps) & (!P.cfg.nearest_init | (dk < P.cfg.max_dis));
        const unsigned long long okm = mm & __builtin_amdgcn_ballot_w64(lb > dk + eps) & (P.cfg.nearest_init ? __builtin_amdgcn_ballot_w64(dk < P.cfg.max_dis) : ~0ull);
        fail = mm & ~okm;
        if (ok) {
            const float s_new = march_update_src_lean<false, false>(P, L, L.idx, dk);
            lb = track_decay(lb, s_new, eps);
            if constexpr (TWO) T.lb3 = track_decay(T.lb3, s_new, eps);
        }
        it++;
        const unsigned long long live = okm & __builtin_amdgcn_ballot_w64(L.state == ST_MARCH);
        if ((live != mm) | (max_it > 0 && it >= max_it)) break;
    }
    if ((fail >> (threadIdx.x & 63u)) & 1ull) lb = -1.0f;
    return it;
}

// The lean loop on TWO objects: every marching lane tracks the same pair {a, b}, a < b, and holds a valid lb3.  Both objects
// are evaluated, the nearer one (the lower index on a tie, as nearest() resolves it: objects are visited in index order with a
// strict `<`) is exactly what nearest() returns while lb3 > min + eps.  lb2 is re-derived on the way (the other object of the
// pair, or lb3): when the second surface recedes the one-object loop can take over again without a full evaluation.
template <int KIND, int NOBJ, uint32_t SIG, int A, int B, bool W1 = false>      // A >= 0: objects A < B of the unrolled table; A < 0: objects a, b of the run-time table
RT_D int march_fast2_src_pair(const Params& P, Lane& L, Trk& T, int a, int b, int max_it) {
    ObjTab tab = obj_table();
    const bool marching = L.state == ST_MARCH;
    const unsigned long long mm = __ballot(marching);
    int it = 0;
    unsigned long long fail = 0ull;
    const float eps = track_eps_loop(P, L.o, T.lb3);      // (as in the one-object loop: the path marched in here stays below lb3)
    for (;;) {
        asm volatile("" : "+s"(tab));
        float dA, dB;
        if constexpr (A >= 0) {
            const ObjM oa = load_obj<SIG, (A >= 0 ? A : 0)>(tab);
            const ObjM ob = load_obj<SIG, (B >= 0 ? B : 0)>(tab);
            dA = fabs_(signed_distance<KIND>(P, oa, L.o, RT_SIG_CLS((A >= 0 ? A : 0)), jit_type(A)));
            dB = fabs_(signed_distance<KIND>(P, ob, L.o, RT_SIG_CLS((B >= 0 ? B : 0)), jit_type(B)));
        } else {
            const ObjM oa = tab[a];
            const ObjM ob = tab[b];
            dA = fabs_(signed_distance<KIND>(P, oa, L.o));
            dB = fabs_(signed_distance<KIND>(P, ob, L.o));
        }
        const bool lt = dB < dA;
        const float d = lt ? dB : dA;
        const float d_other = lt ? dA : dB;
        const bool ok = marching & (T.lb3 > d + eps) & (!P.cfg.nearest_init | (d < P.cfg.max_dis));
        const unsigned long long okm = mm & __builtin_amdgcn_ballot_w64(T.lb3 > d + eps) & (P.cfg.nearest_init ? __builtin_amdgcn_ballot_w64(d < P.cfg.max_dis) : ~0ull);
        fail = mm & ~okm;
        if (ok) {
            const float s_new = march_update_src_lean<W1, true>(P, L, lt ? b : a, d);
            T.lb2 = track_decay(fmin_(d_other - eps, T.lb3), s_new, eps);
            T.lb3 = track_decay(T.lb3, s_new, eps);
            T.k2 = lt ? a : b;
        }
        it++;
        const unsigned long long live = okm & __builtin_amdgcn_ballot_w64(L.state == ST_MARCH);
        if ((live != mm) | (max_it > 0 && it >= max_it)) break;
    }
    if ((fail >> (threadIdx.x & 63u)) & 1ull) T.lb3 = T.lb2 = -1.0f;
    return it;
}
template <int KIND, int NOBJ, uint32_t SIG, bool W1 = false>
RT_D int march_fast2_src(const Params& P, Lane& L, Trk& T, int a, int b, int max_it) {      // a < b, wave-uniform
    int it = 0;
    if constexpr (NOBJ > 0) {
        static_for<NOBJ, 1>([&](auto Ia) {
            constexpr int A = decltype(Ia)::value;
            if (a == A) {
                static_for<NOBJ, 1>([&](auto Ib) {
                    constexpr int B = decltype(Ib)::value;
                    if constexpr (A < B) {
                        if (b == B) it = march_fast2_src_pair<KIND, NOBJ, SIG, A, B, W1>(P, L, T, A, B, max_it);
                    }
                });
            }
        });
    } else {
        it = march_fast2_src_pair<KIND, NOBJ, SIG, -1, -1, W1>(P, L, T, a, b, max_it);
    }
    return it;
}

// One iteration of the tracked march for the wave's marching lanes, whichever of its forms applies (all wave-uniform):
//   1  every lane's lb2 promises to hold and all track the same object: the one-object lean loop (until a lane stops);
//   2  every lane's lb3 promises to hold and all track the same pair: the two-object lean loop;
//   3  lanes with a valid lb2 take one tracked step each on their own object — unless too many would have to wait;
//   4  a full evaluation for everybody (three smallest distances: both bounds fresh);
//   5  (callers that pass an OpView; at most 8 lanes marching, at most 8 objects, option src_op) the object-parallel evaluation
//      in place of 3 and 4;
//   6  (same callers, src_op bit 2) every lane's lb2 promises to hold but the lanes track different objects: the per-lane lean loop.
// "Promises": lb > the lane's LAST distance — a predictor only (the loops test exactly); it keeps a wedge ray from paying
// for a one-object attempt that fails on its first step after every full evaluation.  Returns the form taken; `steps` =
// iterations of a lean loop (1 otherwise).
// TWO (compile time): the kernel keeps both bounds.  The fused pool kernel does not — its 96 registers have no room for the
// second bound and the 21 instances of the two-object loop (measured with both compiled in: 13 spills, 1080p 58.7 -> 60.9 ms
// with the second bound unused, 70.3 ms used: the sparse phases of its light waves are too short to win it back); the march
// kernel of the wavefront split (rt_split.hpp), whose launch is as long as its slowest raycast, does.
template <int KIND, int NOBJ, uint32_t SIG, bool TWO = false>
RT_D int tracked_iteration(const Params& P, Lane& L, Trk& T, int n_march, int max_it, int& steps, unsigned long long* dbg_rounds = nullptr, int* why = nullptr,
                           const OpView V = OpView{nullptr, nullptr}) {
    const bool marching = L.state == ST_MARCH;
    const unsigned long long mm = __ballot(marching);
    const int first = (int)__builtin_ctzll(mm);
    steps = 1;
    const bool two = TWO && P.src_track >= 2;
    // the raycasts that make a launch long lose their over-relaxation within a few steps (w: 1.6 -> 1 at the first overshoot) and
    // then march hundreds of steps with w == 1: the lean loops have an instance without the relaxation bookkeeping for that
    // (kernels that keep both bounds only — the chain kernel and the split march; the fused pool kernel has no room for more code)
    const bool w1 = TWO && (RT_TRK_W1 != 0) && __ballot(marching & (L.w != 1.0f)) == 0ull;
    // (one bound only: any valid lb2 tries the lean loop, as round 4 did)
    if (__ballot(marching & (T.lb2 > (two ? L.dist : 0.0f))) == mm) {
        const int k0 = __builtin_amdgcn_readlane(L.idx, first);
        if (__ballot(marching && L.idx != k0) == 0ull) {
            if (TWO && (RT_TRK_W1 != 0) && w1) steps = march_fast_src<KIND, NOBJ, SIG, TWO, (TWO && RT_TRK_W1 != 0)>(P, L, T, k0, max_it, why);
            else steps = march_fast_src<KIND, NOBJ, SIG, TWO>(P, L, T, k0, max_it, why);
            return 1;
        }
        // every lane's bound promises to hold, but they track different objects: the per-lane lean loop (form 6, round 6)
        if (V.xch != nullptr && (P.src_op & 4) != 0) {
            steps = march_fast_src_lanes<KIND, TWO>(P, V.lds_obj, L, T, max_it);
            return 6;
        }
    }
    if constexpr (TWO) if (two && __ballot(marching & (T.lb3 > L.dist) & (T.k2 != L.idx)) == mm) {
        const int lo = L.idx < T.k2 ? L.idx : T.k2, hi = L.idx < T.k2 ? T.k2 : L.idx;
        const int key = lo | (hi << 8);
        const int key0 = __builtin_amdgcn_readlane(key, first);
        if (__ballot(marching && key != key0) == 0ull) {
            if (marching) T.lb2 = -1.0f;      // (re-derived by the loop's first step)
            if ((RT_TRK_W1 != 0) && w1) steps = march_fast2_src<KIND, NOBJ, SIG, (RT_TRK_W1 != 0)>(P, L, T, key0 & 255, key0 >> 8, max_it);
            else steps = march_fast2_src<KIND, NOBJ, SIG>(P, L, T, key0 & 255, key0 >> 8, max_it);
            return 2;
        }
    }
    // (round 6) a handful of rays that do not share their object(s): one object-parallel evaluation serves them all, in half the
    // instructions of the unrolled one and with both bounds fresh — form 5 replaces forms 3 and 4 while at most 8 lanes march
    if (V.xch != nullptr && n_march <= 8) {      // (the caller passes a buffer only when option src_op asks for it and the scene has <= 8 objects)
        march_step_src_op<KIND>(P, V, L, T);
        return 5;
    }
    const bool can = marching && T.lb2 > 0.0f;
    const int n_can = __popcll(__ballot(can));
    if (n_can > 0 && n_march - n_can < 1 + (n_can >> 2)) {
        march_step_src_tracked<KIND, NOBJ, SIG, TWO>(P, L, T, can, dbg_rounds);
        return 3;
    }
    if (marching) {
        if constexpr (TWO) {
            if (two) march_step_src_full3<KIND, NOBJ, SIG>(P, L, T);
            else march_step_src_full2<KIND, NOBJ, SIG>(P, L, T);
        } else {
            march_step_src_full2<KIND, NOBJ, SIG>(P, L, T);
        }
    }
    return 4;
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
    pixel_of(P, X.q, px, py);
    const size_t pi = (size_t)px * g.height + py;
    // The loop holds only what a roulette kill repeats (stream key, survival probability, one draw): a wave stays in it as
    // long as ANY of its contexts keeps being killed (about 3.4 rounds for 57 contexts at a kill rate of 0.2), so the
    // deposit and the camera-ray regeneration — 200 instructions — come after it, once, for the contexts that survived.
    bool go = false;
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
        go = true;
        break;
    }
    if (go) {
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
    // what this residency cost (one context per pixel at a time: a plain read-modify-write; the plan kernels consume and clear it)
    if (P.cost_buffer) P.cost_buffer[X.q] += X.cost;
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
#if RT_POOL_OP
    __shared__ float4 xch_all[4][8];
#endif
    zero_next_counters(P);
    stage_objects(P, lds_obj);

    const unsigned long long t_wave0 = __builtin_readcyclecounter();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t (*pool)[64] = pool_all[wave];
    uint32_t* sstate = sstate_all[wave];
    const PoolView V = {pool_all[wave], sstate_all[wave], tbl_all[wave]};
    // (RT_POOL_OP: compile-time — the fused pool kernel runs at its register limit, 96 VGPRs for five waves per SIMD: with the
    // object-parallel step compiled in it spills 10 registers instead of 2)
#if RT_POOL_OP
    const OpView OV = {lds_obj, ((P.src_op & 2) != 0 && P.n_obj <= 8) ? xch_all[wave] : nullptr};
#else
    const OpView OV = {lds_obj, nullptr};
#endif
    sstate[lane] = SL_EMPTY;

    // ---- what this wave owns and how it walks it (all wave-uniform).  Heavy waves first: wave 0 of every block, then
    // wave 1, ... so that they spread over the CUs (a block's four waves sit on the four SIMDs of one CU)
    SrcWave Wv;
    Wv.run_inv = 1u | (1u << 26);
    const uint32_t nw = gridDim.x * 4u;
    // Wave index h: heavy waves are h < n_hw.  Blocks are placed round-robin over the CUs in launch order, so block b is the
    // (b / n_cu)-th oldest resident block of CU b % n_cu and its wave w sits on SIMD w.  h enumerates the OLDEST blocks first
    // and, inside a residency class, wave 0 of every CU, then wave 1, ...: heavy waves (the launch's critical path) spread
    // over all SIMDs of the chip one each before any SIMD gets a second one, and are the oldest wave of their SIMD, which the
    // issue arbiter serves first.  (Round 4 used h = wave * gridDim + block: with three blocks per CU the heavy waves were
    // the three wave-0s of a CU — all on one SIMD.)
    const uint32_t hm_cu = (uint32_t)P.n_cu > 0u ? (uint32_t)P.n_cu : gridDim.x;
    const uint32_t hm_cls = blockIdx.x / hm_cu, hm_b0 = hm_cls * hm_cu;
    const uint32_t hm_nb = hm_b0 + hm_cu <= gridDim.x ? hm_cu : gridDim.x - hm_b0;      // blocks in this block's residency class
    const uint32_t h = (uint32_t)__builtin_amdgcn_readfirstlane((int)(hm_b0 * 4u + (uint32_t)wave * hm_nb + (blockIdx.x - hm_b0)));
    // (the chain set, if there is one, is somebody else's: the ownership below deals the REST of the list)
    Wv.skip = chain_skip(P);
    Wv.skip = Wv.skip < (uint32_t)P.np ? Wv.skip : (uint32_t)P.np;
    const uint32_t np_all = (uint32_t)P.np;
    const uint32_t np = np_all - Wv.skip;
    // Heavy waves come in two sizes: the very heaviest pixels — the launch's critical path IS one of their chains — sit in
    // small waves (2 .. tiny_own pixels), where a context never waits for a lane and the lean tracked loop runs most of
    // the time; the other heavy pixels in waves of heavy_own.
    uint32_t n_heavy = 0, n_hw = 0, n_top = 0, n_tw = 0;
    const uint32_t hown = (uint32_t)P.heavy_own;
    uint32_t town = (uint32_t)P.tiny_own;
    if (P.order && hown > 0u) {
        n_heavy = P.plan->n_heavy;
        n_heavy = n_heavy > Wv.skip ? n_heavy - Wv.skip : 0u;
        n_heavy = n_heavy < np ? n_heavy : np;
        const uint32_t budget = P.plan->tiny_waves;
        if (town > 0u && budget > 0u) {
            // the heaviest pixels in small waves, as small as the plan's budget of waves allows
            if (n_heavy <= 2u * budget) town = town < 2u ? town : 2u, n_top = n_heavy;
            else if (n_heavy <= 4u * budget) town = town < 4u ? town : 4u, n_top = n_heavy;
            else n_top = n_heavy < town * budget ? n_heavy : town * budget;
            n_tw = (n_top + town - 1u) / town;
        }
        n_hw = n_tw + (n_heavy - n_top + hown - 1u) / hown;
        if (n_hw > nw / 2u) {        // too many for this grid: without the small waves, ...
            n_top = n_tw = 0;
            n_hw = (n_heavy + hown - 1u) / hown;
        }
        if (n_hw > nw / 2u) {        // ... or with none at all (the plan kernel clamps n_heavy for the grid it was made for)
            n_hw = 0;
            n_heavy = 0;
        }
    }
    const bool heavy = h < n_hw;
    // a heavy wave is the launch's critical path: it issues ahead of the light waves on its SIMD
    if (heavy && P.heavy_prio > 0) __builtin_amdgcn_s_setprio((short)3);
    if (h < n_tw) {
        Wv.base = h * town;
        Wv.stride = 1u;
        const uint32_t left = n_top - Wv.base;
        Wv.n_own = left < town ? left : town;
    } else if (heavy) {
        Wv.base = n_top + (h - n_tw) * hown;
        Wv.stride = 1u;
        const uint32_t left = n_heavy - Wv.base;
        Wv.n_own = left < hown ? left : hown;
    } else {
        const uint32_t l = h - n_hw, n_lw = nw - n_hw, n_light = np - n_heavy;
        Wv.base = n_heavy + l;
        Wv.stride = n_lw;
        Wv.n_own = n_light > l ? (n_light - l - 1u) / n_lw + 1u : 0u;
        // Age-weighted shares.  The k-th resident block of a CU (blockIdx / n_cu: blocks are placed round-robin in launch
        // order) is the k-th OLDEST wave on its SIMD, and the issue arbiter favours the older wave: with equal shares
        // the five waves of a SIMD finish one after the other (75 ... 160 Mcycles at 1080p) and the SIMD runs its last
        // third under-occupied.  Class c takes age_w[c] entries of `order` per round and wave instead of one, so that
        // all of them can end together; neighbouring entries have similar cost, so every wave still gets a fair sample.
        const uint32_t n_cu = (uint32_t)P.n_cu > 0u ? (uint32_t)P.n_cu : gridDim.x;
        const uint32_t n_cls = (gridDim.x + n_cu - 1u) / n_cu;
        const bool age_plan = P.age_on == 1 && P.order && P.plan->age_valid && P.plan->age_cls == n_cls;
        if ((P.age_on == 2 || age_plan) && n_cls > 1u && n_cls <= 8u) {
            const uint32_t G = gridDim.x;
            const uint32_t c = blockIdx.x / n_cu;
            // light waves per class and this wave's rank among the light waves of its class (ordered by wave, then block);
            // heavy waves are h < n_hw (h enumerates class, then wave, then block: see above)
            auto heavy_in = [&](uint32_t w, uint32_t cls) {      // heavy waves with wave index w among the blocks of class cls
                const uint32_t b0 = cls * n_cu, nb = b0 + n_cu <= G ? n_cu : G - b0;
                const uint32_t h0 = b0 * 4u + w * nb;                        // h of (class cls, wave w, first block)
                const uint32_t lim = n_hw > h0 ? n_hw - h0 : 0u;
                return lim < nb ? lim : nb;
            };
            uint32_t R = 0, off = 0, j = 0, wmax = 0, wmin = 0xffffffffu;
            for (uint32_t cls = 0; cls < n_cls; cls++) {
                const uint32_t b0 = cls * n_cu, b1 = (cls + 1u) * n_cu < G ? (cls + 1u) * n_cu : G;
                uint32_t cnt = 0;
                for (uint32_t w = 0; w < 4u; w++) cnt += (b1 - b0) - heavy_in(w, cls);
                const uint32_t wgt = age_plan ? P.plan->age_w[cls] : (P.age_pack >> (4u * cls)) & 15u;
                if (cls < c) off += cnt * wgt;
                R += cnt * wgt;
                wmax = wgt > wmax ? wgt : wmax;
                wmin = wgt < wmin ? wgt : wmin;
            }
            const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane(wave);      // (wave-uniform: keep all of this in scalar registers)
            for (uint32_t w = 0; w < wv; w++) j += ((c + 1u) * n_cu < G ? n_cu : G - c * n_cu) - heavy_in(w, c);
            {
                const uint32_t b0 = c * n_cu;
                const uint32_t hb = heavy_in(wv, c);                         // the heavy blocks of (class, wave) are its first ones
                j += (blockIdx.x - b0) - (hb < blockIdx.x - b0 ? hb : blockIdx.x - b0);
            }
            const uint32_t wgt = age_plan ? P.plan->age_w[c] : (P.age_pack >> (4u * c)) & 15u;
            // Weighted or plain dealing is ONE decision for the whole grid (the two schemes do not tile `order` together: a
            // wave that fell back on its own while others dealt by weight would own pixels twice / leave others unowned).
            // Every wave sees the same R, wmin, wmax: the limits (own_pixel: k < 2^14, run < 64) are tested for the heaviest
            // class, whose waves own at most (full + 1) * wmax pixels.
            const uint32_t full = R > 0u ? n_light / R : 0u;
            if (R > 0u && wmin > 0u && wmax < 64u && (unsigned long long)(full + 1u) * wmax < 16384ull) {
                const uint32_t rem = n_light - full * R;
                const uint32_t start = off + j * wgt;
                const uint32_t part = rem > start ? (rem - start < wgt ? rem - start : wgt) : 0u;
                Wv.base = n_heavy + start;
                Wv.stride = R;
                Wv.run_inv = wgt | ((((1u << 20) + wgt - 1u) / wgt) << 6);
                Wv.n_own = full * wgt + part;
            }
        }
    }
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
    // bookkeeping of the context being marched (a_meta: bounce-step and cost as in G_META, index bits zero; the cost is
    // kept RELATIVE to L.n_steps while the lane marches: subtracted at take-over, added back when the context is parked)
    vec3 a_col = mk(0, 0, 0);
    int a_depth = 0;
    uint32_t a_k = 0, a_q = 0, a_meta = 0, a_key = 0, a_cnt = 0;
    uint32_t n_samples = 0, n_dep = 0;
    unsigned long long m_ready = 0, m_shade = 0;
    const int T = P.shade_lanes;
    const int m_swap = P.swap_lanes;
    // tracked-object march: lower bound of every object but L.idx (<= 0: needs a full evaluation)
    Trk Tk = {-1.0f, -1.0f, 0};
    constexpr bool TRK = KIND == KIND_BOXES || KIND == KIND_GENERIC;
    const bool trk_ok = TRK && P.cull_ok != 0 && P.src_track != 0;

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
    unsigned long long dbg4[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long dbg_passes = 0, dbg_shaded = 0, dbg_march_iters = 0, dbg_march_lanes = 0, dbg_trk_iters = 0, dbg_trk_ok = 0, dbg_full2_iters = 0, dbg_trk_wait = 0, dbg_trk_rounds = 0, dbg_fast_iters = 0, dbg_tb_shade = 0, dbg_tb_load = 0, dbg_tb_adv = 0, dbg_t_disp = 0, dbg_t_trk = 0, dbg_t_full2 = 0, dbg_t_fast = 0, dbg_fast_calls = 0, dbg_s_flag = 0, dbg_s_done = 0, dbg_s_idle = 0, dbg_s_ready = 0, dbg_s_shade = 0;
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
                if (L.state != ST_IDLE) mine = ((uint32_t)gmeta_s(a_meta) >> Wv.lg_s) * Wv.n_own + a_k;
                if (sstate[lane] != SL_EMPTY) {
                    const uint32_t it = ((uint32_t)gmeta_s(pool[G_META][lane]) >> Wv.lg_s) * Wv.n_own + pool[G_K][lane];
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
#if RT_DEBUG_PHASE == 3
                unsigned long long tb0 = __builtin_readcyclecounter();
#endif
                PixCtx X;
                X.o = X.d = X.col = mk(0, 0, 0);
                X.depth = X.idx = X.s = 0;
                X.k = X.q = X.cost = X.key = X.cnt = 0;
                bool have = false, fresh = false;
                if (st == SL_HIT || st == SL_MISS) {
                    X.o = mk(u2f(pool[G_OX][lane]), u2f(pool[G_OY][lane]), u2f(pool[G_OZ][lane]));
                    X.d = mk(u2f(pool[G_DX][lane]), u2f(pool[G_DY][lane]), u2f(pool[G_DZ][lane]));
                    X.col = mk(u2f(pool[G_CR][lane]), u2f(pool[G_CG][lane]), u2f(pool[G_CB][lane]));
                    X.depth = (int)pool[G_DEPTH][lane];
                    const uint32_t meta = pool[G_META][lane];
                    X.idx = gmeta_idx(meta);
                    X.s = gmeta_s(meta);
                    X.cost = gmeta_cost(meta);
                    X.k = pool[G_K][lane];
                    X.q = pool[G_Q][lane];
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
#if RT_DEBUG_PHASE == 3
                { const unsigned long long t = __builtin_readcyclecounter(); dbg_tb_shade += t - tb0; tb0 = t; }
#endif
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
                        const uint32_t q = own_pixel(P, Wv, k);
                        int px, py;
                        if (q < np_all && pixel_of(P, q, px, py)) {
                            const size_t pi = (size_t)px * P.cfg.height + py;
                            bool masked = P.cfg.adaptive_sampling && !(P.diff_pixels[pi] > P.cfg.noise_threshold);
                            if (!masked) {
                                rtpbr_ray rb = P.ray_buffer[pi];
                                X.o = mk(rb.origin[0], rb.origin[1], rb.origin[2]);
                                X.d = mk(rb.direction[0], rb.direction[1], rb.direction[2]);
                                X.col = mk(rb.color[0], rb.color[1], rb.color[2]);
                                X.depth = rb.depth;
                                X.k = k;
                                X.q = q;
                                X.s = (int)(pass << Wv.lg_s);
                                X.cost = 0;
                                have = true;
                                fresh = true;
                            }
                        }
                    }
                    next_item += take;
                }
#if RT_DEBUG_PHASE == 3
                { const unsigned long long t = __builtin_readcyclecounter(); dbg_tb_load += t - tb0; tb0 = t; }
#endif
                bool ready = false;
                if (have) ready = pix_advance<KIND>(P, Wv, X, steps, fresh, n_samples, n_dep);
#if RT_DEBUG_PHASE == 3
                { const unsigned long long t = __builtin_readcyclecounter(); dbg_tb_adv += t - tb0; tb0 = t; }
#endif
                if (ready) {
                    pool[G_OX][lane] = f2u(X.o.x); pool[G_OY][lane] = f2u(X.o.y); pool[G_OZ][lane] = f2u(X.o.z);
                    pool[G_DX][lane] = f2u(X.d.x); pool[G_DY][lane] = f2u(X.d.y); pool[G_DZ][lane] = f2u(X.d.z);
                    pool[G_CR][lane] = f2u(X.col.x); pool[G_CG][lane] = f2u(X.col.y); pool[G_CB][lane] = f2u(X.col.z);
                    pool[G_DEPTH][lane] = (uint32_t)X.depth;
                    pool[G_META][lane] = gmeta_pack(0, X.s, X.cost);
                    pool[G_K][lane] = X.k;
                    pool[G_Q][lane] = X.q;
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
            rec[G_META] = (uint32_t)L.idx | (a_meta + (L.n_steps << 14));
            rec[G_K] = a_k;
            rec[G_Q] = a_q;
            rec[G_KEY] = a_key; rec[G_CNT] = a_cnt;
            const int r = pool_swap(V, lane, is_done, L.state == ST_IDLE, L.state == ST_HIT ? SL_HIT : SL_MISS, rec, m_ready, m_shade);
            if (r & 2) L.state = ST_IDLE;
            if (r & 1) {
                L.o = mk(u2f(rec[G_OX]), u2f(rec[G_OY]), u2f(rec[G_OZ]));
                L.d = mk(u2f(rec[G_DX]), u2f(rec[G_DY]), u2f(rec[G_DZ]));
                a_col = mk(u2f(rec[G_CR]), u2f(rec[G_CG]), u2f(rec[G_CB]));
                a_depth = (int)rec[G_DEPTH];
                a_meta = (rec[G_META] & ~31u) - (L.n_steps << 14);
                a_k = rec[G_K];
                a_q = rec[G_Q];
                a_key = rec[G_KEY]; a_cnt = rec[G_CNT];
                src_march_init(L);
                Tk.lb2 = Tk.lb3 = -1.0f;
            }
        }

#if RT_DEBUG_PHASE == 3
        RT_PHASE(dbg_t_disp)
#else
        RT_PHASE(tB)
#endif
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
            // When to leave the march loop.  Parked READY contexts waiting: as soon as swap_lanes lanes have finished (the
            // swap is cheap) — or, the same ski-rental rule with the swap's cost of about half a march iteration, when
            // the lane-iterations READY contexts have waited for a finished lane reach half the marching lanes: a wave of
            // 2 ... 8 heavy pixels never has swap_lanes finished lanes and used to march each raycast to its end first.  None waiting: finished lanes and parked contexts make no progress until the next shading
            // pass, which stalls the marching lanes for about leave_x8 / 8 march iterations — leave when the lane-iterations
            // wasted by waiting add up to what leaving costs (ski rental: within 2x of the best fixed batch size for ANY
            // arrival rate — grazing rays finish once per hundreds of iterations, ordinary ones every twenty).
            const int n_shade0 = __popcll(m_shade);
            int waste = 0;
            bool tracked = false;
            if constexpr (TRK) tracked = trk_ok && (heavy || n_march <= P.sparse_lanes);
            if (tracked) {
                if constexpr (TRK) {
#ifdef RT_DEBUG_PHASE
                    const unsigned long long ts0 = __builtin_readcyclecounter();
#endif
                    do {
                        // one iteration of the tracked march in whichever form applies (tracked_iteration above): a lean loop on one
                        // or two objects, single tracked steps, or a full evaluation for everybody
#ifdef RT_DEBUG_PHASE
                        dbg_march_iters++;
                        const uint32_t steps_before = L.n_steps;
                        const bool marching = L.state == ST_MARCH;
                        const unsigned long long tt0 = __builtin_readcyclecounter();
#endif
                        // the shading pass that waits (if one does) bounds a lean loop's stay: see the ski-rental rule below
                        int max_it = 1 << 20;
                        if (n_ready == 0 && n_shade0 > 0) {
                            max_it = (P.leave_x8 * n_march - waste * 8 + 8 * n_shade0 - 1) / (8 * n_shade0);
                            max_it = max_it < 1 ? 1 : max_it;
                        }
                        int it = 1;
#if RT_DEBUG_PHASE == 4
                        int why = 0;
                        const int form = tracked_iteration<KIND, NOBJ, SIG, (RT_POOL_TWO != 0)>(P, L, Tk, n_march, max_it, it, &dbg_trk_rounds, &why, OV);
                        if (form == 1) dbg4[why]++;
                        dbg4[3 + form]++;                     // [4..7]: iterations by form
                        dbg4[7 + form] += (unsigned)it;       // [8..11]: steps by form
#elif defined(RT_DEBUG_PHASE)
                        const int form = tracked_iteration<KIND, NOBJ, SIG, (RT_POOL_TWO != 0)>(P, L, Tk, n_march, max_it, it, &dbg_trk_rounds, nullptr, OV);
#else
                        const int form = tracked_iteration<KIND, NOBJ, SIG, (RT_POOL_TWO != 0)>(P, L, Tk, n_march, max_it, it, nullptr, nullptr, OV);
#endif
                        if (form <= 2 && n_ready == 0) waste += n_shade0 * (it - 1);      // (the last step is accounted below)
#ifdef RT_DEBUG_PHASE
                        {
                            const unsigned long long dt = __builtin_readcyclecounter() - tt0;
                            if (form <= 2) { dbg_fast_iters += (unsigned)it; dbg_t_fast += dt; dbg_fast_calls++; }
                            else if (form == 3) { dbg_trk_iters++; dbg_t_trk += dt; dbg_trk_ok += (unsigned)__popcll(__ballot(L.n_steps != steps_before)); }
                            else { dbg_full2_iters++; dbg_t_full2 += dt; }
                            dbg_march_lanes += (unsigned)__popcll(__ballot(L.n_steps != steps_before));
                            dbg_s_flag += (unsigned)__popcll(__ballot(marching && L.n_steps == steps_before));
                            dbg_s_done += (unsigned)__popcll(__ballot(!marching && L.state != ST_IDLE));
                            dbg_s_idle += (unsigned)__popcll(__ballot(L.state == ST_IDLE));
                            dbg_s_ready += (unsigned)__popcll(m_ready);
                            dbg_s_shade += (unsigned)__popcll(m_shade);
                        }
#endif
                        n_march = __popcll(__ballot(L.state == ST_MARCH));
                        n_done = __popcll(__ballot(L.state == ST_HIT || L.state == ST_MISS));
                        waste += n_ready == 0 ? n_done + n_shade0 : (n_done < n_ready ? n_done : n_ready);
                    } while (n_march > 0 && (n_ready > 0 ? (n_done < m_swap && waste * 2 < n_march + 1) : waste * 8 < P.leave_x8 * n_march));
#ifdef RT_DEBUG_PHASE
                    tD += __builtin_readcyclecounter() - ts0;      // (cycles of the tracked march loops, reported in place of the dispatch phase)
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
                    waste += n_ready == 0 ? n_done + n_shade0 : (n_done < n_ready ? n_done : n_ready);
                } while (n_march > 0 && (n_ready > 0 ? (n_done < m_swap && waste * 2 < n_march + 1) : waste * 8 < P.leave_x8 * n_march));
                Tk.lb2 = Tk.lb3 = -1.0f;     // the plain steps did not maintain the bounds
            }
        }
        RT_PHASE(tA)
    }
#ifdef RT_DEBUG_PHASE
    if (lane == 0) {   // cycles per phase and wave lifetime (>> 10), passes, slots shaded, march iterations, lanes marching
        atomicAdd(&P.counters->dbg[0], tB >> 10);
        atomicAdd(&P.counters->dbg[1], dbg_trk_iters | ((tD >> 10) << 32));      // march iterations that ran the tracked step; cycles of the tracked loops >> 10 in the high word
        atomicAdd(&P.counters->dbg[2], tA >> 10);
        atomicAdd(&P.counters->dbg[3], (__builtin_readcyclecounter() - t_start) >> 10);
        atomicAdd(&P.counters->dbg[4], dbg_passes);
        atomicAdd(&P.counters->dbg[5], dbg_shaded);
        atomicAdd(&P.counters->dbg[14], dbg_trk_ok);
        atomicAdd(&P.counters->dbg[6], dbg_march_iters);
        atomicAdd(&P.counters->dbg[7], dbg_march_lanes);
        const unsigned long long life = __builtin_readcyclecounter() - t_start;
        if (heavy) {
            atomicMax(&P.counters->dbg[8], life);
            atomicAdd(&P.counters->dbg[9], life >> 10);
            atomicAdd(&P.counters->dbg[10], 1ull);
        } else {
            atomicMax(&P.counters->dbg[11], life);
        }
#if RT_DEBUG_PHASE == 3
        atomicAdd(&P.counters->dbg[28], dbg_tb_shade >> 10);
        atomicAdd(&P.counters->dbg[29], dbg_tb_load >> 10);
        atomicAdd(&P.counters->dbg[30], dbg_tb_adv >> 10);
        atomicAdd(&P.counters->dbg[31], dbg_t_disp >> 10);
        if (false) {
#elif RT_DEBUG_PHASE == 4
        if (heavy && h == 0) {     // the heaviest wave: why its lean loops end, what its full evaluations are for, lean steps per tracked object
            for (int i = 0; i < 16; i++) P.counters->dbg[16 + i] = dbg4[i];
        }
        if (false) {
#elif RT_DEBUG_PHASE == 2
        {   // histogram of the wave lifetimes (bins of 16 Mcycles): light waves in dbg[16..31]
            const unsigned bin = (unsigned)(life >> 24);
            if (!heavy) atomicAdd(&P.counters->dbg[16 + (bin < 15u ? bin : 15u)], 1ull);
        }
        if (false) {
#else
        if (heavy && h == 0) {      // the wave that owns the heaviest pixels
#endif
            P.counters->dbg[16] = life;
            P.counters->dbg[17] = tB;
            P.counters->dbg[18] = tA;
            P.counters->dbg[19] = dbg_march_iters;
            P.counters->dbg[20] = dbg_trk_iters;
            P.counters->dbg[21] = dbg_full2_iters;
            P.counters->dbg[22] = dbg_passes;
            P.counters->dbg[23] = dbg_trk_ok;
            P.counters->dbg[24] = dbg_march_lanes;
            P.counters->dbg[25] = dbg_fast_iters | (dbg_fast_calls << 32);
            P.counters->dbg[26] = dbg_t_trk;
            P.counters->dbg[27] = dbg_t_full2;
            P.counters->dbg[28] = dbg_s_flag;
            P.counters->dbg[29] = dbg_s_done;
            P.counters->dbg[30] = dbg_t_fast;
            P.counters->dbg[31] = dbg_s_ready | (dbg_s_shade << 32);
        }
        atomicAdd(&P.counters->dbg[12], dbg_full2_iters);
        atomicAdd(&P.counters->dbg[13], dbg_trk_wait);
        atomicAdd(&P.counters->dbg[15], dbg_fast_iters);
    }
#endif
#if RT_DEBUG_PHASE == 5
    {   // per-wave record over diff_buffer (unused without adaptive sampling), after the chain kernel's records: who is slow?
        const uint32_t tot = wave_sum(L.n_steps);
        if (lane == 0) {
            unsigned long long* d = reinterpret_cast<unsigned long long*>(P.diff_buffer) + (size_t)(4096u + blockIdx.x * 4u + (uint32_t)wave) * 8u;
            d[0] = (unsigned long long)h | ((unsigned long long)(blockIdx.x / ((uint32_t)P.n_cu > 0u ? (uint32_t)P.n_cu : gridDim.x)) << 32);
            d[1] = __builtin_readcyclecounter() - t_start;
            d[2] = (heavy ? 1ull : 0ull) | ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 8) | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 15u) << 40);   // | HW_ID << 8 | XCC_ID << 40 (tools/gpu_pool_simd.py)
            d[3] = Wv.n_own;
            d[4] = dbg_march_iters;
            d[5] = dbg_passes;
            d[6] = tot;
            d[7] = 0x7654321ull;
        }
    }
#endif
    // the light waves' lifetimes, per residency slot: what the next plan tunes the age weights with
    if (!heavy && P.age_on == 1 && P.plan && lane == 0 && Wv.n_own > 0u) {
        const uint32_t n_cu = (uint32_t)P.n_cu > 0u ? (uint32_t)P.n_cu : gridDim.x;
        const uint32_t c = blockIdx.x / n_cu;
        if (c < 8u) {
            atomicAdd(&P.plan->life_sum[c], (unsigned long long)(__builtin_readcyclecounter() - t_wave0));
            atomicAdd(&P.plan->life_cnt[c], 1u);
        }
    }
    flush_counters(P, L.n_steps, L.n_raycasts, L.n_hits, L.n_sky, n_samples, n_dep);
}

template <int KIND>
__global__ void __launch_bounds__(256) persistent_steps(const Params P, int steps) { persistent_steps_impl<KIND>(P, steps); }
template <int KIND>
__global__ void __launch_bounds__(256) persistent_pool(const Params P, int steps) { persistent_pool_impl<KIND>(P, steps); }

}  // namespace rt
