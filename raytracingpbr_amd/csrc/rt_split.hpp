// rt_split.hpp — the src/ persistent-ray form as a WAVEFRONT SPLIT, for launches of one (or a few) bounce-steps: the way the
// reference itself calls pathtrace() — once per displayed frame, one bounce-step per pixel (src/renderer.py:29-30,
// src/pathtracer.py:80-103).  A fused launch of many steps is the pool kernel's business (rt_persistent.hpp): there a
// context lives for hundreds of steps and the LDS pool keeps lanes busy.  With ONE step per launch every context needs at most
// one raycast, the pool can never batch its shading (16 of 64 slots per pass, measured: the shading passes cost more issue
// slots than the whole march) and the launch is as long as its longest raycast.  So a bounce-step becomes three kernels over
// the same ray_buffer (T6), each coherent in what it does:
//   src_gen     one lane per pixel, frame order: russian_roulette + track_once (src/pathtracer.py:53-77) — roulette, deposit
//               into image_buffer, camera-ray regeneration — and one word per pixel: "needs a raycast" + the RNG position;
//   src_march   raycast() (src/scene.py:59-84) only.  Persistent waves take groups of 32 pixels from the COST-ORDERED list of
//               the plan kernels (heaviest first: the launch's longest raycasts start early; one claim counter per team of blocks),
//               refill finished lanes in registers from a prefetched group, march with the two-bound tracked march where it
//               applies, and leave the moved origin plus {hit / miss, nearest object} behind;
//   src_shade   one lane per pixel, frame order: the rest of raytrace() (src/pathtracer.py:16-36) — surface interaction or
//               environment lookup, stop tests — and the ray state the next launch starts from.
// Same device functions as the fused kernels, same RNG stream positions: ray_buffer, image_buffer and the counters are bit
// for bit those of the other two schedulers (tests/test_gpu_parity.py).
#pragma once
#include "rt_persistent.hpp"

namespace rt {

enum { MS_NONE = 0, MS_MARCH = 1, MS_HIT = 2, MS_MISS = 3 };
// march word of a local pixel: state (2 bits) | nearest object (5 bits) << 2 | RNG draws of this bounce-step so far << 8
RT_D uint32_t mw_pack(uint32_t state, int idx, uint32_t cnt) { return state | ((uint32_t)idx << 2) | (cnt << 8); }

// The rest of raytrace() (src/pathtracer.py:16-36) after the raycast of bounce-step `base` ended in `word` (MS_HIT / MS_MISS), on the
// pixel's ray record in registers: surface interaction or environment lookup, stop tests — the state the next bounce-step starts from.
template <int KIND>
RT_D void src_shade_core(const Params& P, const ObjFull* lds_obj, int px, int py, uint32_t word, uint32_t base, rtpbr_ray& rb,
                         uint32_t& n_hits, uint32_t& n_sky, uint32_t& n_samples) {
    vec3 o = mk(rb.origin[0], rb.origin[1], rb.origin[2]);
    vec3 d = mk(rb.direction[0], rb.direction[1], rb.direction[2]);
    vec3 col = mk(rb.color[0], rb.color[1], rb.color[2]);
    const uint32_t key = rng_key(P.cfg.seed, (uint32_t)px, (uint32_t)py, base);
    uint32_t cnt = word >> 8;
    // depth += 1 (scene.py:83)
    int depth = rb.depth + 1;
    if ((word & 3u) == MS_HIT) {
        const ObjFull ob = lds_obj[(word >> 2) & 31u];
        surface_interaction<KIND>(P, ob, o, o, d, col, key, cnt);
        n_hits = 1;
        float intensity = brightness(col);
        col = col * mk(ob.emission[0], ob.emission[1], ob.emission[2]);
        float visible = brightness(col);
        bool stop = intensity < visible || visible < P.cfg.vis_lo || visible > P.cfg.vis_hi;
        if (stop) depth = -depth;
    } else {
        depth = -depth;
        col = col * sky_color(P, d);
        n_sky = 1;
        if (P.cfg.primary_miss == RTPBR_PRIMARY_BLACK) col = col * (depth < -1 ? 1.0f : 0.0f);
    }
    n_samples = 1;
    rb.origin[0] = o.x; rb.origin[1] = o.y; rb.origin[2] = o.z;
    rb.direction[0] = d.x; rb.direction[1] = d.y; rb.direction[2] = d.z;
    rb.color[0] = col.x; rb.color[1] = col.y; rb.color[2] = col.z;
    rb.depth = depth;
}

// SHADE_FIRST (round 6, the lazy shading of one-step launches): the SAME pass over ray_buffer first finishes the previous bounce-step
// (P.shade_base) — the shading the host has not launched yet — and then generates this one: one read and one write of the 40-byte ray
// record instead of two each, one kernel instead of two (gen 39 + shade 38 us of a 430 us launch at 1080p).  The arithmetic per pixel
// is the two kernels' in sequence.  COUNT: the shading belongs to THIS rtpbr_sample() call (an earlier step of a call of several) and
// counts into its work counters; a shading left over from the previous call does not (nobody can ask for that call's counters any
// more: rt_capi.hip flushes the pending shading before anything that could see it).
template <int KIND, bool SHADE_FIRST = false, bool COUNT = false>
RT_D void src_gen_impl(const Params& P) {
    __shared__ ObjFull lds_obj[SHADE_FIRST ? MAX_OBJ : 1];
    zero_next_counters(P);
    if constexpr (SHADE_FIRST) stage_objects(P, lds_obj);
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    int px = 0, py = 0;
    const bool in_frame = q < (uint32_t)P.np && pixel_of(P, q, px, py);
    bool valid = in_frame;
    const rtpbr_config& g = P.cfg;
    const size_t pi = (size_t)px * g.height + py;
    // self-adaptive sampling mask (src/pathtracer.py:97-101)
    if (valid && g.adaptive_sampling && !(P.diff_pixels[pi] > g.noise_threshold)) valid = false;
    uint32_t word = MS_NONE, n_samples = 0, n_dep = 0, n_hits = 0, n_sky = 0;
    uint32_t old_word = MS_NONE;
    if constexpr (SHADE_FIRST) old_word = in_frame ? P.march_out[q] : (uint32_t)MS_NONE;
    const bool shade = SHADE_FIRST && ((old_word & 3u) == MS_HIT || (old_word & 3u) == MS_MISS);
    if (valid || shade) {
        rtpbr_ray rb = P.ray_buffer[pi];
        if constexpr (SHADE_FIRST) {
            if (shade) {
                uint32_t h_ = 0, s_ = 0, n_ = 0;
                src_shade_core<KIND>(P, lds_obj, px, py, old_word, P.shade_base, rb, h_, s_, n_);
                if constexpr (COUNT) n_hits = h_, n_sky = s_, n_samples = n_;
            }
        }
        if (valid) {
        const uint32_t key = rng_key(g.seed, (uint32_t)px, (uint32_t)py, P.sample_base);
        uint32_t cnt = 0;
        int depth = rb.depth;
        // russian_roulette :65-77
        float p = (depth == 0) ? 1.0f : g.quality_per_sample;
        p -= (float)depth * (1.0f / (float)g.max_raytrace);
        if (rng_next(key, cnt) > p) {
            rb.color[0] = rb.color[1] = rb.color[2] = 0.0f;
            rb.depth = -depth;
            n_samples += 1;
        } else {
            vec3 col = mk(rb.color[0], rb.color[1], rb.color[2]) * (1.0f / p);
            // track_once :53-62
            if (depth < 1 || depth > g.max_raytrace) {
                float4 acc = P.image_buffer[pi];
                acc.x += col.x;
                acc.y += col.y;
                acc.z += col.z;
                acc.w += 1.0f;
                P.image_buffer[pi] = acc;
                n_dep = 1;
                vec3 o, d;
                gen_ray(P, px, py, key, cnt, o, d);
                rb.origin[0] = o.x; rb.origin[1] = o.y; rb.origin[2] = o.z;
                rb.direction[0] = d.x; rb.direction[1] = d.y; rb.direction[2] = d.z;
                col = mk(1, 1, 1);
                rb.depth = 0;
            }
            rb.color[0] = col.x; rb.color[1] = col.y; rb.color[2] = col.z;
            word = mw_pack(MS_MARCH, 0, cnt);
        }
        }
        P.ray_buffer[pi] = rb;
    }
    if (q < (uint32_t)P.np) P.march_out[q] = word;
    // the march kernel's team counters start from zero
    if (blockIdx.x == 0)
        for (int t = threadIdx.x; t < P.n_teams; t += blockDim.x) P.team_counter[t * 16] = 0u;
    flush_counters(P, 0, 0, n_hits, n_sky, n_samples, n_dep);
}

// One entry of the march list as the march kernel stages it: local pixel, frame index, march word, ray
enum { E_Q = 0, E_PI, E_WORD, E_OX, E_OY, E_OZ, E_DX, E_DY, E_DZ,
#ifdef RT_DEBUG_PHASE
       E_ITEM,      // instrumented build: the entry's position in the list
#endif
       E_COUNT };

template <int KIND, int NOBJ = 0, uint32_t SIG = 0>
RT_D void src_march_impl(const Params& P) {
    __shared__ uint32_t cur_all[4][E_COUNT][64];
    // the object table in LDS and an 8-entry exchange buffer per wave: the object-parallel evaluation of sparse waves (nearest_op3)
    __shared__ ObjFull lds_obj[MAX_OBJ];
    __shared__ float4 xch_all[4][8];
    stage_objects(P, lds_obj);
    const OpView OV = {lds_obj, ((P.src_op & 1) != 0 && P.n_obj <= 8) ? xch_all[threadIdx.x >> 6] : nullptr};
    const int lane = threadIdx.x & 63;
    uint32_t (*cur)[64] = cur_all[threadIdx.x >> 6];
    Lane L;
    L.state = ST_IDLE;
    L.n_steps = L.n_raycasts = L.n_hits = L.n_sky = 0;
    L.o = L.d = mk(0, 0, 0);
    L.t = L.w = L.s = L.dist = L.t_eval = 0.0f;
    L.idx = 0;
    L.steps_left = 0;
    uint32_t a_q = 0, a_pi = 0, a_word = 0, a_steps0 = 0;
    Trk Tk = {-1.0f, -1.0f, 0};
    constexpr bool TRK = KIND == KIND_BOXES || KIND == KIND_GENERIC;
    const bool trk_ok = TRK && P.cull_ok != 0 && P.src_track != 0;
    // the list's first n_heavy entries are the plan's heavy pixels: the waves that work them off track from the start
    const uint32_t n_heavy = (P.order && P.plan) ? P.plan->n_heavy : 0u;
    // DEALING.  The list is cut into groups of GS consecutive entries; group g belongs to team g % NT (every team the same
    // cost profile, its heaviest group first) and the waves of a team — the blocks b with b % NT equal: with NT = the CU count,
    // the blocks resident on one CU — take their team's groups from ONE counter of their own.  Neither of the two simpler
    // schemes works here: one global counter saturates (a 1080p launch is 32 k claims of 64, ~90 claims per microsecond: a third
    // of a millisecond), and static shares ignore that the issue arbiter serves the OLDEST wave of a SIMD first — measured: the
    // eight waves of a SIMD need 1.4 ... 9.8 kcycles per iteration, by age, and the launch ended when the youngest were done.
    // With a counter per team the fast waves simply take more; 32 waves per counter do not contend.
#ifndef RT_SPLIT_GS
#define RT_SPLIT_GS 32      // entries per group (<= 64); measured at 1080p: 16 / 32 / 64 = 0.540 / 0.526 / 0.533 ms per launch (wall)
#endif
    constexpr uint32_t GS = RT_SPLIT_GS;
    const uint32_t NT = (uint32_t)P.n_teams;
    const uint32_t team = blockIdx.x % NT;
    unsigned int* const tc = P.team_counter + team * 16u;                 // one counter per 64 bytes
    // HEAD DEALING (round 6).  The plan's heavy pixels — the head of the list, n_heavy entries: rays that graze the ground sphere or
    // sit in a wedge for hundreds of steps — used to fill the first groups completely: at 768x432 the 2 935 heaviest rays of the
    // frame went to 92 of the 2 048 waves, 32 each, every one tracking its own object, and those waves' tails (200 iterations of
    // tracked rounds for ~14 lanes) WERE the launch (per-wave records, tools/gpu_split_prof.py).  Now the head is INTERLEAVED: the
    // first n_hg groups carry HS head entries each in their first lanes and GS - HS entries of the rest of the list behind them.
    // Every wave's first claims still hold the heaviest rays there are (they start at once), but one or two per wave: in its
    // tail a wave is left with ITS long ray, which runs the lean loops alone, instead of a crowd that cannot.  Measured (one step
    // per launch, object-parallel evaluation on): 768x432 0.2375 -> 0.224 ms; 1080p 0.469 -> 0.478: there the bulk phase is four
    // waves per SIMD at 3 kcycles per iteration and a long ray that rides along takes ONE step per bulk iteration, where a head
    // wave's tracked forms took it further — so the host interleaves for the small frames only (option split_head, rt_capi.hip).
#ifndef RT_SPLIT_HS
#define RT_SPLIT_HS 1       // head entries per group
#endif
    constexpr uint32_t HS = RT_SPLIT_HS;
    const uint32_t n_head = P.split_head ? (n_heavy < P.total_items ? n_heavy : P.total_items) : 0u;
    const uint32_t n_hg = (n_head + HS - 1u) / HS;                       // groups that carry head entries
    const uint32_t bulk_hg = n_hg * (GS - HS);                           // entries of the rest of the list those groups take along
    const uint32_t rest = P.total_items - n_head > bulk_hg ? P.total_items - n_head - bulk_hg : 0u;
    const uint32_t n_groups = n_hg + (rest + GS - 1u) / GS;
    const int kwait = P.wait_lanes;
#ifdef RT_DEBUG_PHASE
    const unsigned long long t_wave0 = __builtin_readcyclecounter();
    unsigned dbg_iters = 0, dbg_iters_seq = 0, dbg_fast_calls = 0, dbg_fast_steps = 0, dbg_fast2_calls = 0, dbg_fast2_steps = 0, dbg_full2 = 0, dbg_trk = 0, dbg_plain = 0, dbg_tail_lanesteps = 0, dbg_op = 0;
    unsigned long long t_seq_done = 0;
    unsigned long long t_form[7] = {0, 0, 0, 0, 0, 0, 0};      // tail only: cycles inside the march step by form (0 = plain steps, 1..5 = tracked_iteration's forms)
    unsigned n_form[7] = {0, 0, 0, 0, 0, 0, 0};
    uint32_t a_item = 0;
    const uint32_t h = (blockIdx.x * 4u + (threadIdx.x >> 6));
#endif

    // Refills read LDS only: the next group is PREFETCHED while the current one is handed out (the loads of a refill are a
    // dependent chain — claim, list entry, march word and ray — and a wave refills ten times per launch: unhidden, those round
    // trips are as long as the marching itself).  `claimed` = the team counter's answer for the group after next (in flight),
    // pf_* = words and rays of the next group (in flight), LDS `cur` = the current group compacted to the entries that need a
    // raycast.
    auto claim = [&]() -> uint32_t {
        uint32_t j = 0;
        if (lane == 0) j = atomicAdd(tc, 1u);
        return j;
    };
    uint32_t pf_q = 0xffffffffu, pf_pi = 0, pf_word = MS_NONE;
    float2 pf_a = make_float2(0, 0), pf_b = make_float2(0, 0), pf_c = make_float2(0, 0);      // origin.xy, (origin.z, direction.x), direction.yz
    uint32_t pf_g = 0xffffffffu, cur_g = 0xffffffffu;      // group numbers (wave-uniform); ~0 = none
    bool more = true;                                      // the team's counter may still hand out groups
    // fetch group number `g` into pf (g = ~0: empty)
    auto load_group = [&](uint32_t g) {
        pf_g = g;
        pf_q = 0xffffffffu;
        pf_word = MS_NONE;
        // list entry of (group g, lane): head groups interleave, the others are GS consecutive entries behind them
        uint32_t item;
        bool have;
        if (g < n_hg) {
            const bool hd = (uint32_t)lane < HS;
            item = hd ? g * HS + (uint32_t)lane : n_head + g * (GS - HS) + ((uint32_t)lane - HS);
            have = hd ? item < n_head : item < P.total_items;
        } else {
            item = n_head + bulk_hg + (g - n_hg) * GS + (uint32_t)lane;
            have = item < P.total_items;
        }
        if (g != 0xffffffffu && (uint32_t)lane < GS && have) {
            const uint32_t q = P.order ? P.order[item] : item;
            int px, py;
            if (pixel_of(P, q, px, py)) {          // (a padding pixel of an edge tile has no ray: src_gen left its word at "none")
                pf_q = q;
                pf_pi = (uint32_t)px * (uint32_t)P.cfg.height + (uint32_t)py;
                pf_word = P.march_out[q];
                const float2* r = reinterpret_cast<const float2*>(P.ray_buffer + pf_pi);      // a ray record is 40 bytes, 8-byte aligned
                pf_a = r[0];
                pf_b = r[1];
                pf_c = r[2];
            }
        }
    };
    auto group_of = [&](uint32_t claimed_v) -> uint32_t {
        const uint32_t j = (uint32_t)__builtin_amdgcn_readfirstlane((int)claimed_v);
        const uint32_t g = j * NT + team;
        return g < n_groups ? g : 0xffffffffu;
    };
    uint32_t n_cur = 0, c_cur = 0;      // entries of the group in `cur`, entries handed out
    uint32_t claimed = claim();
    {
        const uint32_t g0 = group_of(claimed);
        more = g0 != 0xffffffffu;
        load_group(g0);
        claimed = more ? claim() : 0u;
    }
    auto advance_group = [&]() {
        // pf -> LDS, compacted; then start the next load and the claim after it
        const bool keep = (pf_word & 3u) == MS_MARCH;
        const unsigned long long km = __ballot(keep);
        if (keep) {
            const int r = wave_rank(km);
            cur[E_Q][r] = pf_q;
            cur[E_PI][r] = pf_pi;
            cur[E_WORD][r] = pf_word;
            cur[E_OX][r] = __builtin_bit_cast(uint32_t, pf_a.x);
            cur[E_OY][r] = __builtin_bit_cast(uint32_t, pf_a.y);
            cur[E_OZ][r] = __builtin_bit_cast(uint32_t, pf_b.x);
            cur[E_DX][r] = __builtin_bit_cast(uint32_t, pf_b.y);
            cur[E_DY][r] = __builtin_bit_cast(uint32_t, pf_c.x);
            cur[E_DZ][r] = __builtin_bit_cast(uint32_t, pf_c.y);
#ifdef RT_DEBUG_PHASE
            cur[E_ITEM][r] = pf_g * GS + (uint32_t)lane;      // (group-major position, not the list entry, since the head is interleaved)
#endif
        }
        lds_wave_fence();
        n_cur = (uint32_t)__popcll(km);
        c_cur = 0;
        cur_g = pf_g;
        uint32_t g = 0xffffffffu;
        if (more) {
            g = group_of(claimed);
            more = g != 0xffffffffu;
        }
        load_group(g);
        if (more) claimed = claim();
    };
    advance_group();

    for (;;) {
        // ================================================================ retire finished raycasts, refill the lanes
        {
            const bool done = L.state == ST_HIT || L.state == ST_MISS;
            if (done) {
                rtpbr_ray* rb = P.ray_buffer + a_pi;
                rb->origin[0] = L.o.x; rb->origin[1] = L.o.y; rb->origin[2] = L.o.z;
                P.march_out[a_q] = mw_pack(L.state == ST_HIT ? MS_HIT : MS_MISS, L.idx, a_word >> 8);
                // what the raycast cost (fire and forget; the plan kernels consume and clear it)
                if (P.cost_buffer) atomicAdd(&P.cost_buffer[a_q], L.n_steps - a_steps0);
#if RT_DEBUG_PHASE == 2
                {   // histogram of the raycast lengths: dbg[0..7] = <= 16, 32, 64, 128, 256, 511, = 512 (cap), and their step sum per bin in dbg[8..14]
                    const uint32_t n = L.n_steps - a_steps0;
                    const int b = n <= 16u ? 0 : n <= 32u ? 1 : n <= 64u ? 2 : n <= 128u ? 3 : n <= 256u ? 4 : n < (uint32_t)P.cfg.max_raymarch ? 5 : 6;
                    atomicAdd(&P.counters->dbg[b], 1ull);
                    atomicAdd(&P.counters->dbg[8 + b], (unsigned long long)n);
                    // where in the cost-ordered list the LONG raycasts (> 128 steps) sit: position bins 0-1k, -2k, -4k, ... (dbg[16..31]);
                    // raycasts > 256 steps likewise in the high word
                    if (n > 128u) {
                        int pb = 0;
                        for (uint32_t lim = 1024u; pb < 15 && a_item >= lim; lim <<= 1) pb++;
                        atomicAdd(&P.counters->dbg[16 + pb], 1ull + (n > 256u ? (1ull << 32) : 0ull));
                    }
                }
#endif
                L.state = ST_IDLE;
            }
            for (;;) {
                const bool want = L.state == ST_IDLE;
                const unsigned long long wm = __ballot(want);
                if (wm == 0ull) break;
                if (c_cur == n_cur) {
                    if (pf_g == 0xffffffffu) break;        // the team's list is exhausted
                    advance_group();
                    continue;
                }
                const uint32_t r = (uint32_t)wave_rank(wm);
                const uint32_t avail = n_cur - c_cur;
                if (want && r < avail) {
                    const uint32_t e = c_cur + r;
                    a_q = cur[E_Q][e];
                    a_pi = cur[E_PI][e];
                    a_word = cur[E_WORD][e];
                    L.o = mk(__builtin_bit_cast(float, cur[E_OX][e]), __builtin_bit_cast(float, cur[E_OY][e]), __builtin_bit_cast(float, cur[E_OZ][e]));
                    L.d = mk(__builtin_bit_cast(float, cur[E_DX][e]), __builtin_bit_cast(float, cur[E_DY][e]), __builtin_bit_cast(float, cur[E_DZ][e]));
                    a_steps0 = L.n_steps;
#ifdef RT_DEBUG_PHASE
                    a_item = cur[E_ITEM][e];
#endif
                    // start of raycast() src/scene.py:60-63
                    L.t = 0.0f;
                    L.w = P.cfg.omega0;
                    L.s = 0.0f;
                    L.dist = P.cfg.max_dis;
                    L.steps_left = P.cfg.max_raymarch;
                    L.state = ST_MARCH;
                    L.n_raycasts++;
                    Tk.lb2 = Tk.lb3 = -1.0f;
                }
                const uint32_t need = (uint32_t)__popcll(wm);
                c_cur += need < avail ? need : avail;
                lds_wave_fence();        // (the entries are read before a later advance_group overwrites them)
            }
        }
        const bool seq_done = c_cur == n_cur && pf_g == 0xffffffffu;
#ifdef RT_DEBUG_PHASE
        if (seq_done && t_seq_done == 0) { t_seq_done = __builtin_readcyclecounter(); dbg_iters_seq = dbg_iters; }
#endif
        if (L.state == ST_IDLE && seq_done) L.state = ST_EXHAUSTED;
        // ================================================================ march
        int n_march = __popcll(__ballot(L.state == ST_MARCH));
        if (n_march == 0) {
            if (__ballot(L.state != ST_EXHAUSTED) == 0) break;
            continue;
        }
        // leave for a refill when kwait lanes are free — a quarter of the live lanes once the list has run out
        const int n_active = __popcll(__ballot(L.state != ST_EXHAUSTED));
        const int cap = n_active >> 2 > 1 ? n_active >> 2 : 1;
        const int kstar = (seq_done && kwait > cap) ? cap : kwait;
        bool tracked = false;
        // (with the head interleaved no wave is made of heavy rays: the tracked forms are for sparse phases; split_head = 0 is the
        // round-5 dealing — head entries fill the first groups, whose waves track from the start)
        if constexpr (TRK) tracked = trk_ok && ((!P.split_head && cur_g != 0xffffffffu && cur_g * GS <= n_heavy + GS) || n_march <= P.sparse_lanes);
        if (tracked) {
            if constexpr (TRK) {
                do {
#ifdef RT_DEBUG_PHASE
                    const uint32_t steps_before = L.n_steps;
                    const unsigned long long t_f0 = __builtin_readcyclecounter();
#endif
                    int it = 1;
                    const int form = tracked_iteration<KIND, NOBJ, SIG, true>(P, L, Tk, n_march, 0, it, nullptr, nullptr, OV);
                    (void)form;
#ifdef RT_DEBUG_PHASE
                    if (t_seq_done) { t_form[form] += __builtin_readcyclecounter() - t_f0; n_form[form]++; }
                    if (form == 1) { dbg_fast_calls++; dbg_fast_steps += (unsigned)it; }
                    else if (form == 2) { dbg_fast2_calls++; dbg_fast2_steps += (unsigned)it; }
                    else if (form == 3) dbg_trk++;
                    else if (form == 5 || form == 6) dbg_op++;
                    else dbg_full2++;
                    if (t_seq_done) dbg_tail_lanesteps += wave_sum(L.n_steps - steps_before);
                    dbg_iters++;
#endif
                    n_march = __popcll(__ballot(L.state == ST_MARCH));
                } while (n_march > 0 && (n_active - n_march) < kstar);
            }
        } else {
            do {
#ifdef RT_DEBUG_PHASE
                const unsigned long long t_f0 = __builtin_readcyclecounter();
#endif
                if (L.state == ST_MARCH) march_step_src<KIND, NOBJ, SIG>(P, L);
#ifdef RT_DEBUG_PHASE
                if (t_seq_done) { t_form[0] += __builtin_readcyclecounter() - t_f0; n_form[0]++; }
                dbg_iters++;
                dbg_plain++;
                if (t_seq_done) dbg_tail_lanesteps += (unsigned)n_march;
#endif
                n_march = __popcll(__ballot(L.state == ST_MARCH));
            } while (n_march > 0 && (n_active - n_march) < kstar);
            Tk.lb2 = Tk.lb3 = -1.0f;     // the plain steps did not maintain the bounds
        }
    }
#ifdef RT_DEBUG_PHASE
    if (lane == 0) {   // per-wave timeline, written over diff_buffer (unused without adaptive sampling; the host reads it back)
        unsigned long long* w = reinterpret_cast<unsigned long long*>(P.diff_buffer) + (size_t)h * 16u;
        for (int f = 0; f < 7; f++) w[8 + f] = t_form[f] | ((unsigned long long)n_form[f] << 40);      // tail: cycles | calls << 40, by form
        w[0] = t_wave0;
        w[1] = t_seq_done;
        w[2] = __builtin_readcyclecounter();
        w[3] = (unsigned long long)dbg_iters | ((unsigned long long)dbg_iters_seq << 32);
        w[4] = (unsigned long long)dbg_fast_calls | ((unsigned long long)dbg_fast_steps << 32);
        w[5] = (unsigned long long)(dbg_full2 & 0xffffu) | ((unsigned long long)(dbg_op & 0xffffu) << 16) | ((unsigned long long)dbg_trk << 32);      // full evaluations | object-parallel ones << 16 | tracked rounds << 32
        w[6] = (unsigned long long)dbg_plain | ((unsigned long long)dbg_tail_lanesteps << 32);
        w[7] = (unsigned long long)dbg_fast2_calls | ((unsigned long long)dbg_fast2_steps << 32);
    }
#endif
    flush_counters(P, L.n_steps, L.n_raycasts, 0, 0, 0, 0);
}

template <int KIND>
RT_D void src_shade_impl(const Params& P) {
    __shared__ ObjFull lds_obj[MAX_OBJ];
    stage_objects(P, lds_obj);
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_hits = 0, n_sky = 0, n_samples = 0;
    const uint32_t word = q < (uint32_t)P.np ? P.march_out[q] : 0u;
    const uint32_t st = word & 3u;
    if (st == MS_HIT || st == MS_MISS) {
        int px, py;
        pixel_of(P, q, px, py);
        const size_t pi = (size_t)px * P.cfg.height + py;
        rtpbr_ray rb = P.ray_buffer[pi];
        src_shade_core<KIND>(P, lds_obj, px, py, word, P.sample_base, rb, n_hits, n_sky, n_samples);
        P.ray_buffer[pi] = rb;
    }
    flush_counters(P, 0, 0, n_hits, n_sky, n_samples, 0);
}

template <int KIND>
__global__ void __launch_bounds__(256) src_gen(const Params P) { src_gen_impl<KIND>(P); }
template <int KIND, bool COUNT>
__global__ void __launch_bounds__(256) src_shade_gen(const Params P) { src_gen_impl<KIND, true, COUNT>(P); }
template <int KIND>
__global__ void __launch_bounds__(256) src_march(const Params P) { src_march_impl<KIND>(P); }
template <int KIND>
__global__ void __launch_bounds__(256) src_shade(const Params P) { src_shade_impl<KIND>(P); }

}  // namespace rt
