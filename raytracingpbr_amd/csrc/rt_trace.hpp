// rt_trace.hpp — the complete-path kernels (scheduler 0, primary_rays, the LDS ray pool) as templates; the src/ persistent-ray
// kernels that share the pool live in rt_persistent.hpp (included at the end).
// Included by rt_kernels.hip (ahead-of-time instances + launchers) and by rt_jit_tu.hip (the translation unit that
// rtpbr compiles at run time for one scene: object count, shape types and rotation classes as compile-time constants).
#pragma once
#include <hip/hip_runtime.h>

#include "rt_device.hpp"

namespace rt {

RT_D uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Work counters: wave sum -> block sum in LDS -> ONE atomic per counter per block (a persistent
// grid has thousands of waves; per-wave atomics on the same six addresses serialise for ~0.5 ms).
// Every wave of the block must call it exactly once (they all do, at the end of the kernel).
RT_D void flush_counters(const Params& P, uint32_t steps, uint32_t raycasts, uint32_t hits, uint32_t sky,
                         uint32_t samples, uint32_t deposits) {
    __shared__ unsigned long long blk[6];
    if (threadIdx.x < 6) blk[threadIdx.x] = 0;
    __syncthreads();
    steps = wave_sum(steps);
    raycasts = wave_sum(raycasts);
    hits = wave_sum(hits);
    sky = wave_sum(sky);
    samples = wave_sum(samples);
    deposits = wave_sum(deposits);
    if ((threadIdx.x & 63) == 0) {
        if (steps) atomicAdd(&blk[0], (unsigned long long)steps);
        if (raycasts) atomicAdd(&blk[1], (unsigned long long)raycasts);
        if (hits) atomicAdd(&blk[2], (unsigned long long)hits);
        if (sky) atomicAdd(&blk[3], (unsigned long long)sky);
        if (samples) atomicAdd(&blk[4], (unsigned long long)samples);
        if (deposits) atomicAdd(&blk[5], (unsigned long long)deposits);
    }
    __syncthreads();
    if (threadIdx.x < 6 && blk[threadIdx.x] != 0) atomicAdd(&P.counters->shard[blockIdx.x & 63u][threadIdx.x], blk[threadIdx.x]);
}

// Stage the per-lane-indexed object table (T4: transform + material) in LDS.
// (the src/ kernels only — persistent pool, persistent steps, src_gen: their launches are the 0.2 ms ones; in the complete-path pool kernel
// the extra pointer cost the ahead-of-time instances 2 % — 3 505 -> 3 430 Msamples/s — for 4.5 us per launch: those calls fill with launch_zero)
RT_D void zero_next_counters(const Params& P) {
    if (P.counters_next != nullptr && blockIdx.x == 0) {
        uint4* p = reinterpret_cast<uint4*>(P.counters_next);
        for (int i = threadIdx.x; i < (int)((sizeof(Counters) + 64) / 16); i += 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}
RT_D void stage_objects(const Params& P, ObjFull* lds_obj) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(P.objfull);
    uint32_t* dst = reinterpret_cast<uint32_t*>(lds_obj);
    const int nw = P.n_obj * (int)(sizeof(ObjFull) / 4);
    for (int i = threadIdx.x; i < nw; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

// -------------------------------------------------------------------------------------------
// Pieces of the bounce loop shared by both schedulers (cornell_box_v3/pathtracer.py:81-106).
struct PathRay {
    vec3 o, d, col;
    float t_eval;      // record.position = o + t_eval*d
    int idx;           // record.object
    int bounce;        // i of "for i in range(MAX_RAYTRACE)"
    uint32_t key, cnt; // RNG stream
    uint32_t item;     // work item = q*K + k
};

// after a hit: surface interaction, emission / stop test, next-bounce roulette (:84-89,:97-104).
// returns true if the path continues with another raycast.
template <int KIND, bool HAVE_NORMAL = false>
RT_D bool shade_hit(const Params& P, const ObjFull* lds_obj, PathRay& R, vec3 given_normal = vec3{0, 0, 0}) {
    const ObjFull& o = lds_obj[R.idx];   // read field by field where it is used (see surface_interaction)
    vec3 pos = fma3(R.t_eval, R.d, R.o);
    surface_interaction<KIND, HAVE_NORMAL>(P, o, pos, R.o, R.d, R.col, R.key, R.cnt, given_normal);
    float intensity = brightness(R.col);
    R.col = R.col * mk(o.emission[0], o.emission[1], o.emission[2]);
    float visible = brightness(R.col);
    bool stop = intensity < visible || visible < P.cfg.vis_lo || visible > P.cfg.vis_hi;
    if (stop) return false;
    R.bounce++;
    if (R.bounce >= P.cfg.max_raytrace) return false;  // falls out of the loop keeping the throughput (G4)
    float inv_pdf = exp_((float)R.bounce / P.cfg.light_quality);
    float p = 1.0f - 1.0f / inv_pdf;
    if (rng_next(R.key, R.cnt) < p) {
        R.col = R.col * p;
        return false;
    }
    return true;
}

// after a miss (:93-95; sky variants tokyo_ibl.py:352-354, bunny_sdf.py:351-354, bunny_sdf_v2.py:355-360)
RT_D void shade_miss(const Params& P, PathRay& R, uint32_t& n_sky) {
    if (P.cfg.sky_kind == RTPBR_SKY_BLACK) {
        R.col = mk(0, 0, 0);
    } else if (R.bounce == 0 && P.cfg.primary_miss == RTPBR_PRIMARY_BLACK) {
        R.col = mk(0, 0, 0);
    } else if (R.bounce == 0 && P.cfg.primary_miss == RTPBR_PRIMARY_WHITE) {
    } else {
        R.col = R.col * sky_color(P, R.d);
        n_sky++;
    }
}

// Staging layout [q][k] = item-linear: the items a wave has in flight are (nearly) consecutive,
// so its stores fall into the same few cache lines and merge in L2 before they reach HBM.  A record is the sample's
// three colour words, 12 bytes (round 3; it was a float4 whose fourth word nobody read: the count a sample adds is 1 by
// definition and padding pixels of an edge tile are known from the geometry, so they are not written at all).
// (NOT a non-temporal store: the records of neighbouring samples share 64-byte lines and finish up to a path's lifetime apart —
// the L2 is what merges them; measured with `nt` stores: WRITE_SIZE of the trace kernel 12.5 -> 15.5 GB per headline step)
RT_D void write_sample(const Params& P, uint32_t item, vec3 col) {
    reinterpret_cast<StageRec*>(P.stage)[item] = StageRec{col.x, col.y, col.z};
}
// Dense staging (round 6; pool kernel, wave-uniform call).  The item-linear layout leaves the 6.4 GB of records of a headline
// step as 11.1 GB of HBM writes: the records of one 128-byte line finish up to a path's lifetime apart and the line leaves
// the L2 in between.  Here the lanes that finished in this pass append their records to the region of the chunk the item was claimed
// with — a chunk belongs to ONE wave, so its fill count is that wave's to read-modify-write — in completion order, one contiguous
// run per chunk and pass, and note which sample each is (one byte: chunk <= 256).  accumulate_dense undoes the permutation.
// Compiled in by -DRT_STAGE_DENSE=1 (run-time instances, option stage_dense): merely present, the loop costs the ahead-of-time and the
// unbaked instances 5 % (registers), so it is not.
#ifndef RT_STAGE_DENSE
#define RT_STAGE_DENSE 0
#endif
RT_D void stage_sample(const Params& P, bool fin, uint32_t item, vec3 col, int lane) {
#if !RT_STAGE_DENSE
    if (fin) write_sample(P, item, col);
#else
    if (!P.stage_dense) {
        if (fin) write_sample(P, item, col);
        return;
    }
    unsigned long long m = __ballot(fin);
    const uint32_t cid = item / P.chunk;
    while (m) {
        const int first = __ffsll((long long)m) - 1;
        const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cid, first);
        const bool mine = fin && cid == c;
        const unsigned long long mm = __ballot(mine);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm, 0));
        uint32_t base = 0;
        // (an atomic: the count lives in L2, whichever lane of the wave touched it last)
        if (lane == first) base = atomicAdd(&P.stage_fill[c], (uint32_t)__popcll(mm));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
        if (mine) {
            const uint32_t c0 = c * P.chunk;
            const uint32_t slot = c0 + base + rank;
            reinterpret_cast<StageRec*>(P.stage)[slot] = StageRec{col.x, col.y, col.z};
            P.stage_idx[slot] = (uint8_t)(item - c0);
        }
        m &= ~mm;
    }
#endif
}

RT_D void lds_wave_fence();
// TOLERANCE FLAVOUR (RT_FAST_MATH): no staging.  The exact kernels stage one record per sample because image_buffer must be
// summed in SAMPLE ORDER to stay bit-identical with `image_buffer[i,j] += vec4(color,1)` (samples finish out of order); a
// flavour that is held to a tolerance needs no order.  Every wave keeps 16 per-pixel accumulators in LDS (direct-mapped by
// local pixel; a wave's samples in flight belong to a handful of neighbouring pixels: consecutive work items are consecutive
// samples of one pixel), finished samples are added there with LDS float atomics, and an accumulator goes to image_buffer —
// four f32 atomics — when its slot is taken by another pixel or the kernel ends: 16 bytes per pixel and wave instead of 24
// bytes per SAMPLE, and no accumulate kernel.  The order of the additions depends on the schedule: results agree to rounding
// from run to run, not bit for bit (the flavour promises a tolerance).
constexpr int ACC_SLOTS = 16;
struct AccView {
    float (*acc)[4];       // [slot][r, g, b, count]
    uint32_t* tag;         // local pixel held by the slot, ~0 = empty
};
RT_D void acc_flush_slot(const Params& P, const AccView& A, uint32_t s, uint32_t q, int lane, uint32_t& n_dep) {
    int px, py;
    if (pixel_of(P, q, px, py) && lane < 4) {
        float* dst = reinterpret_cast<float*>(P.image_buffer + ((size_t)px * P.cfg.height + py)) + lane;
        atomicAdd(dst, A.acc[s][lane]);
    }
    if (lane == 0) n_dep += (uint32_t)A.acc[s][3];
}
// all lanes call it (wave-uniform control flow); `fin`: this lane holds a finished sample of local pixel q
RT_D void acc_add(const Params& P, const AccView& A, bool fin, uint32_t q, vec3 col, int lane, uint32_t& n_dep) {
    unsigned long long m = __ballot(fin);
    while (m) {
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_readlane((int)q, (int)__builtin_ctzll(m));
        const bool mine = fin && q == q0;
        const uint32_t s = q0 & (uint32_t)(ACC_SLOTS - 1);
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.tag[s]);
        if (t != q0) {
            if (t != 0xffffffffu) acc_flush_slot(P, A, s, t, lane, n_dep);
            lds_wave_fence();
            if (lane < 4) A.acc[s][lane] = 0.0f;
            if (lane == 0) A.tag[s] = q0;
            lds_wave_fence();
        }
        if (mine) {
            atomicAdd(&A.acc[s][0], col.x);
            atomicAdd(&A.acc[s][1], col.y);
            atomicAdd(&A.acc[s][2], col.z);
            atomicAdd(&A.acc[s][3], 1.0f);
        }
        lds_wave_fence();
        m &= ~__ballot(mine);
    }
}
RT_D void acc_flush_all(const Params& P, const AccView& A, int lane, uint32_t& n_dep) {
    lds_wave_fence();
    for (uint32_t s = 0; s < (uint32_t)ACC_SLOTS; s++) {
        const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.tag[s]);
        if (t != 0xffffffffu) acc_flush_slot(P, A, s, t, lane, n_dep);
    }
}

// renderer.py:32-35: jitter, get_ray, color = 1, then roulette at i = 0 (p = 0, the draw is consumed).
// returns 1 = ray ready to march, 0 = finished already, -1 = padding pixel of an edge tile (nothing to trace)
RT_D int start_item(const Params& P, PathRay& R) {
    uint32_t q = R.item / (uint32_t)P.K;
    uint32_t k = R.item - q * (uint32_t)P.K;
    int px, py;
    if (!pixel_of(P, q, px, py)) return -1;
    R.key = rng_key(P.cfg.seed, (uint32_t)px, (uint32_t)py, P.sample_base + k);
    R.cnt = 0;
    gen_ray(P, px, py, R.key, R.cnt, R.o, R.d);
    R.col = mk(1, 1, 1);
    R.bounce = 0;
    float inv_pdf = exp_(0.0f / P.cfg.light_quality);
    float p = 1.0f - 1.0f / inv_pdf;
    if (rng_next(R.key, R.cnt) < p) {
        R.col = R.col * p;
        return 0;
    }
    return 1;
}

// Wave-uniform work range; items are claimed in chunks from one global atomic.
struct WorkRange {
    uint32_t next, end;
    bool drained;
};

// Hand fresh work items to the lanes with `want` set: ballot + mbcnt prefix rank into the wave's
// range (wave-uniform control flow).  Returns true for the lanes that received an item.
RT_D bool claim_items(const Params& P, WorkRange& wr, bool want, int lane, uint32_t& item) {
    const unsigned long long m = __ballot(want);
    int need = __popcll(m);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
    int assigned = 0;
    bool got = false;
    while (need > 0) {
        if (wr.next == wr.end) {
            if (wr.drained) break;
            uint32_t start = 0;
            if (lane == 0) start = atomicAdd(P.work_counter, P.chunk);
            start = __builtin_amdgcn_readfirstlane(start);
            if (start >= P.total_items) {
                wr.drained = true;
                break;
            }
            wr.next = start;
            uint32_t e = start + P.chunk;
            wr.end = e < P.total_items ? e : P.total_items;
        }
        int avail = (int)(wr.end - wr.next);
        int take = need < avail ? need : avail;
        if (want && !got && rank >= assigned && rank < assigned + take) {
            item = wr.next + (uint32_t)(rank - assigned);
            got = true;
        }
        wr.next += (uint32_t)take;
        assigned += take;
        need -= take;
    }
    return got;
}

// -------------------------------------------------------------------------------------------
// Scheduler 0: in-register refill (no LDS ray pool).
template <int KIND, int NOBJ, uint32_t SIG = 0>
__global__ void __launch_bounds__(256) trace_paths(const Params P) {
    __shared__ ObjFull lds_obj[MAX_OBJ];
    stage_objects(P, lds_obj);

    const int lane = threadIdx.x & 63;
    Lane L;
    L.state = ST_IDLE;
    L.n_steps = L.n_raycasts = L.n_hits = L.n_sky = 0;
    L.o = L.d = mk(0, 0, 0);
    L.t = L.w = L.s = L.dist = L.t_eval = 0.0f;
    L.idx = 0;
    L.steps_left = 0;
    PathRay R;
    R.o = R.d = R.col = mk(0, 0, 0);
    R.t_eval = 0.0f;
    R.idx = R.bounce = 0;
    R.key = R.cnt = R.item = 0;
    uint32_t n_samples = 0;
    WorkRange wr = {0, 0, false};

    for (;;) {
        // ================================================================ phase B
        bool finished = false;
        bool alive = false;
        if (L.state == ST_HIT) {
            R.o = L.o; R.d = L.d; R.t_eval = L.t_eval; R.idx = L.idx;
            alive = shade_hit<KIND>(P, lds_obj, R);
            L.n_hits++;
            finished = !alive;
        } else if (L.state == ST_MISS) {
            R.d = L.d;
            shade_miss(P, R, L.n_sky);
            finished = true;
        }
        if (finished) {
            write_sample(P, R.item, R.col);
            n_samples++;
            L.state = ST_IDLE;
        }
        // ---- refill idle lanes with fresh pixel-samples
        {
            bool got = claim_items(P, wr, L.state == ST_IDLE, lane, R.item);
            if (L.state == ST_IDLE) {
                if (got) {
                    int r = start_item(P, R);
                    if (r == 1) alive = true;
                    else if (r == 0) { write_sample(P, R.item, R.col); n_samples++; }
                    // (r < 0: padding pixel of an edge tile, nothing to record: the lane stays idle until the next refill)
                } else if (wr.drained) {
                    L.state = ST_EXHAUSTED;
                }
            }
        }
        if (alive) {
            L.o = R.o; L.d = R.d;
            march_init(P, L);
        }

        // ================================================================ phase A
        unsigned long long marching = __ballot(L.state == ST_MARCH);
        if (marching == 0) {
            if (__ballot(L.state != ST_EXHAUSTED) == 0) break;
            continue;
        }
        const int n_active = __popcll(__ballot(L.state != ST_EXHAUSTED));
        int kstar = P.wait_lanes;
        const int cap = n_active >> 2 > 1 ? n_active >> 2 : 1;
        kstar = (wr.drained && kstar > cap) ? cap : kstar;
        int n_march;
        do {
            if (L.state == ST_MARCH) march_step<KIND, NOBJ, SIG>(P, L);
            n_march = __popcll(__ballot(L.state == ST_MARCH));
        } while (n_march > 0 && (n_active - n_march) < kstar);
    }
    flush_counters(P, L.n_steps, L.n_raycasts, L.n_hits, L.n_sky, n_samples, 0);
}

// -------------------------------------------------------------------------------------------
// Primary raycasts in their own kernel (option primary_split).  A wave takes 64 CONSECUTIVE work
// items = consecutive samples of one pixel, so its 64 camera rays differ only by the sub-pixel
// jitter and the lens offset: they march in lock step (no pool needed, nearly no divergence) and
// the object loop can be culled at wave level (nearest_culled).  The result of the raycast —
// {t_eval, nearest index, hit/miss} — is written per item and the pool kernel resumes the path
// from there (it regenerates the camera ray from the same RNG stream: cheaper than moving 24 more
// bytes per sample).  Same arithmetic as the pool kernel's march, bit-identical results.
// The ONE-OBJECT loop of the coherent primary march (round 5).  Measured (tools/gpu_dbg_primary.py): after wave-level culling
// 43 % of the wave-steps of the Cornell headline (38 % of the Tokyo frame's) evaluate exactly ONE object — the 64 camera rays of
// a pixel fly towards the same wall — and still pay the eight skip tests and eight bound updates of the culled step (~170
// instructions against ~65 for the object itself plus the raycast bookkeeping).  When a culled step has evaluated only object K
// for the whole wave, the march continues HERE: one bound for all other objects (lbo = the smallest of their lower bounds,
// decayed by the path marched), one wave vote per step — the same criterion nearest_culled applies per object, on the minimum —
// object K evaluated exactly as nearest_culled evaluates it, so (index, distance) and everything downstream are bit for bit
// the same.  The loop ends when some lane can no longer exclude the others (nothing has been stepped then) or all rays are done;
// the per-object bounds are brought up to date on the way out (the others by the path marched inside, rounded UP).
template <int KIND, int NOBJ, uint32_t SIG, int K>
RT_D void primary_lean_obj(const Params& P, Lane& L, float& ub, float (&lb)[NOBJ > 0 ? NOBJ : 1], uint32_t& n_iter) {
    ObjTab tab = obj_table();
    float lbo = 3.0e38f;
#pragma unroll
    for (int j = 0; j < NOBJ; j++)
        if (j != K && (SIG != 0 || j < P.n_obj)) lbo = fmin_(lbo, lb[j]);
    float acc = 0.0f, lbk = lb[K];
    for (;;) {
        asm volatile("" : "+s"(tab));
        const bool active = L.state == ST_MARCH;
        const unsigned long long act_mask = __builtin_amdgcn_ballot_w64(active);
        if (act_mask == 0ull) break;
        const float eps = 1.9073486328125e-06f * (fabs_(L.t) + P.cull_extent);
        const float bound = ub + eps;
        // some lane cannot exclude every other object: back to the culled step (13 = "unordered or <=", i.e. !(lbo > bound))
        if ((__builtin_amdgcn_fcmpf(lbo, bound, 13) & act_mask) != 0ull) break;
        const vec3 pos = fma3(L.t, L.d, L.o);
        const float t_before = L.t;
        const ObjM o = load_obj<SIG, K>(tab);
        const float d = fabs_(signed_distance<KIND>(P, o, pos, RT_SIG_CLS(K), jit_type(K)));
        // what nearest_culled returns when K is the only object it visits: K initialises the search (nearest_init = 0) or has
        // to beat MAX_DIS (nearest_init = 1)
        const bool take = !P.cfg.nearest_init | (d < P.cfg.max_dis);
        if (active) {
            L.t_eval = L.t;
            march_update(P, L, take ? K : 0, take ? d : P.cfg.max_dis);
        }
        const float moved = fabs_(L.t - t_before) * 1.000001f;
        ub = (take ? d : P.cfg.max_dis) + moved;
        lbo -= moved;
        acc += moved;
        lbk = (d - eps) - moved;
        n_iter++;
    }
    // bounds of the culled step, as it would have left them (the others: minus the whole path, rounded up — a lower bound stays one)
    const float dec = acc * 1.000002f;
#pragma unroll
    for (int j = 0; j < NOBJ; j++) lb[j] = j == K ? lbk : lb[j] - dec;
}

// The wave's marching lanes marched TO THE END OF THEIR RAYCASTS with wave-level culling: every lane keeps a lower bound per
// object (last exact distance minus the path marched since) and an upper bound of its minimum; an object no marching lane can
// need is skipped (nearest_culled, exact), and while the whole wave needs one object only the march goes on in that object's
// lean loop (primary_lean_obj).  Used by the coherent primary kernel (64 camera rays of one pixel) and by the pool kernel's
// DRAIN (the work has run out, a handful of lanes still march: with so few rays nearly every object can be excluded).
// dbg_hist (measurement build): objects evaluated per wave-step, bin 0 = steps inside lean loops.
template <int KIND, int NOBJ, uint32_t SIG>
RT_D void culled_march_wave(const Params& P, Lane& L, uint32_t* dbg_hist) {
    float lb[NOBJ > 0 ? NOBJ : 1];
#pragma unroll
    for (int i = 0; i < (NOBJ > 0 ? NOBJ : 1); i++) lb[i] = -1.0f;   // nothing known yet: everything is evaluated
    float ub = 3.0e38f;
    while (__any(L.state == ST_MARCH)) {
        const bool active = L.state == ST_MARCH;
        vec3 pos = fma3(L.t, L.d, L.o);
        const float t_before = L.t;
        int idx;
        float dist;
        // all lanes run the (wave-uniform) object loop; finished lanes just do not commit
        uint32_t ev_mask = 0;
        uint32_t dbg_ev = 0;
        nearest_culled<KIND, NOBJ, SIG>(P, pos, L.t, active, ub, lb, idx, dist, dbg_hist ? &dbg_ev : nullptr, &ev_mask);
        if (dbg_hist) dbg_hist[dbg_ev < 8u ? dbg_ev : 8u]++;
        if (active) {
            L.t_eval = L.t;
            march_update(P, L, idx, dist);
        }
        // the next evaluation point is |dt| * |d| away; |d| <= 1 + 2^-20
        const float moved = fabs_(L.t - t_before) * 1.000001f;
        ub = dist + moved;
#pragma unroll
        for (int i = 0; i < (NOBJ > 0 ? NOBJ : 1); i++) lb[i] -= moved;
        // the wave needed exactly one object: go on in its lean loop (primary_lean_obj above)
        if (P.primary_lean && (ev_mask & (ev_mask - 1u)) == 0u && ev_mask != 0u) {
            const int k = (int)__builtin_ctz(ev_mask);
            uint32_t lean_iters = 0;
            static_for<(NOBJ > 0 ? NOBJ : 1), 1>([&](auto Ic) {
                constexpr int i = decltype(Ic)::value;
                if (k == i) primary_lean_obj<KIND, NOBJ, SIG, i>(P, L, ub, lb, lean_iters);
            });
            if (dbg_hist) dbg_hist[0] += lean_iters;      // (bin 0 is otherwise empty: wave-steps taken inside the lean loop)
        }
    }
}

template <int KIND, int NOBJ, uint32_t SIG = 0, bool CULL = true>
RT_D void primary_rays_impl(const Params& P) {
    const int lane = threadIdx.x & 63;
    const uint32_t n_groups = (P.total_items + 63u) / 64u;
    const uint32_t n_waves = gridDim.x * 4u;
    uint32_t n_steps = 0, n_raycasts = 0;
#ifdef RT_DEBUG_PRIMARY
    uint32_t dbg_hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // persistent waves (a few thousand resident blocks; one block per 256 items would be
    // dispatch-bound: 2 M blocks of ~10 us each) that CLAIM runs of up to 16 consecutive groups of 64
    // items from a counter: a static stride leaves the slowest wave's tail exposed (average wave
    // lifetime was 60 % of the kernel)
    // run length: 16 groups when there is plenty of work, fewer when that would leave waves idle
    uint32_t RUN = n_groups / (n_waves * 4u);
    RUN = RUN < 1u ? 1u : (RUN > 16u ? 16u : RUN);
    uint32_t g = 0, g_end = 0;
    for (;;) {
        if (g == g_end) {
            uint32_t start = 0;
            if (lane == 0) start = atomicAdd(P.work_counter + 1, RUN);
            g = __builtin_amdgcn_readfirstlane(start);
            if (g >= n_groups) break;
            g_end = g + RUN < n_groups ? g + RUN : n_groups;
        }
        const uint32_t item = g * 64u + (uint32_t)lane;
        g++;
        Lane L;
        L.state = ST_IDLE;
        L.n_steps = L.n_raycasts = L.n_hits = L.n_sky = 0;
        L.o = L.d = mk(0, 0, 0);
        L.t = L.w = L.s = L.dist = L.t_eval = 0.0f;
        L.idx = 0;
        L.steps_left = 0;
        PathRay R;
        R.item = item;
        bool valid = false;
        if (item < P.total_items) {
            int r = start_item(P, R);
            valid = r == 1;
            if (valid) {
                L.o = R.o;
                L.d = R.d;
                march_init(P, L);
            }
        }
        if constexpr (CULL && NOBJ > 0 && KIND != KIND_BUNNY && KIND != KIND_MIXED) {
#ifdef RT_DEBUG_PRIMARY
            culled_march_wave<KIND, NOBJ, SIG>(P, L, dbg_hist);
#else
            culled_march_wave<KIND, NOBJ, SIG>(P, L, nullptr);
#endif
        } else {
            while (__any(L.state == ST_MARCH)) {
                if (L.state == ST_MARCH) march_step<KIND, NOBJ, SIG>(P, L);
            }
        }
        if (item < P.total_items) {
            P.primary[item] = L.t_eval;
            P.primary_code[item] = (uint8_t)((uint32_t)L.idx | ((uint32_t)(valid ? L.state : ST_IDLE) << 5));
        }
        n_steps += L.n_steps;
        n_raycasts += L.n_raycasts;
    }
#ifdef RT_DEBUG_PRIMARY
    if (lane == 0)
        for (int i = 0; i < 9; i++) atomicAdd(&P.counters->dbg[i], (unsigned long long)dbg_hist[i]);
#endif
    flush_counters(P, n_steps, n_raycasts, 0, 0, 0, 0);
}

template <int KIND, int NOBJ, uint32_t SIG = 0>
__global__ void __launch_bounds__(256) primary_rays(const Params P) { primary_rays_impl<KIND, NOBJ, SIG>(P); }

// -------------------------------------------------------------------------------------------
// Shared by both pool kernels: the wave-private LDS ray pool and the rank-matched swap.
enum { SL_EMPTY = 0, SL_READY = 1, SL_HIT = 2, SL_MISS = 3 };
constexpr int POOL_WORDS = 15;   // dwords per parked record; what they mean is the kernel's business

struct PoolView {
    uint32_t (*pool)[64];   // [word][slot]: lane s <-> slot s accesses are conflict-free
    uint32_t* sstate;       // slot state, SL_*
    uint32_t* tbl;          // rank -> slot table for the matching
};

// Wave-private LDS data passes between lanes of ONE wave (rank table, slot states, pool records).  The
// hardware executes a wave's LDS instructions in order; this fence pins the same order for the compiler
// (no reordering of may-alias LDS accesses across it) and costs no instruction.
RT_D void lds_wave_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
RT_D int wave_rank(unsigned long long m) {   // number of set bits below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}
template <int W>
RT_D void pool_load(const PoolView& V, uint32_t slot, uint32_t (&rec)[W]) {
#pragma unroll
    for (int w = 0; w < W; w++) rec[w] = V.pool[w][slot];
}
template <int W>
RT_D void pool_store(const PoolView& V, uint32_t slot, const uint32_t (&rec)[W]) {
#pragma unroll
    for (int w = 0; w < W; w++) V.pool[w][slot] = rec[w];
}

// Swap finished lanes with parked READY records (wave-uniform control flow; all lanes call it).
//   is_done: this lane holds a finished record in rec[] that must be parked as `done_state`;
//   is_idle: this lane holds nothing and wants a READY record.
// Finished lanes are served first (they need ANY non-occupied slot: READY ones are swapped, free
// ones just receive the record), then idle lanes take the remaining READY records.  The j-th
// requester gets the j-th listed slot: both sides are ranked with ballot + mbcnt prefix counts and
// matched through the 64-entry table.  Returns bit 0 = rec[] now holds a taken READY record,
// bit 1 = this lane's record was parked.  m_ready / m_shade are refreshed from the slot states.
template <int W>
RT_D int pool_swap(const PoolView& V, int lane, bool is_done, bool is_idle, uint32_t done_state,
                   uint32_t (&rec)[W], unsigned long long& m_ready, unsigned long long& m_shade) {
    const unsigned long long done = __ballot(is_done);
    const unsigned long long idle = __ballot(is_idle);
    const int n_done = __popcll(done);
    const int n_ready = __popcll(m_ready);
    if (!(n_done > 0 || (idle != 0 && n_ready > 0))) return 0;
    const unsigned long long m_free = ~(m_ready | m_shade);
    const int n_free = __popcll(m_free);
    // slot side: READY slots first, then free slots, listed by rank
    if ((m_ready >> lane) & 1ull) V.tbl[wave_rank(m_ready)] = (uint32_t)lane;
    else if ((m_free >> lane) & 1ull) V.tbl[n_ready + wave_rank(m_free)] = (uint32_t)lane;
    lds_wave_fence();   // the table is read by other lanes of this wave
    // lane side: finished lanes first, then idle lanes
    const int req = is_done ? wave_rank(done) : n_done + wave_rank(idle);
    const bool served = (is_done || is_idle) && req < n_ready + n_free;
    const bool takes = served && req < n_ready;
    const bool parks = served && is_done;
    uint32_t slot = 0;
    if (served) slot = V.tbl[req];
    uint32_t got[W];
#pragma unroll
    for (int w = 0; w < W; w++) got[w] = rec[w];
    if (takes) pool_load(V, slot, got);          // read the READY record before overwriting its slot
    lds_wave_fence();
    if (parks) {
        pool_store(V, slot, rec);
        V.sstate[slot] = done_state;
    } else if (takes) {
        V.sstate[slot] = SL_EMPTY;
    }
#pragma unroll
    for (int w = 0; w < W; w++) rec[w] = got[w];
    lds_wave_fence();   // slot states written by the lanes that parked / took
    const uint32_t st = V.sstate[lane];
    m_ready = __ballot(st == SL_READY);
    m_shade = __ballot(st == SL_HIT || st == SL_MISS);
    return (takes ? 1 : 0) | (parks ? 2 : 0);
}

// -------------------------------------------------------------------------------------------
// Scheduler 1: per-wave LDS ray pool ("parked rays").  Every wave owns 64 register lanes (the
// rays being marched) plus 64 LDS slots holding parked rays that are either READY to start a
// raycast or waiting to be shaded (HIT / MISS).  A lane whose raycast finishes swaps its ray
// with a READY slot (ballot + mbcnt rank matching through a small LDS table) and keeps marching,
// so the march loop stays (nearly) full; shading runs on the SLOTS (lane s <-> slot s) only when
// >= shade_lanes of them wait, so it runs on (nearly) full waves too.  Slots freed by finished
// samples are refilled with fresh pixel-samples.  Wave-private: no cross-wave synchronisation.
// F_META packs the nearest-object index (5 bits), the bounce number (11 bits) and the RNG draw count (16 bits):
// the host falls back to scheduler 0 for MAX_RAYTRACE > 2047 (a path draws < 8 numbers per bounce).
enum { F_OX = 0, F_OY, F_OZ, F_DX, F_DY, F_DZ, F_CR, F_CG, F_CB, F_TEVAL, F_META, F_KEY, F_ITEM, F_COUNT };
static_assert(F_COUNT <= POOL_WORDS, "trace record must fit the pool record");
RT_D uint32_t pack_meta(int idx, int bounce, uint32_t cnt) { return (uint32_t)idx | ((uint32_t)bounce << 5) | (cnt << 16); }
RT_D int meta_idx(uint32_t m) { return (int)(m & 31u); }
RT_D int meta_bounce(uint32_t m) { return (int)((m >> 5) & 2047u); }
RT_D uint32_t meta_cnt(uint32_t m) { return m >> 16; }

// Waves per SIMD the box instances are compiled for.  The kernel is latency-bound per wave (PMC at 4
// waves: a wave issues during 44 % of its cycles, waits on s_waitcnt 27 %, on the issue arbiter
// 29 %), so more resident waves pay.  Measured on the headline frame (pool kernel time / HBM-side
// bytes per launch, of which 18 GB are staging stores and primary records):
//   4 waves (116 VGPRs)            147.1 ms / 18 GB      5 waves (96 VGPRs, no spill)  137.9 ms / 19 GB
//   6 waves (80 VGPRs, 13 spills)  132.4 ms / 23 GB      7 waves (72, 18 spills; rank table aliased so that LDS fits) 131.5 ms / 29 GB
// Two things keep the 80-register build cheap: the marching ray's origin, direction and last distance
// are parked in LDS during shading (7 dwords per lane), and the material is fetched after the normal
// (compiler barrier in surface_interaction).  Without the barrier the 6-wave build saved and restored
// 13 registers around every shading pass through scratch: 100 GB of extra traffic per launch.
#ifndef RT_POOL_WAVES
#define RT_POOL_WAVES 6
#endif
#ifndef RT_POOL_WAVES_GENERIC
#define RT_POOL_WAVES_GENERIC 5   // 114 -> 96 VGPRs, 13 spills: C4 (Tokyo IBL 4K) trace kernel 196 -> 180 ms; 6 waves: 187
#endif
#ifndef RT_POOL_WAVES_BUNNY
#define RT_POOL_WAVES_BUNNY 4
#endif
constexpr int pool_waves(int kind) {
    return kind == KIND_BOXES ? RT_POOL_WAVES : kind == KIND_GENERIC ? RT_POOL_WAVES_GENERIC : kind == KIND_BUNNY ? RT_POOL_WAVES_BUNNY : 1;
}
template <int KIND, int NOBJ, uint32_t SIG = 0>
RT_D void trace_paths_pool_impl(const Params& P) {
    __shared__ ObjFull lds_obj[NOBJ > 0 ? NOBJ : (KIND == KIND_BUNNY ? 1 : MAX_OBJ)];   // KIND_BUNNY: exactly one object
    __shared__ uint32_t pool_all[4][F_COUNT][64];
    constexpr bool PARK = KIND == KIND_BOXES || KIND == KIND_GENERIC;
    __shared__ float save_all[PARK ? 4 : 1][PARK ? 7 : 1][PARK ? 64 : 1];   // marching state parked during shading
    // the swap's rank table (dispatch phase) shares its words with the parked marching state (shading phase) where that
    // exists: never live together, and 1 KB less per block lets a seventh block fit into a CU's 160 KB
    __shared__ uint32_t tbl_all[PARK ? 1 : 4][64];
    __shared__ uint32_t sstate_all[4][64];
#if RT_FAST_MATH
    __shared__ float acc_all[4][ACC_SLOTS][4];
    __shared__ uint32_t acc_tag_all[4][ACC_SLOTS];
#endif
    stage_objects(P, lds_obj);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
#if RT_FAST_MATH
    const AccView A = {acc_all[wave], acc_tag_all[wave]};
    if (lane < ACC_SLOTS) A.tag[lane] = 0xffffffffu;
    lds_wave_fence();
    uint32_t n_dep = 0;
#endif
    uint32_t (*pool)[64] = pool_all[wave];
    uint32_t* sstate = sstate_all[wave];
    uint32_t* const tbl_w = PARK ? reinterpret_cast<uint32_t*>(&save_all[PARK ? wave : 0][0][0]) : tbl_all[PARK ? 0 : wave];
    const PoolView V = {pool_all[wave], sstate_all[wave], tbl_w};
    sstate[lane] = SL_EMPTY;

    Lane L;
    L.state = ST_IDLE;   // ST_IDLE = no ray, ST_MARCH, ST_HIT / ST_MISS = raycast done, ray still in registers
    L.n_steps = L.n_raycasts = L.n_hits = L.n_sky = 0;
    L.o = L.d = mk(0, 0, 0);
    L.t = L.w = L.s = L.dist = L.t_eval = 0.0f;
    L.idx = 0;
    L.steps_left = 0;
    // the marching ray's bookkeeping (travels with the ray through the pool)
    vec3 a_col = mk(0, 0, 0);
    uint32_t a_meta = 0, a_key = 0, a_item = 0;   // a_meta: bounce and draw count, packed like F_META
    // work counters kept wave-uniform (scalar registers, scalar adds of ballot popcounts) where the
    // control flow allows it: the per-lane v_add per march step and 3 VGPRs go away
    uint32_t w_steps = 0, w_raycasts = 0, w_hits = 0, w_samples = 0, w_sky = 0;
    uint32_t w_mlp_wave = 0, w_mlp_lane = 0;
    WorkRange wr = {0, 0, false};
    unsigned long long m_ready = 0, m_shade = 0;   // slot masks (wave-uniform)
    __shared__ __attribute__((aligned(16))) float b_lds_all[(KIND == KIND_BUNNY) ? 4 * BUNNY_LDS_WORDS : 4];
    __shared__ __attribute__((aligned(16))) float b_bias[32];
    float* b_lds = &b_lds_all[(KIND == KIND_BUNNY) ? wave * BUNNY_LDS_WORDS : 0];
    BunnyFrag b_frag = {};
    if (KIND == KIND_BUNNY && P.bunny != nullptr) {
        bunny_frag_load(P.bunny, lane, b_frag);
        bunny_bias_stage(P.bunny, b_bias);
        __syncthreads();
    }
    const int T = P.shade_lanes;
    const int m_swap = P.swap_lanes;

    auto f2u = [](float x) { return __builtin_bit_cast(uint32_t, x); };
    auto u2f = [](uint32_t x) { return __builtin_bit_cast(float, x); };

#ifdef RT_DEBUG_CULL
    float dbg_lb[NOBJ > 0 ? NOBJ : 1];
#pragma unroll
    for (int i = 0; i < (NOBJ > 0 ? NOBJ : 1); i++) dbg_lb[i] = -1.0f;
    float dbg_ub = 3.0e38f;
    uint32_t dbg_cull_eval = 0, dbg_cull_all = 0;
#endif
#ifdef RT_DEBUG_PHASE
    unsigned long long tB = 0, tD = 0, tA = 0, tc = __builtin_readcyclecounter();
    unsigned long long dbg_march_lanes = 0, dbg_march_steps = 0, dbg_shade_lanes = 0, dbg_shade_passes = 0;
#define RT_PHASE(acc) { unsigned long long tn = __builtin_readcyclecounter(); acc += tn - tc; tc = tn; }
#else
#define RT_PHASE(acc)
#endif
    for (;;) {
        // ================================================================ phase B: shade / refill the slots
        {
            const int n_shade = __popcll(m_shade);
            const int n_ready = __popcll(m_ready);
            const int n_free = 64 - n_shade - n_ready;
            const bool run_b = n_shade >= T || (n_ready <= P.ready_low && (n_shade > 0 || (n_free > 0 && !wr.drained)));
            if (run_b) {
                // Shading needs ~60 registers of its own; the marching lanes' ray (origin, direction, relaxation
                // state) is parked in LDS meanwhile instead of being spilled to scratch by the register cap
                if constexpr (KIND == KIND_BOXES || KIND == KIND_GENERIC) {
                    float (*sv)[64] = save_all[wave];
                    sv[0][lane] = L.o.x, sv[1][lane] = L.o.y, sv[2][lane] = L.o.z;
                    sv[3][lane] = L.d.x, sv[4][lane] = L.d.y, sv[5][lane] = L.d.z;
                    sv[6][lane] = L.dist;
                }
                uint32_t st = sstate[lane];
                w_hits += (uint32_t)__popcll(__ballot(st == SL_HIT));
                PathRay R;
                R.o = R.d = R.col = mk(0, 0, 0);
                R.t_eval = 0.0f;
                R.idx = R.bounce = 0;
                R.key = R.cnt = R.item = 0;
                bool alive = false;
                uint32_t sky1 = 0;   // this slot did an environment lookup
                if (st == SL_HIT || st == SL_MISS) {
                    R.o = mk(u2f(pool[F_OX][lane]), u2f(pool[F_OY][lane]), u2f(pool[F_OZ][lane]));
                    R.d = mk(u2f(pool[F_DX][lane]), u2f(pool[F_DY][lane]), u2f(pool[F_DZ][lane]));
                    R.col = mk(u2f(pool[F_CR][lane]), u2f(pool[F_CG][lane]), u2f(pool[F_CB][lane]));
                    R.t_eval = u2f(pool[F_TEVAL][lane]);
                    const uint32_t meta = pool[F_META][lane];
                    R.idx = meta_idx(meta);
                    R.bounce = meta_bounce(meta);
                    R.cnt = meta_cnt(meta);
                    R.key = pool[F_KEY][lane];
                    R.item = pool[F_ITEM][lane];
                    if (KIND != KIND_BUNNY) {
                        if (st == SL_HIT) {
                            alive = shade_hit<KIND>(P, lds_obj, R);
                        } else {
                            shade_miss(P, R, sky1);
                        }
                    }
                }
                if (KIND == KIND_BUNNY) {
                    // the normal's four MLP evaluations run on the matrix cores for the whole wave
                    // (uniform control flow); only the lanes whose slot holds a hit use the result
                    vec3 hp = fma3(R.t_eval, R.d, R.o);
                    vec3 nrm = mk(0, 0, 0);
                    if (__any(st == SL_HIT)) {
                        w_mlp_lane += 4u * (uint32_t)__popcll(__ballot(st == SL_HIT));
                        if (P.mlp_mfma) nrm = bunny_normal_wave(P, b_frag, b_lds, b_bias, tbl_w, lane, st == SL_HIT, hp, w_mlp_wave);
                        else if (st == SL_HIT) nrm = calc_normal<KIND_BUNNY>(P, lds_obj[0], hp);
                    }
                    if (st == SL_HIT) {
                        alive = shade_hit<KIND, true>(P, lds_obj, R, nrm);
                    } else if (st == SL_MISS) {
                        shade_miss(P, R, sky1);
                    }
                }
                w_sky += (uint32_t)__popcll(__ballot(sky1 != 0));
#if RT_FAST_MATH
                acc_add(P, A, (st == SL_HIT || st == SL_MISS) && !alive, R.item / (uint32_t)P.K, R.col, lane, n_dep);
#else
                stage_sample(P, (st == SL_HIT || st == SL_MISS) && !alive, R.item, R.col, lane);
#endif
                w_samples += (uint32_t)__popcll(__ballot((st == SL_HIT || st == SL_MISS) && !alive));
                if (st == SL_HIT || st == SL_MISS) st = SL_EMPTY;
                // refill free slots with fresh pixel-samples.  start_item costs ~200 instructions and a shading pass frees
                // only about a third of its slots, so the refill may wait until refill_lanes slots are free — unless this
                // pass was started because the pool ran dry
                const bool is_free = st == SL_EMPTY && !alive;
                const bool do_refill = __popcll(__ballot(is_free)) >= P.refill_lanes || n_shade < T;
                bool got = claim_items(P, wr, is_free && do_refill, lane, R.item);
                uint32_t resumed = 0;   // primary_split: state the primary kernel left this item in
                bool roulette0 = false;
                // the primary record is requested BEFORE the camera ray is regenerated: the ~150 instructions of
                // start_item cover part of the global-load latency
                float rec_t = 0.0f;
                uint32_t rec_code = 0;
                // (read ONCE, streaming: a non-temporal load keeps the 4.3 GB of primary records of a headline step from turning the
                // L2 over under the staging records, whose partially written lines then live long enough to be completed —
                // WRITE_SIZE of this kernel 12.49 -> 11.11 GB per step, time unchanged; round 6)
                if (got && P.primary_split) {
#if defined(__HIP_DEVICE_COMPILE__)
                    rec_t = __builtin_nontemporal_load(P.primary + R.item);
                    rec_code = __builtin_nontemporal_load(P.primary_code + R.item);
#else
                    rec_t = P.primary[R.item];
                    rec_code = P.primary_code[R.item];
#endif
                }
                if (got) {
                    int r = start_item(P, R);
                    if (r == 1) {
                        if (P.primary_split) {
                            R.t_eval = rec_t;
                            R.idx = (int)(rec_code & 31u);
                            resumed = rec_code >> 5;             // ST_HIT or ST_MISS
                        } else {
                            alive = true;
                        }
                    }
                    roulette0 = r == 0;
                }
#if RT_FAST_MATH
                acc_add(P, A, roulette0, R.item / (uint32_t)P.K, R.col, lane, n_dep);
#else
#if RT_STAGE_DENSE
                if (__any(roulette0)) stage_sample(P, roulette0, R.item, R.col, lane);
#else
                if (roulette0) write_sample(P, R.item, R.col);
#endif
#endif
                w_samples += (uint32_t)__popcll(__ballot(roulette0));
                if (resumed == ST_HIT || resumed == ST_MISS) {
                    // park the finished primary raycast; it is shaded with the next batch
                    pool[F_OX][lane] = f2u(R.o.x); pool[F_OY][lane] = f2u(R.o.y); pool[F_OZ][lane] = f2u(R.o.z);
                    pool[F_DX][lane] = f2u(R.d.x); pool[F_DY][lane] = f2u(R.d.y); pool[F_DZ][lane] = f2u(R.d.z);
                    pool[F_CR][lane] = f2u(R.col.x); pool[F_CG][lane] = f2u(R.col.y); pool[F_CB][lane] = f2u(R.col.z);
                    pool[F_TEVAL][lane] = f2u(R.t_eval);
                    pool[F_META][lane] = pack_meta(R.idx, R.bounce, R.cnt);
                    pool[F_KEY][lane] = R.key;
                    pool[F_ITEM][lane] = R.item;
                    st = resumed == ST_HIT ? SL_HIT : SL_MISS;
                }
                if (alive) {
                    pool[F_OX][lane] = f2u(R.o.x); pool[F_OY][lane] = f2u(R.o.y); pool[F_OZ][lane] = f2u(R.o.z);
                    pool[F_DX][lane] = f2u(R.d.x); pool[F_DY][lane] = f2u(R.d.y); pool[F_DZ][lane] = f2u(R.d.z);
                    pool[F_CR][lane] = f2u(R.col.x); pool[F_CG][lane] = f2u(R.col.y); pool[F_CB][lane] = f2u(R.col.z);
                    pool[F_META][lane] = pack_meta(0, R.bounce, R.cnt);
                    pool[F_KEY][lane] = R.key;
                    pool[F_ITEM][lane] = R.item;
                    st = SL_READY;
                }
                sstate[lane] = st;
                m_ready = __ballot(st == SL_READY);
                m_shade = __ballot(st == SL_HIT || st == SL_MISS);   // non-zero only with primary_split
                if constexpr (KIND == KIND_BOXES || KIND == KIND_GENERIC) {
                    float (*sv)[64] = save_all[wave];
                    L.o = mk(sv[0][lane], sv[1][lane], sv[2][lane]);
                    L.d = mk(sv[3][lane], sv[4][lane], sv[5][lane]);
                    L.dist = sv[6][lane];
                }
            }
        }

        RT_PHASE(tB)
        // ================================================================ dispatch: swap finished lanes with parked rays
        {
            const bool is_done = L.state == ST_HIT || L.state == ST_MISS;
            uint32_t rec[F_COUNT];
            rec[F_OX] = f2u(L.o.x); rec[F_OY] = f2u(L.o.y); rec[F_OZ] = f2u(L.o.z);
            rec[F_DX] = f2u(L.d.x); rec[F_DY] = f2u(L.d.y); rec[F_DZ] = f2u(L.d.z);
            rec[F_CR] = f2u(a_col.x); rec[F_CG] = f2u(a_col.y); rec[F_CB] = f2u(a_col.z);
            rec[F_TEVAL] = f2u(L.t_eval);
            rec[F_META] = (a_meta & ~31u) | (uint32_t)L.idx;
            rec[F_KEY] = a_key; rec[F_ITEM] = a_item;
            const int r = pool_swap(V, lane, is_done, L.state == ST_IDLE, L.state == ST_HIT ? SL_HIT : SL_MISS, rec, m_ready, m_shade);
            if (r & 2) L.state = ST_IDLE;
            w_raycasts += (uint32_t)__popcll(__ballot((r & 1) != 0));
            if (r & 1) {
                L.o = mk(u2f(rec[F_OX]), u2f(rec[F_OY]), u2f(rec[F_OZ]));
                L.d = mk(u2f(rec[F_DX]), u2f(rec[F_DY]), u2f(rec[F_DZ]));
                a_col = mk(u2f(rec[F_CR]), u2f(rec[F_CG]), u2f(rec[F_CB]));
                a_meta = rec[F_META];
                a_key = rec[F_KEY]; a_item = rec[F_ITEM];
                march_init(P, L);
            }
        }

        RT_PHASE(tD)
        // ================================================================ phase A: march
        {
            int n_march = __popcll(__ballot(L.state == ST_MARCH));
            if (n_march == 0) {
                const bool any_ray = __ballot(L.state != ST_IDLE) != 0;
                if (!any_ray && m_ready == 0 && m_shade == 0 && wr.drained) break;
                continue;
            }
            const int n_ready = __popcll(m_ready);
            // DRAIN: the work has run out, nothing is parked READY and a handful of lanes still march — what is left of the launch
            // is these rays' dependent steps.  They go on to the end of their raycasts in the culled wave march of the primary
            // kernel (exact: same (index, distance) per step): with so few rays nearly every object can be excluded, and a lone
            // ray runs its object's lean loop.
            if constexpr (NOBJ > 0 && (KIND == KIND_BOXES || KIND == KIND_GENERIC)) {
                if (wr.drained && n_ready == 0 && n_march <= P.drain_lanes && P.cull_ok) {
                    const uint32_t s0 = L.n_steps;
                    culled_march_wave<KIND, NOBJ, SIG>(P, L, nullptr);
                    w_steps += wave_sum(L.n_steps - s0);
                    continue;
                }
            }
            int n_done;
            // bunny: position evaluated (local point b_lp), MLP still to run.  Lanes still waiting when the march phase
            // is left re-derive their point on re-entry (the cheap half of the step; nothing has been counted yet)
            bool b_pending = false;
            vec3 b_lp = mk(0, 0, 0);
            do {
                if (KIND == KIND_BUNNY) {
                    // The neural SDF costs ~1700 instructions, the bounding-sphere branch ~40.  Lanes outside
                    // the unit sphere RUN AHEAD with cheap steps until they enter it (pending) or finish, so
                    // the MLP is evaluated once for as many lanes as possible instead of once per step for
                    // whichever few lanes happen to be inside.
                    for (;;) {
                        const bool can = L.state == ST_MARCH && !b_pending;
                        if (!__any(can)) break;
                        if (can) {
                            float dist;
                            if (bunny_pre(P, L, b_lp, dist)) b_pending = true;
                            else march_update(P, L, 0, dist);
                        }
                        if (__popcll(__ballot(b_pending)) >= P.mlp_lanes) break;
                    }
                    const unsigned long long pm = __ballot(b_pending);
                    if (pm) {
                        // The waiting lanes are COMPACTED into the low ray slots and the matrix cores compute whole
                        // halves of 32 slots: all of them when >= mlp_full wait (two halves), else the first 32 by
                        // rank — from the bottom and from the top of the wave in turn, so nobody starves; the others
                        // keep waiting (they cost nothing) and join the next pass.  Uniform control flow.
                        const int n_pend = __popcll(pm);
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0));
                        const int halves = n_pend >= P.mlp_full ? 2 : 1;
                        const int take = n_pend < 32 * halves ? n_pend : 32 * halves;
                        const int first = (w_mlp_wave & 1u) ? n_pend - take : 0;
                        const bool sel = b_pending && rank >= first && rank < first + take;
                        w_mlp_wave += (uint32_t)halves;
                        w_mlp_lane += (uint32_t)take;
                        // (option mlp_mfma = 0: the same network on the vector ALU, per lane, under the same pass policy — the A/B
                        // that isolates the matrix cores; the chain order is the same, so are the bits)
                        float sd;
                        if (P.mlp_mfma) sd = bunny_mlp_wave(b_frag, P.bunny, b_lds, b_bias, lane, b_lp, sel ? rank - first : -1, halves);
                        else sd = sel ? bunny_mlp(P.bunny, b_lp) : 0.0f;
                        if (sel) {
                            march_update(P, L, 0, bunny_post_value(P, sd));
                            b_pending = false;
                        }
                    }
                } else {
                    w_steps += (uint32_t)n_march;
#ifdef RT_DEBUG_PHASE
                    dbg_march_lanes += (unsigned)n_march;
                    dbg_march_steps += 1;
#endif
#ifdef RT_DEBUG_CULL
                    // MEASUREMENT BUILD (-DRT_DEBUG_CULL): how many (wave-step, object) evaluations would vanish if the pool
                    // kernel's march loop kept per-lane Lipschitz bounds and skipped an object when NO marching lane needs it
                    // (what primary_rays does for its coherent waves)?  The culled step is exact, so results do not change;
                    // dbg[0] += objects evaluated, dbg[1] += objects x wave-steps.
                    if constexpr (NOBJ > 0 && (KIND == KIND_BOXES || KIND == KIND_GENERIC)) {
                        const bool active = L.state == ST_MARCH;
                        if (active && L.steps_left == P.cfg.max_raymarch) {      // a ray that starts its raycast knows nothing yet
#pragma unroll
                            for (int i = 0; i < NOBJ; i++) dbg_lb[i] = -1.0f;
                            dbg_ub = 3.0e38f;
                        }
                        vec3 pos = fma3(L.t, L.d, L.o);
                        const float t_before = L.t;
                        int idx;
                        float dist;
                        uint32_t ev = 0;
                        nearest_culled<KIND, NOBJ, SIG>(P, pos, L.t, active, dbg_ub, dbg_lb, idx, dist, &ev);
                        dbg_cull_eval += ev;
                        dbg_cull_all += NOBJ;
                        float moved = 0.0f;
                        if (active) {
                            L.t_eval = L.t;
                            march_update(P, L, idx, dist);
                            moved = fabs_(L.t - t_before) * 1.000001f;
                            dbg_ub = dist + moved;
                        }
#pragma unroll
                        for (int i = 0; i < NOBJ; i++) dbg_lb[i] -= moved;
                    } else
#endif
                    if (L.state == ST_MARCH) march_step<KIND, NOBJ, SIG>(P, L);
                }
                n_march = __popcll(__ballot(L.state == ST_MARCH));
                n_done = __popcll(__ballot(L.state == ST_HIT || L.state == ST_MISS));
                // keep marching until enough lanes want a swap; with no READY ray parked, finished lanes
                // can only be parked, so wait for more of them (bounded by the march lanes running out)
            } while (n_march > 0 && n_done < (n_ready > 0 ? m_swap : 2 * m_swap));
        }
        RT_PHASE(tA)
    }
#ifdef RT_DEBUG_PHASE
    if (lane == 0) {   // DEBUG: cycles per phase >> 10, summed over waves, in the mlp_wave / sky / mlp_lane counters (scenes without sky or MLP)
        atomicAdd(&P.counters->mlp_wave_evals, tB >> 10);
        atomicAdd(&P.counters->sky_lookups, tD >> 10);
        atomicAdd(&P.counters->mlp_lane_evals, tA >> 10);
        atomicAdd(&P.counters->hits, dbg_march_lanes);          // (replaces the hit count in this build)
        atomicAdd(&P.counters->deposits, dbg_march_steps);
    }
#endif
#ifdef RT_DEBUG_CULL
    if (lane == 0) {
        atomicAdd(&P.counters->dbg[0], (unsigned long long)dbg_cull_eval);
        atomicAdd(&P.counters->dbg[1], (unsigned long long)dbg_cull_all);
    }
#endif
    if (KIND == KIND_BUNNY && lane == 0 && w_mlp_wave) {
        atomicAdd(&P.counters->mlp_wave_evals, (unsigned long long)w_mlp_wave);
        atomicAdd(&P.counters->mlp_lane_evals, (unsigned long long)w_mlp_lane);
    }
#if RT_FAST_MATH
    acc_flush_all(P, A, lane, n_dep);
    const uint32_t dep_out = n_dep;
#else
    const uint32_t dep_out = 0;
#endif
    // (the neural-SDF march counts its steps per lane: run-ahead lanes step at different times)
    flush_counters(P, KIND == KIND_BUNNY ? L.n_steps : (lane == 0 ? w_steps : 0u), lane == 0 ? w_raycasts : 0u,
#ifdef RT_DEBUG_PHASE
                   0u,
#else
                   lane == 0 ? w_hits : 0u,
#endif
                   lane == 0 ? w_sky : 0u, lane == 0 ? w_samples : 0u, dep_out);
}
template <int KIND, int NOBJ, uint32_t SIG = 0>
__global__ void __launch_bounds__(256, pool_waves(KIND)) trace_paths_pool(const Params P) { trace_paths_pool_impl<KIND, NOBJ, SIG>(P); }

}  // namespace rt

#include "rt_persistent.hpp"   // the src/ persistent-ray kernels (same namespace; uses the pool above)
#include "rt_split.hpp"        // ... and their wavefront split for launches of one bounce-step
#include "rt_chain.hpp"        // ... and the chain kernel: the heaviest pixels of a chain-bound launch, beside the pool kernel
