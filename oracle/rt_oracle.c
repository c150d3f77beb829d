/*
 * rt_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the per-pixel Monte Carlo sample path of HK-SHAO/RayTracingPBR
 * (Python + Taichi).  It exists to CHECK the HIP kernels; nothing in the product path
 * (raytracingpbr_amd/) may import, link or call it.  Allowed users: tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * PARITY PIN STATUS: pinned to the reference's own code, at rounding tolerance.  Taichi is not installable here and
 * the reference ships no tests or golden vectors, so the pin is: (P1) the reference's own
 * @ti.func / @ti.kernel bodies, imported from /root/reference and executed on a stand-in runtime by
 * tools/ref_crosscheck.py, whose inputs and outputs are the fixtures tests/golden/ref_*.npz that
 * tests/test_oracle_refpin.py compares this file with — function tables, per-raycast and per-interaction
 * observations, per-sample colours and counts, frame buffers; every sample that differs is traced to a decision whose
 * operands agree to a stated number of ulps (the decision log below) or the test fails; (P2) the reference's committed
 * Cornell image per pixel on the GPU, (P3) the published bunny image's silhouette; (K1) analytic known answers.
 * What the stand-in cannot show — Taichi's own lowering of ti.random, casts and min/max of NaN — is listed in
 * SURVEY.md Appendix D and DESIGN.md section 2.
 *
 * Each function cites the reference lines it follows (paths relative to the reference
 * root).  Numerics: IEEE f32, operation order as written, see rt_oracle_math.h.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -mfma -fopenmp).
 */
#include "rt_oracle.h"
#include "rt_oracle_math.h"

#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[256];
static int fail(int code, const char* msg) { snprintf(g_err, sizeof g_err, "%s", msg); return code; }

/* ------------------------------------------------------------------ decision log (test hook)
 * Every data-dependent branch of the sample path goes through DEC(kind, a, b, scale) = (a < b).  While a log is
 * active on this thread (rto_test_sample_decisions) the operands are recorded, and the decisions whose ordinal is in
 * the flip list return the opposite: tests/test_oracle_refpin.py uses this to show that a sample which differs from
 * the reference's own run differs by a decision whose operands were within a few ulps of `scale` (the magnitude of
 * the largest intermediate that went into them), and that taking that decision the other way reproduces the
 * reference.  EVENT entries mark what happened (end of a raycast, of a surface interaction) so that a log can be
 * lined up with the reference's recorded events.  The -O3 timing build (make fast) compiles the hook out. */
enum { RTO_D_NEAREST = 1, RTO_D_HIT, RTO_D_FALLBACK, RTO_D_ESCAPE, RTO_D_OUTER, RTO_D_REFLECT, RTO_D_TIR, RTO_D_TRANSMIT,
       RTO_D_HORIZON, RTO_D_STOP_GAIN, RTO_D_STOP_LO, RTO_D_STOP_HI, RTO_D_ROULETTE, RTO_D_ENV_X, RTO_D_ENV_Y, RTO_D_BOUND,
       RTO_E_RAYCAST = 100, RTO_E_SURFACE = 101, RTO_E_DIR = 102 };
typedef struct { int kind, outcome; float a, b, scale; } rto_decision;
#ifndef RTO_NO_DECISIONS
static __thread struct { int active, n, cap, n_flip; const int* flip; rto_decision* log; } g_dec;
/* surface-interaction outputs to adopt instead of the computed ones (rows of 10 floats: direction, colour, origin, draw
 * count), by ordinal within the sample: lets a test keep the oracle on the reference's recorded trajectory, so that what
 * is compared downstream is not the amplification of an upstream rounding difference (tests/test_oracle_refpin.py) */
static __thread struct { const float* rows; int n, next; } g_inj;
static int dec_slow(int kind, float a, float b, float scale, int r) {
    const int i = g_dec.n++;
    for (int k = 0; k < g_dec.n_flip; k++) if (g_dec.flip[k] == i) r = !r;
    if (g_dec.log && i < g_dec.cap) { rto_decision d = {kind, r, a, b, scale}; g_dec.log[i] = d; }
    return r;
}
static inline int DEC(int kind, float a, float b, float scale) {
    const int r = a < b;
    return __builtin_expect(g_dec.active, 0) ? dec_slow(kind, a, b, scale, r) : r;
}
/* floor(v) for a texel index; a flip moves to the other side of the nearest integer boundary */
static inline int DEC_FLOOR(int kind, float v, float scale) {
    int x = (int)v;
    if (__builtin_expect(g_dec.active, 0)) {
        const float fr = v - (float)x;
        const float edge = fr < 0.5f ? (float)x : (float)(x + 1);
        if (!dec_slow(kind, v, edge, scale, 1)) x = fr < 0.5f ? x - 1 : x + 1;
    }
    return x;
}
static inline void EVENT(int kind, float a, float b, float c) {
    if (__builtin_expect(g_dec.active, 0) && g_dec.log && g_dec.n < g_dec.cap) {
        rto_decision d = {kind, 0, a, b, c};
        g_dec.log[g_dec.n] = d;
    }
    if (g_dec.active) g_dec.n++;
}
#else
#define DEC(kind, a, b, scale) ((void)(scale), (a) < (b))
#define DEC_FLOOR(kind, v, scale) ((void)(scale), (int)(v))
#define EVENT(kind, a, b, c) ((void)(a), (void)(b), (void)(c))
#endif
static inline float v3_maxabs(v3 p) { return fmaxf(fmaxf(fabsf(p.x), fabsf(p.y)), fabsf(p.z)); }

struct rto_ctx {
    rtpbr_config cfg;
    int have_cfg, have_scene, have_cam;
    rtpbr_object obj[RTPBR_MAX_OBJECTS];
    int n_obj;
    rtpbr_camera cam;
    float extent;                         /* max_i(|centre_i| + |size_i|), inf-norm: magnitude of the SDF intermediates (decision log) */
    float* env; int env_w, env_h;         /* T9: (W_e,H_e,3) f32, [x][y] */
    float* image_buffer;                  /* T7 */
    float* image_pixels;                  /* T8 */
    rtpbr_ray* ray_buffer;                /* T6 */
    float* diff_buffer;                   /* T11 (W,H,2) */
    float* diff_pixels;                   /* T11 (W,H) */
    int tile_w, tile_h, rank, world;
    uint32_t sample_base;                 /* samples (or bounce-steps) done since create */
    rtpbr_counters ctr;
    unsigned long long mlp_evals;         /* of the last rto_sample() call */
    int threads;
};

/* ------------------------------------------------------------------ bunny weights
 * examples/bunny/bunny_sdf_glass.py:157-201 — the 625 literals are DATA, supplied at run
 * time through rto_set_bunny_weights() (tests load them from the committed data file). */
static float g_bunny[625];
static int g_bunny_set = 0;
static __thread unsigned long long t_mlp_evals;      /* network evaluations by this thread (sd_bunny inside the unit sphere) */

/* ------------------------------------------------------------------ Euler -> matrix
 * src/util.py:36-42 rotate(); examples: angle().  M = Rz @ Ry @ Rx, row major. */
static void m3_mul(const float* a, const float* b, float* o) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            o[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}
void rto_rotate(const float* rad, float* m) {
    float sx = (float)sin((double)rad[0]), cx = (float)cos((double)rad[0]);
    float sy = (float)sin((double)rad[1]), cy = (float)cos((double)rad[1]);
    float sz = (float)sin((double)rad[2]), cz = (float)cos((double)rad[2]);
    float rz[9] = {cz, sz, 0, -sz, cz, 0, 0, 0, 1};
    float ry[9] = {cy, 0, -sy, 0, 1, 0, sy, 0, cy};
    float rx[9] = {1, 0, 0, 0, cx, sx, 0, -sx, cx};
    float t[9];
    m3_mul(rz, ry, t);
    m3_mul(t, rx, m);
}

/* ------------------------------------------------------------------ SDF primitives
 * src/sdf.py:21-51 (p local, s = transform.scale) */
static float sd_bunny(v3 p);

static float sdf_shape(int type, v3 p, const float* s, float rho, float max_dis) {
    switch (type) {
    case RTPBR_SHAPE_SPHERE:                                   /* sdf.py:26-28 */
        return v3_length(p) - s[0];
    case RTPBR_SHAPE_BOX: {                                    /* sdf.py:31-34 */
        float qx = fabsf(p.x) - s[0], qy = fabsf(p.y) - s[1], qz = fabsf(p.z) - s[2];
        v3 m = v3_make(fmaxf(qx, 0.0f), fmaxf(qy, 0.0f), fmaxf(qz, 0.0f));
        return (v3_length(m) + fminf(fmaxf(qx, fmaxf(qy, qz)), 0.0f)) - rho;
    }
    case RTPBR_SHAPE_CYLINDER: {                               /* sdf.py:37-40 */
        float l = sqrtf(fmaf(p.z, p.z, p.x * p.x));
        float dx = fabsf(l) - s[0], dy = fabsf(p.y) - s[1];
        float mx = fmaxf(dx, 0.0f), my = fmaxf(dy, 0.0f);
        return fminf(fmaxf(dx, dy), 0.0f) + sqrtf(fmaf(my, my, mx * mx));
    }
    case RTPBR_SHAPE_CONE: {                                   /* sdf.py:43-46 */
        float q = sqrtf(fmaf(p.z, p.z, p.x * p.x));
        return fmaxf(fmaf(s[2], p.y, s[0] * q), -s[1] - p.y);
    }
    case RTPBR_SHAPE_PLANE:                                    /* sdf.py:49-51 */
        return p.y - s[1];
    case RTPBR_SHAPE_BUNNY:
        return sd_bunny(p);
    default:                                                   /* sdf.py:21-23 */
        return max_dis;
    }
}

/* world -> local: src/sdf.py:64-68 transform(); examples: angle(radians(rot)) @ (pos - position).
 * The bunny adds its per-frame animation (bunny_sdf_glass.py:213-217). */
static v3 to_local(const struct rto_ctx* c, const rtpbr_object* o, v3 p) {
    v3 d = v3_sub(p, v3_make(o->transform.position[0], o->transform.position[1], o->transform.position[2]));
    v3 l = m3_mulv(o->transform.matrix, d);
    if (o->type == RTPBR_SHAPE_BUNNY) {
        float t = RTO_PI * (float)c->cfg.frame / 120.0f;
        float st, ct;
        rto_sincosf(t, &st, &ct);
        /* angle(vec3(0,0,t)) = [[c,s,0],[-s,c,0],[0,0,1]] */
        v3 r = v3_make(fmaf(st, l.y, ct * l.x), fmaf(ct, l.y, -st * l.x), l.z);
        r.z = r.z + c->cfg.anim_bob * st;      /* bunny_sdf_glass.py:216, bunny_sdf_v2.py:216 (0.1); bunny_sdf.py has no bob (0) */
        l = r;
    }
    return l;
}

static float signed_distance(const struct rto_ctx* c, const rtpbr_object* o, v3 p) {
    return sdf_shape(o->type, to_local(c, o, p), o->transform.scale, c->cfg.box_round, c->cfg.max_dis);
}

/* nearest: min_i |sdf_i|, ties to the lowest index.
 * examples: cornell_box_v3/pathtracer.py:41-49 (start from object 0);
 * src/scene.py:44-56 and tokyo_ibl.py:221-236 (start from (0, MAX_DIS)) — cfg.nearest_init. */
static int nearest(struct rto_ctx* c, v3 p, float* dist, rtpbr_counters* ctr) {
    int idx = 0;
    float best;
    int start;
    if (c->cfg.nearest_init) { best = c->cfg.max_dis; start = 0; }
    else { best = fabsf(signed_distance(c, &c->obj[0], p)); start = 1; }
    for (int i = start; i < c->n_obj; i++) {
        float d = fabsf(signed_distance(c, &c->obj[i], p));
        if (DEC(RTO_D_NEAREST, d, best, v3_maxabs(p) + c->extent)) { best = d; idx = i; }
    }
    ctr->march_steps++;
    *dist = best;
    return idx;
}

typedef struct { v3 origin, direction, color; int depth; } ray_t;

/* ------------------------------------------------------------------ sphere tracing */

/* examples, plain: cornell_box_v2.py:186-196, cornell_box.py:213-223, shortest:63-72 */
static int raycast_plain(struct rto_ctx* c, const ray_t* ray, v3* pos, int* hit, rtpbr_counters* ctr) {
    float t = c->cfg.min_dis;
    int idx = 0, steps = 0; *hit = 0; *pos = ray->origin;
    for (int i = 0; i < c->cfg.max_raymarch; i++) {
        float d;
        *pos = v3_fma(t, ray->direction, ray->origin);
        idx = nearest(c, *pos, &d, ctr);
        t += d;
        steps = i + 1;
        *hit = DEC(RTO_D_HIT, d, c->cfg.hit_eps, v3_maxabs(*pos) + c->extent);
        if (DEC(RTO_D_ESCAPE, c->cfg.max_dis, t, t) || *hit) break;
    }
    ctr->raycasts++;
    EVENT(RTO_E_RAYCAST, (float)*hit, (float)steps, (float)idx);
    return idx;
}

/* examples, relaxed: cornell_box_v3/pathtracer.py:52-78; tokyo_ibl.py:246-265 (no guard,
 * omega <- 0.5+0.5*omega); bunny_sdf_glass.py:248-267 (omega 0.5 constant) */
static int raycast_relaxed(struct rto_ctx* c, const ray_t* ray, v3* pos, int* hit, rtpbr_counters* ctr) {
    float t = c->cfg.min_dis;
    float w = c->cfg.omega0, s = 0.0f, d = 0.0f;
    int idx = 0, steps = 0; *hit = 0; *pos = ray->origin;
    for (int i = 0; i < c->cfg.max_raymarch; i++) {
        float dist;
        *pos = v3_fma(t, ray->direction, ray->origin);
        idx = nearest(c, *pos, &dist, ctr);
        steps = i + 1;
        float ld = d;
        d = dist;
        const float mag = v3_maxabs(*pos) + c->extent;
        if ((!c->cfg.omega_guard || w > 1.0f) && DEC(RTO_D_FALLBACK, ld + d, s, mag)) {
            s -= w * s;
            t += s;
            w = c->cfg.omega_fb_a + c->cfg.omega_fb_b * w;
            continue;
        }
        float err = d / t;
        s = w * d;
        t += s;
        *hit = DEC(RTO_D_HIT, err, c->cfg.hit_eps, mag / (t - s));      /* |d - t eps| in units of the position's rounding */
        if (DEC(RTO_D_ESCAPE, c->cfg.max_dis, t, t) || *hit) break;
    }
    ctr->raycasts++;
    EVENT(RTO_E_RAYCAST, (float)*hit, (float)steps, (float)idx);
    return idx;
}

/* src/scene.py:59-84: the ray origin itself moves, hit = d < t*PIXEL_RADIUS, depth += 1 */
static int raycast_src(struct rto_ctx* c, ray_t* ray, int* hit, rtpbr_counters* ctr) {
    float t = 0.0f, w = c->cfg.omega0, s = 0.0f, d = c->cfg.max_dis;
    int idx = 0, steps = 0; *hit = 0;
    for (int i = 0; i < c->cfg.max_raymarch; i++) {
        float ld = d;
        idx = nearest(c, ray->origin, &d, ctr);
        steps = i + 1;
        const float mag = v3_maxabs(ray->origin) + c->extent;
        if (w > 1.0f && DEC(RTO_D_FALLBACK, ld + d, s, mag)) {
            s -= w * s;
            w = 1.0f;
            t += s;
            ray->origin = v3_fma(s, ray->direction, ray->origin);
            continue;
        }
        s = w * d;
        t += s;
        ray->origin = v3_fma(s, ray->direction, ray->origin);
        *hit = DEC(RTO_D_HIT, d, t * c->cfg.hit_eps, mag);
        if (*hit || !DEC(RTO_D_ESCAPE, t, c->cfg.max_dis, t)) break;
    }
    EVENT(RTO_E_RAYCAST, (float)*hit, (float)steps, (float)idx);
    ray->depth += 1;
    ctr->raycasts++;
    return idx;
}

/* ------------------------------------------------------------------ normal
 * world: cornell_box_v3/sdf.py:26-31; local: src/sdf.py:77-87 + src/scene.py:87-96 */
static v3 calc_normal(struct rto_ctx* c, const rtpbr_object* o, v3 p) {
    float h = c->cfg.normal_h;
    if (c->cfg.normal_space == RTPBR_NORMAL_WORLD) {
        /* e = vec2(1,-1)*h; e.xyy, e.yyx, e.yxy, e.xxx */
        v3 e0 = v3_make(h, -h, -h), e1 = v3_make(-h, -h, h), e2 = v3_make(-h, h, -h), e3 = v3_make(h, h, h);
        float d0 = signed_distance(c, o, v3_add(p, e0));
        float d1 = signed_distance(c, o, v3_add(p, e1));
        float d2 = signed_distance(c, o, v3_add(p, e2));
        float d3 = signed_distance(c, o, v3_add(p, e3));
        v3 n = v3_add(v3_add(v3_add(v3_scale(e0, d0), v3_scale(e1, d1)), v3_scale(e2, d2)), v3_scale(e3, d3));
        return v3_normalize(n);
    } else {
        v3 q = to_local(c, o, p);
        const float sg[4][3] = {{1, -1, -1}, {-1, -1, 1}, {-1, 1, -1}, {1, 1, 1}};
        v3 n = v3_make(0, 0, 0);
        for (int i = 0; i < 4; i++) {
            v3 e = v3_make(sg[i][0], sg[i][1], sg[i][2]);
            float d = sdf_shape(o->type, v3_add(q, v3_scale(e, h)), o->transform.scale, c->cfg.box_round, c->cfg.max_dis);
            n = v3_add(n, v3_scale(e, d));
        }
        return v3_normalize(n);
    }
}

/* ------------------------------------------------------------------ sampling helpers */
static inline float brightness(v3 c) { return v3_dot(c, v3_make(0.299f, 0.587f, 0.114f)); } /* util.py:31-33 */

/* util.py:21-28 random_in_unit_sphere (surface); pbr.py:16-19 hemispheric_sampling */
static v3 hemispheric_sampling(v3 n, uint32_t key, uint32_t* cnt) {
    float a = rto_rand(key, cnt), b = rto_rand(key, cnt);
    float z = 2.0f * a - 1.0f;
    float ang = b * 2.0f * RTO_PI;
    float sn, cs;
    rto_sincosf(ang, &sn, &cs);
    float sq = sqrtf(1.0f - z * z);
    v3 u = v3_make(sq * sn, sq * cs, z);
    return v3_normalize(v3_add(n, u));
}

/* ------------------------------------------------------------------ surface interaction
 * src/pbr.py:22-62; examples: cornell_box_v3/pbr.py:30-66; shortest:91-94 */
static void surface_interaction(struct rto_ctx* c, ray_t* ray, const rtpbr_object* o, v3 pos,
                                uint32_t key, uint32_t* cnt) {
    const rtpbr_config* g = &c->cfg;
    const rtpbr_material* m = &o->material;
    v3 albedo = v3_make(m->albedo[0], m->albedo[1], m->albedo[2]);
    v3 n = calc_normal(c, o, pos);
    if (g->surface_kind == RTPBR_SURFACE_DIFFUSE) {
        ray->direction = hemispheric_sampling(n, key, cnt);
        ray->color = v3_mul(ray->color, albedo);
        ray->origin = pos;
        EVENT(RTO_E_SURFACE, (float)*cnt, 0.0f, 0.0f);
        EVENT(RTO_E_DIR, ray->direction.x, ray->direction.y, ray->direction.z);
        return;
    }
    v3 I = ray->direction;
    int outer = DEC(RTO_D_OUTER, v3_dot(I, n), 0.0f, 1.0f);
    if (!outer) n = v3_neg(n);
    v3 hemi = hemispheric_sampling(n, key, cnt);
    float alpha = m->roughness * m->roughness;
    v3 N = v3_normalize(v3_mix(n, hemi, alpha));
    float NoI = v3_dot(N, I);
    float eta = outer ? g->env_ior / m->ior : m->ior / g->env_ior;
    float k = 1.0f - eta * eta * (1.0f - NoI * NoI);
    float F0;
    if (g->fresnel_kind == RTPBR_FRESNEL_C2) { F0 = (eta - 1.0f) / (eta + 1.0f); F0 = F0 * (2.0f * F0); }
    else { F0 = 2.0f * (eta - 1.0f) / (eta + 1.0f); F0 = F0 * F0; }
    float x1 = fabsf(1.0f + NoI), x2 = x1 * x1, x5 = x2 * x2 * x1;
    float F = rto_mix(x5, 1.0f, F0);
    if (g->fresnel_roughness_mix) F = rto_mix(F, F0, m->roughness);
    v3 D;
    float c1 = rto_rand(key, cnt);
    if (DEC(RTO_D_REFLECT, c1, F + m->metallic, 1.0f) || DEC(RTO_D_TIR, k, 0.0f, 1.0f)) {
        float tn = 2.0f * NoI;
        D = v3_make(I.x - tn * N.x, I.y - tn * N.y, I.z - tn * N.z);
        if (g->below_horizon == RTPBR_HORIZON_KILL) {
            float keep = DEC(RTO_D_HORIZON, 0.0f, v3_dot(D, n), 1.0f) ? 1.0f : 0.0f;
            ray->color = v3_scale(ray->color, keep);
        } else if (DEC(RTO_D_HORIZON, v3_dot(D, n), 0.0f, 1.0f)) D = v3_neg(D);
    } else {
        float c2 = rto_rand(key, cnt);
        if (DEC(RTO_D_TRANSMIT, c2, m->transmission, 1.0f)) {
            float f = sqrtf(k) + eta * NoI;
            D = v3_make(eta * I.x - f * N.x, eta * I.y - f * N.y, eta * I.z - f * N.z);
        } else D = hemi;
    }
    ray->direction = D;
    ray->color = v3_mul(ray->color, albedo);
    if (g->origin_mode == RTPBR_ORIGIN_HIT) ray->origin = pos;
    else {
        float sgn = DEC(RTO_D_HORIZON, v3_dot(D, n), 0.0f, 1.0f) ? -1.0f : 1.0f;
        v3 off = v3_scale(v3_scale(n, g->min_dis), sgn);
        ray->origin = v3_add(ray->origin, off);
    }
    EVENT(RTO_E_SURFACE, (float)*cnt, 0.0f, 0.0f);
    EVENT(RTO_E_DIR, D.x, D.y, D.z);
#ifndef RTO_NO_DECISIONS
    if (g_inj.rows && g_inj.next < g_inj.n) {
        const float* r = g_inj.rows + 10 * (size_t)g_inj.next++;
        ray->direction = v3_make(r[0], r[1], r[2]);
        ray->color = v3_make(r[3], r[4], r[5]);
        ray->origin = v3_make(r[6], r[7], r[8]);
        *cnt = (uint32_t)r[9];
    }
#endif
}

/* ------------------------------------------------------------------ sky
 * src/ibl.py:25-29,36-40 + src/util.py:45-50; scene_demo/main.py:245-248 gradient */
static v3 sky_color(struct rto_ctx* c, v3 D, rtpbr_counters* ctr) {
    ctr->sky_lookups++;
    if (c->cfg.sky_kind == RTPBR_SKY_GRADIENT) {
        float t = 0.5f * D.y + 0.5f;
        v3 g = v3_mix(v3_make(1.0f, 1.0f, 0.5f), v3_make(0.25f, 0.35f, 1.0f), t);
        return v3_scale(g, 1.8f);
    }
    if (c->cfg.sky_kind == RTPBR_SKY_ENVMAP && c->env) {
        float u = rto_atan2f(D.z, D.x) * RTO_INV_2PI + 0.5f;
        float v = rto_asinf(D.y) * RTO_INV_PI + 0.5f;
        int x = DEC_FLOOR(RTO_D_ENV_X, u * (float)c->env_w, (float)c->env_w), y = DEC_FLOOR(RTO_D_ENV_Y, v * (float)c->env_h, (float)c->env_h);
        if (x < 0) x = 0; if (x > c->env_w - 1) x = c->env_w - 1;   /* G6: clamped */
        if (y < 0) y = 0; if (y > c->env_h - 1) y = c->env_h - 1;
        const float* t = c->env + ((size_t)x * c->env_h + y) * 3;
        return v3_make(t[0], t[1], t[2]);
    }
    return v3_make(0, 0, 0);
}

/* ------------------------------------------------------------------ camera
 * src/camera.py:11-36 get_ray; shortest:102-118 pinhole */
typedef struct { v3 lookfrom, x, y, llc, horizontal, vertical; float lens_radius; } cam_frame;

static void camera_frame(const struct rto_ctx* c, cam_frame* f) {
    const rtpbr_camera* m = &c->cam;
    v3 lf = v3_make(m->lookfrom[0], m->lookfrom[1], m->lookfrom[2]);
    v3 la = v3_make(m->lookat[0], m->lookat[1], m->lookat[2]);
    v3 up = v3_make(m->vup[0], m->vup[1], m->vup[2]);
    float theta = m->vfov * RTO_DEG2RAD;
    float hh = tanf(theta * 0.5f);
    float hw = m->aspect * hh;
    v3 z = v3_normalize(v3_sub(lf, la));
    v3 x = v3_normalize(v3_cross(up, z));
    v3 y = v3_cross(z, x);
    v3 hwfx = v3_scale(x, hw * m->focus);
    v3 hhfy = v3_scale(y, hh * m->focus);
    f->lookfrom = lf; f->x = x; f->y = y;
    f->llc = v3_sub(v3_sub(v3_sub(lf, hwfx), hhfy), v3_scale(z, m->focus));
    f->horizontal = v3_scale(hwfx, 2.0f);
    f->vertical = v3_scale(hhfy, 2.0f);
    f->lens_radius = m->aperture * 0.5f;
}

static void gen_ray(const struct rto_ctx* c, const cam_frame* f, int px, int py,
                    uint32_t key, uint32_t* cnt, ray_t* ray) {
    float j1 = rto_rand(key, cnt), j2 = rto_rand(key, cnt);
    float u, v;
    v3 ro = f->lookfrom;
    if (c->cfg.camera_kind == RTPBR_CAMERA_PINHOLE) {
        u = ((float)px + j1) / (float)c->cfg.width;
        v = ((float)py + j2) / (float)c->cfg.height;
    } else {
        u = ((float)px + j1) * (1.0f / (float)c->cfg.width);
        v = ((float)py + j2) * (1.0f / (float)c->cfg.height);
        float a = rto_rand(key, cnt), b = rto_rand(key, cnt);
        float ang = b * 2.0f * RTO_PI;
        float sn, cs;
        rto_sincosf(ang, &sn, &cs);
        float r = sqrtf(a);
        float rx = f->lens_radius * (r * sn), ry = f->lens_radius * (r * cs);
        v3 off = v3_fma(ry, f->y, v3_scale(f->x, rx));
        ro = v3_add(f->lookfrom, off);
    }
    v3 po = v3_fma(v, f->vertical, v3_fma(u, f->horizontal, f->llc));
    ray->origin = ro;
    ray->direction = v3_normalize(v3_sub(po, ro));
    ray->color = v3_make(1, 1, 1);
    ray->depth = 0;
}

/* ------------------------------------------------------------------ complete-path sample
 * cornell_box_v3/pathtracer.py:81-106 raytrace + renderer.py:31-36 */
static v3 sample_complete_n(struct rto_ctx* c, const cam_frame* f, int px, int py, uint32_t sidx, rtpbr_counters* ctr, uint32_t* draws) {
    const rtpbr_config* g = &c->cfg;
    uint32_t key = rto_rng_key(g->seed, (uint32_t)px, (uint32_t)py, sidx), cnt = 0;
    ray_t ray;
    gen_ray(c, f, px, py, key, &cnt, &ray);
    for (int i = 0; i < g->max_raytrace; i++) {
        float inv_pdf = rto_expf((float)i / g->light_quality);
        float p = 1.0f - 1.0f / inv_pdf;
        if (DEC(RTO_D_ROULETTE, rto_rand(key, &cnt), p, 1.0f)) { ray.color = v3_scale(ray.color, p); break; }
        v3 pos; int hit, idx;
        if (g->march_kind == RTPBR_MARCH_PLAIN) idx = raycast_plain(c, &ray, &pos, &hit, ctr);
        else idx = raycast_relaxed(c, &ray, &pos, &hit, ctr);
        if (!hit) {
            if (g->sky_kind == RTPBR_SKY_BLACK) ray.color = v3_make(0, 0, 0);
            else if (i == 0 && g->primary_miss == RTPBR_PRIMARY_BLACK) ray.color = v3_make(0, 0, 0);
            else if (i == 0 && g->primary_miss == RTPBR_PRIMARY_WHITE) { /* color stays */ }
            else ray.color = v3_mul(ray.color, sky_color(c, ray.direction, ctr));
            break;
        }
        const rtpbr_object* o = &c->obj[idx];
        surface_interaction(c, &ray, o, pos, key, &cnt);
        ctr->hits++;
        float intensity = brightness(ray.color);
        ray.color = v3_mul(ray.color, v3_make(o->material.emission[0], o->material.emission[1], o->material.emission[2]));
        float visible = brightness(ray.color);
        if (DEC(RTO_D_STOP_GAIN, intensity, visible, fmaxf(intensity, visible)) || DEC(RTO_D_STOP_LO, visible, g->vis_lo, g->vis_lo)
            || DEC(RTO_D_STOP_HI, g->vis_hi, visible, g->vis_hi)) break;
    }
    ctr->samples++;
    if (draws) *draws = cnt;
    return ray.color;
}
static v3 sample_complete(struct rto_ctx* c, const cam_frame* f, int px, int py, uint32_t sidx, rtpbr_counters* ctr) {
    return sample_complete_n(c, f, px, py, sidx, ctr, NULL);
}

/* ------------------------------------------------------------------ persistent-ray step
 * src/pathtracer.py:16-91: russian_roulette -> track_once -> raytrace */
static void step_persistent(struct rto_ctx* c, const cam_frame* f, int px, int py, uint32_t step, rtpbr_counters* ctr) {
    const rtpbr_config* g = &c->cfg;
    size_t pi = (size_t)px * g->height + py;
    rtpbr_ray* rb = &c->ray_buffer[pi];
    ray_t ray;
    ray.origin = v3_make(rb->origin[0], rb->origin[1], rb->origin[2]);
    ray.direction = v3_make(rb->direction[0], rb->direction[1], rb->direction[2]);
    ray.color = v3_make(rb->color[0], rb->color[1], rb->color[2]);
    ray.depth = rb->depth;
    uint32_t key = rto_rng_key(g->seed, (uint32_t)px, (uint32_t)py, step), cnt = 0;
    /* russian_roulette :65-77 */
    float p = (ray.depth == 0) ? 1.0f : g->quality_per_sample;
    p -= (float)ray.depth * (1.0f / (float)g->max_raytrace);
    if (DEC(RTO_D_ROULETTE, p, rto_rand(key, &cnt), 1.0f)) {
        ray.color = v3_make(0, 0, 0);
        ray.depth = -ray.depth;
    } else {
        ray.color = v3_scale(ray.color, 1.0f / p);
        /* track_once :53-62 */
        if (ray.depth < 1 || ray.depth > g->max_raytrace) {
            float* ib = c->image_buffer + pi * 4;
            ib[0] += ray.color.x; ib[1] += ray.color.y; ib[2] += ray.color.z; ib[3] += 1.0f;
            ctr->deposits++;
            gen_ray(c, f, px, py, key, &cnt, &ray);
        }
        /* raytrace :16-36 */
        int hit;
        int idx = raycast_src(c, &ray, &hit, ctr);
        if (hit) {
            const rtpbr_object* o = &c->obj[idx];
            surface_interaction(c, &ray, o, ray.origin, key, &cnt);
            ctr->hits++;
            float intensity = brightness(ray.color);
            ray.color = v3_mul(ray.color, v3_make(o->material.emission[0], o->material.emission[1], o->material.emission[2]));
            float visible = brightness(ray.color);
            int stop = DEC(RTO_D_STOP_GAIN, intensity, visible, fmaxf(intensity, visible)) || DEC(RTO_D_STOP_LO, visible, g->vis_lo, g->vis_lo)
                       || DEC(RTO_D_STOP_HI, g->vis_hi, visible, g->vis_hi);
            if (stop) ray.depth = -ray.depth;
        } else {
            ray.depth = -ray.depth;
            ray.color = v3_mul(ray.color, sky_color(c, ray.direction, ctr));
            if (g->primary_miss == RTPBR_PRIMARY_BLACK)
                ray.color = v3_scale(ray.color, ray.depth < -1 ? 1.0f : 0.0f);
        }
    }
    rb->origin[0] = ray.origin.x; rb->origin[1] = ray.origin.y; rb->origin[2] = ray.origin.z;
    rb->direction[0] = ray.direction.x; rb->direction[1] = ray.direction.y; rb->direction[2] = ray.direction.z;
    rb->color[0] = ray.color.x; rb->color[1] = ray.color.y; rb->color[2] = ray.color.z;
    rb->depth = ray.depth;
    ctr->samples++;
}

/* ------------------------------------------------------------------ tone map
 * src/postprocessor.py:12-38, src/aces.py:5-30; per-variant order SURVEY.md A.9 */
static v3 aces_fit(v3 c, int trunc) {
    float a1 = trunc ? 0.024578f : 0.0245786f, a2 = trunc ? 0.0000905f : 0.000090537f;
    const float mi[9] = {0.59719f, 0.35458f, 0.04823f, 0.07600f, 0.90834f, 0.01566f, 0.02840f, 0.13383f, 0.83777f};
    float mo[9] = {1.60475f, -0.53108f, -0.07367f, -0.10208f, 1.10813f, -0.00605f, -0.00327f, -0.07276f, 1.07602f};
    if (trunc) { mo[1] = -0.531f; mo[2] = -0.0736f; mo[3] = -0.102f; }
    v3 v = m3_mulv(mi, c);
    float in[3] = {v.x, v.y, v.z}, out[3];
    for (int i = 0; i < 3; i++) {
        float x = in[i];
        float a = x * (x + a1) - a2;
        float b = x * (0.983729f * x + 0.4329510f) + 0.238081f;
        out[i] = a / b;
    }
    return m3_mulv(mo, v3_make(out[0], out[1], out[2]));
}
static inline float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline v3 v3_clamp01(v3 c) { return v3_make(clamp01(c.x), clamp01(c.y), clamp01(c.z)); }
static inline v3 v3_pow(v3 c, float e) { return v3_make(rto_powf(c.x, e), rto_powf(c.y, e), rto_powf(c.z, e)); }

static v3 tone_map(const rtpbr_config* g, const float* b) {
    v3 c = v3_make(b[0] / b[3], b[1] / b[3], b[2] / b[3]);
    c = v3_scale(c, g->exposure);
    float ig = 1.0f / g->gamma;
    switch (g->tonemap_order) {
    case RTPBR_TONEMAP_GAMMA_ACES_CLAMP: return v3_clamp01(aces_fit(v3_pow(c, ig), g->aces_truncated));
    case RTPBR_TONEMAP_ACES_GAMMA:       return v3_pow(aces_fit(c, g->aces_truncated), ig);
    case RTPBR_TONEMAP_ACES_CLAMP_GAMMA: return v3_pow(v3_clamp01(aces_fit(c, g->aces_truncated)), ig);
    default:                             return v3_clamp01(v3_pow(aces_fit(c, g->aces_truncated), ig));
    }
}

/* ------------------------------------------------------------------ bunny MLP
 * bunny_sdf_glass.py:149-203.  Weight layout of g_bunny (see tools/extract_bunny_weights.py):
 *   layer 0: 4 blocks x {wy[4], wz[4], wx[4], b[4]}           (64)   f0k = sin(p.y*wy + p.z*wz - p.x*wx + b)
 *   layer 1: 4 blocks x {4 mat4 (row major, 64), bias[4]}     (272)  f1k = sin(sum_j f0j @ M_kj + b)/1.0 + f0k
 *   layer 2: 4 blocks x {4 mat4, bias[4]}                     (272)  f2k = sin(sum_j f1j @ M_kj + b)/1.4 + f1k
 *   output : 4 x vec4 + bias                                   (17)
 * v @ M is the row-vector product: (v@M)_j = sum_i v_i M_ij.
 * Sums are evaluated as fma chains (order: see below) and the activations with rto_sin_pi (exactly specified). */
static float sd_bunny(v3 p) {
    float len = v3_length(p);
    if (DEC(RTO_D_BOUND, 1.0f, len, 1.0f)) return len - 0.8f;
    if (!g_bunny_set) return len - 0.8f;
    t_mlp_evals++;
    const float* w = g_bunny;
    float f0[16], f1[16], f2[16];
    /* layer 0: sin(p.y*wy + p.z*wz - p.x*wx + b), as one fma chain */
    for (int k = 0; k < 4; k++) {
        const float* b = w + k * 16;
        for (int j = 0; j < 4; j++) {
            float a = fmaf(p.z, b[4 + j], fmaf(p.y, b[j], 0.0f));
            a = fmaf(-p.x, b[8 + j], a);
            f0[k * 4 + j] = rto_sin_pi(a + b[12 + j]);
        }
    }
    /* layers 1, 2: out_kj = sin(bias + sum_m sum_i v_mi * M_m[i][j]) [/1.4] + in_kj.  The 16-term sum is ONE fma
     * chain that STARTS FROM THE BIAS, in the order (i outer, m inner) — the order in which a 16x16x4 f32 MFMA that
     * takes the bias as its C operand and the previous layer's result registers directly as its B operand accumulates
     * (rt_device.hpp bunny_mlp_wave); "sin(..) / 1.4 + in" is one fma with f32(1/1.4) (what LLVM's arcp + contract
     * fast-math flags, on by default in Taichi, make of it).  All are choices inside the rounding freedom the
     * reference leaves;
     * tests/test_oracle_refpin.py::test_bunny_sdf_and_raycast holds the result to 2e-6 of the reference's own code. */
    const float* src = f0; float* dst = f1;
    for (int layer = 0; layer < 2; layer++) {
        const float* lw = w + 64 + layer * 272;
        for (int k = 0; k < 4; k++) {
            const float* bw = lw + k * 68;
            for (int j = 0; j < 4; j++) {
                float acc = bw[64 + j];                          /* the accumulator starts from the bias */
                for (int i = 0; i < 4; i++)
                    for (int m = 0; m < 4; m++) acc = fmaf(src[m * 4 + i], bw[m * 16 + i * 4 + j], acc);
                float sn = rto_sin_pi(acc);
                dst[k * 4 + j] = layer == 1 ? fmaf(sn, 0.714285731f, src[k * 4 + j])      /* f32(1/1.4), contracted */
                                            : sn + src[k * 4 + j];
            }
        }
        src = f1; dst = f2;
    }
    /* output: dot(f2, ow) - 0.16 */
    const float* ow = w + 64 + 544;
    float sd = f2[0] * ow[0];
    for (int t = 1; t < 16; t++) sd = fmaf(f2[t], ow[t], sd);
    return sd + ow[16];
}

/* ================================================================== C interface */
const char* rto_last_error(void) { return g_err; }
const char* rto_backend(void) { return "cpu-oracle"; }

int rto_create(int device, struct rto_ctx** out) {
    (void)device;
    if (!out) return fail(RTPBR_EINVAL, "out is NULL");
    struct rto_ctx* c = (struct rto_ctx*)calloc(1, sizeof *c);
    if (!c) return fail(RTPBR_ENOMEM, "calloc");
    c->world = 1;
    *out = c;
    return RTPBR_OK;
}
int rto_destroy(struct rto_ctx* c) {
    if (!c) return RTPBR_OK;
    free(c->env); free(c->image_buffer); free(c->image_pixels); free(c->ray_buffer);
    free(c->diff_buffer); free(c->diff_pixels); free(c);
    return RTPBR_OK;
}
int rto_set_config(struct rto_ctx* c, const rtpbr_config* cfg) {
    if (!c || !cfg) return fail(RTPBR_EINVAL, "null");
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->width > 65535 || cfg->height > 65535) return fail(RTPBR_EINVAL, "bad resolution");
    int realloc_buf = !c->have_cfg || c->cfg.width != cfg->width || c->cfg.height != cfg->height;
    c->cfg = *cfg;
    c->have_cfg = 1;
    if (realloc_buf) {
        size_t P = (size_t)cfg->width * cfg->height;
        free(c->image_buffer); free(c->image_pixels); free(c->ray_buffer); free(c->diff_buffer); free(c->diff_pixels);
        c->image_buffer = (float*)calloc(P * 4, sizeof(float));
        c->image_pixels = (float*)calloc(P * 3, sizeof(float));
        c->ray_buffer = (rtpbr_ray*)calloc(P, sizeof(rtpbr_ray));
        c->diff_buffer = (float*)calloc(P * 2, sizeof(float));
        c->diff_pixels = (float*)calloc(P, sizeof(float));
        if (!c->image_buffer || !c->image_pixels || !c->ray_buffer || !c->diff_buffer || !c->diff_pixels) return fail(RTPBR_ENOMEM, "buffers");
    }
    return RTPBR_OK;
}
int rto_set_scene(struct rto_ctx* c, const rtpbr_object* objs, int n, int scale10) {
    if (!c || !objs || n <= 0 || n > RTPBR_MAX_OBJECTS) return fail(RTPBR_EINVAL, "bad scene");
    for (int i = 0; i < n; i++) {
        c->obj[i] = objs[i];
        rtpbr_transform* t = &c->obj[i].transform;
        if (scale10) for (int k = 0; k < 3; k++) { t->position[k] *= 10.0f; t->scale[k] *= 10.0f; }
        float rad[3] = {t->rotation[0] * RTO_DEG2RAD, t->rotation[1] * RTO_DEG2RAD, t->rotation[2] * RTO_DEG2RAD};
        rto_rotate(rad, t->matrix);
    }
    c->n_obj = n;
    c->extent = 1.0f;
    for (int i = 0; i < n; i++) {
        const rtpbr_transform* t = &c->obj[i].transform;
        for (int k = 0; k < 3; k++) c->extent = fmaxf(c->extent, fabsf(t->position[k]) + fabsf(t->scale[k]));
    }
    c->have_scene = 1;
    return RTPBR_OK;
}
int rto_get_scene(struct rto_ctx* c, rtpbr_object* objs, int n) {
    if (!c || !objs || n > c->n_obj) return fail(RTPBR_EINVAL, "bad get_scene");
    memcpy(objs, c->obj, (size_t)n * sizeof *objs);
    return RTPBR_OK;
}
int rto_set_camera(struct rto_ctx* c, const rtpbr_camera* cam) {
    if (!c || !cam) return fail(RTPBR_EINVAL, "null");
    c->cam = *cam; c->have_cam = 1;
    return RTPBR_OK;
}
/* src/ibl.py:14-23 + postprocessor.adjust :17-21: (c/255*exposure)^gamma */
int rto_set_env(struct rto_ctx* c, const void* texels, int w, int h, int fmt, float exposure, float gamma) {
    if (!c || !texels || w <= 0 || h <= 0) return fail(RTPBR_EINVAL, "bad env");
    free(c->env);
    size_t n = (size_t)w * h * 3;
    c->env = (float*)malloc(n * sizeof(float));
    if (!c->env) return fail(RTPBR_ENOMEM, "env");
    if (fmt == RTPBR_ENV_RGB8) {
        const uint8_t* s = (const uint8_t*)texels;
        float lut[256];
        for (int i = 0; i < 256; i++) lut[i] = rto_powf(((float)i / 255.0f) * exposure, gamma);
        for (size_t i = 0; i < n; i++) c->env[i] = lut[s[i]];
    } else memcpy(c->env, texels, n * sizeof(float));
    c->env_w = w; c->env_h = h;
    return RTPBR_OK;
}
int rto_set_tiles(struct rto_ctx* c, int tw, int th, int rank, int world) {
    if (!c || world < 1 || rank < 0 || rank >= world) return fail(RTPBR_EINVAL, "bad tiles");
    if (world > 1 && (tw <= 0 || th <= 0)) return fail(RTPBR_EINVAL, "bad tile size");
    c->tile_w = tw; c->tile_h = th; c->rank = rank; c->world = world;
    return RTPBR_OK;
}
static inline int owns(const struct rto_ctx* c, int x, int y) {
    if (c->world <= 1) return 1;
    int ntx = (c->cfg.width + c->tile_w - 1) / c->tile_w;
    int t = (y / c->tile_h) * ntx + (x / c->tile_w);
    return t % c->world == c->rank;
}
/* src/renderer.py:12-22 */
int rto_refresh(struct rto_ctx* c) {
    if (!c || !c->have_cfg) return fail(RTPBR_ESTATE, "no config");
    size_t P = (size_t)c->cfg.width * c->cfg.height;
    memset(c->image_buffer, 0, P * 4 * sizeof(float));
    for (size_t i = 0; i < P; i++) c->ray_buffer[i].depth = 0;
    if (c->cfg.adaptive_sampling)                              /* src/renderer.py:19-21 */
        for (size_t i = 0; i < P; i++) { c->diff_buffer[i * 2] = 1.0f; c->diff_buffer[i * 2 + 1] = 1.0f; c->diff_pixels[i] = 1e32f; }
    return RTPBR_OK;
}
int rto_set_threads(struct rto_ctx* c, int n) { c->threads = n; return RTPBR_OK; }

int rto_sample(struct rto_ctx* c, int n) {
    if (!c || !c->have_cfg || !c->have_scene || !c->have_cam) return fail(RTPBR_ESTATE, "config/scene/camera missing");
    if (n < 0) return fail(RTPBR_EINVAL, "n < 0");
    if (c->cfg.sky_kind == RTPBR_SKY_ENVMAP && !c->env) return fail(RTPBR_ESTATE, "sky_kind ENVMAP needs set_env first");
    for (int i = 0; i < c->n_obj; i++)
        if (c->obj[i].type == RTPBR_SHAPE_BUNNY && !g_bunny_set) return fail(RTPBR_ESTATE, "bunny shape needs set_shape_data first");
    const int W = c->cfg.width, H = c->cfg.height;
    cam_frame f;
    camera_frame(c, &f);
    rtpbr_counters tot; memset(&tot, 0, sizeof tot);
    unsigned long long mlp_total = 0;
    int persistent = c->cfg.kernel_form == RTPBR_FORM_PERSISTENT_RAY;
    int steps = persistent ? n * c->cfg.steps_per_launch : n;
    uint32_t base = c->sample_base;
#ifdef _OPENMP
    int nt = c->threads > 0 ? c->threads : omp_get_max_threads();
#pragma omp parallel num_threads(nt)
#endif
    {
        rtpbr_counters ctr; memset(&ctr, 0, sizeof ctr);
        t_mlp_evals = 0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int x = 0; x < W; x++) {
            for (int y = 0; y < H; y++) {
                if (!owns(c, x, y)) continue;
                if (persistent) {
                    /* src/pathtracer.py:97-101: adaptive mask, evaluated once per pathtrace() launch;
                     * diff_pixels only changes in post_process(), so once per call is the same */
                    if (c->cfg.adaptive_sampling && !(c->diff_pixels[(size_t)x * H + y] > c->cfg.noise_threshold)) continue;
                    for (int s = 0; s < steps; s++) step_persistent(c, &f, x, y, base + (uint32_t)s, &ctr);
                } else {
                    float* ib = c->image_buffer + ((size_t)x * H + y) * 4;
                    for (int s = 0; s < steps; s++) {
                        v3 col = sample_complete(c, &f, x, y, base + (uint32_t)s, &ctr);
                        ib[0] += col.x; ib[1] += col.y; ib[2] += col.z; ib[3] += 1.0f;
                        ctr.deposits++;
                    }
                }
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            tot.samples += ctr.samples; tot.raycasts += ctr.raycasts; tot.march_steps += ctr.march_steps;
            tot.hits += ctr.hits; tot.sky_lookups += ctr.sky_lookups; tot.deposits += ctr.deposits;
            mlp_total += t_mlp_evals;
        }
    }
    c->sample_base = base + (uint32_t)steps;
    c->ctr = tot;
    c->mlp_evals = mlp_total;
    return RTPBR_OK;
}
int rto_post_process(struct rto_ctx* c) {
    if (!c || !c->have_cfg) return fail(RTPBR_ESTATE, "no config");
    size_t P = (size_t)c->cfg.width * c->cfg.height;
    for (size_t i = 0; i < P; i++) {
        v3 last = v3_make(c->image_pixels[i * 3], c->image_pixels[i * 3 + 1], c->image_pixels[i * 3 + 2]);
        v3 t = tone_map(&c->cfg, c->image_buffer + i * 4);
        c->image_pixels[i * 3 + 0] = t.x; c->image_pixels[i * 3 + 1] = t.y; c->image_pixels[i * 3 + 2] = t.z;
        if (c->cfg.adaptive_sampling) {                         /* src/postprocessor.py:40-43 */
            v3 dc = v3_make(fabsf(t.x - last.x), fabsf(t.y - last.y), fabsf(t.z - last.z));
            c->diff_buffer[i * 2] += brightness(dc);
            c->diff_buffer[i * 2 + 1] += 1.0f;
            c->diff_pixels[i] = c->diff_buffer[i * 2] / c->diff_buffer[i * 2 + 1];
        }
    }
    return RTPBR_OK;
}
int rto_sync(struct rto_ctx* c) { (void)c; return RTPBR_OK; }

static int buf_ptr(struct rto_ctx* c, int which, void** p, size_t* n) {
    if (!c || !c->have_cfg) return fail(RTPBR_ESTATE, "no config");
    size_t P = (size_t)c->cfg.width * c->cfg.height;
    switch (which) {
    case RTPBR_BUF_IMAGE_BUFFER: *p = c->image_buffer; *n = P * 16; return 0;
    case RTPBR_BUF_IMAGE_PIXELS: *p = c->image_pixels; *n = P * 12; return 0;
    case RTPBR_BUF_RAY_BUFFER:   *p = c->ray_buffer;   *n = P * sizeof(rtpbr_ray); return 0;
    case RTPBR_BUF_DIFF_BUFFER:  *p = c->diff_buffer;  *n = P * 8; return 0;
    case RTPBR_BUF_DIFF_PIXELS:  *p = c->diff_pixels;  *n = P * 4; return 0;
    }
    return fail(RTPBR_EINVAL, "bad buffer id");
}
int rto_read_buffer(struct rto_ctx* c, int which, void* dst, size_t nbytes) {
    void* p; size_t n; int r = buf_ptr(c, which, &p, &n); if (r) return r;
    if (nbytes != n) return fail(RTPBR_EINVAL, "size mismatch");
    memcpy(dst, p, n); return RTPBR_OK;
}
int rto_write_buffer(struct rto_ctx* c, int which, const void* src, size_t nbytes) {
    void* p; size_t n; int r = buf_ptr(c, which, &p, &n); if (r) return r;
    if (nbytes != n) return fail(RTPBR_EINVAL, "size mismatch");
    memcpy(p, src, n); return RTPBR_OK;
}
int rto_get_counters(struct rto_ctx* c, rtpbr_counters* out) { *out = c->ctr; return RTPBR_OK; }
int rto_get_counter(struct rto_ctx* c, const char* name, unsigned long long* out) {
    if (!c || !name || !out) return fail(RTPBR_EINVAL, "null");
    if (!strcmp(name, "samples")) *out = c->ctr.samples;
    else if (!strcmp(name, "raycasts")) *out = c->ctr.raycasts;
    else if (!strcmp(name, "march_steps")) *out = c->ctr.march_steps;
    else if (!strcmp(name, "hits")) *out = c->ctr.hits;
    else if (!strcmp(name, "sky_lookups")) *out = c->ctr.sky_lookups;
    else if (!strcmp(name, "deposits")) *out = c->ctr.deposits;
    else if (!strcmp(name, "mlp_lane_evals")) *out = c->mlp_evals;       /* the oracle evaluates ray by ray */
    else if (!strcmp(name, "mlp_wave_evals")) *out = 0;
    else return fail(RTPBR_EINVAL, "unknown counter");
    return RTPBR_OK;
}
int rto_set_sample_base(struct rto_ctx* c, uint32_t base) { c->sample_base = base; return RTPBR_OK; }
int rto_set_bunny_weights(const float* w, int n) {
    if (n != 625) return fail(RTPBR_EINVAL, "need 625 weights");
    memcpy(g_bunny, w, sizeof g_bunny); g_bunny_set = 1; return RTPBR_OK;
}

int rto_set_shape_data(struct rto_ctx* c, int shape, const float* data, int n) {
    (void)c;
    if (shape != RTPBR_SHAPE_BUNNY) return fail(RTPBR_EINVAL, "only the bunny takes shape data");
    return rto_set_bunny_weights(data, n);
}

/* ------------------------------------------------------------------ test hooks (K1 known answers) */
float rto_test_sdf(int type, const float* p, const float* s, float rho) {
    return sdf_shape(type, v3_make(p[0], p[1], p[2]), s, rho, 1e3f);
}
void rto_test_sincos(float a, float* s, float* c) { rto_sincosf(a, s, c); }
float rto_test_exp(float x) { return rto_expf(x); }
float rto_test_sin_pi(float x) { return rto_sin_pi(x); }
float rto_test_log(float x) { return rto_logf(x); }
float rto_test_pow(float x, float y) { return rto_powf(x, y); }
float rto_test_atan2(float y, float x) { return rto_atan2f(y, x); }
float rto_test_asin(float x) { return rto_asinf(x); }
float rto_test_rand(uint32_t seed, uint32_t x, uint32_t y, uint32_t s, uint32_t n) {
    uint32_t k = rto_rng_key(seed, x, y, s); return rto_rand(k, &n);
}
void rto_test_spherical_map(const float* d, float* uv) {
    uv[0] = rto_atan2f(d[2], d[0]) * RTO_INV_2PI + 0.5f;
    uv[1] = rto_asinf(d[1]) * RTO_INV_PI + 0.5f;
}
void rto_test_aces(const float* in, int trunc, float* out) {
    v3 r = aces_fit(v3_make(in[0], in[1], in[2]), trunc); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void rto_test_tonemap(const rtpbr_config* g, const float* rgba, float* out) {
    v3 r = tone_map(g, rgba); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float rto_test_brightness(const float* c) { return brightness(v3_make(c[0], c[1], c[2])); }
/* camera ray for pixel (px,py), sample sidx: origin[3], direction[3] */
int rto_test_get_ray(struct rto_ctx* c, int px, int py, uint32_t sidx, float* out6) {
    cam_frame f; camera_frame(c, &f);
    uint32_t key = rto_rng_key(c->cfg.seed, (uint32_t)px, (uint32_t)py, sidx), cnt = 0;
    ray_t r; gen_ray(c, &f, px, py, key, &cnt, &r);
    out6[0] = r.origin.x; out6[1] = r.origin.y; out6[2] = r.origin.z;
    out6[3] = r.direction.x; out6[4] = r.direction.y; out6[5] = r.direction.z;
    return RTPBR_OK;
}
int rto_test_normal(struct rto_ctx* c, int obj, const float* p, float* n3) {
    v3 n = calc_normal(c, &c->obj[obj], v3_make(p[0], p[1], p[2]));
    n3[0] = n.x; n3[1] = n.y; n3[2] = n.z; return RTPBR_OK;
}
int rto_test_nearest(struct rto_ctx* c, const float* p, float* dist) {
    rtpbr_counters k; memset(&k, 0, sizeof k);
    return nearest(c, v3_make(p[0], p[1], p[2]), dist, &k);
}
/* one raycast from (o,d): returns object index, fills pos[3], hit, steps */
int rto_test_raycast(struct rto_ctx* c, const float* o, const float* d, float* pos3, int* hit, int* steps) {
    rtpbr_counters k; memset(&k, 0, sizeof k);
    ray_t r; r.origin = v3_make(o[0], o[1], o[2]); r.direction = v3_make(d[0], d[1], d[2]); r.color = v3_make(1, 1, 1); r.depth = 0;
    v3 pos; int idx;
    if (c->cfg.march_kind == RTPBR_MARCH_SRC) { idx = raycast_src(c, &r, hit, &k); pos = r.origin; }
    else if (c->cfg.march_kind == RTPBR_MARCH_PLAIN) idx = raycast_plain(c, &r, &pos, hit, &k);
    else idx = raycast_relaxed(c, &r, &pos, hit, &k);
    pos3[0] = pos.x; pos3[1] = pos.y; pos3[2] = pos.z; *steps = (int)k.march_steps;
    return idx;
}
/* surface interaction at pos for object obj with incoming dir; stream (px,py,sidx) from draw n0.
 * out: direction[3], color[3], origin[3] */
int rto_test_surface(struct rto_ctx* c, int obj, const float* pos, const float* dir, uint32_t sidx, float* out9) {
    ray_t r; r.origin = v3_make(pos[0], pos[1], pos[2]); r.direction = v3_make(dir[0], dir[1], dir[2]);
    r.color = v3_make(1, 1, 1); r.depth = 1;
    uint32_t key = rto_rng_key(c->cfg.seed, 0, 0, sidx), cnt = 0;
    surface_interaction(c, &r, &c->obj[obj], r.origin, key, &cnt);
    out9[0] = r.direction.x; out9[1] = r.direction.y; out9[2] = r.direction.z;
    out9[3] = r.color.x; out9[4] = r.color.y; out9[5] = r.color.z;
    out9[6] = r.origin.x; out9[7] = r.origin.y; out9[8] = r.origin.z;
    return (int)cnt;
}
float rto_test_bunny(const float* p) { return sd_bunny(v3_make(p[0], p[1], p[2])); }

/* ---- hooks for tests/test_oracle_refpin.py (fixtures produced by the reference's own code) ---- */
float rto_test_signed_distance(struct rto_ctx* c, int obj, const float* p) {
    return signed_distance(c, &c->obj[obj], v3_make(p[0], p[1], p[2]));
}
/* one complete-path sample: colour, and {raycasts, march steps, RNG draws} */
int rto_test_sample(struct rto_ctx* c, int px, int py, uint32_t sidx, float* color3, uint32_t* stats3) {
    cam_frame f; camera_frame(c, &f);
    rtpbr_counters k; memset(&k, 0, sizeof k);
    uint32_t draws = 0;
    v3 col = sample_complete_n(c, &f, px, py, sidx, &k, &draws);
    color3[0] = col.x; color3[1] = col.y; color3[2] = col.z;
    stats3[0] = (uint32_t)k.raycasts; stats3[1] = (uint32_t)k.march_steps; stats3[2] = draws;
    return RTPBR_OK;
}
/* Decision session for ANY of the rto_test_* hooks called next on this thread (ctypes calls stay on the caller's thread):
 * begin() arms the log / the flips / the surface-output injection, end() disarms and returns the number of log entries. */
int rto_test_decisions_begin(const int* flips, int n_flips, rto_decision* log, int cap, const float* inject_rows, int n_inject) {
#ifdef RTO_NO_DECISIONS
    (void)flips; (void)n_flips; (void)log; (void)cap; (void)inject_rows; (void)n_inject;
    return fail(RTPBR_ESTATE, "built without the decision log");
#else
    g_dec.active = 1; g_dec.n = 0; g_dec.cap = cap; g_dec.log = log; g_dec.flip = flips; g_dec.n_flip = n_flips;
    g_inj.rows = inject_rows; g_inj.n = n_inject; g_inj.next = 0;
    return RTPBR_OK;
#endif
}
int rto_test_decisions_end(void) {
#ifdef RTO_NO_DECISIONS
    return 0;
#else
    const int n = g_dec.n;
    g_dec.active = 0; g_dec.log = NULL; g_dec.n_flip = 0; g_dec.flip = NULL;
    g_inj.rows = NULL; g_inj.n = g_inj.next = 0;
    return n;
#endif
}
/* The same sample with the decision log on: decisions whose ordinals are listed in flips[] are taken the other way.
 * log (may be NULL) receives up to cap entries; *n_log = number of decisions and events the sample went through. */
int rto_test_sample_decisions(struct rto_ctx* c, int px, int py, uint32_t sidx, const int* flips, int n_flips,
                              rto_decision* log, int cap, int* n_log, float* color3, uint32_t* stats3) {
#ifdef RTO_NO_DECISIONS
    (void)c; (void)px; (void)py; (void)sidx; (void)flips; (void)n_flips; (void)log; (void)cap; (void)n_log; (void)color3; (void)stats3;
    return fail(RTPBR_ESTATE, "built without the decision log");
#else
    g_dec.active = 1; g_dec.n = 0; g_dec.cap = cap; g_dec.log = log; g_dec.flip = flips; g_dec.n_flip = n_flips;
    int r = rto_test_sample(c, px, py, sidx, color3, stats3);
    if (n_log) *n_log = g_dec.n;
    g_dec.active = 0; g_dec.log = NULL; g_dec.n_flip = 0;
    return r;
#endif
}
/* One bounce-step of the persistent-ray form for one pixel from a given ray state (10 words: origin, direction, colour,
 * depth bits): returns the new state and what the step deposited (4 floats, zero if nothing).  The context's own buffers
 * are left as they were.  (Runs inside a decision session like every other hook.) */
int rto_test_step(struct rto_ctx* c, int px, int py, uint32_t step, const float* ray_in10, float* ray_out10, float* deposit4) {
    if (!c || !c->ray_buffer || px < 0 || py < 0 || px >= c->cfg.width || py >= c->cfg.height) return fail(RTPBR_EINVAL, "bad pixel");
    cam_frame f; camera_frame(c, &f);
    rtpbr_counters k; memset(&k, 0, sizeof k);
    const size_t pi = (size_t)px * c->cfg.height + py;
    rtpbr_ray saved = c->ray_buffer[pi];
    float ib[4]; memcpy(ib, c->image_buffer + pi * 4, sizeof ib);
    memcpy(&c->ray_buffer[pi], ray_in10, sizeof(rtpbr_ray));
    memset(c->image_buffer + pi * 4, 0, sizeof ib);
    step_persistent(c, &f, px, py, step, &k);
    memcpy(ray_out10, &c->ray_buffer[pi], sizeof(rtpbr_ray));
    memcpy(deposit4, c->image_buffer + pi * 4, sizeof ib);
    c->ray_buffer[pi] = saved;
    memcpy(c->image_buffer + pi * 4, ib, sizeof ib);
    return RTPBR_OK;
}
/* surface interaction with explicit incoming colour and RNG position (stream (px,py,sidx), draw n0).
 * out12: direction, colour, origin, geometric normal as calc_normal returns it.  Returns the draw
 * count after the call. */
int rto_test_surface_at(struct rto_ctx* c, int obj, const float* pos, const float* origin, const float* dir, const float* color,
                        int px, int py, uint32_t sidx, uint32_t n0, float* out12) {
    ray_t r; r.origin = v3_make(origin[0], origin[1], origin[2]); r.direction = v3_make(dir[0], dir[1], dir[2]);
    r.color = v3_make(color[0], color[1], color[2]); r.depth = 1;
    uint32_t key = rto_rng_key(c->cfg.seed, (uint32_t)px, (uint32_t)py, sidx), cnt = n0;
    v3 p = v3_make(pos[0], pos[1], pos[2]);
    v3 n = calc_normal(c, &c->obj[obj], p);
    surface_interaction(c, &r, &c->obj[obj], p, key, &cnt);
    out12[0] = r.direction.x; out12[1] = r.direction.y; out12[2] = r.direction.z;
    out12[3] = r.color.x; out12[4] = r.color.y; out12[5] = r.color.z;
    out12[6] = r.origin.x; out12[7] = r.origin.y; out12[8] = r.origin.z;
    out12[9] = n.x; out12[10] = n.y; out12[11] = n.z;
    return (int)cnt;
}
void rto_test_sky(struct rto_ctx* c, const float* d, float* rgb) {
    rtpbr_counters k; memset(&k, 0, sizeof k);
    v3 s = sky_color(c, v3_make(d[0], d[1], d[2]), &k); rgb[0] = s.x; rgb[1] = s.y; rgb[2] = s.z;
}
/* src-form raycast: the ray origin moves; out: origin[3]; returns the object index */
int rto_test_raycast_src(struct rto_ctx* c, const float* o, const float* d, float* origin3, int* hit, int* steps) {
    rtpbr_counters k; memset(&k, 0, sizeof k);
    ray_t r; r.origin = v3_make(o[0], o[1], o[2]); r.direction = v3_make(d[0], d[1], d[2]); r.color = v3_make(1, 1, 1); r.depth = 0;
    int idx = raycast_src(c, &r, hit, &k);
    origin3[0] = r.origin.x; origin3[1] = r.origin.y; origin3[2] = r.origin.z; *steps = (int)k.march_steps;
    return idx;
}
/* camera ray drawn from draw n0 of stream (px,py,sidx): the src form reaches gen_ray after the roulette draw */
int rto_test_get_ray_at(struct rto_ctx* c, int px, int py, uint32_t sidx, uint32_t n0, float* out6) {
    cam_frame f; camera_frame(c, &f);
    uint32_t key = rto_rng_key(c->cfg.seed, (uint32_t)px, (uint32_t)py, sidx), cnt = n0;
    ray_t r; gen_ray(c, &f, px, py, key, &cnt, &r);
    out6[0] = r.origin.x; out6[1] = r.origin.y; out6[2] = r.origin.z;
    out6[3] = r.direction.x; out6[4] = r.direction.y; out6[5] = r.direction.z;
    return (int)cnt;
}
