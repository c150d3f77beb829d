/*
 * rt_oracle_math.h — TEST INFRASTRUCTURE (part of the CPU oracle, see rt_oracle.c).
 *
 * Exactly specified f32 arithmetic used by the oracle.  The reference computes in IEEE f32
 * (ti.init(default_fp=ti.f32), src/config.py:5) with Taichi's math library, whose rounding
 * is not reproducible outside Taichi.  To make "same inputs ->
 * identical results" a testable statement between this oracle and the HIP kernel, every
 * operation is pinned to something IEEE-754 defines exactly:
 *   +, -, *, /, sqrtf, fmaf  : correctly rounded (compile with -ffp-contract=off so that
 *                              only the fmaf() written below are fused);
 *   sin, cos, exp, atan2, asin: fixed polynomial kernels (Cephes single precision
 *                              coefficients) written with the exact ops above.
 * tests/test_oracle_math.py pins these against libm/numpy (<= 2 ulp-ish absolute error).
 */
#ifndef RT_ORACLE_MATH_H
#define RT_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#define RTO_PI      3.14159274f          /* f32(pi)                         */
#define RTO_2_PI    0.636619747f         /* f32(2/pi)                       */
#define RTO_INV_2PI 0.159154937f         /* f32(0.5/pi)  util.py:48         */
#define RTO_INV_PI  0.318309873f         /* f32(1/pi)                       */
#define RTO_DEG2RAD 0.0174532924f        /* f32(pi/180)  taichi.math.radians */

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
/* a + s*b, fused per component */
static inline v3 v3_fma(float s, v3 b, v3 a) {
    return v3_make(fmaf(s, b.x, a.x), fmaf(s, b.y, a.y), fmaf(s, b.z, a.z));
}
/* dot = fma(z,z', fma(y,y', x*x')) */
static inline float v3_dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline float v3_length(v3 a) { return sqrtf(v3_dot(a, a)); }
static inline v3 v3_normalize(v3 a) { float inv = 1.0f / sqrtf(v3_dot(a, a)); return v3_scale(a, inv); }
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* taichi.math.mix(x, y, a) = x*(1-a) + y*a */
static inline float rto_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
static inline v3 v3_mix(v3 x, v3 y, float a) {
    float b = 1.0f - a;
    return v3_make(x.x * b + y.x * a, x.y * b + y.y * a, x.z * b + y.z * a);
}
/* row-major 3x3 times column vector */
static inline v3 m3_mulv(const float* m, v3 v) {
    return v3_make(fmaf(m[2], v.z, fmaf(m[1], v.y, m[0] * v.x)),
                   fmaf(m[5], v.z, fmaf(m[4], v.y, m[3] * v.x)),
                   fmaf(m[8], v.z, fmaf(m[7], v.y, m[6] * v.x)));
}

/* ---- sin / cos: Cody-Waite reduction by pi/2 (3 terms) + Cephes sinf/cosf kernels ---- */
static inline void rto_sincosf(float a, float* s_out, float* c_out) {
    float kf = rintf(a * RTO_2_PI);
    int k = (int)kf;
    float r = fmaf(kf, -1.5703125f, a);
    r = fmaf(kf, -4.83751296997070312e-4f, r);
    r = fmaf(kf, -7.54978995489188e-8f, r);
    float r2 = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    float sn = fmaf(r * r2, ps, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    float cs = fmaf(r2 * r2, pc, fmaf(-0.5f, r2, 1.0f));
    float s, c;
    switch (k & 3) {
        case 0: s = sn;  c = cs;  break;
        case 1: s = cs;  c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = s; *c_out = c;
}
static inline float rto_sinf(float a) { float s, c; rto_sincosf(a, &s, &c); return s; }

/* ---- sin for the neural-SDF activations (48 per bunny SDF evaluation): 11 operations.
 * k = round(a/pi) is formed by adding 1.5*2^23: the fma rounds a/pi to the nearest integer in one step and leaves k's
 * parity in the lowest mantissa bit of t; reduction by pi in two Cody-Waite terms (3.140625 has 11 significant bits,
 * so k*3.140625 is exact for |k| < 2^13; the second term carries pi - 3.140625 to 2^-35 absolute: fine for the
 * |k| <= ~100 the MLP's pre-activations reach, error grows as 6e-11 |k|); one odd degree-9 minimax polynomial on
 * [-pi/2, pi/2] (max abs error 1.2e-7); the sign is applied by ADDING parity << 31 to the bit pattern.
 * Defined (deterministically) for every input; accurate for |a| < ~1e3.  tests/test_oracle_math.py pins it. */
static inline float rto_sin_pi(float a) {
    const float magic = 12582912.0f;                       /* 1.5 * 2^23 */
    float t = fmaf(a, RTO_INV_PI, magic);
    float kf = t - magic;
    float r = fmaf(kf, -3.140625f, a);
    r = fmaf(kf, -9.676535846665502e-4f, r);
    float r2 = r * r;
    float p = fmaf(fmaf(fmaf(2.6073803383042105e-06f, r2, -0.00019809493096545339f), r2, 0.008333046920597553f), r2,
                   -0.16666658222675323f);
    float s = fmaf(r * r2, p, r);
    uint32_t tb, sb;
    memcpy(&tb, &t, 4); memcpy(&sb, &s, 4);
    sb += tb << 31;
    memcpy(&s, &sb, 4);
    return s;
}

/* ---- exp: Cephes expf ---- */
static inline float rto_expf(float x) {
    float kf = rintf(x * 1.44269504088896341f);
    float r = fmaf(kf, -0.693359375f, x);
    r = fmaf(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float e = fmaf(p, r * r, r) + 1.0f;
    int k = (int)kf;
    if (k > 127) k = 127;
    if (k < -126) k = -126;
    uint32_t bits = (uint32_t)(k + 127) << 23;
    float sc; memcpy(&sc, &bits, 4);
    return e * sc;
}

/* ---- log (Cephes logf) and pow(x, y) = exp(y*log(x)) for x >= 0: the tone map's and the env
 * preprocess's pow(c, gamma) (src/postprocessor.py:17-21).  x < 0 gives NaN like powf. ---- */
static inline float rto_logf(float x) {
    int e = 0;
    if (x < 1.17549435e-38f) { x = x * 16777216.0f; e = -24; }
    uint32_t b; memcpy(&b, &x, 4);
    e += (int)((b >> 23) & 0xffu) - 126;
    b = (b & 0x807fffffu) | 0x3f000000u;
    float m; memcpy(&m, &b, 4);                       /* m in [0.5, 1) */
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else m = m - 1.0f;
    float z = m * m;
    float y = 7.0376836292e-2f;
    y = fmaf(y, m, -1.1514610310e-1f);
    y = fmaf(y, m, 1.1676998740e-1f);
    y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f);
    y = fmaf(y, m, -1.6668057665e-1f);
    y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f);
    y = fmaf(y, m, 3.3333331174e-1f);
    y = y * m * z;
    float fe = (float)e;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    z = m + y;
    return fmaf(0.693359375f, fe, z);
}
static inline float rto_powf(float x, float y) {
    if (x == 0.0f) return 0.0f;
    if (!(x > 0.0f)) return NAN;
    return rto_expf(y * rto_logf(x));
}

/* ---- atan on [0, inf): Cephes atanf ---- */
static inline float rto_atanf_pos(float x) {
    float y0, z;
    if (x > 2.414213562373095f) { y0 = 1.5707963267948966f; z = -1.0f / x; }
    else if (x > 0.4142135623730950f) { y0 = 0.7853981633974483f; z = (x - 1.0f) / (x + 1.0f); }
    else { y0 = 0.0f; z = x; }
    float zz = z * z;
    float p = 8.05374449538e-2f;
    p = fmaf(p, zz, -1.38776856032e-1f);
    p = fmaf(p, zz, 1.99777106478e-1f);
    p = fmaf(p, zz, -3.33329491539e-1f);
    return y0 + fmaf(p * zz, z, z);
}
static inline float rto_atan2f(float y, float x) {
    if (x == 0.0f && y == 0.0f) return 0.0f;
    float ax = fabsf(x), ay = fabsf(y);
    float a;
    if (ax == 0.0f) a = 1.5707963267948966f;
    else a = rto_atanf_pos(ay / ax);
    if (x < 0.0f) a = RTO_PI - a;
    return (y < 0.0f) ? -a : a;
}

/* ---- asin: Cephes asinf; the argument is clamped to [-1,1] (the reference would give NaN) ---- */
static inline float rto_asinf(float x) {
    float a = fabsf(x);
    if (a > 1.0f) a = 1.0f;
    float z, w;
    int big = a > 0.5f;
    if (big) { z = 0.5f * (1.0f - a); w = sqrtf(z); }
    else { w = a; z = a * a; }
    float p = 4.2163199048e-2f;
    p = fmaf(p, z, 2.4181311049e-2f);
    p = fmaf(p, z, 4.5470025998e-2f);
    p = fmaf(p, z, 7.4953002686e-2f);
    p = fmaf(p, z, 1.6666752422e-1f);
    float r = fmaf(p * z, w, w);
    if (big) r = 1.5707963267948966f - (r + r);
    return (x < 0.0f) ? -r : r;
}

/* ---- counter-based RNG (replaces ti.random()) ----
 * A stream is keyed by (seed, pixel x, pixel y, sample index); draw number n of the stream
 * is a murmur3-finalised counter.  24-bit mantissa uniform in [0,1), like ti.random(f32). */
static inline uint32_t rto_mix32(uint32_t z) {
    z ^= z >> 16; z *= 0x85ebca6bU; z ^= z >> 13; z *= 0xc2b2ae35U; z ^= z >> 16;
    return z;
}
static inline uint32_t rto_rng_key(uint32_t seed, uint32_t x, uint32_t y, uint32_t sample) {
    uint32_t k = rto_mix32(seed + 0x9E3779B9U);
    k = rto_mix32(k ^ (x | (y << 16)));
    k = rto_mix32(k ^ sample);
    return k;
}
static inline float rto_rand(uint32_t key, uint32_t* n) {
    uint32_t z = rto_mix32(key + (*n) * 0x9E3779B9U);
    (*n)++;
    return (float)(z >> 8) * 5.9604644775390625e-8f;
}

#endif
