// rt_math.hpp — exactly specified f32 arithmetic for the gfx950 kernels.
//
// The reference computes in IEEE f32 through Taichi (src/config.py:5 default_fp=ti.f32);
// Taichi's own math library rounding is not reproducible.  This
// build pins every operation to something IEEE-754 defines exactly so that results are a
// pure function of the inputs on any conforming machine:
//   + - * /  sqrt  fma : correctly rounded (hipcc default for f32 divide/sqrt; the
//                        translation unit is compiled with -ffp-contract=off, so the only
//                        fused operations are the __builtin_fmaf written here);
//   sin cos exp atan2 asin : fixed polynomial kernels (Cephes single-precision
//                        coefficients) evaluated with the exact ops above.
// These run once per path segment (lens, hemisphere, roulette, sky), never in the march
// loop, so they cost <1 % of a sample; the march loop itself is mul/add/fma/min/max/sqrt.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_HD __host__ __device__ __forceinline__
#define RT_D __device__ __forceinline__

// RT_FAST_MATH = 1 (run-time compiled instances only, option "precision" = 1): the TOLERANCE flavour.  The reference runs
// under Taichi's default fast-math (src/config.py:5, ti.init without fast_math=False: LLVM reassociation / contraction,
// approximate sin/cos/pow/rsqrt on GPU backends — SURVEY.md D4), i.e. it never promised more than this: square roots and
// reciprocal square roots are the hardware approximations with ONE correction step (v_sqrt / v_rsq), the neural SDF's 48
// sines per evaluation are v_sin_f32, exp / log the hardware v_exp / v_log, products and sums may be
// contracted (-ffp-contract=fast), f32 divide is v_rcp-based (-fno-hip-fp32-correctly-rounded-divide-sqrt), and the exact
// decision bands of the march loop are dropped.  Not bit-exact with anything; held to the north star's per-pixel L2 bar
// against the oracle by tests/test_gpu_fast.py.  The exact flavour (0, default) stays the parity anchor.
#ifndef RT_FAST_MATH
#define RT_FAST_MATH 0
#endif
#if RT_FAST_MATH && !defined(__HIP_DEVICE_COMPILE__)
#undef RT_FAST_MATH
#define RT_FAST_MATH 0          // the host pass only parses the device functions
#endif

namespace rt {

constexpr float PI = 3.14159274f;
constexpr float TWO_OVER_PI = 0.636619747f;
constexpr float INV_2PI = 0.159154937f;   // src/util.py:48  0.5/pi
constexpr float INV_PI = 0.318309873f;
constexpr float DEG2RAD = 0.0174532924f;  // taichi.math.radians

struct vec3 {
    float x, y, z;
};

RT_HD vec3 mk(float x, float y, float z) { return vec3{x, y, z}; }
RT_HD vec3 operator+(vec3 a, vec3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD vec3 operator-(vec3 a, vec3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD vec3 operator*(vec3 a, vec3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD vec3 operator*(vec3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RT_HD vec3 operator-(vec3 a) { return mk(-a.x, -a.y, -a.z); }
RT_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RT_HD float sqrt_(float x);
// a + s*b, fused per component
RT_HD vec3 fma3(float s, vec3 b, vec3 a) { return mk(fma_(s, b.x, a.x), fma_(s, b.y, a.y), fma_(s, b.z, a.z)); }
RT_HD float dot(vec3 a, vec3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
// Correctly rounded sqrt.  Host: libm/IEEE.  Device: v_sqrt_f32 (<= 1 ulp) + the two-candidate
// fma correction LLVM uses for its own IEEE lowering, but with an UNCONDITIONAL exact
// power-of-two pre/post scaling (2^32 / 2^-16) instead of the compare+select denormal and
// zero/inf handling: 11 instructions instead of ~17.  Valid for 0 <= x < 2^95 (every length
// on this path; distances are < MAX_DIS = 2000).  tests/test_gpu_parity.py checks it against
// the compiler's IEEE sqrt for ALL 2^31 non-negative bit patterns below 2^95.
// tolerance flavour: v_sqrt_f32 (<= 1 ulp) + one residual correction y + (x - y^2) / (2 y): within 0.5 ulp + a few 2^-24
// of an ulp, i.e. the correctly rounded result for all but about one argument in a thousand, in 6 instructions instead
// of 11 — a bare v_sqrt_f32 is NOT enough here: |sqrt(s2) - r| cancels against the 100-unit ground sphere of the src /
// Tokyo scenes, 1 ulp of 100 is 7.6e-6, and hit thresholds are ~1e-4 t (measured: 0.8 % of the samples changed colour)
RT_HD float sqrt_fast_(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(RT_FAST_HW_SQRT)
    return __builtin_amdgcn_sqrtf(x);      // all-box scenes: no large cancellation behind the root (see rt_jit.hip)
#elif defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(__builtin_fmaxf(x, 1.0e-36f));
    return __builtin_fmaf(__builtin_fmaf(-y, y, x), h, y);
#else
    return __builtin_sqrtf(x);
#endif
}
RT_HD float sqrt_(float x) {
#if RT_FAST_MATH && !defined(RT_FAST_EXACT_SQRT)
    return sqrt_fast_(x);
#elif defined(__HIP_DEVICE_COMPILE__)
    float xs = x * 4294967296.0f;
    float y = __builtin_amdgcn_sqrtf(xs);
    float ym = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1);
    float yp = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
    float rm = __builtin_fmaf(-ym, y, xs);
    float rp = __builtin_fmaf(-yp, y, xs);
    y = (0.0f >= rm) ? ym : y;
    y = (0.0f < rp) ? yp : y;
    return y * 1.52587890625e-05f;
#else
    return __builtin_sqrtf(x);
#endif
}
// sqrt_(x / 4) for an x that is an exact multiple-of-4 scaling: the same v_sqrt_f32 input (x/4 * 2^32 = x * 2^30)
RT_HD float sqrt_quarter_(float x) {
#if RT_FAST_MATH && !defined(RT_FAST_EXACT_SQRT)
    return 0.5f * sqrt_fast_(x);
#elif defined(__HIP_DEVICE_COMPILE__)
    float xs = x * 1073741824.0f;
    float y = __builtin_amdgcn_sqrtf(xs);
    float ym = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1);
    float yp = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
    float rm = __builtin_fmaf(-ym, y, xs);
    float rp = __builtin_fmaf(-yp, y, xs);
    y = (0.0f >= rm) ? ym : y;
    y = (0.0f < rp) ? yp : y;
    return y * 1.52587890625e-05f;
#else
    return __builtin_sqrtf(x * 0.25f);
#endif
}
// sqrt_(x) + c and sqrt_quarter_(x) + c in one rounding less to ISSUE, not to compute: the root's power-of-two post-scaling is exact
// (no underflow: the root of the smallest denormal is 2^-74.5), so fma(y, 2^-16, c) rounds exactly the sum sqrt_(x) + c that the two
// separate instructions round — one instruction instead of two behind every root whose next operation is an addition (a sphere's
// `- r`, a box's `+ min(max3 q, 0)`, a cylinder's `- r` and `+ min(..)`, the lazy box search's `- rho`).
RT_HD float sqrt_add_(float x, float c) {
#if defined(__HIP_DEVICE_COMPILE__) && !(RT_FAST_MATH && !defined(RT_FAST_EXACT_SQRT))
    float xs = x * 4294967296.0f;
    float y = __builtin_amdgcn_sqrtf(xs);
    float ym = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1);
    float yp = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
    float rm = __builtin_fmaf(-ym, y, xs);
    float rp = __builtin_fmaf(-yp, y, xs);
    y = (0.0f >= rm) ? ym : y;
    y = (0.0f < rp) ? yp : y;
    return __builtin_fmaf(y, 1.52587890625e-05f, c);
#else
    return sqrt_(x) + c;
#endif
}
RT_HD float sqrt_quarter_add_(float x, float c) {
#if defined(__HIP_DEVICE_COMPILE__) && !(RT_FAST_MATH && !defined(RT_FAST_EXACT_SQRT))
    float xs = x * 1073741824.0f;
    float y = __builtin_amdgcn_sqrtf(xs);
    float ym = __builtin_bit_cast(float, __builtin_bit_cast(int, y) - 1);
    float yp = __builtin_bit_cast(float, __builtin_bit_cast(int, y) + 1);
    float rm = __builtin_fmaf(-ym, y, xs);
    float rp = __builtin_fmaf(-yp, y, xs);
    y = (0.0f >= rm) ? ym : y;
    y = (0.0f < rp) ? yp : y;
    return __builtin_fmaf(y, 1.52587890625e-05f, c);
#else
    return sqrt_quarter_(x) + c;
#endif
}
RT_HD float sqrt_ieee_(float x) { return __builtin_sqrtf(x); }
// The root inside a shape's distance.  Exact flavour: sqrt_.  Tolerance flavour: the correction step of sqrt_fast_ only where
// the root cancels against something LARGE (`big`: the shape's radius / half-height above RT_BIG_EXTENT — the 100-unit ground
// spheres of the src/ and Tokyo scenes, whose 1 ulp is 7.6e-6 against hit thresholds of ~1e-4 t); for a shape of a few units
// a bare v_sqrt_f32 is off by less than the rounding of the position that went in.
#define RT_BIG_EXTENT 16.0f
// (tolerance flavour) the root of a LARGE sphere's distance, radius r: v_sqrt_f32 + one residual step whose 1 / (2 sqrt x) is
// taken as 1 / (2 r) — exact enough where it matters (x - y^2 is at most an ulp's worth and sqrt x = r + the distance; far from
// the sphere the step over-corrects by sqrt x / r ulps of a distance that decides nothing): 3 instructions, one quarter-rate,
// instead of sqrt_fast_'s 6 with two
RT_HD float sqrt_big_sphere_(float x, float r) {
#if RT_FAST_MATH && !defined(RT_FAST_EXACT_SQRT) && defined(__HIP_DEVICE_COMPILE__)
    const float y = __builtin_amdgcn_sqrtf(x);
    return __builtin_fmaf(__builtin_fmaf(-y, y, x), 0.5f * __builtin_amdgcn_rcpf(r), y);
#else
    (void)r;
    return sqrt_(x);
#endif
}
RT_HD float sqrt_shape_(float x, bool big) {
#if RT_FAST_MATH && !defined(RT_FAST_EXACT_SQRT) && defined(__HIP_DEVICE_COMPILE__)
    if (!big) return __builtin_amdgcn_sqrtf(x);
#endif
    (void)big;
    return sqrt_(x);
}
// sqrt_shape_(x, big) + c (exact flavour: the fused form above)
RT_HD float sqrt_shape_add_(float x, bool big, float c) {
#if RT_FAST_MATH && !defined(RT_FAST_EXACT_SQRT) && defined(__HIP_DEVICE_COMPILE__)
    return sqrt_shape_(x, big) + c;
#else
    (void)big;
    return sqrt_add_(x, c);
#endif
}
RT_HD float length(vec3 a) { return sqrt_(dot(a, a)); }
RT_HD vec3 normalize(vec3 a) {
#if RT_FAST_MATH
    // v_rsq_f32 + one Newton step (directions feed positions at t ~ 10: 1 ulp of a direction is 1e-6 of a position)
    const float s2 = dot(a, a);
    float inv = __builtin_amdgcn_rsqf(s2);
    inv = inv * __builtin_fmaf(-0.5f * s2, inv * inv, 1.5f);
#else
    float inv = 1.0f / sqrt_(dot(a, a));
#endif
    return a * inv;
}
RT_HD vec3 cross(vec3 a, vec3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_HD float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
RT_HD vec3 mix(vec3 x, vec3 y, float a) {
    float b = 1.0f - a;
    return mk(x.x * b + y.x * a, x.y * b + y.y * a, x.z * b + y.z * a);
}
RT_HD float fmin_(float a, float b) { return __builtin_fminf(a, b); }
RT_HD float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
RT_HD float fabs_(float a) { return __builtin_fabsf(a); }
// row-major 3x3 times column vector
RT_HD vec3 mulv(const float* m, vec3 v) {
    return mk(fma_(m[2], v.z, fma_(m[1], v.y, m[0] * v.x)), fma_(m[5], v.z, fma_(m[4], v.y, m[3] * v.x)),
              fma_(m[8], v.z, fma_(m[7], v.y, m[6] * v.x)));
}

// sin/cos: Cody-Waite reduction by pi/2 + Cephes sinf/cosf kernels on [-pi/4, pi/4]
RT_HD void sincos_(float a, float* s_out, float* c_out) {
    // (the tolerance flavour keeps this polynomial pair: it runs twice per surface interaction, not in the march loop, and
    // v_sin_f32 / v_cos_f32 are good to ~1e-6 ABSOLUTE only — as a direction that is 1e-5 of a position ten units away)
    float kf = __builtin_rintf(a * TWO_OVER_PI);
    int k = (int)kf;
    float r = fma_(kf, -1.5703125f, a);
    r = fma_(kf, -4.83751296997070312e-4f, r);
    r = fma_(kf, -7.54978995489188e-8f, r);
    float r2 = r * r;
    float ps = fma_(fma_(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    float sn = fma_(r * r2, ps, r);
    float pc = fma_(fma_(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    float cs = fma_(r2 * r2, pc, fma_(-0.5f, r2, 1.0f));
    bool swap = (k & 1) != 0;
    float s = swap ? cs : sn;
    float c = swap ? sn : cs;
    if (k & 2) s = -s;
    if ((k + 1) & 2) c = -c;
    *s_out = s;
    *c_out = c;
}
RT_HD float sin_(float a) {
    float s, c;
    sincos_(a, &s, &c);
    return s;
}

// sin for the neural-SDF activations (48 per network evaluation): 11 operations.
// k = round(a/pi) is formed by adding 1.5*2^23: the fma rounds a/pi to the nearest integer in one step and leaves k's
// parity in the lowest mantissa bit of t; reduction by pi in two Cody-Waite terms (3.140625 has 11 significant bits, so
// k*3.140625 is exact for |k| < 2^13; the second term carries pi - 3.140625 to 2^-35: error 6e-11 |k|, the MLP's
// pre-activations reach |k| ~ 100); one odd degree-9 minimax polynomial on [-pi/2, pi/2] (max abs error 1.2e-7); the
// sign is applied with one v_lshl_add_u32 (parity << 31 ADDED to the bit pattern).  Deterministic for every input.
RT_HD float sin_pi_(float a) {
#if RT_FAST_MATH
    return __builtin_amdgcn_sinf(a * INV_2PI);      // |a| stays below ~100 pi in the network: inside the instruction's +-256 revolutions
#endif
    const float magic = 12582912.0f;
    float t = fma_(a, INV_PI, magic);
    float kf = t - magic;
    float r = fma_(kf, -3.140625f, a);
    r = fma_(kf, -9.676535846665502e-4f, r);
    float r2 = r * r;
    float p = fma_(fma_(fma_(2.6073803383042105e-06f, r2, -0.00019809493096545339f), r2, 0.008333046920597553f), r2,
                   -0.16666658222675323f);
    float s = fma_(r * r2, p, r);
    return __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, t) << 31) + __builtin_bit_cast(uint32_t, s));
}

// exp: Cephes expf
RT_HD float exp_(float x) {
#if RT_FAST_MATH
    return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
#endif
    float kf = __builtin_rintf(x * 1.44269504088896341f);
    float r = fma_(kf, -0.693359375f, x);
    r = fma_(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fma_(p, r, 1.3981999507e-3f);
    p = fma_(p, r, 8.3334519073e-3f);
    p = fma_(p, r, 4.1665795894e-2f);
    p = fma_(p, r, 1.6666665459e-1f);
    p = fma_(p, r, 5.0000001201e-1f);
    float e = fma_(p, r * r, r) + 1.0f;
    int k = (int)kf;
    k = k > 127 ? 127 : k;
    k = k < -126 ? -126 : k;
    uint32_t bits = (uint32_t)(k + 127) << 23;
    return e * __builtin_bit_cast(float, bits);
}

// log (Cephes logf) and pow(x, y) = exp(y*log(x)) for x >= 0 (tone map, env preprocess);
// x < 0 gives NaN like powf.
RT_HD float log_(float x) {
#if RT_FAST_MATH
    return __builtin_amdgcn_logf(x) * 0.693147180559945309f;
#endif
    int e = 0;
    if (x < 1.17549435e-38f) {
        x = x * 16777216.0f;
        e = -24;
    }
    uint32_t b = __builtin_bit_cast(uint32_t, x);
    e += (int)((b >> 23) & 0xffu) - 126;
    b = (b & 0x807fffffu) | 0x3f000000u;
    float m = __builtin_bit_cast(float, b);
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = m + m - 1.0f;
    } else {
        m = m - 1.0f;
    }
    float z = m * m;
    float y = 7.0376836292e-2f;
    y = fma_(y, m, -1.1514610310e-1f);
    y = fma_(y, m, 1.1676998740e-1f);
    y = fma_(y, m, -1.2420140846e-1f);
    y = fma_(y, m, 1.4249322787e-1f);
    y = fma_(y, m, -1.6668057665e-1f);
    y = fma_(y, m, 2.0000714765e-1f);
    y = fma_(y, m, -2.4999993993e-1f);
    y = fma_(y, m, 3.3333331174e-1f);
    y = y * m * z;
    float fe = (float)e;
    y = fma_(-2.12194440e-4f, fe, y);
    y = fma_(-0.5f, z, y);
    z = m + y;
    return fma_(0.693359375f, fe, z);
}
RT_HD float pow_(float x, float y) {
    if (x == 0.0f) return 0.0f;
    if (!(x > 0.0f)) return __builtin_nanf("");
    return exp_(y * log_(x));
}

RT_HD float atan_pos_(float x) {
    float y0, z;
    if (x > 2.414213562373095f) {
        y0 = 1.5707963267948966f;
        z = -1.0f / x;
    } else if (x > 0.4142135623730950f) {
        y0 = 0.7853981633974483f;
        z = (x - 1.0f) / (x + 1.0f);
    } else {
        y0 = 0.0f;
        z = x;
    }
    float zz = z * z;
    float p = 8.05374449538e-2f;
    p = fma_(p, zz, -1.38776856032e-1f);
    p = fma_(p, zz, 1.99777106478e-1f);
    p = fma_(p, zz, -3.33329491539e-1f);
    return y0 + fma_(p * zz, z, z);
}
RT_HD float atan2_(float y, float x) {
    if (x == 0.0f && y == 0.0f) return 0.0f;
    float ax = fabs_(x), ay = fabs_(y);
    float a = (ax == 0.0f) ? 1.5707963267948966f : atan_pos_(ay / ax);
    if (x < 0.0f) a = PI - a;
    return (y < 0.0f) ? -a : a;
}
// asin with the argument clamped to [-1,1] (the reference would return NaN outside)
RT_HD float asin_(float x) {
    float a = fabs_(x);
    if (a > 1.0f) a = 1.0f;
    bool big = a > 0.5f;
    float z, w;
    if (big) {
        z = 0.5f * (1.0f - a);
        w = sqrt_(z);
    } else {
        w = a;
        z = a * a;
    }
    float p = 4.2163199048e-2f;
    p = fma_(p, z, 2.4181311049e-2f);
    p = fma_(p, z, 4.5470025998e-2f);
    p = fma_(p, z, 7.4953002686e-2f);
    p = fma_(p, z, 1.6666752422e-1f);
    float r = fma_(p * z, w, w);
    if (big) r = 1.5707963267948966f - (r + r);
    return (x < 0.0f) ? -r : r;
}

// counter-based RNG replacing ti.random(): stream key from
// (seed, pixel x, pixel y, sample index); draw n = murmur3-finalised counter; 24-bit mantissa.
RT_HD uint32_t mix32(uint32_t z) {
    z ^= z >> 16;
    z *= 0x85ebca6bU;
    z ^= z >> 13;
    z *= 0xc2b2ae35U;
    z ^= z >> 16;
    return z;
}
RT_HD uint32_t rng_key(uint32_t seed, uint32_t x, uint32_t y, uint32_t sample) {
    uint32_t k = mix32(seed + 0x9E3779B9U);
    k = mix32(k ^ (x | (y << 16)));
    k = mix32(k ^ sample);
    return k;
}
RT_HD float rng_next(uint32_t key, uint32_t& n) {
    uint32_t z = mix32(key + n * 0x9E3779B9U);
    n++;
    return (float)(z >> 8) * 5.9604644775390625e-8f;
}

}  // namespace rt
