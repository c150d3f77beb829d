// rt_device.hpp — device functions of the sample path (gfx950, wave64).
//
// Stage functions of SURVEY.md §8(a): F5 camera ray, F7/F21 sphere tracing step, F8 nearest,
// F9/F10 transform + SDF primitives, F11 normal, F12/F22 surface interaction, F13 sky,
// F15 sampling helpers, F19 tone map, F23 neural bunny SDF.  Each cites the reference lines
// whose behaviour it reproduces; arithmetic follows rt_math.hpp (exact ops, fixed order).
#pragma once
#include <cstddef>
#include <type_traits>

#include "rt_types.hpp"

namespace rt {

// scene specialisations: GENERIC = analytic primitives (no neural SDF code), BOXES = all boxes
// (Cornell), BUNNY = all objects are the neural bunny, MIXED = bunny next to analytic primitives
enum { KIND_GENERIC = 0, KIND_BOXES = 1, KIND_BUNNY = 2, KIND_MIXED = 3 };

// ---------------------------------------------------------------- F23 bunny MLP
// examples/bunny/bunny_sdf_glass.py:149-203; weight layout: see tools/extract_bunny_weights.py.
// 3 -> 16 -> 16 -> 16 -> 1 with sine activations and residual adds.  Sums are fma chains in
// (block, row) order, activations use sin_pi_.  The 625 weights are read at wave-uniform
// addresses through a constant-address-space pointer (scalar loads -> SGPR operands); the pointer
// is laundered so the loads are not hoisted out of the march loop (that would spill them).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) float* CFloatPtr;
#else
typedef const float* CFloatPtr;
#endif

// the MLP proper; valid inside the unit sphere (the caller has done the bounding-sphere test).
// Weights stream through SGPRs in 16-dword blocks (one s_load_dwordx16 = one mat4 = 16 fmas);
// the loads are software-pipelined two blocks ahead of their use (sched_barrier pins the issue
// point) so the scalar-cache latency hides behind the arithmetic of the previous blocks.
typedef float f16v __attribute__((ext_vector_type(16)));
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) f16v* CF16Ptr;
#define RT_LD16(ptr, off) (*(CF16Ptr)((ptr) + (off)))
#define RT_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define RT_LD16(ptr, off) (*(const f16v*)((ptr) + (off)))
#define RT_PIN()
#endif

// Device layout of the 625 weights (rtpbr_set_shape_data permutes the caller's array, layout of
// tools/extract_bunny_weights.py, into this one): the hidden layers' 16x16 matrices are stored in CHAIN order,
//   lw[k*68 + i*16 + m*4 + j] = M_{k,m}[i][j]      (caller: lw[k*68 + m*16 + i*4 + j])
// because a neuron's 16-term sum is one fma chain in the order (i outer, m inner) — see bunny_mlp_wave.
constexpr float INV_1_4 = 0.714285731f;   // f32(1/1.4): the reference's "/ 1.4" (bunny_sdf_glass.py:190-193) as a multiplication

// one 16 -> 16 layer: out[k*4+j] = sin(b_k[j] + chain_{i,m} in[m*4+i]*M_{k,m}[i][j]) (*1/1.4, fused) + in[k*4+j]
template <bool DIV>
RT_D void bunny_layer(CFloatPtr lw, const float* in, float* out) {
    // block b = k*4 + i (16 dwords [m][j]) lives at lw + k*68 + i*16
    f16v w0 = RT_LD16(lw, 0);
    f16v w1 = RT_LD16(lw, 16);
    RT_PIN();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float acc[4];
        CFloatPtr bias = lw + k * 68 + 64;      // the chain starts from the bias
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int b = k * 4 + i;
            // request block b+2 before block b is consumed
            const int nb = b + 2 < 16 ? b + 2 : 15;
            f16v w2 = RT_LD16(lw, (nb >> 2) * 68 + (nb & 3) * 16);
            RT_PIN();
#pragma unroll
            for (int m = 0; m < 4; m++) {
#pragma unroll
                for (int jj = 0; jj < 4; jj++) {
                    const float x = in[m * 4 + i];
                    acc[jj] = fma_(x, w0[m * 4 + jj], (m == 0 && i == 0) ? bias[jj] : acc[jj]);
                }
            }
            w0 = w1;
            w1 = w2;
        }
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const float sn = sin_pi_(acc[jj]);
            out[k * 4 + jj] = DIV ? fma_(sn, INV_1_4, in[k * 4 + jj]) : sn + in[k * 4 + jj];
        }
    }
}

RT_D float bunny_mlp(const float* __restrict__ wg, vec3 p) {
    CFloatPtr w = (CFloatPtr)wg;
    asm volatile("" : "+s"(w));
    float f0[16], f1[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        CFloatPtr b = w + k * 16;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float a = fma_(p.z, b[4 + j], fma_(p.y, b[j], 0.0f));
            a = fma_(-p.x, b[8 + j], a);
            f0[k * 4 + j] = sin_pi_(a + b[12 + j]);
        }
    }
    bunny_layer<false>(w + 64, f0, f1);         // f1 = sin(f0 @ W1 + b1) + f0
    bunny_layer<true>(w + 64 + 272, f1, f0);    // f2 = sin(f1 @ W2 + b2) / 1.4 + f1   (f2 overwrites f0)
    CFloatPtr ow = w + 64 + 544;
    float sd = f0[0] * ow[0];
#pragma unroll
    for (int t = 1; t < 16; t++) sd = fma_(f0[t], ow[t], sd);
    return sd + ow[16];
}

// ---- wave-cooperative MLP on the matrix cores -------------------------------------------------
// The two 16x16 layers (and the 4x16 input layer) are dense contractions over the 64 rays of a wave, so they
// run as v_mfma_f32_16x16x4_f32: f32 in / f32 accumulate, bit-for-bit a k-ordered fmaf chain from C (MI355X
// guide section 3) — the same chain the VALU version and the oracle compute, so results stay bit-identical.
// TRANSPOSED product: D[i = neuron][j = ray] = sum_k W^T[i][k] * act[k][j], i.e. the WEIGHTS are the A operand
// (lane l holds A[i = l&15][k = l>>4], 9 VGPRs per lane, loaded once per kernel) and the ACTIVATIONS the B
// operand (lane l holds B[k = l>>4][j = l&15]).  The result layout — lane l, register v holds
// D[i = 4*(l>>4) + v][j = l&15] — then IS the next layer's B operand: instruction kb of the next layer takes
// register v = kb, i.e. lane group g contributes neuron 4g + kb as its k-th term.  No LDS round trip, no
// transposition between layers; the price is the summation order (kb outer, g inner) = (i outer, m inner) in
// the reference's (block m, row i) indexing; the CPU checker uses the same order.  A layer's bias
// is the C operand of its first MFMA (the chain starts from the bias), residual adds stay in the same registers.
// Only the 3 input coordinates (ray -> slot's lane group) and the 16 outputs per ray (lane group -> ray) cross lanes,
// through a 4 KB wave-private LDS buffer.  The 64 slots are processed as two halves of 2 x 16 rays (8 + 8 live
// activation / accumulator registers instead of 16 + 16); callers compact their rays so that often one half suffices.
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int BUNNY_LDS_WORDS = 64 * 16;      // output staging [ray][16]; the input staging [3][64] aliases its upper half
constexpr int BUNNY_LDS_IN = 512;

struct BunnyFrag {
    float a0;        // input layer  A[i = neuron][k]: rows k = (wy, wz, wx, bias)
    float a1[4];     // layer 1      A_kb[i][k = g] = W1(out i, in t = 4g + kb)
    float a2[4];     // layer 2
};

// w = device layout (hidden layers in chain order, see above)
RT_D void bunny_frag_load(const float* __restrict__ w, int lane, BunnyFrag& F) {
    const int g = lane >> 4, io = lane & 15, blk = io >> 2, jj = io & 3;
    F.a0 = w[blk * 16 + g * 4 + jj];
#pragma unroll
    for (int kb = 0; kb < 4; kb++) {
        F.a1[kb] = w[64 + blk * 68 + kb * 16 + g * 4 + jj];          // M_{blk, m = g}[i = kb][jj]
        F.a2[kb] = w[64 + 272 + blk * 68 + kb * 16 + g * 4 + jj];
    }
}
// the 2 x 16 biases, [layer][neuron], staged once per block in LDS (read as one b128 per lane group)
RT_D void bunny_bias_stage(const float* __restrict__ w, float* bias_lds) {
    if (threadIdx.x < 32) {
        const int layer = threadIdx.x >> 4, n = threadIdx.x & 15;
        bias_lds[threadIdx.x] = w[64 + layer * 272 + (n >> 2) * 68 + 64 + (n & 3)];
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
RT_D f4v mfma4(float a, float b, f4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
#define RT_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
RT_D void bunny_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#else
RT_D f4v mfma4(float, float, f4v c) { return c; }
#define RT_SCHED_BARRIER()
RT_D void bunny_lds_fence() {}
#endif

// MUST be called by all 64 lanes (wave-uniform control flow).  lp = this lane's local point; `slot` = the ray slot
// (0 .. 32*halves-1) this lane's point is evaluated in, or -1 when the lane has no point (slots nobody writes hold
// stale values: rays are independent).  `halves` (1 or 2, wave-uniform) = how many 32-slot halves are computed:
// the callers COMPACT the lanes that need a value into the low slots, so a pass costs what its rays need in units of
// 32, not always 64.  Returns the MLP value of this lane's point (garbage for slot < 0).
RT_D float bunny_mlp_wave(const BunnyFrag& F, const float* __restrict__ wg, float* lds, const float* bias_lds, int lane, vec3 lp,
                          int slot, int halves) {
    const int io = lane & 15, g = lane >> 4;
    float* in = lds + BUNNY_LDS_IN;
    if (slot >= 0) {
        in[0 * 64 + slot] = lp.y;
        in[1 * 64 + slot] = lp.z;
        in[2 * 64 + slot] = -lp.x;
    }
    bunny_lds_fence();
    const f4v bias1 = *reinterpret_cast<const f4v*>(&bias_lds[4 * g]);
    const f4v bias2 = *reinterpret_cast<const f4v*>(&bias_lds[16 + 4 * g]);
    // Two ray blocks per half.  The f32 matrix instructions do NOT overlap with VALU work on this chip — neither from
    // the same wave nor from the other waves of the SIMD (tools/ubench/mfma_overlap.hip: 8 MFMA + 96 fma cost 510
    // cycles back to back, 262 + 248 alone, and 604 when interleaved one MFMA : one sine) — so a layer is issued as
    // one uninterrupted run of MFMAs (two dependent chains of four, back to back) followed by one run of sines,
    // pinned with scheduling barriers; interleaving them costs 16 %.
#pragma nounroll
    for (int h = 0; h < halves; h++) {
        const f4v z = {0.0f, 0.0f, 0.0f, 0.0f};
        f4v act0, act1, c0, c1;
        {   // input layer: (p.y, p.z, -p.x, 1) . (wy, wz, wx, b); lane group g supplies component g of ray io
            const float x0 = in[(g < 3 ? g : 0) * 64 + (2 * h) * 16 + io];
            const float x1 = in[(g < 3 ? g : 0) * 64 + (2 * h + 1) * 16 + io];
            RT_SCHED_BARRIER();
            c0 = mfma4(F.a0, g < 3 ? x0 : 1.0f, z);
            c1 = mfma4(F.a0, g < 3 ? x1 : 1.0f, z);
        }
        RT_SCHED_BARRIER();
#pragma unroll
        for (int v = 0; v < 4; v++) {
            act0[v] = sin_pi_(c0[v]);
            act1[v] = sin_pi_(c1[v]);
        }
        RT_SCHED_BARRIER();
        c0 = mfma4(F.a1[0], act0[0], bias1);                                     // layer 1: the bias is the C operand
#pragma unroll
        for (int kb = 1; kb < 4; kb++) c0 = mfma4(F.a1[kb], act0[kb], c0);
        c1 = mfma4(F.a1[0], act1[0], bias1);
#pragma unroll
        for (int kb = 1; kb < 4; kb++) c1 = mfma4(F.a1[kb], act1[kb], c1);
        RT_SCHED_BARRIER();
#pragma unroll
        for (int v = 0; v < 4; v++) {
            act0[v] = sin_pi_(c0[v]) + act0[v];
            act1[v] = sin_pi_(c1[v]) + act1[v];
        }
        RT_SCHED_BARRIER();
        c0 = mfma4(F.a2[0], act0[0], bias2);                                     // layer 2
#pragma unroll
        for (int kb = 1; kb < 4; kb++) c0 = mfma4(F.a2[kb], act0[kb], c0);
        c1 = mfma4(F.a2[0], act1[0], bias2);
#pragma unroll
        for (int kb = 1; kb < 4; kb++) c1 = mfma4(F.a2[kb], act1[kb], c1);
        RT_SCHED_BARRIER();
#pragma unroll
        for (int v = 0; v < 4; v++) {
            act0[v] = fma_(sin_pi_(c0[v]), INV_1_4, act0[v]);
            act1[v] = fma_(sin_pi_(c1[v]), INV_1_4, act1[v]);
        }
        // ---- back to lane = ray: [ray][neuron 4g .. 4g+3]
        *reinterpret_cast<f4v*>(&lds[((2 * h) * 16 + io) * 16 + 4 * g]) = act0;
        *reinterpret_cast<f4v*>(&lds[((2 * h + 1) * 16 + io) * 16 + 4 * g]) = act1;
    }
    bunny_lds_fence();
    CFloatPtr ow = (CFloatPtr)wg + 64 + 544;
    f4v o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) o[q] = *reinterpret_cast<const f4v*>(&lds[(slot >= 0 ? slot : lane) * 16 + 4 * q]);
    bunny_lds_fence();    // the buffer is rewritten by the next call
    float sd = o[0][0] * ow[0];
#pragma unroll
    for (int t = 1; t < 16; t++) sd = fma_(o[t >> 2][t & 3], ow[t], sd);
    return sd + ow[16];
}

// bunny_sdf_glass.py:150-153: outside the unit sphere the SDF is the distance to a 0.8 sphere
RT_D float sd_bunny(const float* __restrict__ wg, vec3 p) {
    float len = length(p);
    if (len > 1.0f || wg == nullptr) return len - 0.8f;
    return bunny_mlp(wg, p);
}

// ---------------------------------------------------------------- F10 SDF primitives
// src/sdf.py:21-51 (l = local position, s = transform.scale); box rounding rho per variant
RT_D float sd_box(vec3 l, float sx, float sy, float sz, float rho) {
    float qx = fabs_(l.x) - sx, qy = fabs_(l.y) - sy, qz = fabs_(l.z) - sz;
#if defined(__HIP_DEVICE_COMPILE__)
    // |max(q, 0)| as sqrt(|q + |q||^2 / 4): q + |q| = 2 max(q, 0) exactly, the scaling by 4 commutes with every rounding of the
    // dot product and sqrt_quarter_ takes it back inside the root's own scaling — bit for bit the reference's length(max(q, 0)),
    // with three full-rate adds (abs source modifier) in place of three half-rate v_max_f32 (as in nearest_boxes_lazy)
    vec3 u = mk(qx + fabs_(qx), qy + fabs_(qy), qz + fabs_(qz));
#if RT_FAST_MATH      // (the root is the distance to the box's core: never large where it matters — the bare instruction)
    const float len = 0.5f * sqrt_shape_(dot(u, u), false);
    return (len + fmin_(fmax_(qx, fmax_(qy, qz)), 0.0f)) - rho;
#else
    return sqrt_quarter_add_(dot(u, u), fmin_(fmax_(qx, fmax_(qy, qz)), 0.0f)) - rho;      // (root + addend in one fma: rt_math.hpp)
#endif
#else
    vec3 m = mk(fmax_(qx, 0.0f), fmax_(qy, 0.0f), fmax_(qz, 0.0f));
    return (sqrt_(dot(m, m)) + fmin_(fmax_(qx, fmax_(qy, qz)), 0.0f)) - rho;
#endif
}

template <int KIND>
RT_D float sdf_local(const Params& P, int type, vec3 l, float sx, float sy, float sz) {
    if (KIND == KIND_BOXES) return sd_box(l, sx, sy, sz, P.cfg.box_round);
    if (KIND == KIND_BUNNY) return sd_bunny(P.bunny, l);
    // (a run-time instance knows which shapes its scene holds: a switch on a per-lane type keeps only those cases)
#if defined(__HIP_DEVICE_COMPILE__)
#define RT_SHAPE_CASE(T) case T: if (!jit_has_type(T)) __builtin_unreachable();
#else
#define RT_SHAPE_CASE(T) case T:
#endif
    switch (type) {
        RT_SHAPE_CASE(RTPBR_SHAPE_SPHERE)
#if RT_FAST_MATH && !defined(RT_FAST_EXACT_SQRT)
            return (sx > RT_BIG_EXTENT ? sqrt_big_sphere_(dot(l, l), sx) : sqrt_shape_(dot(l, l), false)) - sx;
#else
            return sqrt_add_(dot(l, l), -sx);
#endif
        RT_SHAPE_CASE(RTPBR_SHAPE_BOX)
            return sd_box(l, sx, sy, sz, P.cfg.box_round);
        RT_SHAPE_CASE(RTPBR_SHAPE_CYLINDER) {
            const bool big = sx > RT_BIG_EXTENT || sy > RT_BIG_EXTENT;
            // (|r| - sx with r a root: r >= +0, the abs is the identity; both roots take their addend in the root's own fma)
            float dx = sqrt_shape_add_(fma_(l.z, l.z, l.x * l.x), big, -sx), dy = fabs_(l.y) - sy;
            float mx = fmax_(dx, 0.0f), my = fmax_(dy, 0.0f);
            return sqrt_shape_add_(fma_(my, my, mx * mx), big, fmin_(fmax_(dx, dy), 0.0f));
        }
        RT_SHAPE_CASE(RTPBR_SHAPE_CONE) {
            float q = sqrt_shape_(fma_(l.z, l.z, l.x * l.x), sy > RT_BIG_EXTENT);
            return fmax_(fma_(sz, l.y, sx * q), -sy - l.y);
        }
        RT_SHAPE_CASE(RTPBR_SHAPE_PLANE)
            return l.y - sy;
        case RTPBR_SHAPE_BUNNY:
            if (KIND == KIND_MIXED) return sd_bunny(P.bunny, l);
            return P.cfg.max_dis;
        default:
            return P.cfg.max_dis;
    }
}

// a x + b y of a single-axis rotation.  Tolerance flavour, baked table: the run-time compiler snaps matrix entries within 2^-20 of
// 0 / +-1 (rt_jit.hip: cosf of a right angle is -4.4e-8, not 0) and a rotation by a right angle becomes what it is, a signed
// permutation — no arithmetic at all; dropping a 4.4e-8 x term moves a position by less than half an ulp of the room.
RT_D float rot2(float a, float x, float b, float y) {
#if RT_FAST_MATH && defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_constant_p(a) && __builtin_constant_p(b)) {
        if (a == 0.0f && b == 1.0f) return y;
        if (a == 0.0f && b == -1.0f) return -y;
        if (b == 0.0f && a == 1.0f) return x;
        if (b == 0.0f && a == -1.0f) return -x;
    }
#endif
    return fma_(b, y, a * x);
}

// F9 world -> local (src/sdf.py:64-68) + the bunny's per-frame animation (bunny_sdf_glass.py:213-217)
template <int KIND, typename OBJ>
RT_D vec3 to_local(const Params& P, const OBJ& o, vec3 p, int cls = ROT_GENERAL) {
    vec3 d = p - mk(o.px, o.py, o.pz);
    if constexpr (KIND == KIND_BOXES || KIND == KIND_GENERIC) {
        // Sparse rotations (every analytic shape: sphere / cylinder / cone see squares or |l|, a plane l.y - h).  `cls` is a COMPILE-TIME constant after unrolling (rotation signature of
        // the scene, see nearest); matrices whose off-axis entries are EXACTLY 0 and whose axis
        // entry is EXACTLY 1 (rotation about one coordinate axis, or none) are classified by
        // rtpbr_set_scene.  Dropping the x*0 and x*1 terms of the fma chain is exact for finite
        // positions up to the sign of a zero result, and the box SDF only sees |l|.
        if (cls == ROT_IDENT) return d;
        if (cls == ROT_X) return mk(d.x, rot2(o.m[4], d.y, o.m[5], d.z), rot2(o.m[7], d.y, o.m[8], d.z));
        if (cls == ROT_Y) return mk(rot2(o.m[0], d.x, o.m[2], d.z), d.y, rot2(o.m[6], d.x, o.m[8], d.z));
        if (cls == ROT_Z) return mk(rot2(o.m[0], d.x, o.m[1], d.y), rot2(o.m[3], d.x, o.m[4], d.y), d.z);
    }
    vec3 l = mulv(o.m, d);
    if (KIND == KIND_BUNNY || (KIND == KIND_MIXED && o.type == RTPBR_SHAPE_BUNNY)) {
        float st = P.anim_s, ct = P.anim_c;
        vec3 r = mk(fma_(st, l.y, ct * l.x), fma_(ct, l.y, -st * l.x), l.z);
        r.z = r.z + P.anim_bz;
        l = r;
    }
    return l;
}

// type_known >= 0: the shape type is a compile-time constant of the unrolled object loop (run-time compiled instances)
template <int KIND, typename OBJ>
RT_D float signed_distance(const Params& P, const OBJ& o, vec3 p, int cls = ROT_GENERAL, int type_known = -1) {
    return sdf_local<KIND>(P, type_known >= 0 ? type_known : o.type, to_local<KIND>(P, o, p, cls), o.sx, o.sy, o.sz);
}

// ---------------------------------------------------------------- F8 nearest
// min_i |sdf_i|, ties to the lowest index (cornell_box_v3/pathtracer.py:41-49; src/scene.py:44-56).
// The march-loop object table lives in the kernarg segment (constant address space) and is
// read at wave-uniform indices -> s_load_dwordx16 into SGPRs, consumed as scalar operands.
// The pointer is laundered through an empty asm every call so the loads stay INSIDE the march
// loop: hoisting 8 x 16 constants out of it needs more SGPRs than exist and the compiler then
// spills them into VGPR lanes (v_writelane/v_readlane per constant per step).
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) ObjM* ObjTab;
RT_D ObjTab obj_table() {
    const __attribute__((address_space(4))) char* k =
        (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    return (ObjTab)(k + offsetof(Params, objm));
}
#else
typedef const ObjM* ObjTab;   // host pass only parses the device functions
RT_D ObjTab obj_table() { return nullptr; }
#endif

// rotation class of object i in the kernel's compile-time signature (3 bits per object, 0 = general)
#define RT_SIG_CLS(i) (int)((SIG >> (3 * (i))) & 7u)

// compile-time loop: f(std::integral_constant<int, I>) for I = 0, STEP, 2 STEP, ... < N
template <int N, int STEP, int I = 0, typename F>
RT_D void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, STEP, I + STEP>(static_cast<F&&>(f));
    }
}

// Object I of the march table.  Signature 0: the general 64-byte block.  Otherwise the packed table
// (rt_types.hpp): wide scalar loads of exactly the dwords the object's rotation class uses.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f8u __attribute__((ext_vector_type(8), aligned(4)));
typedef float f2u_ __attribute__((ext_vector_type(2), aligned(4)));
template <uint32_t SIG, int I>
RT_D ObjM load_obj(ObjTab tab) {
#if RT_JIT_BAKED
    {
        auto w = [](int k) { return __builtin_bit_cast(float, RT_JIT_TABLE_BITS[I][k]); };
        ObjM o = {};
        o.px = w(0), o.py = w(1), o.pz = w(2);
#pragma unroll
        for (int k = 0; k < 9; k++) o.m[k] = w(3 + k);
        o.sx = w(12), o.sy = w(13), o.sz = w(14);
        o.type = (int32_t)RT_JIT_TABLE_BITS[I][15];
        return o;
    }
#endif
    if constexpr (SIG == 0) {
        return tab[I];
    } else {
        constexpr int cls = sig_cls(SIG, I);
        const __attribute__((address_space(4))) float* f = (const __attribute__((address_space(4))) float*)tab + sig_offset(SIG, I);
        ObjM o = {};
        if constexpr (cls == ROT_GENERAL) {
            return *(const __attribute__((address_space(4))) ObjM*)f;
        } else if constexpr (cls == ROT_IDENT) {
            const f8u a = *(const __attribute__((address_space(4))) f8u*)f;
            o.px = a[0], o.py = a[1], o.pz = a[2], o.sx = a[3], o.sy = a[4], o.sz = a[5];
        } else {
            // (one s_load_dwordx16 instead of x8 + x2 was measured 2 % slower: more SGPR pressure)
            const f8u a = *(const __attribute__((address_space(4))) f8u*)f;
            const f2u_ b = *(const __attribute__((address_space(4))) f2u_*)(f + 8);
            o.px = a[0], o.py = a[1], o.pz = a[2], o.sx = a[7], o.sy = b[0], o.sz = b[1];
            if constexpr (cls == ROT_X) o.m[4] = a[3], o.m[5] = a[4], o.m[7] = a[5], o.m[8] = a[6];
            if constexpr (cls == ROT_Y) o.m[0] = a[3], o.m[2] = a[4], o.m[6] = a[5], o.m[8] = a[6];
            if constexpr (cls == ROT_Z) o.m[0] = a[3], o.m[1] = a[4], o.m[3] = a[5], o.m[4] = a[6];
        }
        return o;
    }
}
#else
template <uint32_t SIG, int I>
RT_D ObjM load_obj(ObjTab) { return ObjM{}; }   // host pass only parses the device functions
#endif
// ---- all-box scenes: pick the nearest box on SQUARED distances, one square root per step.
// The correctly rounded square root is a third of a box evaluation and only the winner's distance
// is used.  With q = |l| - s, s2 = |max(q,0)|^2 and mx = max3(q) the reference distance is
//     d = | (sqrt(s2) + min(mx,0)) - rho |
// and, per object, ONE of three cases holds:
//   outside  (mx > 0, sqrt(s2) > rho):  d = sqrt(s2) - rho          increasing in s2
//   core     (mx <= 0, s2 = 0):         d = rho - mx                increasing in -mx
//   shell    (mx > 0, sqrt(s2) <= rho): d = rho - sqrt(s2) <= rho   (an over-relaxed step ended inside the rounding)
// key = s2 (outside, shell) or (2 rho - mx)^2 (core) orders outside and core objects exactly like d
// does: d_core < d_out  <=>  rho - mx < sqrt(s2) - rho  <=>  (2 rho - mx)^2 < s2.  The three
// smallest keys k1 <= k2 <= k3 are tracked (two v_med3 per object).  Rounding can only change an
// order when two keys agree to ~2^-22, so the lane is SUSPECT when
//   * k3 <= k1 (1 + 2^-20), or k2 <= k1 (1 + 2^-20) unless k2 == k1 EXACTLY and the lane has no core
//     object: then both keys are the same s2, hence the same d, and the strict `<` below has kept the
//     lower index like the reference does (the room's rim corners belong to two slabs: 0.1 % of the
//     lane-steps are such ties);
//   * an object is in its shell (k1 < rho^2) and another one is within 2 rho (k2 < 4 rho^2): otherwise
//     the shell object is the nearest for sure (d <= rho < every other d).
// If any marching lane of the wave is suspect the function returns false and the caller evaluates the
// full expression for the wave (~1 % of the wave-steps).  Otherwise (idx, best) are bit-for-bit the
// reference's: the winner's distance is computed with the same operations, |fl(sqrt_(s2) - rho)| or
// fl(rho - mx).
// (Measured dead end: a first tier that assumes "no marching lane is inside a box" and drops the core key saves 5 of
// 27 instructions per box, but over-relaxed steps (omega = 1.6) land inside a box once per raycast on purpose, so half
// of the wave-steps have such a lane and pay both tiers: 130 -> 177 ms.)
template <int NOBJ, uint32_t SIG>
RT_D bool nearest_boxes_lazy(const Params& P, vec3 p, int& idx, float& best) {
    ObjTab tab = obj_table();
    asm volatile("" : "+s"(tab));
    const float rho = P.cfg.box_round;
    const float four_rho = P.box_four_rho;
    float k1 = 3.0e38f, k2 = 3.0e38f, k3 = 3.0e38f;   // the three smallest keys
    float pmx = 1.0f;                                  // max3(q) of the object with the smallest key (<= 0: core)
    bool has_core = false;
    idx = 0;
    // All keys are carried SCALED BY 4 (exact): max(q,0) is formed as (q + |q|) = 2 max(q,0) — an add with an abs
    // source modifier runs at the full FP32 rate on gfx950, v_max_f32 at half of it (tools/ubench/valu_rate.hip:
    // fma/add/mul/sub on VGPRs 2.45 cycles per wave instruction; min/max/med3/cmp/cndmask/shifts and every
    // instruction with an SGPR, DPP or SDWA operand 4.2) — and the core key as (2 (2 rho - mx))^2.  Scaling by a power
    // of two commutes with every rounding here, so the order of the keys, the 2^-20 closeness band and the winner's
    // distance (sqrt of the key / 4: same v_sqrt_f32 input, see sqrt_quarter_) are bit-for-bit what they were.
    auto visit = [&](const ObjM& o, int i, int cls) {
        vec3 l = to_local<KIND_BOXES>(P, o, p, cls);
        float qx = fabs_(l.x) - o.sx, qy = fabs_(l.y) - o.sy, qz = fabs_(l.z) - o.sz;
        float mx = fmax_(qx, fmax_(qy, qz));
        vec3 u = mk(qx + fabs_(qx), qy + fabs_(qy), qz + fabs_(qz));
        float s2 = dot(u, u);
        float t = fma_(mx, -2.0f, four_rho);
        const bool out = mx > 0.0f;
        has_core = has_core | !out;
        float key = out ? s2 : t * t;
        bool lt = key < k1;
        k3 = __builtin_amdgcn_fmed3f(key, k2, k3);   // k1 <= k2 <= k3 always: medians insert the new key
        k2 = __builtin_amdgcn_fmed3f(key, k1, k2);
        k1 = lt ? key : k1;
        idx = lt ? i : idx;
        pmx = lt ? mx : pmx;
    };
    static_for<NOBJ, 2>([&](auto I) {   // two objects' constants requested per scalar-load group
        constexpr int i = decltype(I)::value;
        const ObjM oa = load_obj<SIG, i>(tab);
        const ObjM ob = load_obj<SIG, (i + 1 < NOBJ ? i + 1 : i)>(tab);
        visit(oa, i, RT_SIG_CLS(i));
        if constexpr (i + 1 < NOBJ) visit(ob, i + 1, RT_SIG_CLS(i + 1));
    });
#if !RT_FAST_MATH      // (tolerance flavour: near-ties go either way; an object in its rounding shell is off by <= rho)
    const float lim = k1 * 1.00000095367431640625f;    // 1 + 2^-20
    const bool suspect = k3 <= lim || (k2 <= lim && (k2 != k1 || has_core)) || (k1 < P.box_rho2m && k2 < P.box_4rho2m);
    if (__any(suspect)) return false;
#endif
    const bool core = !(pmx > 0.0f);
    float d = core ? rho - pmx : fabs_(sqrt_quarter_add_(k1, -rho));
    if (P.cfg.nearest_init && !(d < P.cfg.max_dis)) {   // src/ form: the search starts from (0, MAX_DIS)
        d = P.cfg.max_dis;
        idx = 0;
    }
    best = d;
    return true;
}

template <int KIND, int NOBJ, uint32_t SIG = 0>
RT_D void nearest_exact(const Params& P, vec3 p, int& idx, float& best);

template <int KIND, int NOBJ, uint32_t SIG = 0>
RT_D void nearest(const Params& P, vec3 p, int& idx, float& best) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (KIND == KIND_BOXES && NOBJ > 0) {
        if (P.box_lazy && nearest_boxes_lazy<NOBJ, SIG>(P, p, idx, best)) return;
#ifdef RT_DEBUG_LAZY   // wave-steps that fell back to the exact expression (read with rtpbr_get_counter "mlp_lane_evals")
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0)) == 0) atomicAdd(&P.counters->mlp_lane_evals, 1ull);
#endif
    }
#endif
    nearest_exact<KIND, NOBJ, SIG>(P, p, idx, best);
}

template <int KIND, int NOBJ, uint32_t SIG>
RT_D void nearest_exact(const Params& P, vec3 p, int& idx, float& best) {
    const int n = NOBJ > 0 ? NOBJ : P.n_obj;
    ObjTab tab = obj_table();
    asm volatile("" : "+s"(tab));
    int start;
    idx = 0;
    if (P.cfg.nearest_init) {
        best = P.cfg.max_dis;
        start = 0;
    } else {
        const ObjM o = load_obj<SIG, 0>(tab);
        best = fabs_(signed_distance<KIND>(P, o, p, RT_SIG_CLS(0), jit_type(0)));
        start = 1;
    }
    if constexpr (NOBJ > 0) {
        // two objects per iteration: both constant blocks are requested before either is used, so
        // the scalar-load latency of object i+1 hides behind the arithmetic of object i
        static_for<NOBJ, 2>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const ObjM oa = load_obj<SIG, i>(tab);
            const ObjM ob = load_obj<SIG, (i + 1 < NOBJ ? i + 1 : i)>(tab);
            if (i >= start) {
                float d = fabs_(signed_distance<KIND>(P, oa, p, RT_SIG_CLS(i), jit_type(i)));
                bool lt = d < best;
                best = lt ? d : best;
                idx = lt ? i : idx;
            }
            if constexpr (i + 1 < NOBJ) {
                float d = fabs_(signed_distance<KIND>(P, ob, p, RT_SIG_CLS(i + 1), jit_type(i + 1)));
                bool lt = d < best;
                best = lt ? d : best;
                idx = lt ? i + 1 : idx;
            }
        });
    } else {
        for (int i = start; i < n; i++) {
            const ObjM o = tab[i];
            float d = fabs_(signed_distance<KIND>(P, o, p));
            bool lt = d < best;
            best = lt ? d : best;
            idx = lt ? i : idx;
        }
    }
}

// ---------------------------------------------------------------- F8 nearest for the tracked-object march (rt_persistent.hpp)
// nearest_exact that also returns the SECOND smallest |sdf_i| (3e38 when there is no second object): what a lane needs
// to know to skip every object but the nearest one on its next steps.  best <= second always, so the two smallest of
// {best, second, d} are min(best, d) and med3(best, second, d): one v_med3 per object on top of nearest_exact.
template <int KIND, int NOBJ, uint32_t SIG>
RT_D void nearest_exact2(const Params& P, vec3 p, int& idx, float& best, float& second) {
    const int n = NOBJ > 0 ? NOBJ : P.n_obj;
    ObjTab tab = obj_table();
    asm volatile("" : "+s"(tab));
    idx = 0;
    second = 3.0e38f;
    // the reference starts from (0, MAX_DIS) (src/scene.py:46, nearest_init) or from object 0: an initial best of
    // 3e38 that the first object always beats gives the second form
    best = P.cfg.nearest_init ? P.cfg.max_dis : 3.0e38f;
    bool first = !P.cfg.nearest_init;
    auto visit = [&](float d, int i) {
        second = __builtin_amdgcn_fmed3f(best, second, d);
        bool lt = first || d < best;
        best = lt ? d : best;
        idx = lt ? i : idx;
        first = false;
    };
    if constexpr (NOBJ > 0) {
        static_for<NOBJ, 2>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const ObjM oa = load_obj<SIG, i>(tab);
            const ObjM ob = load_obj<SIG, (i + 1 < NOBJ ? i + 1 : i)>(tab);
            if (SIG != 0 || i < P.n_obj) visit(fabs_(signed_distance<KIND>(P, oa, p, RT_SIG_CLS(i), jit_type(i))), i);
            if constexpr (i + 1 < NOBJ) {
                if (SIG != 0 || i + 1 < P.n_obj) visit(fabs_(signed_distance<KIND>(P, ob, p, RT_SIG_CLS(i + 1), jit_type(i + 1))), i + 1);
            }
        });
    } else {
        for (int i = 0; i < n; i++) {
            const ObjM o = tab[i];
            visit(fabs_(signed_distance<KIND>(P, o, p)), i);
        }
    }
    // (nearest_init: the initial MAX_DIS took part in the med3 as `best`; an object at >= MAX_DIS never wins, and a
    // `second` of MAX_DIS in place of a larger distance only makes the tracked steps re-evaluate earlier)
}

// ... and the THIRD smallest, with the index of the second (rt_persistent.hpp: the two-object lean loop keeps marching on the
// two nearest objects while a bound proves every other one farther).  best <= second <= third always: inserting d gives
// min(best, d), med3(best, second, d), med3(second, third, d).  idx2 = an object that attains `second` (any of them when
// several do: the bound that is derived from `third` then covers the others).
template <int KIND, int NOBJ, uint32_t SIG>
RT_D void nearest_exact3(const Params& P, vec3 p, int& idx, float& best, int& idx2, float& second, float& third) {
    const int n = NOBJ > 0 ? NOBJ : P.n_obj;
    ObjTab tab = obj_table();
    asm volatile("" : "+s"(tab));
    idx = 0;
    idx2 = -1;      // no real second object yet (nearest_init: the initial (0, MAX_DIS) is not an object)
    second = third = 3.0e38f;
    best = P.cfg.nearest_init ? P.cfg.max_dis : 3.0e38f;
    bool first = !P.cfg.nearest_init;
    bool virt = true;       // `best` is not an object's distance yet (the 3e38 / MAX_DIS start value)
    auto visit = [&](float d, int i) {
        third = __builtin_amdgcn_fmed3f(second, third, d);
        const bool lt = first || d < best;
        const bool mid = !lt && d < second;
        second = __builtin_amdgcn_fmed3f(best, second, d);
        idx2 = lt ? (virt ? -1 : idx) : (mid ? i : idx2);
        best = lt ? d : best;
        idx = lt ? i : idx;
        virt = virt && !lt;
        first = false;
    };
    if constexpr (NOBJ > 0) {
        static_for<NOBJ, 2>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const ObjM oa = load_obj<SIG, i>(tab);
            const ObjM ob = load_obj<SIG, (i + 1 < NOBJ ? i + 1 : i)>(tab);
            if (SIG != 0 || i < P.n_obj) visit(fabs_(signed_distance<KIND>(P, oa, p, RT_SIG_CLS(i), jit_type(i))), i);
            if constexpr (i + 1 < NOBJ) {
                if (SIG != 0 || i + 1 < P.n_obj) visit(fabs_(signed_distance<KIND>(P, ob, p, RT_SIG_CLS(i + 1), jit_type(i + 1))), i + 1);
            }
        });
    } else {
        for (int i = 0; i < n; i++) {
            const ObjM o = tab[i];
            visit(fabs_(signed_distance<KIND>(P, o, p)), i);
        }
    }
}

// |sdf| of ONE object chosen by a wave-uniform index (scalar compare-and-branch chain over the unrolled table)
template <int KIND, int NOBJ, uint32_t SIG>
RT_D float sdf_object(const Params& P, int kw, vec3 p) {
    ObjTab tab = obj_table();
    asm volatile("" : "+s"(tab));
    float r = 0.0f;
    if constexpr (NOBJ > 0) {
        static_for<NOBJ, 1>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if (kw == i) {
                // (the empty asm pins the evaluation inside its branch: with a baked table nothing else keeps the compiler
                // from evaluating all objects speculatively ahead of the chain)
                vec3 q = p;
                asm volatile("" : "+v"(q.x));
                const ObjM o = load_obj<SIG, i>(tab);
                r = fabs_(signed_distance<KIND>(P, o, q, RT_SIG_CLS(i), jit_type(i)));
            }
        });
    } else {
        const ObjM o = tab[kw];
        r = fabs_(signed_distance<KIND>(P, o, p));
    }
    return r;
}

// ---------------------------------------------------------------- F8 nearest with wave-level culling
// For COHERENT waves (the 64 lanes march almost the same ray: consecutive samples of one pixel)
// most objects are far from every lane, and exact bounds prove it without evaluating them.
// |sdf_i| is 1-Lipschitz in the position, so with `moved` = distance marched since the last step:
//   lb_i = (last exact |sdf_i|) - (everything marched since)     is a lower bound of |sdf_i| now,
//   ub   = (last exact minimum) + moved                          is an upper bound of the new minimum.
// Object i is skipped when lb_i > ub for EVERY lane (wave-uniform branch): then |sdf_i| > min strictly,
// so it is neither the nearest nor a tie and (index, distance) are exactly what the full loop gives.
// Objects are still visited in index order with the strict `<` of the reference, so ties resolve
// identically.  eps covers the rounding of the computed distances (|error| <~ 10 ulp of the largest
// intermediate, |pos - centre| + size); we allow 2^-19 (t + extent) on each side, extent = max(64,
// 4 max_i(|centre_i| + |size_i|)) from the host.  lb[] and ub are maintained by the caller across
// steps.  Works for every shape whose SDF is 1-Lipschitz (sphere, box, cylinder, plane, none; the host
// checks the cone's slope vector and excludes the neural SDF).  With signature 0 the table may hold
// fewer than NOBJ objects (runtime count, compile-time unrolling).
template <int KIND, int NOBJ, uint32_t SIG = 0>
RT_D void nearest_culled(const Params& P, vec3 p, float t, bool active, float ub, float (&lb)[NOBJ > 0 ? NOBJ : 1],
                         int& idx, float& best, uint32_t* dbg_evaluated = nullptr, uint32_t* ev_mask = nullptr) {
    static_assert(NOBJ > 0, "culling needs a compile-time object count");
    ObjTab tab = obj_table();
    asm volatile("" : "+s"(tab));
    const float eps = 1.9073486328125e-06f * (fabs_(t) + P.cull_extent);
    const float bound = ub + eps;
    const unsigned long long act_mask = __builtin_amdgcn_ballot_w64(active);
    best = P.cfg.max_dis;   // the reference starts from object 0 or from MAX_DIS (nearest_init); see below
    idx = 0;
    bool first = !P.cfg.nearest_init;
    static_for<NOBJ, 1>([&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (SIG == 0) {
            if (i >= P.n_obj) return;
        }
        // provably not the nearest for any lane: skip.  (v_cmp straight into a scalar mask: 13 = "unordered or <=", i.e.
        // !(lb > bound); __all / __ballot cost two more VALU instructions per object here)
        if ((__builtin_amdgcn_fcmpf(lb[i], bound, 13) & act_mask) == 0ull) return;
        const ObjM o = load_obj<SIG, i>(tab);
#if defined(RT_DEBUG_PHASE) || defined(RT_DEBUG_CULL) || defined(RT_DEBUG_PRIMARY)
        if (dbg_evaluated) (*dbg_evaluated)++;
#endif
        if (ev_mask) *ev_mask |= 1u << i;        // (wave-uniform: a scalar or)
        float d = fabs_(signed_distance<KIND>(P, o, p, RT_SIG_CLS(i), jit_type(i)));
        lb[i] = d - eps;
        bool take = first || d < best;                            // nearest_init = 0: the first visited object initialises
        best = take ? d : best;
        idx = take ? i : idx;
        first = false;
    });
}

// ---------------------------------------------------------------- per-lane path state
enum { ST_IDLE = 0, ST_MARCH = 1, ST_HIT = 2, ST_MISS = 3, ST_EXHAUSTED = 4 };

struct Lane {
    vec3 o, d, col;
    float t, w, s, dist;  // sphere-tracing state (raycast locals of the reference)
    float t_eval;         // t of the last evaluated position (record.position = o + t_eval*d)
    int idx;              // nearest object at the last evaluation (record.object)
    int steps_left;
    int bounce;           // i of "for i in range(MAX_RAYTRACE)"
    uint32_t key, cnt;    // RNG stream
    uint32_t item;        // work item = q*K + k
    int state;
    uint32_t n_steps, n_raycasts, n_hits, n_sky;  // work counters
};

// start of raycast(): cornell_box_v3/pathtracer.py:54-56 / cornell_box_v2.py:187
RT_D void march_init(const Params& P, Lane& L) {
    L.t = P.cfg.min_dis;
    L.w = P.cfg.omega0;
    L.s = 0.0f;
    L.dist = 0.0f;
    L.steps_left = P.cfg.max_raymarch;
    L.state = ST_MARCH;
    L.n_raycasts++;
}

// One iteration of the examples' raycast loop.  plain: cornell_box_v2.py:186-196;
// relaxed: cornell_box_v3/pathtracer.py:57-76, tokyo_ibl.py:249-263, bunny_sdf_glass.py:252-265.
// The part of one raycast iteration after nearest(): relaxation / hit / escape bookkeeping.
RT_D void march_update(const Params& P, Lane& L, int idx, float dist) {
    L.idx = idx;
    L.n_steps++;
    L.steps_left--;
    bool hit, done;
    if (P.cfg.march_kind == RTPBR_MARCH_PLAIN) {
        L.t += dist;
        hit = dist < P.cfg.hit_eps;
        done = hit || L.t > P.cfg.max_dis;
    } else {
        float ld = L.dist;
        L.dist = dist;
        bool fb = (!P.cfg.omega_guard || L.w > 1.0f) && (ld + dist < L.s);
        // fallback branch: s -= w*s; t += s; w = a + b*w; continue
        float s_fb = L.s - L.w * L.s;
        // (omega_fb_b == 0: w is finite and positive, so a + 0*w == a exactly; saves the two instructions a baked 0 cannot fold)
        float w_fb = P.cfg.omega_fb_b == 0.0f ? P.cfg.omega_fb_a : P.cfg.omega_fb_a + P.cfg.omega_fb_b * L.w;
        // normal branch: err = d / t; hit = err < PIXEL_RADIUS.  The correctly rounded quotient
        // is only needed when d is within 2^-20 (relative) of t*eps; otherwise the comparison is
        // decided by the product (monotone rounding), which saves the 11-instruction divide on
        // practically every step.  Wave-uniform branch: taken if ANY active lane is in the band.
        // (thresholds t * (eps (1 -+ 2^-20)): a conservative pre-filter, 16 ulp wide against two roundings; any
        // lane inside the band sends the wave to the exact quotient, so the decision is the reference's either way)
#if RT_FAST_MATH
        bool hit_n = dist < L.t * P.cfg.hit_eps;      // tolerance flavour: the product decides
#else
        bool sure_hit = dist < L.t * (P.cfg.hit_eps * 0.99999905f);
        bool sure_miss = dist > L.t * (P.cfg.hit_eps * 1.00000095f);
        bool unsure = !(sure_hit || sure_miss) || !(L.t > 0.0f);
        bool hit_n = sure_hit;
        if (__any(unsure)) {
            float err = dist / L.t;
            hit_n = err < P.cfg.hit_eps;
        }
#endif
        float s_nm = L.w * dist;
        float s_new = fb ? s_fb : s_nm;
        L.s = s_new;
        L.t += s_new;
        L.w = fb ? w_fb : L.w;
        // (bitwise on purpose: keeps the lane masks in scalar registers instead of 0/1 selects in VGPRs)
        const bool nfb = !fb;
        hit = nfb & hit_n;
        done = nfb & (hit_n | (L.t > P.cfg.max_dis));
    }
    done = done | (L.steps_left == 0);
    if (done) L.state = hit ? ST_HIT : ST_MISS;
}

template <int KIND, int NOBJ, uint32_t SIG = 0>
RT_D void march_step(const Params& P, Lane& L) {
    vec3 pos = fma3(L.t, L.d, L.o);
    L.t_eval = L.t;
    int idx;
    float dist;
    nearest<KIND, NOBJ, SIG>(P, pos, idx, dist);
    march_update(P, L, idx, dist);
}

// Single-bunny scenes: the cheap half of nearest().  Returns true when the position is inside the
// unit sphere, i.e. the MLP is needed (local position in lp); otherwise dist is final.
RT_D bool bunny_pre(const Params& P, Lane& L, vec3& lp, float& dist) {
    ObjTab tab = obj_table();
    asm volatile("" : "+s"(tab));
    const ObjM o = tab[0];
    vec3 pos = fma3(L.t, L.d, L.o);
    L.t_eval = L.t;
    vec3 l = to_local<KIND_BUNNY>(P, o, pos);
    float len = length(l);
    if (len > 1.0f || P.bunny == nullptr) {
        dist = fabs_(len - 0.8f);
        if (P.cfg.nearest_init) dist = fmin_(dist, P.cfg.max_dis);
        return false;
    }
    lp = l;
    return true;
}
RT_D float bunny_post(const Params& P, vec3 lp) {
    float dist = fabs_(bunny_mlp(P.bunny, lp));
    if (P.cfg.nearest_init) dist = fmin_(dist, P.cfg.max_dis);
    return dist;
}
RT_D float bunny_post_value(const Params& P, float sd) {
    float dist = fabs_(sd);
    if (P.cfg.nearest_init) dist = fmin_(dist, P.cfg.max_dis);
    return dist;
}

// ---------------------------------------------------------------- F11 normal
// world: cornell_box_v3/sdf.py:26-31; local: src/sdf.py:77-87 + src/scene.py:87-96
template <int KIND>
RT_D vec3 calc_normal(const Params& P, const ObjFull& o, vec3 p) {
    float h = P.cfg.normal_h;
    if (KIND == KIND_BUNNY || KIND == KIND_MIXED) {
        // same arithmetic as below, but as a rolled loop: one copy of the (large) MLP code instead of four
        const bool world = P.cfg.normal_space == RTPBR_NORMAL_WORLD;
        vec3 q = world ? p : to_local<KIND>(P, o, p);
        vec3 n = mk(0, 0, 0);
#pragma nounroll
        for (int i = 0; i < 4; i++) {
            // tetrahedron offsets (1,-1,-1), (-1,-1,1), (-1,1,-1), (1,1,1)
            float ex = (i == 0 || i == 3) ? 1.0f : -1.0f;
            float ey = (i >= 2) ? 1.0f : -1.0f;
            float ez = (i & 1) ? 1.0f : -1.0f;
            vec3 e = world ? mk(ex * h, ey * h, ez * h) : mk(ex, ey, ez);
            float d = world ? signed_distance<KIND>(P, o, q + e) : sdf_local<KIND>(P, o.type, q + e * h, o.sx, o.sy, o.sz);
            vec3 t = e * d;
            n = (i == 0 && world) ? t : n + t;
        }
        return normalize(n);
    }
    if (P.cfg.normal_space == RTPBR_NORMAL_WORLD) {
        vec3 e0 = mk(h, -h, -h), e1 = mk(-h, -h, h), e2 = mk(-h, h, -h), e3 = mk(h, h, h);
        float d0 = signed_distance<KIND>(P, o, p + e0);
        float d1 = signed_distance<KIND>(P, o, p + e1);
        float d2 = signed_distance<KIND>(P, o, p + e2);
        float d3 = signed_distance<KIND>(P, o, p + e3);
        vec3 n = ((e0 * d0 + e1 * d1) + e2 * d2) + e3 * d3;
        return normalize(n);
    } else {
        vec3 q = to_local<KIND>(P, o, p);
        vec3 e0 = mk(1, -1, -1), e1 = mk(-1, -1, 1), e2 = mk(-1, 1, -1), e3 = mk(1, 1, 1);
        float d0 = sdf_local<KIND>(P, o.type, q + e0 * h, o.sx, o.sy, o.sz);
        float d1 = sdf_local<KIND>(P, o.type, q + e1 * h, o.sx, o.sy, o.sz);
        float d2 = sdf_local<KIND>(P, o.type, q + e2 * h, o.sx, o.sy, o.sz);
        float d3 = sdf_local<KIND>(P, o.type, q + e3 * h, o.sx, o.sy, o.sz);
        vec3 n = mk(0, 0, 0);
        n = n + e0 * d0;
        n = n + e1 * d1;
        n = n + e2 * d2;
        n = n + e3 * d3;
        return normalize(n);
    }
}

// calc_normal for the single-bunny kind with the MLP on the matrix cores: same arithmetic as the rolled loop in
// calc_normal (tetrahedron offsets, signed distance of the one object), wave-cooperative: ALL 64 lanes call it.
// `hit` = this lane's slot holds a hit at `p`.  Only some of the 64 slots hold hits when a shading pass runs, so the
// (hit, offset) pairs are COMPACTED: pair j = 4*rank + e is evaluated by lane j & 63 of batch j >> 6 — one MLP pass
// per 16 hits instead of four passes per shading pass.  Hit positions travel by ds_bpermute (rank -> lane table in
// `tbl`, 64 words of wave-private LDS that are idle during shading), the four distances travel back the same way.
RT_D vec3 bunny_normal_wave(const Params& P, const BunnyFrag& F, float* lds, const float* bias_lds, uint32_t* tbl, int lane,
                            bool hit, vec3 p, uint32_t& n_passes) {
    ObjTab tab = obj_table();
    asm volatile("" : "+s"(tab));
    const ObjM o = tab[0];                 // the single object's transform: scalar operands
    const float h = P.cfg.normal_h;
    const bool world = P.cfg.normal_space == RTPBR_NORMAL_WORLD;
    const unsigned long long m = __ballot(hit);
    const int n_hit = __popcll(m);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
    if (hit) tbl[rank] = (uint32_t)lane;
    bunny_lds_fence();
    const int e_i = lane & 3;                       // tetrahedron offsets (1,-1,-1), (-1,-1,1), (-1,1,-1), (1,1,1)
    const float ex = (e_i == 0 || e_i == 3) ? 1.0f : -1.0f, ey = (e_i >= 2) ? 1.0f : -1.0f, ez = (e_i & 1) ? 1.0f : -1.0f;
    const vec3 e = world ? mk(ex * h, ey * h, ez * h) : mk(ex, ey, ez);
    float d4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int n_batches = (n_hit + 15) >> 4;
    for (int b = 0; b < n_batches; b++) {
        const int halves = (n_hit - b * 16) > 8 ? 2 : 1;      // a last batch of <= 8 hits fills 32 slots only
        const int hidx = b * 16 + (lane >> 2);
        const int src = (int)tbl[hidx < n_hit ? hidx : 0];
        const vec3 hp = mk(__shfl(p.x, src, 64), __shfl(p.y, src, 64), __shfl(p.z, src, 64));
        const vec3 q = world ? hp : to_local<KIND_BUNNY>(P, o, hp);
        const vec3 l = world ? to_local<KIND_BUNNY>(P, o, q + e) : q + e * h;
        const float len = length(l);
        const float sd = bunny_mlp_wave(F, P.bunny, lds, bias_lds, lane, l, lane < 32 * halves ? lane : -1, halves);
        n_passes += (uint32_t)halves;
        const float d = (len > 1.0f) ? len - 0.8f : sd;
        const bool mine = hit && (rank >> 4) == b;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float v = __shfl(d, (4 * rank + k) & 63, 64);
            d4[k] = mine ? v : d4[k];
        }
    }
    vec3 n = mk(0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float kx = (k == 0 || k == 3) ? 1.0f : -1.0f, ky = (k >= 2) ? 1.0f : -1.0f, kz = (k & 1) ? 1.0f : -1.0f;
        const vec3 ek = world ? mk(kx * h, ky * h, kz * h) : mk(kx, ky, kz);
        const vec3 t = ek * d4[k];
        n = (k == 0 && world) ? t : n + t;
    }
    return normalize(n);
}

RT_D float brightness(vec3 c) { return dot(c, mk(0.299f, 0.587f, 0.114f)); }  // src/util.py:31-33

// F15: src/util.py:21-28 random_in_unit_sphere + src/pbr.py:16-19 hemispheric_sampling
RT_D vec3 hemispheric_sampling(vec3 n, uint32_t key, uint32_t& cnt) {
    float a = rng_next(key, cnt), b = rng_next(key, cnt);
    float z = 2.0f * a - 1.0f;
    float ang = b * 2.0f * PI;
    float sn, cs;
    sincos_(ang, &sn, &cs);
    float sq = sqrt_shape_(1.0f - z * z, false);      // (a direction component: no cancellation behind the root)
    vec3 u = mk(sq * sn, sq * cs, z);
    return normalize(n + u);
}

// ---------------------------------------------------------------- F12/F22 surface interaction
// src/pbr.py:22-62; cornell_box_v3/pbr.py:30-66; cornell_box_shortest.py:91-94.
// `origin` is the ray origin (src form: already marched), `pos` the hit position.
template <int KIND, bool HAVE_NORMAL = false>
RT_D void surface_interaction(const Params& P, const ObjFull& o, vec3 pos, vec3& origin, vec3& dir, vec3& col,
                              uint32_t key, uint32_t& cnt, vec3 given_normal = vec3{0, 0, 0}) {
    const rtpbr_config& g = P.cfg;
    vec3 n = HAVE_NORMAL ? given_normal : calc_normal<KIND>(P, o, pos);
    // The material is fetched from LDS only after the normal is done: without this compiler barrier the
    // scheduler requests all 28 dwords of the object record up front and the pool kernel needs 14 more
    // registers across the shading pass (96-VGPR build: 14 spills -> 0; 80-VGPR build: 100 GB of
    // scratch traffic per launch -> 5 GB).
    asm volatile("" ::: "memory");
    vec3 albedo = mk(o.albedo[0], o.albedo[1], o.albedo[2]);
    if (g.surface_kind == RTPBR_SURFACE_DIFFUSE) {
        dir = hemispheric_sampling(n, key, cnt);
        col = col * albedo;
        origin = pos;
        return;
    }
    vec3 I = dir;
    bool outer = dot(I, n) < 0.0f;
    if (!outer) n = -n;
    vec3 hemi = hemispheric_sampling(n, key, cnt);
    float alpha = o.roughness * o.roughness;
    vec3 N = normalize(mix(n, hemi, alpha));
    float NoI = dot(N, I);
    float eta = outer ? g.env_ior / o.ior : o.ior / g.env_ior;
    float k = 1.0f - eta * eta * (1.0f - NoI * NoI);
    float F0;
    if (g.fresnel_kind == RTPBR_FRESNEL_C2) {
        F0 = (eta - 1.0f) / (eta + 1.0f);
        F0 = F0 * (2.0f * F0);
    } else {
        F0 = 2.0f * (eta - 1.0f) / (eta + 1.0f);
        F0 = F0 * F0;
    }
    float x1 = fabs_(1.0f + NoI), x2 = x1 * x1, x5 = x2 * x2 * x1;
    float F = mix(x5, 1.0f, F0);
    if (g.fresnel_roughness_mix) F = mix(F, F0, o.roughness);
    vec3 D;
    float c1 = rng_next(key, cnt);
    if (c1 < F + o.metallic || k < 0.0f) {
        float tn = 2.0f * NoI;
        D = mk(I.x - tn * N.x, I.y - tn * N.y, I.z - tn * N.z);
        if (g.below_horizon == RTPBR_HORIZON_KILL) {
            float keep = dot(D, n) > 0.0f ? 1.0f : 0.0f;
            col = col * keep;
        } else if (dot(D, n) < 0.0f) {
            D = -D;
        }
    } else {
        float c2 = rng_next(key, cnt);
        if (c2 < o.transmission) {
            float f = sqrt_shape_(k, false) + eta * NoI;
            D = mk(eta * I.x - f * N.x, eta * I.y - f * N.y, eta * I.z - f * N.z);
        } else {
            D = hemi;
        }
    }
    dir = D;
    col = col * albedo;
    if (g.origin_mode == RTPBR_ORIGIN_HIT) {
        origin = pos;
    } else {
        float sgn = dot(D, n) < 0.0f ? -1.0f : 1.0f;
        vec3 off = (n * g.min_dis) * sgn;
        origin = origin + off;
    }
}

// ---------------------------------------------------------------- F13 sky
// src/ibl.py:25-29,36-40 + src/util.py:45-50; scene_demo/main.py:245-248,322.  Index clamped (G6).
RT_D vec3 sky_color(const Params& P, vec3 D) {
    if (P.cfg.sky_kind == RTPBR_SKY_GRADIENT) {
        float t = 0.5f * D.y + 0.5f;
        vec3 g = mix(mk(1.0f, 1.0f, 0.5f), mk(0.25f, 0.35f, 1.0f), t);
        return g * 1.8f;
    }
    if (P.cfg.sky_kind == RTPBR_SKY_ENVMAP && P.env != nullptr) {
        float u = atan2_(D.z, D.x) * INV_2PI + 0.5f;
        float v = asin_(D.y) * INV_PI + 0.5f;
        int ew = P.env_w, eh = P.env_h;
#if defined(__HIP_DEVICE_COMPILE__)
        // (the int -> float conversions stay HERE, one instruction each: hoisted out of a persistent kernel's main loop they hold two
        // VGPRs for its whole length — the two registers the src/ pool kernel spilled to scratch, `Scratch_Size 12` in round 5's traces)
        asm volatile("" : "+s"(ew), "+s"(eh));
#endif
        int x = (int)(u * (float)ew), y = (int)(v * (float)eh);
        x = x < 0 ? 0 : (x > ew - 1 ? ew - 1 : x);
        y = y < 0 ? 0 : (y > eh - 1 ? eh - 1 : y);
        if (P.env8 != nullptr) {       // (wave-uniform) RGBA8 texel + the 256-entry table: bit for bit the float texel
            const uint32_t w = P.env8[(size_t)x * eh + y];
            return mk(P.env_lut[w & 255u], P.env_lut[(w >> 8) & 255u], P.env_lut[(w >> 16) & 255u]);
        }
        float4 t = P.env[(size_t)x * eh + y];
        return mk(t.x, t.y, t.z);
    }
    return mk(0, 0, 0);
}

// ---------------------------------------------------------------- F5 camera ray
// src/camera.py:11-36 (frame precomputed on the host); cornell_box_shortest.py:102-118 pinhole.
// RNG order: jitter x, jitter y, lens a, lens b.
RT_D void gen_ray(const Params& P, int px, int py, uint32_t key, uint32_t& cnt, vec3& ro, vec3& rd) {
    const CamFrame& f = P.cam;
    float j1 = rng_next(key, cnt), j2 = rng_next(key, cnt);
    float u, v;
    vec3 lf = mk(f.lf[0], f.lf[1], f.lf[2]);
    ro = lf;
    if (P.cfg.camera_kind == RTPBR_CAMERA_PINHOLE) {
        u = ((float)px + j1) / (float)P.cfg.width;
        v = ((float)py + j2) / (float)P.cfg.height;
    } else {
        u = ((float)px + j1) * f.inv_w;
        v = ((float)py + j2) * f.inv_h;
        float a = rng_next(key, cnt), b = rng_next(key, cnt);
        float ang = b * 2.0f * PI;
        float sn, cs;
        sincos_(ang, &sn, &cs);
        float r = sqrt_shape_(a, false);
        float rx = f.lens_radius * (r * sn), ry = f.lens_radius * (r * cs);
        vec3 off = fma3(ry, mk(f.y[0], f.y[1], f.y[2]), mk(f.x[0], f.x[1], f.x[2]) * rx);
        ro = lf + off;
    }
    vec3 po = fma3(v, mk(f.ver[0], f.ver[1], f.ver[2]), fma3(u, mk(f.hor[0], f.hor[1], f.hor[2]), mk(f.llc[0], f.llc[1], f.llc[2])));
    rd = normalize(po - ro);
}

// ---------------------------------------------------------------- F19 tone map
// src/postprocessor.py:12-38, src/aces.py:5-30; order per variant
RT_D vec3 aces_fit(vec3 c, int trunc) {
    float a1 = trunc ? 0.024578f : 0.0245786f, a2 = trunc ? 0.0000905f : 0.000090537f;
    const float mi[9] = {0.59719f, 0.35458f, 0.04823f, 0.07600f, 0.90834f, 0.01566f, 0.02840f, 0.13383f, 0.83777f};
    float mo[9] = {1.60475f, -0.53108f, -0.07367f, -0.10208f, 1.10813f, -0.00605f, -0.00327f, -0.07276f, 1.07602f};
    if (trunc) {
        mo[1] = -0.531f;
        mo[2] = -0.0736f;
        mo[3] = -0.102f;
    }
    vec3 v = mulv(mi, c);
    float in[3] = {v.x, v.y, v.z}, out[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float x = in[i];
        float a = x * (x + a1) - a2;
        float b = x * (0.983729f * x + 0.4329510f) + 0.238081f;
        out[i] = a / b;
    }
    return mulv(mo, mk(out[0], out[1], out[2]));
}
RT_D float clamp01(float x) { return fmin_(fmax_(x, 0.0f), 1.0f); }
RT_D vec3 clamp01(vec3 c) { return mk(clamp01(c.x), clamp01(c.y), clamp01(c.z)); }
RT_D vec3 pow3(vec3 c, float e) { return mk(pow_(c.x, e), pow_(c.y, e), pow_(c.z, e)); }

RT_D vec3 tone_map(const rtpbr_config& g, float4 b) {
    vec3 c = mk(b.x / b.w, b.y / b.w, b.z / b.w);
    c = c * g.exposure;
    float ig = 1.0f / g.gamma;
    switch (g.tonemap_order) {
        case RTPBR_TONEMAP_GAMMA_ACES_CLAMP:
            return clamp01(aces_fit(pow3(c, ig), g.aces_truncated));
        case RTPBR_TONEMAP_ACES_GAMMA:
            return pow3(aces_fit(c, g.aces_truncated), ig);
        case RTPBR_TONEMAP_ACES_CLAMP_GAMMA:
            return pow3(clamp01(aces_fit(c, g.aces_truncated)), ig);
        default:
            return clamp01(pow3(aces_fit(c, g.aces_truncated), ig));
    }
}

// ---------------------------------------------------------------- pixel enumeration
// Local pixel q (tile-major over the tiles this rank owns, x-major inside a tile, y fastest)
// -> absolute pixel.  Tile t belongs to rank t % world.
RT_D bool pixel_of(const Params& P, uint32_t q, int& x, int& y) {
    uint32_t tpix = (uint32_t)(P.tile_w * P.tile_h);
    uint32_t tl = q / tpix;
    uint32_t r = q - tl * tpix;
    uint32_t lx = r / (uint32_t)P.tile_h;
    uint32_t ly = r - lx * (uint32_t)P.tile_h;
    uint32_t tid = (uint32_t)P.rank + tl * (uint32_t)P.world;
    uint32_t ty = tid / (uint32_t)P.ntx;
    uint32_t tx = tid - ty * (uint32_t)P.ntx;
    x = (int)(tx * (uint32_t)P.tile_w + lx);
    y = (int)(ty * (uint32_t)P.tile_h + ly);
    return x < P.cfg.width && y < P.cfg.height && ty < (uint32_t)P.nty;
}

}  // namespace rt
