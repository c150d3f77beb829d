// rt_jit_tu.hip — the translation unit rtpbr compiles AT RUN TIME for one scene (rt_jit.hip drives hipcc --genco).
// Everything the ahead-of-time library specialises only for listed cases is a compile-time constant here:
//   RT_JIT_KIND   KIND_BOXES (all boxes: nearest box on squared distances) or KIND_GENERIC (analytic shapes)
//   RT_JIT_NOBJ   the object count (the object loop is fully unrolled, two objects per scalar-load group)
//   RT_JIT_TYPES  (shape type + 1) in 4 bits per object: the per-object shape switch disappears
//   RT_JIT_SIG    rotation class in 3 bits per object: identity / single-axis rotations use 0 / 4 of the 9 products
//                 and the packed object table (rt_types.hpp)
//   RT_JIT_CULL   the camera rays' wave-level Lipschitz culling is valid for this scene (host check)
//   RT_JIT_WAVES  waves per SIMD the pool kernel is compiled for
// Same arithmetic as the ahead-of-time instances: results are bit-identical (tests/test_gpu_jit.py).
#include "rt_trace.hpp"

// With option "jit_bake" the generated header (RT_JIT_TABLE_FILE) also carries the march table and the render
// configuration: RT_JIT_BAKE_PARAMS overwrites those fields of a local copy of the launch arguments with compile-time
// constants, and constant propagation does the rest (variant branches fold, thresholds become literals).
#ifndef RT_JIT_BAKE_PARAMS
#define RT_JIT_BAKE_PARAMS(Q)
#endif
namespace rt {
// the neural-SDF kinds keep their own object handling (one object, table row 0): no unrolled object loop
constexpr int TU_NOBJ = (RT_JIT_KIND == KIND_BUNNY || RT_JIT_KIND == KIND_MIXED) ? 0 : RT_JIT_NOBJ;
// RT_JIT_FORM: 0 = the complete-path kernels only, 1 = the persistent-ray kernels only (halves the compile time of a scene's
// first use), anything else = all four
#ifndef RT_JIT_FORM
#define RT_JIT_FORM 2
#endif
#if RT_JIT_FORM != 1
extern "C" __global__ void __launch_bounds__(256, RT_JIT_WAVES) rt_jit_trace(const Params P) {
    Params Q = P;
    RT_JIT_BAKE_PARAMS(Q);
    trace_paths_pool_impl<RT_JIT_KIND, TU_NOBJ, RT_JIT_SIG>(Q);
}
extern "C" __global__ void __launch_bounds__(256) rt_jit_primary(const Params P) {
    Params Q = P;
    RT_JIT_BAKE_PARAMS(Q);
    primary_rays_impl<RT_JIT_KIND, TU_NOBJ, RT_JIT_SIG, (RT_JIT_CULL != 0)>(Q);
}
#endif
#if RT_JIT_FORM != 0
// src/ persistent-ray form (pathtrace() of src/pathtracer.py:94-103): `steps` bounce-steps per pixel and launch
// (RT_JIT_WAVES_SRC: the persistent pool kernel's own occupancy target; the complete-path box instances use 6, this kernel 5)
#ifndef RT_JIT_WAVES_SRC
#define RT_JIT_WAVES_SRC (RT_JIT_WAVES > 5 ? 5 : RT_JIT_WAVES)
#endif
extern "C" __global__ void __launch_bounds__(256, RT_JIT_WAVES_SRC) rt_jit_persistent_pool(const Params P, int steps) {
    Params Q = P;
    RT_JIT_BAKE_PARAMS(Q);
    persistent_pool_impl<RT_JIT_KIND, TU_NOBJ, RT_JIT_SIG>(Q, steps);
}
// ... and one bounce-step as a wavefront split (rt_split.hpp): what a launch of ONE step runs
extern "C" __global__ void __launch_bounds__(256) rt_jit_src_gen(const Params P) {
    Params Q = P;
    RT_JIT_BAKE_PARAMS(Q);
    src_gen_impl<RT_JIT_KIND>(Q);
}
extern "C" __global__ void __launch_bounds__(256) rt_jit_src_shade_gen(const Params P) {
    Params Q = P;
    RT_JIT_BAKE_PARAMS(Q);
    src_gen_impl<RT_JIT_KIND, true, false>(Q);
}
extern "C" __global__ void __launch_bounds__(256) rt_jit_src_shade_gen_count(const Params P) {
    Params Q = P;
    RT_JIT_BAKE_PARAMS(Q);
    src_gen_impl<RT_JIT_KIND, true, true>(Q);
}
extern "C" __global__ void __launch_bounds__(256) rt_jit_src_march(const Params P) {
    Params Q = P;
    RT_JIT_BAKE_PARAMS(Q);
    src_march_impl<RT_JIT_KIND, TU_NOBJ, RT_JIT_SIG>(Q);
}
extern "C" __global__ void __launch_bounds__(256) rt_jit_src_shade(const Params P) {
    Params Q = P;
    RT_JIT_BAKE_PARAMS(Q);
    src_shade_impl<RT_JIT_KIND>(Q);
}
extern "C" __global__ void __launch_bounds__(256) rt_jit_chain_steps(const Params P, int steps) {
    Params Q = P;
    RT_JIT_BAKE_PARAMS(Q);
    chain_steps_impl<RT_JIT_KIND, TU_NOBJ, RT_JIT_SIG>(Q, steps);
}
extern "C" __global__ void __launch_bounds__(256) rt_jit_persistent_steps(const Params P, int steps) {
    Params Q = P;
    RT_JIT_BAKE_PARAMS(Q);
    persistent_steps_impl<RT_JIT_KIND, TU_NOBJ, RT_JIT_SIG>(Q, steps);
}
#endif
}  // namespace rt
