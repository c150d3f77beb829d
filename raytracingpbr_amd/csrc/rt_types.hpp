// rt_types.hpp — device-side layout of the scene, camera frame and launch parameters.
//
// HBM / kernarg / LDS placement:
//   Params (kernarg, constant address space -> s_load into SGPRs): config knobs, camera
//     frame, work geometry, and ObjM[n]: the per-object constants the MARCH LOOP reads
//     (position, world->local matrix, shape params).  They are wave-uniform, so they are
//     consumed as scalar operands and cost no VGPRs and no LDS bandwidth.
//   ObjFull[n] (HBM -> LDS once per workgroup): transform + material of every object,
//     gathered PER LANE by hit-object index in the shading phase (T4).
//   stage (HBM): one StageRec (12 B) per pixel-sample of the current sub-launch, [q][k] (k fastest),
//     reduced in sample order into image_buffer (T7) by the accumulate kernel.
#pragma once
#include <stdint.h>

#include "../../include/rtpbr.h"
#include "rt_math.hpp"

namespace rt {

constexpr int MAX_OBJ = RTPBR_MAX_OBJECTS;
static_assert(sizeof(rtpbr_config) == 164 && sizeof(rtpbr_object) == 116 && sizeof(rtpbr_camera) == 52 &&
                  sizeof(rtpbr_ray) == 40,
              "C-ABI struct layout changed");

// Rotation class of an object's world->local matrix (to_local): a matrix whose off-axis entries are
// exactly +-0 and whose axis entry is exactly 1 needs 4 of the 9 products of the general chain.
enum { ROT_GENERAL = 0, ROT_IDENT = 1, ROT_X = 2, ROT_Y = 3, ROT_Z = 4 };
// Rotation SIGNATURE of an 8-box scene: 3 bits per object.  The march kernels are additionally
// instantiated ahead of time for the signatures listed here (the class of every object is then a
// compile-time constant of the unrolled object loop: no branches — a scalar branch per object costs
// more than the products it saves).  rtpbr_set_scene picks the first listed signature the scene is
// compatible with, else the general instance (signature 0).
//   0x4db691: I X X Y Y Y Y X — a room of axis-aligned slabs (90-degree rotations about x or y) with
//             yawed blocks inside, i.e. the Cornell Box layouts of the reference's examples
#define RT_BOX_SIGNATURES(X, ...) X(0x4db691u, __VA_ARGS__)

// A signature instance reads a PACKED object table (host: pack_objects in rt_capi.hip): per object only
// the dwords its rotation class needs, contiguous and 16-byte granular, so that one or two wide scalar
// loads fetch them (the general 64-byte block with its unused matrix entries removed by the compiler
// decays into up to six narrow s_load instructions per object — and every instruction, scalar or
// vector, costs an issue slot):
//   general: px py pz m0..m8 sx sy sz type                     16 dwords
//   X:       px py pz m4 | m5 m7 m8 sx | sy sz - -             12
//   Y:       px py pz m0 | m2 m6 m8 sx | sy sz - -             12
//   Z:       px py pz m0 | m1 m3 m4 sx | sy sz - -             12
//   identity px py pz sx | sy sz - -                            8
constexpr int sig_cls(uint32_t sig, int i) { return (int)((sig >> (3 * i)) & 7u); }
constexpr int sig_words(int cls) { return cls == ROT_GENERAL ? 16 : (cls == ROT_IDENT ? 8 : 12); }
constexpr int sig_offset(uint32_t sig, int i) {
    int o = 0;
    for (int j = 0; j < i; j++) o += sig_words(sig_cls(sig, j));
    return o;
}

// Run-time compiled instances (rt_jit.hip) know every object's SHAPE TYPE at compile time as well: RT_JIT_TYPES holds
// (type + 1) in 4 bits per object, 0 = "read the type from the table" (ahead-of-time instances: always 0).
#ifndef RT_JIT_TYPES
#define RT_JIT_TYPES 0ull
#endif
constexpr int jit_type(int i) { return i < 16 ? (int)((RT_JIT_TYPES >> (4 * i)) & 15ull) - 1 : -1; }
// ... so a shape switch on a PER-LANE type (the object-parallel evaluation of sparse waves, rt_persistent.hpp nearest_op3) only
// needs the cases of the shapes the scene holds
constexpr bool jit_has_type(int t) {
    if (RT_JIT_TYPES == 0ull) return true;
    for (int i = 0; i < 16; i++)
        if ((int)((RT_JIT_TYPES >> (4 * i)) & 15ull) - 1 == t) return true;
    return false;
}
// ... and, with option "jit_bake", the march table itself: RT_JIT_TABLE_FILE is a generated header that defines
//   static constexpr uint32_t RT_JIT_TABLE_BITS[RT_JIT_NOBJ][16]     (the ObjM blocks, bit patterns)
// so that positions, matrix entries and sizes become instruction literals (no scalar loads, no SGPRs for them).
#ifdef RT_JIT_TABLE_FILE
#include RT_JIT_TABLE_FILE
#define RT_JIT_BAKED 1
#else
#define RT_JIT_BAKED 0
#endif

// 16 dwords: what one march step needs from one object
struct ObjM {
    float px, py, pz;
    float m[9];
    float sx, sy, sz;
    int32_t type;
};
static_assert(sizeof(ObjM) == 64, "ObjM must be 64 bytes");

// transform + material for the shading phase, gathered per lane (stored in LDS)
struct ObjFull {
    float px, py, pz;
    float m[9];
    float sx, sy, sz;
    int32_t type;
    float albedo[3];
    float emission[3];
    float roughness, metallic, transmission, ior;
    float pad[2];
};
static_assert(sizeof(ObjFull) == 112, "ObjFull must be 112 bytes");

// camera frame precomputed on the host with the same op order the oracle uses
// (src/camera.py:11-36): only the per-sample part runs on the device.
struct CamFrame {
    float lf[3], x[3], y[3], llc[3], hor[3], ver[3];
    float lens_radius;
    float inv_w, inv_h;
};
static_assert(sizeof(CamFrame) == 21 * 4, "RtJitKey::cam_words holds a CamFrame");

// one staged sample: its three colour words (the count it adds is 1 by definition)
struct __attribute__((packed, aligned(4))) StageRec { float r, g, b; };
static_assert(sizeof(StageRec) == 12, "staging record is 12 bytes");

struct Counters {
    unsigned long long samples, raycasts, march_steps, hits, sky_lookups, deposits;
    // neural SDF (matrix-core path): MLP passes over a wave, and the ray-evaluations those passes were needed for
    unsigned long long mlp_wave_evals, mlp_lane_evals;
    // instrumented builds only (-DRT_DEBUG_PHASE through RTPBR_JIT_EXTRA_FLAGS): cycles per phase, passes, lanes; counters "dbg0".."dbg9", "dbga".."dbgv"
    unsigned long long dbg[32];
    // The six work counters are ADDED UP here by the kernels: 64 shards of one cache line each, shard = blockIdx % 64, word k =
    // march_steps, raycasts, hits, sky_lookups, samples, deposits (flush_counters).  One word per counter saturates at ~90
    // atomics per microsecond: the 8 100 blocks of a one-lane-per-pixel kernel at 1080p took 0.1 ms for that alone.  The host
    // folds the shards into the fields above when it reads them (rt_capi.hip).
    unsigned long long shard[64][16];
};

// The plan of the src/ pool kernel's cost-ordered ownership (device memory; written by the plan kernels, rt_kernels.hip)
struct PlanBuf {
    uint32_t hist[256];           // pixels per cost bucket (log scale, 8 buckets per octave)
    uint32_t cursor[256];         // where the next pixel of a bucket goes in `order` (heaviest bucket first)
    unsigned long long total;     // sum of the costs
    uint32_t n_heavy;             // the first n_heavy entries of `order` are walked by heavy waves
    uint32_t tiny_waves;          // how many waves the small heavy waves may take (the kernel sizes them: 2, 4 or tiny_own pixels each)
    // The CHAIN SET (rt_chain.hpp): when a launch is as long as its heaviest pixel's dependency chain, the first n_chain entries
    // of `order` leave the pool kernel altogether and run in the chain kernel beside it, wave w on entries
    // chain_start[w] .. chain_start[w + 1]: the heaviest pixels alone in a wave, lighter ones in twos, fours, eights.
    uint32_t n_chain, n_chain_waves;
    uint32_t chain_start[2049];
    // ---- everything below survives a re-plan (launch_plan clears the part above only)
    // Age-weighted shares of the light waves (rt_persistent.hpp): residency slot c of a CU (blockIdx / n_cu) takes
    // age_w[c] entries of `order` per round.  Self-tuning: every light wave adds its lifetime (cycles) to its slot's sum,
    // the next plan moves the weights towards equal mean lifetimes.
    uint32_t age_valid;           // age_w holds tuned weights for age_cls slots
    uint32_t age_cls;
    uint32_t age_w[8];
    float age_wf[8];              // the weights before rounding
    unsigned long long life_sum[8];
    uint32_t life_cnt[8];
};

struct Params {
    rtpbr_config cfg;
    CamFrame cam;
    int32_t n_obj;
    // bunny animation (bunny_sdf_glass.py:213-217): sin/cos of t = pi*frame/120, host-computed
    float anim_s, anim_c;
    float anim_bz;          // cfg.anim_bob * anim_s: the frame's vertical bob (bunny_sdf_glass.py:216), host-computed
    // work geometry
    uint32_t sample_base;   // absolute index of the first sample (or bounce-step) of this launch
    int32_t K;              // samples per pixel in this sub-launch
    int32_t tile_w, tile_h, ntx, nty, rank, world;
    int32_t n_local_tiles;  // tiles owned by this rank
    int32_t np;             // padded local pixel count = n_local_tiles*tile_w*tile_h
    uint32_t total_items;   // np*K
    uint32_t chunk;         // work items claimed per atomic
    int32_t wait_lanes;     // k*: leave the march phase when this many lanes wait for shading / a swap
    int32_t shade_lanes;    // pool scheduler: shade when this many parked rays wait
    int32_t refill_lanes;   // pool scheduler: start new pixel-samples when this many slots are free (or the pool runs dry)
    int32_t ready_low;      // pool scheduler: also shade when no more than this many READY rays are parked
    int32_t swap_lanes;     // pool scheduler: swap when this many lanes finished their raycast
    int32_t sparse_lanes;   // src/ pool kernel: tracked-object march steps when at most this many lanes march (0 = only in heavy waves)
    int32_t heavy_own;      // src/ pool kernel: pixels a heavy wave owns (<= 128: they stay resident for the whole launch)
    int32_t n_cu;           // src/ pool kernel: compute units of the device (blocks are placed round-robin: blockIdx / n_cu = residency slot)
    int32_t age_on;         // src/ pool kernel: age-weighted shares of the light waves: 0 off, 1 self-tuned (weights in the plan), 2 = age_pack
    uint32_t age_pack;      // ... entries of `order` per round for the waves of residency slot c: 4 bits each, slot 0 in the lowest
    int32_t tiny_own;       // src/ pool kernel: pixels per small heavy wave (the plan says how many such waves there may be)
    int32_t leave_x8;       // src/ pool kernel: cost of leaving the march loop for a shading pass, in eighths of a march iteration of the wave
    int32_t chain_on;       // src/ pool kernel: 1 = the plan's chain set is walked by the chain kernel (rt_chain.hpp): skip it here
    int32_t src_track;      // src/ pool kernel: 1 = tracked-object march steps enabled
    int32_t src_op;         // src/ kernels: bit 0 = object-parallel evaluation while at most 8 lanes march (rt_persistent.hpp nearest_op3) in the split march
                            // and chain kernels, bit 1 = in the fused pool kernel as well
    int32_t split_head;     // split march kernel: 1 = the list's heavy head is interleaved over the groups (one entry per group), 0 = it fills the first groups
    int32_t heavy_prio;     // src/ pool kernel: 1 = heavy waves raise their issue priority (s_setprio)
    int32_t scheduler;      // 0 = in-register refill, 1 = per-wave LDS ray pool
    int32_t mlp_mfma;       // bunny: 1 = hidden layers on the matrix cores (f32 MFMA, bit-identical), 0 = VALU
    int32_t mlp_lanes;      // bunny: run the MLP when this many lanes wait for it (or none can run ahead)
    int32_t mlp_full;       // bunny: compute both 32-slot halves when at least this many wait, else the first 32
    // pointers
    float* stage;           // 3 floats per work item (StageRec)
    // dense staging (round 6): a wave appends the records of a claimed chunk in COMPLETION order to the chunk's own region of `stage`
    // (coalesced, no line stays partially written for a path's lifetime) with the sample's offset in the chunk beside it;
    // accumulate_dense puts them back in sample order in LDS.  0 = item-linear records (stage[item])
    int32_t stage_dense;
    uint32_t acc_batch;     // accumulate_dense: items per block pass (a multiple of both K and chunk)
    uint32_t acc_magic_k, acc_magic_chunk;   // floor(2^32 / d) + 1: __umulhi(i, magic) == i / d for i * d < 2^32
    uint8_t* stage_idx;     // per record: item - first item of its chunk (chunk <= 256)
    uint32_t* stage_fill;   // per chunk: records appended so far (zeroed before the launch)
    float* primary;         // per item: t_eval of the primary raycast, written by primary_rays, read by the pool kernel
    uint8_t* primary_code;  // per item: idx | state << 5 (32 objects, 5 states) — 5 bytes per item in all, not a float2 (round 6)
    int32_t primary_split;  // 1 = primary raycasts run in their own coherent lock-step kernel
    int32_t drain_lanes;    // pool kernel, work exhausted: at most this many marching lanes go on in the culled wave march (0 = never)
    int32_t primary_lean;   // ... whose march goes on in a one-object loop while the whole wave needs one object only (primary_lean_obj)
    uint32_t box_sig;       // host side: which RT_BOX_SIGNATURES instance to launch (0 = general)
    int32_t cull_ok;        // host side: every shape is 1-Lipschitz and n_obj <= 8: primary_rays may cull (nearest_culled)
    float cull_extent;      // nearest_culled: scale of the rounding allowance
    int32_t box_lazy;       // nearest_boxes_lazy enabled (option lazy_sqrt, box_round >= 0)
    float box_four_rho;     // 4 * box_round (keys are carried scaled by 4, see nearest_boxes_lazy)
    float box_rho2m;        // 4 * box_round^2 * (1 + 2^-19): a smaller key means an object is in its rounding shell
    float box_4rho2m;       // 4 * (2 box_round)^2 * (1 + 2^-19): a larger key means that object is farther than box_round
    float4* image_buffer;   // T7 (W,H) float4
    float* image_pixels;    // T8 (W,H,3)
    rtpbr_ray* ray_buffer;  // T6
    float2* diff_buffer;    // T11
    float* diff_pixels;     // T11
    const ObjFull* objfull;
    const float4* env;      // T9 as float4 texels [x][y]
    int32_t env_w, env_h;
    // T9 as SURVEY.md names it for an 8-bit source (src/ibl.py:14-23: the image is uint8, every channel takes one of 256 values after
    // Image.process): RGBA8 texels + the 256-entry table of (c / 255 * exposure)^gamma — the same floats, a quarter of the bytes.
    // nullptr = use `env` (option env_packed, or the map was uploaded as floats)
    const uint32_t* env8;
    const float* env_lut;
    const float* bunny;     // 625 weights
    unsigned int* work_counter;
    Counters* counters;
    // the work counters (and the claim counters behind them) of the NEXT rtpbr_sample() call, zeroed by the first block of this call's
    // kernels (zero_next_counters): two buffers taken in turn — a fill of its own between two small launches is 4.5 us of 160
    Counters* counters_next;
    // src/ pool kernel, cost-ordered ownership (rt_plan.hpp): march steps per local pixel since the last plan (accumulated at
    // write-back), the local pixels ordered by that cost (heaviest first; nullptr = no plan yet: identity), and the plan
    uint32_t* cost_buffer;
    unsigned int* team_counter;   // split march kernel: one claim counter per team of blocks, 64 bytes apart (zeroed by src_gen)
    int32_t n_teams;
    uint32_t shade_base;    // src_shade_gen: the bounce-step whose shading this launch does first (the lazy shading of one-step launches; fills a hole:
                            // a field in front of K cost the unbaked complete-path instances 4 % — scalar loads regrouped in a kernel at its register budget)
    uint32_t* march_out;    // wavefront split of the src/ form (rt_split.hpp): one word per local pixel between its three kernels
    const uint32_t* order;
    struct PlanBuf* plan;
    ObjM objm[MAX_OBJ];
};
static_assert(sizeof(Params) < 3900, "Params must fit the 4 KB kernarg segment");

}  // namespace rt
