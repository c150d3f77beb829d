// rt_ctx.hpp — the context behind the C ABI (shared by rt_capi.hip and rt_rccl.hip).
//
// One rtpbr_ctx owns one HIP stream and every device buffer of one renderer instance (what the Taichi runtime owns
// in the reference: src/fileds.py:7-15, src/scene.py:38-41, src/ibl.py:16-17).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "rt_device.hpp"

namespace rt {
void launch_trace(const Params& P, int kind, int grid, hipStream_t st);
void launch_accumulate(const Params& P, int n_cu, hipStream_t st);
void launch_zero(void* p, size_t bytes, hipStream_t st);      // a small fill as a kernel of our own (bytes % 16 == 0)
void launch_persistent(const Params& P, int kind, int steps, hipStream_t st);
void launch_persistent_pool(const Params& P, int kind, int steps, int grid, hipStream_t st);
int persistent_pool_blocks_per_cu(int kind);
void launch_chain_steps(const Params& P, int kind, int steps, int grid, hipStream_t st);
void launch_src_gen(const Params& P, int kind, hipStream_t st);
void launch_src_march(const Params& P, int kind, int grid, hipStream_t st);
int src_march_blocks_per_cu(int kind);
void launch_src_shade(const Params& P, int kind, hipStream_t st);
void launch_src_shade_gen(const Params& P, int kind, bool count, hipStream_t st);      // the previous bounce-step's shading + this one's gen in one pass
void launch_refresh(float4* ib, rtpbr_ray* rb, float2* db, float* dp, int adaptive, size_t n, hipStream_t st);
void launch_post_process(const Params& P, hipStream_t st);
void launch_pack(const Params& P, float4* dst, hipStream_t st);
void launch_unpack(const Params& P, const float4* src, hipStream_t st);
void launch_math_probe(int op, const float* a, const float* b, float* out, float* out2, int n, hipStream_t st);
int trace_blocks_per_cu(int kind, int n_obj, uint32_t box_sig, int scheduler);
void launch_primary(const Params& P, int kind, int n_cu, hipStream_t st);
void launch_sqrt_exhaustive(unsigned long long* mismatches, hipStream_t st);
void launch_plan(uint32_t* cost, uint32_t* order, PlanBuf* plan, uint32_t np, uint32_t n_waves, int heavy_own, int mean_x16, int bulk_x16,
                 int tiny_waves, int n_cu, int n_cls, int chain_on, uint32_t chain_ref_waves, hipStream_t st);      // chain_ref_waves: waves of the pool grid beside the chain kernel
}  // namespace rt

// run-time compiled per-scene instances (rt_jit.hip)
struct RtJitKey {
    int kind, n_obj;
    unsigned long long types;   // (shape type + 1) in 4 bits per object
    unsigned sig;               // rotation class in 3 bits per object
    int cull, waves;
    int form;                   // 0 = complete-path kernels, 1 = persistent-ray kernels (only those are compiled)
    int baked;                  // 1: the march table and the render configuration are baked into the code object
    int fast;                   // 1: the tolerance flavour (RT_FAST_MATH, rt_math.hpp): hardware sqrt / rcp / sin / exp, contraction
    int dense;                  // 1: complete-path pool kernel compiled with the dense staging (RT_STAGE_DENSE, rt_trace.hpp stage_sample)
    const unsigned* table;      // n_obj x 16 words (ObjM blocks)
    unsigned cfg_words[sizeof(rtpbr_config) / 4];   // rtpbr_config with seed and frame zeroed
    unsigned extra[4];          // box_lazy, box_four_rho, box_rho2m, box_4rho2m (bit patterns)
    int ints[8];                // tile_w, tile_h, ntx, nty, world, shade_lanes, swap_lanes, mlp_mfma
    unsigned cam_words[21];     // baked == 2: the camera frame (rt::CamFrame) as well — fixed-camera offline renders
};
struct RtJitModule {
    hipModule_t module = nullptr;
    hipFunction_t trace = nullptr, primary = nullptr, persistent_pool = nullptr, persistent_steps = nullptr;
    hipFunction_t chain_steps = nullptr;                                           // the chain kernel (rt_chain.hpp)
    hipFunction_t src_shade_gen = nullptr, src_shade_gen_count = nullptr;
    hipFunction_t src_gen = nullptr, src_march = nullptr, src_shade = nullptr;      // the wavefront split of one src/ bounce-step (rt_split.hpp)
    int trace_blocks_per_cu = 0, persistent_blocks_per_cu = 0, march_blocks_per_cu = 0;
    std::string path;
    int device = 0;
    int pins = 0;                       // contexts whose last rtpbr_sample() used this instance (never unloaded while > 0)
    unsigned long long last_use = 0;    // LRU clock of the registry
};
struct rtpbr_ctx;
int rt_jit_build(const RtJitKey& key, std::string* out, bool* deterministic);
std::string rt_jit_catalog_dir();       // code objects shipped with the library (read-only; filled by `python -m raytracingpbr_amd.prebuild`)
int rt_jit_acquire(rtpbr_ctx* c, const RtJitKey& key, RtJitModule** out);   // returns the instance pinned
void rt_jit_release(RtJitModule* m);
int rt_jit_launch(hipFunction_t f, const rt::Params& P, unsigned grid, hipStream_t st);
int rt_jit_launch_steps(hipFunction_t f, const rt::Params& P, int steps, unsigned grid, hipStream_t st);

// error channel: thread-local message behind rtpbr_last_error() (defined in rt_capi.hip)
int rt_fail(int code, const char* fmt, const char* a = "");
int rt_fail_hip(const char* expr, hipError_t e);
#define RT_HIP_TRY(expr)                                         \
    do {                                                         \
        hipError_t e_ = (expr);                                  \
        if (e_ != hipSuccess) return rt_fail_hip(#expr, e_);     \
    } while (0)

struct rtpbr_ctx {
    using Params = rt::Params;
    using ObjFull = rt::ObjFull;
    using ObjM = rt::ObjM;
    using Counters = rt::Counters;
    static constexpr int MAX_OBJ = rt::MAX_OBJ;
    int device = 0;
    bool headless = false;            // no device behind this context (rtpbr_jit_prebuild): host-side state and derivations only
    hipStream_t stream = nullptr;
    bool have_cfg = false, have_scene = false, have_cam = false;
    rtpbr_config cfg{};
    rtpbr_object obj[MAX_OBJ];
    int n_obj = 0;
    int kind = rt::KIND_GENERIC;
    rtpbr_camera cam{};
    Params P{};
    // device buffers
    float4* image_buffer = nullptr;
    float* image_pixels = nullptr;
    rtpbr_ray* ray_buffer = nullptr;
    float2* diff_buffer = nullptr;
    float* diff_pixels = nullptr;
    ObjFull* objfull = nullptr;
    float4* env = nullptr;
    uint32_t* env8 = nullptr;        // the same map as RGBA8 texels + a 256-entry table (8-bit sources only; option env_packed)
    float* env_lut = nullptr;
    int env_packed = 1;              // 1 (default): the kernels read env8 + env_lut instead of the float4 texels when the map came as 8-bit texels — same values, same speed (C4 5640-5650 Msamples/s either way), FETCH_SIZE of the trace kernel 13.3 -> 10.5 GB per launch
    float* bunny = nullptr;
    float* stage = nullptr;          // 12 bytes per pixel-sample of a launch (rt::StageRec)
    size_t stage_cap = 0;  // bytes
    float* primary = nullptr;   // primary records: items floats (t_eval), then items bytes (idx | state << 5)
    size_t primary_cap = 0;
    int drain_lanes = 16;    // complete-path pool kernel: culled wave march for the drain (<= this many lanes marching, work exhausted)
    int primary_lean = 1;    // one-object lean loop in the primary kernel
    int primary_split = 1;
    int specialize = 1;      // use the RT_BOX_SIGNATURES instance the scene fits
    int lazy_sqrt = 1;       // all-box scenes: nearest box on squared distances (nearest_boxes_lazy)
    uint32_t scene_sig = 0;
    ObjM objm[MAX_OBJ];      // march table in its general layout; P.objm is filled per launch (pack_objects)
    unsigned int* work_counter = nullptr;   // (behind `counters`, same allocation)
    int timing = 1;               // option timing: events around the kernels of rtpbr_sample()
    bool total1_recorded = false; // the last rtpbr_sample() recorded ev_total1 (work followed its last timed kernel)
    Counters* counters = nullptr;        // the buffer of the LAST rtpbr_sample() call (what rtpbr_get_counters reads): counters_buf[0] or [1]
    Counters* counters_buf[2] = {nullptr, nullptr};   // taken in turn; a call's kernels zero the other one for the next call (zero_next_counters)
    bool counters_clean[2] = {false, false};
    int counters_turn = 0;
    // tiles
    int tile_w = 0, tile_h = 0, rank = 0, world = 1;
    // progress
    uint32_t sample_base = 0;
    // options
    long long staging_bytes = 16LL << 30;  // 288 GB of HBM: a whole 1080p x 256 spp step (8.5 GB of samples + 4.2 GB of primary records) is one launch
    int wait_lanes = 24;
    int shade_lanes = 56;
    int refill_lanes = 24;
    int ready_low = 4;
    int jit_waves = 0;            // waves per SIMD the run-time pool kernel is compiled for (0 = as the ahead-of-time instances)
    int chunk = 0;                // work items claimed per atomic by the pool kernels (0 = automatic)
    // complete-path pool kernel: 1 = records appended per claim in completion order (rt_trace.hpp stage_sample; run-time instances only: the
    // code is compiled in by -DRT_STAGE_DENSE=1).  Measured on the headline step (round 6): trace-kernel WRITE 11.06 -> 8.75 GB, but trace
    // 96.3 -> 101.6 ms (the append is ~6 instructions per sample in a kernel that issues 190 per sample, plus a fill-count round trip per
    // pass) and accumulate 1.5 -> 4.0 ms: off by default
    int stage_dense = 0;
    unsigned long long dense_launches = 0;      // ... launches of the last rtpbr_sample() that did (counter "dense_launches")
    int residency = 32;           // src/ form, pool scheduler: bounce-steps a pixel stays resident when a wave owns more pixels than it holds
    int sparse_lanes = 24;        // src/ form, pool scheduler: tracked-object march steps when at most this many lanes march (heavy waves: always)
    // src/ form, pool scheduler: cost-ordered ownership (rt_persistent.hpp, plan kernels in rt_kernels.hip)
    int src_plan = 1;             // 1 = re-plan the ownership from the measured per-pixel cost
    int plan_interval = 64;       // ... once at least this many bounce-steps have been recorded since the last plan
    int heavy_own = 80;           // pixels per heavy wave (<= 128)
    int age_on = 1;               // age-weighted shares of the light waves (residency slot k of a CU = k-th oldest wave of its SIMD)
    int age_w[8] = {8, 8, 8, 8, 8, 8, 8, 8};
    int tiny_waves = 64;          // small heavy waves for the very heaviest pixels (a quarter of the grid when the launch is chain-bound)
    int tiny_own = 8;             // pixels per small heavy wave (2 or 4 when the budget allows)
    int leave_x8 = 24;            // a shading pass costs the marching lanes about 3 march iterations
    int src_track = 2;            // tracked-object march steps (heavy waves, sparse phases): 0 off, 1 one-object bounds, 2 also the two-object lean loop
    int src_op = 7;               // sparse-wave evaluations of round 6: bit 0 object-parallel nearest() while at most 8 lanes march (split march + chain kernel), bit 1 the same in the fused pool kernel (builds with RT_POOL_OP), bit 2 the per-lane lean loop
    int heavy_prio = 1;           // heavy waves run at raised issue priority
    int heavy_mean_x16 = 48;      // a pixel is heavy when its cost exceeds 3 x the mean pixel ...
    int heavy_bulk_x16 = 8;       // ... and half a wave's share of the frame (in march iterations)
    uint32_t* cost_buffer = nullptr;   // np x u32
    unsigned int* team_counter = nullptr;   // 1024 counters x 64 bytes (split march kernel)
    std::vector<void*> host_blocks;         // page-locked host memory handed out by rtpbr_host_alloc (freed with the context)
    std::vector<size_t> host_sizes;         // ... their sizes (rtpbr_read_buffer_async checks its destination against them)
    // asynchronous read-back (rtpbr_read_buffer_async): a copy stream, a ring of tickets, and per buffer the copy that still reads it
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_read_ready = nullptr;     // "everything enqueued so far" on the context's stream, as the copy stream sees it
    hipEvent_t ev_read_done[8] = {};        // ticket t -> slot t % 8
    int read_issued = 0;                    // tickets handed out so far (the next ticket)
    int read_pending[5] = {-1, -1, -1, -1, -1};   // per RTPBR_BUF_*: the newest ticket whose copy reads it (-1: none that a writer would have to wait for)
    uint32_t* march_out = nullptr;     // np x u32 (wavefront split, rt_split.hpp); sized with cost_buffer
    size_t march_np = 0;
    int src_chain = 1;            // src/ form, fused launches: the plan's chain set runs in the chain kernel beside the pool kernel (rt_chain.hpp)
    long long chain_np_max = 2500000;   // ... frames of more local pixels than this are throughput-bound: no chain set
    int chain_waves = 1024;       // ... the most waves the chain set may take (<= 2048)
    int plan_chain_waves = 0;     // ... waves of the chain set the LAST plan made (0: not chain-bound / no plan yet), read back asynchronously:
    uint32_t* plan_rb_host = nullptr;   // page-locked word the copy lands in
    hipEvent_t ev_plan_rb = nullptr;
    bool plan_rb_pending = false;
    hipStream_t stream2 = nullptr;      // ... on this stream
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int src_split = 1;            // src/ form: launches of at most this many bounce-steps run as the wavefront split (gen / march / shade per step); 0 = never (measured at 1080p: one step 0.51 against 0.60 ms fused; two steps 1.3 against 0.65)
    int split_wait = 24;          // ... its march kernel refills when this many lanes are free
    // LAZY SHADING of the split's launches (option src_lazy): the shading of a bounce-step is not launched with it — the next launch's gen
    // pass does it first (src_shade_gen: one pass over ray_buffer instead of two), and anything that could see the difference flushes it
    // (flush_shade in rt_capi.hip: readers and writers of ray_buffer, the work counters, every setter, refresh)
    int src_lazy = 1;
    bool shade_pending = false;
    uint32_t shade_pending_base = 0;
    bool shade_pending_same_call = false;      // ... of an earlier step of the rtpbr_sample() call in progress: it counts into this call's counters
    bool ray_ptr_out = false;     // rtpbr_buffer_device_ptr handed ray_buffer out: its holder reads it unannounced, no lazy shading any more
    int split_head = -1;          // ... the list's heavy head interleaved over the groups: -1 = for small frames (two waves per SIMD), 0 never, 1 always
    uint32_t* order = nullptr;         // np x u32
    rt::PlanBuf* plan = nullptr;
    size_t plan_np = 0;                // pixels the three buffers are sized for
    bool order_valid = false;
    long long cost_steps = 0;          // bounce-steps recorded in cost_buffer since the last plan
    int grid_blocks = 0;          // src/ form, pool scheduler: workgroups to launch (0 = automatic); tuning / test knob
    int swap_lanes = 0;           // 0 = automatic: 8 in the complete-path pool kernel, 12 in the src/ pool kernel (measured: rt_capi.hip)
    int mlp_lanes = 24;
    int mlp_full = 56;
    int mlp_mfma = 1;
    int scheduler = -1;  // -1 = auto (the LDS ray pool in both kernel forms), 0 = one lane per item / pixel, 1 = pool
    int waves_per_cu = 0;  // 0 = from the occupancy query
    // timing
    std::vector<hipEvent_t> ev;
    int ev_used = 0;
    std::vector<hipEvent_t> evp;   // pairs around the primary_rays launches
    int evp_used = 0;
    hipEvent_t ev_total1 = nullptr;      // end of the last rtpbr_sample() where work follows its last timed kernel (its start = its first event)
    bool timed = false;
    int n_cu = 256;
    // run-time compiled instance of the current scene (rt_jit.hip): -1 = when no ahead-of-time specialisation serves
    // the scene, 0 = never, 1 = always (falling back to the ahead-of-time kernels if it cannot be built), 2 = always, an
    // error otherwise
    int jit = -1;
    int precision = 0;                // 0 = exactly rounded arithmetic (bit-identical with the oracle), 1 = the tolerance flavour (run-time instances only)
    int jit_bake = 0;                 // 1: run-time instances carry the scene's constants as literals
    RtJitModule* jit_mod = nullptr;   // the one the last rtpbr_sample() used, pinned until the next one (nullptr = ahead-of-time instance)
    unsigned jit_sig = 0;
    // multi-GPU gather (rt_rccl.hip): communicator handle (ncclComm_t) and the packed-tile buffers
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    void* gather_send = nullptr;
    void* gather_recv = nullptr;
    size_t gather_cap = 0;        // bytes of gather_send
    size_t gather_recv_cap = 0;   // bytes of gather_recv (rank 0: local share x world)
};
int rt_order_after_reads(rtpbr_ctx* c, unsigned mask);      // rt_capi.hip: writers of the buffers in `mask` wait (on the device) for asynchronous read-backs of them
void rt_rccl_release(rtpbr_ctx* c);
int rt_rccl_check_async(rtpbr_ctx* c);      // RTPBR_OK when the context has no communicator

