/*
 * rtpbr.h — C ABI of the MI355X-native SDF path-tracing sample path.
 *
 * This is the drop-in boundary of SURVEY.md §8(b).  In the reference the boundary is the
 * Taichi kernel launch from Python: the kernels `pathtrace()` (src/pathtracer.py:94-103),
 * `refresh()` (src/renderer.py:12-22) and `post_process()` (src/postprocessor.py:24-43) take
 * no arguments and read/write global Taichi fields; the examples pass the camera by value
 * (examples/cornell_box/cornell_box_v3/renderer.py:11-42,
 *  examples/bunny/bunny_sdf_glass.py:393-432, examples/scene_demo/tokyo_ibl.py:403-439).
 * Here every field becomes a buffer owned by an opaque context and every kernel launch
 * becomes one plain C function, so that a ctypes / cgo / JNI / N-API stub can bind it.
 *
 * Conventions
 *   - every function returns 0 on success and a negative RTPBR_E* code on failure;
 *     rtpbr_last_error() returns a thread-local human readable message;
 *   - host pointers passed in are copied; the caller keeps ownership;
 *   - calls on one context are serialised on one HIP stream; different contexts
 *     (e.g. one per GPU) are independent; a context is not thread safe;
 *   - image buffers use the reference's field layout: dense (W, H, C) float32,
 *     element [i][j][c] with i = column from the left, j = row from the BOTTOM,
 *     j fastest.
 *
 * All structs are plain-old-data with 4-byte members only (no padding).
 */
#ifndef RTPBR_H
#define RTPBR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ error codes */
#define RTPBR_OK            0
#define RTPBR_EINVAL       -1   /* bad argument / inconsistent state            */
#define RTPBR_EHIP         -2   /* a HIP runtime call or kernel launch failed    */
#define RTPBR_ENOMEM       -3   /* allocation failed                             */
#define RTPBR_ESTATE       -4   /* call order violated (e.g. sample before scene)*/

/* ------------------------------------------------------------------ shapes
 * reference: class SHAPE(IntEnum) src/sdf.py:12-18.  RTPBR_SHAPE_BUNNY is the neural
 * SDF of examples/bunny/bunny_sdf_glass.py:149-203 (its own file calls it SHAPE_BUNNY=1). */
enum {
    RTPBR_SHAPE_NONE = 0,
    RTPBR_SHAPE_SPHERE = 1,
    RTPBR_SHAPE_BOX = 2,
    RTPBR_SHAPE_CYLINDER = 3,
    RTPBR_SHAPE_CONE = 4,
    RTPBR_SHAPE_PLANE = 5,
    RTPBR_SHAPE_BUNNY = 6
};

#define RTPBR_MAX_OBJECTS 32

/* ------------------------------------------------------------------ data model (T1..T5) */

/* reference: Ray src/dataclass.py:5-10 (40 B).  depth = raycast count, its SIGN is the
 * "finished" flag, 0 = fresh after refresh().  The examples' Ray has no depth. */
typedef struct rtpbr_ray {
    float   origin[3];
    float   direction[3];
    float   color[3];
    int32_t depth;
} rtpbr_ray;

/* reference: Material src/dataclass.py:13-20 (40 B).  emission is MULTIPLICATIVE,
 * (1,1,1) means "not a light". */
typedef struct rtpbr_material {
    float albedo[3];
    float emission[3];
    float roughness;
    float metallic;
    float transmission;
    float ior;
} rtpbr_material;

/* reference: Transform src/dataclass.py:23-28 (72 B).  rotation = Euler DEGREES,
 * scale = shape parameters (never a real scale), matrix = world->local rotation,
 * row major, filled by rtpbr_set_scene() like update_transform() src/scene.py:99-103. */
typedef struct rtpbr_transform {
    float position[3];
    float rotation[3];
    float scale[3];
    float matrix[9];
} rtpbr_transform;

/* reference: SDFObject src/dataclass.py:31-35 (116 B). */
typedef struct rtpbr_object {
    int32_t         type;
    rtpbr_transform transform;
    rtpbr_material  material;
} rtpbr_object;

/* reference: Camera src/dataclass.py:38-46 (52 B); vfov in degrees. */
typedef struct rtpbr_camera {
    float lookfrom[3];
    float lookat[3];
    float vup[3];
    float vfov;
    float aspect;
    float aperture;
    float focus;
} rtpbr_camera;

/* ------------------------------------------------------------------ variant knobs
 * One field per row of SURVEY.md Appendix B.  The reference fixes these as Python module
 * constants consumed through ti.static at JIT time (src/config.py:7-28 and the constant
 * block at the top of every example); here they are runtime values of one POD struct. */

enum { RTPBR_FORM_COMPLETE_PATH = 0,   /* examples: for i in range(MAX_RAYTRACE) inside the kernel */
       RTPBR_FORM_PERSISTENT_RAY = 1 };/* src/: one bounce-step per launch, state in ray_buffer   */

enum { RTPBR_MARCH_PLAIN = 0,          /* cornell_box.py:213-223, cornell_box_v2.py:186-196, shortest:63-72 */
       RTPBR_MARCH_RELAXED = 1,        /* cornell_box_v3/pathtracer.py:52-78 (err = d/t), tokyo, bunny      */
       RTPBR_MARCH_SRC = 2 };          /* src/scene.py:59-84 (origin moves, d < t*PIXEL_RADIUS)             */

enum { RTPBR_NORMAL_WORLD = 0,         /* examples: offsets in world space                   */
       RTPBR_NORMAL_LOCAL = 1 };       /* src/sdf.py:77-87: local frame, not rotated back    */

enum { RTPBR_RR_EXAMPLES = 0,          /* p = 1-exp(-i/light_quality); "kill" = color*=p, keep */
       RTPBR_RR_SRC = 1 };             /* p = (depth==0?1:q) - depth/MAX_RAYTRACE; kill = 0    */

enum { RTPBR_FRESNEL_C2 = 0,           /* F0=(e-1)/(e+1); F0*=2*F0  (cornell v1/v2/v3, bunny)  */
       RTPBR_FRESNEL_C4 = 1 };         /* F0=2(e-1)/(e+1); F0*=F0   (src, scene_demo, tokyo)   */

enum { RTPBR_HORIZON_KILL = 0,         /* examples: color *= (dot(D,n) > 0)                    */
       RTPBR_HORIZON_FLIP = 1 };       /* src/pbr.py:50-51: D = -D                             */

enum { RTPBR_ORIGIN_HIT = 0,           /* examples: origin = hit position                      */
       RTPBR_ORIGIN_OFFSET = 1 };      /* src/pbr.py:59-60: origin += +-n*MIN_DIS              */

enum { RTPBR_SURFACE_FULL = 0,
       RTPBR_SURFACE_DIFFUSE = 1 };    /* cornell_box_shortest.py:91-94                        */

enum { RTPBR_SKY_BLACK = 0,            /* cornell: miss => color = 0                           */
       RTPBR_SKY_ENVMAP = 1,           /* src/ibl.py:36-40, tokyo, bunny                       */
       RTPBR_SKY_GRADIENT = 2 };       /* scene_demo/main.py:245-248                           */

enum { RTPBR_PRIMARY_AS_SKY = 0,
       RTPBR_PRIMARY_BLACK = 1,        /* src BLACK_BACKGROUND, bunny_sdf.py:352               */
       RTPBR_PRIMARY_WHITE = 2 };      /* bunny_sdf_v2.py:355-356                              */

enum { RTPBR_TONEMAP_GAMMA_ACES_CLAMP = 0,  /* src, shortest, v3                               */
       RTPBR_TONEMAP_ACES_GAMMA = 1,        /* v1, v2                                          */
       RTPBR_TONEMAP_ACES_CLAMP_GAMMA = 2,  /* bunny*                                          */
       RTPBR_TONEMAP_ACES_GAMMA_CLAMP = 3 };/* tokyo, scene_demo                               */

enum { RTPBR_CAMERA_THIN_LENS = 0,     /* src/camera.py:11-36                                  */
       RTPBR_CAMERA_PINHOLE = 1 };     /* cornell_box_shortest.py:107-118 (no lens draws)      */

typedef struct rtpbr_config {
    int32_t  width, height;        /* image_resolution                                        */
    uint32_t seed;                 /* counter-based RNG seed                                  */
    int32_t  kernel_form;          /* RTPBR_FORM_*                                            */
    int32_t  max_raymarch;         /* MAX_RAYMARCH                                            */
    int32_t  max_raytrace;         /* MAX_RAYTRACE                                            */
    /* sphere tracing */
    int32_t  march_kind;           /* RTPBR_MARCH_*                                           */
    float    min_dis;              /* examples: march start t0; src: normal offset MIN_DIS    */
    float    max_dis;              /* MAX_DIS                                                 */
    float    hit_eps;              /* PRECISION (plain) or PIXEL_RADIUS (relaxed / src)       */
    float    omega0;               /* initial over-relaxation                                 */
    int32_t  omega_guard;          /* 1: fall back only while omega > 1                       */
    float    omega_fb_a, omega_fb_b; /* on overshoot: omega <- a + b*omega                    */
    /* sdf */
    float    box_round;            /* rounding rho of sd_box                                  */
    int32_t  nearest_init;         /* 0: start from object 0 (cornell_box_v3/pathtracer.py:41-49)
                                      1: start from (0, MAX_DIS) (src/scene.py:45-46, tokyo_ibl.py:222) */
    /* normal */
    float    normal_h;
    int32_t  normal_space;         /* RTPBR_NORMAL_*                                          */
    /* russian roulette */
    int32_t  rr_kind;              /* RTPBR_RR_*                                              */
    float    light_quality;        /* examples                                                */
    float    quality_per_sample;   /* src QUALITY_PER_SAMPLE                                  */
    /* surface model */
    int32_t  surface_kind;         /* RTPBR_SURFACE_*                                         */
    int32_t  fresnel_kind;         /* RTPBR_FRESNEL_*                                         */
    int32_t  fresnel_roughness_mix;/* examples 1, src 0                                       */
    int32_t  below_horizon;        /* RTPBR_HORIZON_*                                         */
    int32_t  origin_mode;          /* RTPBR_ORIGIN_*                                          */
    float    env_ior;              /* ENV_IOR                                                 */
    /* miss / sky */
    int32_t  sky_kind;             /* RTPBR_SKY_*                                             */
    int32_t  primary_miss;         /* RTPBR_PRIMARY_*                                         */
    /* stop test: stop if brighter, or visible < vis_lo, or visible > vis_hi */
    float    vis_lo, vis_hi;
    /* camera */
    int32_t  camera_kind;          /* RTPBR_CAMERA_*                                          */
    /* tone map */
    int32_t  tonemap_order;        /* RTPBR_TONEMAP_*                                         */
    int32_t  aces_truncated;       /* 1: cornell_box_shortest.py:126-128 literals             */
    float    exposure;
    float    gamma;
    /* animation uniform (u_frame, bunny_sdf_glass.py:213-217) */
    int32_t  frame;
    /* persistent-ray form: bounce-steps per pixel per launch (SAMPLES_PER_PIXEL) */
    int32_t  steps_per_launch;
    /* self-adaptive sampling (src/config.py:14,17; persistent-ray form only): a pixel is sampled
     * only while its running mean display-space change diff_pixels exceeds noise_threshold */
    int32_t  adaptive_sampling;
    float    noise_threshold;
    /* neural-bunny animation: amplitude of the vertical bob p.z += anim_bob * sin(t) after the frame rotation
     * (bunny_sdf_glass.py:216 and bunny_sdf_v2.py:216: 0.1; bunny_sdf.py:213-214 rotates only: 0) */
    float    anim_bob;
} rtpbr_config;

/* ------------------------------------------------------------------ buffers */
enum { RTPBR_BUF_IMAGE_BUFFER = 0,   /* T7 image_buffer  (W,H,4) f32: (sum r, sum g, sum b, count) */
       RTPBR_BUF_IMAGE_PIXELS = 1,   /* T8 image_pixels  (W,H,3) f32 display colour                */
       RTPBR_BUF_RAY_BUFFER   = 2,   /* T6 ray_buffer    (W,H) of rtpbr_ray (persistent-ray form)  */
       RTPBR_BUF_DIFF_BUFFER  = 3,   /* T11 diff_buffer  (W,H,2) f32 (sum of display change, count) src/fileds.py:21 */
       RTPBR_BUF_DIFF_PIXELS  = 4 }; /* T11 diff_pixels  (W,H) f32                                  src/fileds.py:22 */

enum { RTPBR_ENV_RGB8 = 0,           /* uint8 (W_e,H_e,3), [x][y], y=0 bottom: what ti.tools.imread gives */
       RTPBR_ENV_RGB32F = 1 };       /* float32 (W_e,H_e,3) already preprocessed (T9 as is)               */

/* Work counters of the last rtpbr_sample() call (B-bar, S-bar). */
typedef struct rtpbr_counters {
    uint64_t samples;       /* pixel-samples (complete-path) or bounce-steps (persistent) */
    uint64_t raycasts;      /* raycast() calls                                            */
    uint64_t march_steps;   /* nearest() evaluations inside raycast loops                 */
    uint64_t hits;          /* surface interactions                                       */
    uint64_t sky_lookups;   /* environment lookups                                        */
    uint64_t deposits;      /* image_buffer accumulations                                 */
} rtpbr_counters;

typedef struct rtpbr_ctx rtpbr_ctx;

/* ------------------------------------------------------------------ entry points */

/* Create a context on HIP device `device`.  Replaces ti.init(arch=...) (src/config.py:5)
 * plus field allocation (src/fileds.py:7-15). */
int rtpbr_create(int device, rtpbr_ctx** out);
int rtpbr_destroy(rtpbr_ctx* ctx);
const char* rtpbr_last_error(void);
/* Name of the backend ("hip-gfx950"); lets callers assert which library they loaded. */
const char* rtpbr_backend(void);

/* Replaces the module constants of src/config.py:7-28.  (Re)allocates the per-pixel
 * buffers zero-initialised when the resolution changes. */
int rtpbr_set_config(rtpbr_ctx* ctx, const rtpbr_config* cfg);

/* Replaces the element-wise copy into the `objects` field (src/scene.py:38-41) and
 * build_scene()/update_all_transform() (src/scene.py:106-113): the world->local
 * matrices are computed here from `rotation`.  `scale10` != 0 multiplies position and
 * scale by 10 (cornell_box_v3/sdf.py:16-18). */
int rtpbr_set_scene(rtpbr_ctx* ctx, const rtpbr_object* objects, int n, int scale10);
/* Read back the uploaded table (with matrices filled), n entries. */
int rtpbr_get_scene(rtpbr_ctx* ctx, rtpbr_object* objects, int n);

/* Replaces the 0-d camera fields (src/camera.py:119-129) / the camera kernel arguments
 * (cornell_box_v3/renderer.py:12-16). */
int rtpbr_set_camera(rtpbr_ctx* ctx, const rtpbr_camera* cam);

/* Replaces Image(path) + Image.process(exposure, gamma) (src/ibl.py:14-23,32-33).
 * RGB8 texels are converted as (c/255*exposure)^gamma; RGB32F texels are used as is. */
int rtpbr_set_env(rtpbr_ctx* ctx, const void* texels, int w, int h, int fmt,
                  float exposure, float gamma);

/* Per-shape data blobs.  Only RTPBR_SHAPE_BUNNY takes one: the 625 weights of the neural
 * SDF (examples/bunny/bunny_sdf_glass.py:157-201, inline literals in the reference). */
int rtpbr_set_shape_data(rtpbr_ctx* ctx, int shape, const float* data, int n);

/* Tile partition of the frame for multi-GPU rendering: tiles of
 * tile_w x tile_h pixels are dealt round-robin, tile t belongs to rank t % world.
 * (0,0,0,1) or world==1 means "the whole frame". */
int rtpbr_set_tiles(rtpbr_ctx* ctx, int tile_w, int tile_h, int rank, int world);

/* refresh() src/renderer.py:12-22: zero image_buffer, ray_buffer.depth = 0. */
int rtpbr_refresh(rtpbr_ctx* ctx);

/* The hot path.  complete-path form: `n` samples per owned pixel are traced and
 * accumulated (the spp loop of cornell_box_v3/renderer.py:31-36).  persistent-ray form:
 * `n` launches of pathtrace() (src/renderer.py:29-30), each advancing every pixel by
 * cfg.steps_per_launch bounce-steps.  Asynchronous: returns after enqueue. */
int rtpbr_sample(rtpbr_ctx* ctx, int n);

/* post_process() src/postprocessor.py:24-43: image_buffer -> image_pixels. */
int rtpbr_post_process(rtpbr_ctx* ctx);

/* Block until everything enqueued on the context has finished (its stream, and the copies of rtpbr_read_buffer_async). */
int rtpbr_sync(rtpbr_ctx* ctx);

/* field.to_numpy() / from_numpy(): copy a whole buffer to/from host memory (blocking). */
int rtpbr_read_buffer(rtpbr_ctx* ctx, int which, void* dst, size_t nbytes);
int rtpbr_write_buffer(rtpbr_ctx* ctx, int which, const void* src, size_t nbytes);

/* Page-locked host memory for the destination of rtpbr_read_buffer(): a host that shows every frame (src/main.py:62-64 hands
 * image_pixels to the window once per render()) reads into the SAME buffer again and again, and a page-locked one takes the copy
 * at the link's rate without the driver's staging.  Plain memory works as before; this is an optimisation the caller opts into.
 * Freed by rtpbr_host_free() or with the context. */
int rtpbr_host_alloc(rtpbr_ctx* ctx, size_t nbytes, void** ptr);
int rtpbr_host_free(rtpbr_ctx* ctx, void* ptr);

/* Handing a frame over WITHOUT stalling the device (round 6).  The reference's window takes image_pixels on the device
 * (`canvas.set_image(image_pixels)`, src/main.py:64: the frame never visits the host) and its loop goes straight on to the
 * next render(); rtpbr_read_buffer() above is the blocking `field.to_numpy()`.  Two more ways out:
 *
 * rtpbr_buffer_device_ptr: the DEVICE address and size of a buffer — zero copy for a consumer on the same GPU (display
 *   interop, a video encoder, a torch tensor through __cuda_array_interface__).  The consumer orders itself behind the
 *   context's work with rtpbr_get_stream() (or calls rtpbr_sync()); the address stays valid until rtpbr_set_config()
 *   changes the resolution or the context is destroyed, and the next rtpbr_post_process() / rtpbr_sample() overwrites
 *   the contents, as the reference's next render() does.
 *
 * rtpbr_read_buffer_async: enqueue the copy of a buffer into PAGE-LOCKED host memory (a block of rtpbr_host_alloc; EINVAL
 *   otherwise — a pageable destination would make the copy synchronous) behind everything enqueued on the context so
 *   far, on a copy stream of its own, and return a ticket at once.  The context's later work does NOT wait for the copy —
 *   frame k's read-back overlaps frame k+1's sample kernels — except the first call that would overwrite the buffer
 *   being read (the next rtpbr_post_process() for image_pixels): that one is ordered behind the copy on the device, the
 *   host still does not block.  rtpbr_read_wait(ticket) blocks the HOST until that copy has landed in `dst`.  Up to 8
 *   reads may be outstanding; taking a 9th ticket first waits for the oldest. */
int rtpbr_buffer_device_ptr(rtpbr_ctx* ctx, int which, void** device_ptr, size_t* nbytes);
int rtpbr_read_buffer_async(rtpbr_ctx* ctx, int which, void* dst, size_t nbytes, int* ticket);
int rtpbr_read_wait(rtpbr_ctx* ctx, int ticket);

/* Multi-GPU gather support.  pack: copy this rank's tiles of image_buffer, tile-major,
 * into a DEVICE buffer of rtpbr_packed_bytes() bytes (all ranks get the same padded size
 * so one RCCL gather moves them).  unpack: scatter a packed buffer that belongs to rank
 * `src_rank` into this context's image_buffer.  Both run on the context's stream. */
int rtpbr_packed_bytes(rtpbr_ctx* ctx, size_t* nbytes);
int rtpbr_pack_tiles(rtpbr_ctx* ctx, void* device_dst);
int rtpbr_unpack_tiles(rtpbr_ctx* ctx, const void* device_src, int src_rank);

/* ---- the ONE collective of the multi-GPU path, on RCCL directly (rt_rccl.hip).
 * Replaces nothing in the reference (it is single-device); completes rtpbr_set_tiles / pack / unpack
 * so that hosts without PyTorch can render on N GPUs: every rank packs its tiles, one ncclGather (rccl.h:745) moves
 * them to rank 0, rank 0 scatters them into its full image_buffer (T7) — all enqueued on the context's stream.
 * One process per GPU: rank 0 obtains a 128-byte id, ships it to the other ranks by the host's own means, every rank
 * calls rtpbr_rccl_init (collective, like ncclCommInitRank, rccl.h:220), then rtpbr_gather_tiles (collective).
 * One process driving G contexts: rtpbr_rccl_init_all (ncclCommInitAll, rccl.h:236) and rtpbr_gather_tiles_all.
 * rank/world must equal the ones given to rtpbr_set_tiles.  librccl is dlopen-ed at the first of these calls. */
int rtpbr_rccl_unique_id(void* out, size_t nbytes);                       /* nbytes >= 128 */
int rtpbr_rccl_init(rtpbr_ctx* ctx, const void* unique_id, size_t nbytes, int rank, int world);
int rtpbr_rccl_init_all(rtpbr_ctx** ctxs, int n);
int rtpbr_gather_tiles(rtpbr_ctx* ctx);
int rtpbr_gather_tiles_all(rtpbr_ctx** ctxs, int n);
/* What RCCL itself reports for this context's communicator: ncclCommCount (rccl.h:378), ncclCommUserRank (:400) and the
 * library's version (ncclGetVersion, :164) — so that a caller can show that the collective really spans `world` ranks. */
int rtpbr_rccl_info(rtpbr_ctx* ctx, int* nranks, int* rank, int* version);

/* Measurement hooks (SURVEY.md §5 tracing row, §8(d)). */
int rtpbr_get_counters(rtpbr_ctx* ctx, rtpbr_counters* out);         /* blocking */
/* One counter by name: the six above, plus "mlp_wave_evals" / "mlp_lane_evals" — 32-ray half passes of the
 * wave-cooperative neural-SDF network (bunny_sdf_glass.py:149-203 on the matrix cores) and the ray evaluations they
 * were needed for; their ratio / 32 is the slot utilisation of the MLP.  EINVAL for an unknown name. */
int rtpbr_get_counter(rtpbr_ctx* ctx, const char* name, unsigned long long* out);
/* Device time (HIP events on the context's stream) of the trace kernel launches and of
 * all kernels of the last rtpbr_sample() call, in milliseconds (blocking).  Option "timing" = 0 records no events
 * (every event is a few microseconds of idle queue between two small kernels: 8 us of a 160 us one-step launch) —
 * these two calls then return RTPBR_ESTATE. */
int rtpbr_last_sample_ms(rtpbr_ctx* ctx, float* trace_ms, float* total_ms, int* launches);
/* Device time of the primary_rays launches of the last rtpbr_sample() call (0 launches when
 * the primary raycasts ran inside the trace kernel: option "primary_split" 0, neural SDF). */
int rtpbr_last_primary_ms(rtpbr_ctx* ctx, float* primary_ms, int* launches);
/* Raw stream handle (hipStream_t) so callers can order their own work after ours. */
int rtpbr_get_stream(rtpbr_ctx* ctx, void** stream);
/* Tuning knobs that do not change results.  Keys: "staging_bytes" (sub-launch staging budget),
 * "scheduler" (-1 auto, 0 in-register refill / lock-step, 1 per-wave LDS ray pool),
 * "wait_lanes" (scheduler 0), "shade_lanes", "swap_lanes" (0 = automatic, the default: 8 in the complete-path pool kernel, 12 in the
 * persistent-ray one), "refill_lanes", "ready_low" (scheduler 1), "waves_per_cu",
 * "residency" (persistent-ray form, pool scheduler: bounce-steps a pixel stays resident in a wave that owns more pixels
 * than the 128 it can hold; a power of two, default 32), "grid_blocks" (same kernel: workgroups to launch, 0 = automatic),
 * "drain_lanes" (complete-path pool kernel: once the work has run out and nothing is parked, a wave with at most this many marching
 * lanes finishes their raycasts in the culled wave march of the primary kernel; default 16, 0 = never),
 * "primary_lean" (1, default: the coherent primary-ray kernel marches on in a one-object loop while its whole wave needs one object),
 * "src_chain" (persistent-ray form, fused launches: 1, default = when the plan finds the launch as long as its heaviest pixel's
 * dependency chain AND the device has room beside the pool kernel's grid, the heaviest pixels — the chain set — run in the chain
 * kernel on a second stream, alone or in small groups per wave; 0 = never, 2 = whenever the plan says so), "chain_waves" (the most
 * waves the chain set may take, default 1024; the pool kernel's grid makes room for them), "chain_np_max" (frames of more local
 * pixels than this are throughput-bound and get no chain set, default 2 500 000),
 * "src_split" (persistent-ray form: a launch of at most this many bounce-steps runs as a wavefront split — per step one
 * coherent kernel for roulette / deposit / camera ray, one for the raycasts on the cost-ordered pixel list, one for shading —
 * instead of the fused pool kernel; 0 = never, default 1), "split_wait" (its march kernel refills lanes when this many are
 * free, default 24),
 * "src_op" (round 6: bit 0 = while at most 8 lanes of a wave march — the tail of a one-step launch, a chain wave of a few pixels —
 * the wave evaluates nearest() OBJECT-PARALLEL: lane (r, j) evaluates object j for the r-th marching ray from the LDS table and a DPP
 * butterfly over each group of eight lanes returns nearest / second / third, bit-identical; split march and chain kernels; bit 1 =
 * the fused pool kernel too, only in builds with -DRT_POOL_OP=1: it spills there; bit 2 = the per-lane lean loop: when every marching
 * lane holds a valid bound but for a DIFFERENT object, each lane evaluates its own object from the LDS table in one loop; default 7),
 * "split_head" (split march kernel: the
 * cost-ordered list's heavy head interleaved over the groups, one entry per group, instead of filling the first groups: -1 = for
 * frames of at most 600 000 pixels (default), 0 never, 1 always),
 * "src_lazy" (one-step launches of the persistent-ray form, i.e. the wavefront split; 1, default: a launch leaves its shading to the
 * next launch's gen pass — one pass over ray_buffer instead of two, one kernel less per launch — and every call that could see the
 * difference launches it first: readers and writers of ray_buffer (rtpbr_buffer_device_ptr of ray_buffer ends the lazy mode for the
 * context), rtpbr_get_counter(s), every setter, rtpbr_refresh, a launch of another kind; rtpbr_post_process and reads of the image
 * buffers do not need it.  0: every launch shades.  Same bits either way),
 * "env_packed" (1, default: an environment uploaded as 8-bit texels is read as RGBA8 texels + the 256-entry table of
 * (c / 255 * exposure)^gamma — the same floats, a quarter of the bytes, same speed; 0 = float4 texels),
 * "src_track" (same kernel: 1 = tracked-object march steps — a lane that knows a lower bound of every object but the
 * nearest one evaluates only that one, exactly; heavy waves always use them), "sparse_lanes" (... other waves while at
 * most this many lanes march, 0 = never; default 24), "leave_x8" (cost of a shading pass in eighths of a march iteration:
 * the march loop is left when the lane-iterations wasted by finished lanes and parked contexts reach it; default 24),
 * "src_plan" (1: the pool kernel records every pixel's march steps and re-orders its ownership by them — heaviest pixels
 * first, the very heaviest in waves of their own), "plan_interval" (bounce-steps on record before a re-plan, default 64),
 * "heavy_mean_x16" / "heavy_bulk_x16" (a pixel is heavy when its cost exceeds both that many sixteenths of the mean pixel
 * and of a wave's share of the frame in march iterations; defaults 48, 8), "heavy_own" (pixels per heavy wave, <= 128,
 * default 80), "tiny_own" / "tiny_waves" (the very heaviest pixels: waves of at most tiny_own pixels, tiny_waves of them —
 * a quarter of the grid when the launch is as long as its longest chain), "heavy_prio" (heavy waves raise their issue
 * priority), "age_tune" (1, default: the light waves' shares are weighted by the residency slot of their block — the k-th
 * block of a CU is the k-th oldest wave of its SIMD and the issue arbiter favours the older wave — with weights the library
 * tunes from the lifetimes it measures, so that the waves of a SIMD end together; 0 = equal shares), "age_weights" (fixed
 * weights instead, one hex digit per slot, oldest first),
 * "primary_split" (primary raycasts in their own coherent lock-step kernel with wave-level
 * object culling; pool scheduler, analytic shapes: 0 never, 1 for launches of >= 2^23 samples
 * (default), 2 always), "specialize" (1: use the instance compiled
 * for the scene's rotation signature when there is one), "lazy_sqrt" (1: all-box scenes pick the
 * nearest box on squared distances and take one exact square root per march step),
 * "mlp_mfma", "mlp_lanes" (neural SDF: run the network when this many lanes wait for it), "mlp_full" (compute both
 * 32-slot halves of a pass when at least this many wait, else the first 32 by rank),
 * "jit" (per-scene kernels compiled at run time by hipcc --genco from the sources next to the library — what Taichi's
 * JIT does for the reference: object loop unrolled, each object's shape function and rotation class fixed at compile
 * time, src/scene.py:44-56 — cached under $RTPBR_JIT_CACHE / ~/.cache/rtpbr: -1 (default) when no ahead-of-time
 * specialisation serves the scene, 0 never, 1 always but falling back to the ahead-of-time kernels if compilation is
 * impossible, 2 always and an error otherwise), "jit_bake" (1: the run-time instance also carries the scene's object
 * table and the whole rtpbr_config except seed and frame as compile-time constants — one code object per scene and
 * configuration; for offline renders of a fixed scene),
 * "jit_waves" (waves per SIMD the run-time pool kernel is compiled for; 0 = as the ahead-of-time instances),
 * "chunk" (work items a wave claims per atomic, at most 8192; 0 = automatic: total / (waves x 64) clamped to [256, 1024]),
 * "timing" (1, the default: rtpbr_sample() brackets its kernels with HIP events for rtpbr_last_sample_ms /
 * rtpbr_last_primary_ms; 0: none),
 * "stage_dense" (complete-path pool kernel, run-time instances only — the code is compiled in on request; 1: a wave appends the finished samples of a claim to the claim's own
 * stretch of the staging in completion order, with one byte that says which sample each is, and the accumulate kernel puts them
 * back in sample order — the claim is then the largest size <= 256 that is a whole multiple or a whole fraction of the launch's
 * samples per pixel, and a launch that has none (or a "chunk" that is none) keeps the item-linear records; 0, the default:
 * item-linear records always.  Same bits either way; 21 % fewer HBM writes and 7 % more time on the 1080p x 256 spp step),
 * "reserve_spp" (allocate the staging of a call of that many samples per pixel now instead of on
 * first use), "sample_base" (absolute index of the next sample: checkpoint/resume).
 * Returns RTPBR_EINVAL for unknown keys or out-of-range values. */
int rtpbr_set_option(rtpbr_ctx* ctx, const char* key, long long value);

/* Ahead-of-time compilation of the scene-specialised kernels (round 6) — NO DEVICE NEEDED.  Taichi compiles the reference's
 * kernels at first call on the machine that runs them (ti.init, src/config.py:5; ti.static unrolling, src/scene.py:44-56);
 * option "jit" does the same here and needs hipcc + the kernel sources on the target.  rtpbr_jit_prebuild compiles, on a BUILD
 * machine, exactly the code object rtpbr_sample() would ask for with this scene (objects, scale10), configuration, camera, tile
 * partition (tile_w, tile_h, world; rank does not matter) and options ("key=value key=value ...", as rtpbr_set_option:
 * jit, jit_bake, precision, mlp_mfma, ...) into $RTPBR_JIT_CACHE and returns its path.  Code objects placed in the catalog
 * directory next to the library (raytracingpbr_amd/data/jit, or $RTPBR_JIT_CATALOG) are found by rtpbr_sample() BEFORE it
 * forks a compiler: a target without hipcc runs the baked kernels of the scenes it ships with.  `python -m
 * raytracingpbr_amd.prebuild` fills the catalog for the BASELINE scenes (called by the package's build step). */
int rtpbr_jit_prebuild(const rtpbr_object* objects, int n, int scale10, const rtpbr_config* cfg, const rtpbr_camera* cam,
                       int tile_w, int tile_h, int world, const char* options, char* path_out, size_t path_cap);

#ifdef __cplusplus
}
#endif
#endif /* RTPBR_H */
