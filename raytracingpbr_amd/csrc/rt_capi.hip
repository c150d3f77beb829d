// rt_capi.hip — host side of the C ABI declared in include/rtpbr.h.
//
// One rtpbr_ctx owns one HIP stream and every device buffer of one renderer instance (what
// the Taichi runtime owns in the reference: src/fileds.py:7-15, src/scene.py:38-41,
// src/ibl.py:16-17).  Each entry point replaces one Taichi kernel launch or field access of
// the reference; see the header for the line-by-line mapping.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_ctx.hpp"

using namespace rt;

static void derive_launch(rtpbr_ctx* c, RtJitKey* key, bool* want, bool* strict_error);
extern "C" int rtpbr_set_option(rtpbr_ctx* c, const char* key, long long value);
extern "C" int rtpbr_set_tiles(rtpbr_ctx* c, int tw, int th, int rank, int world);
extern "C" int rtpbr_set_camera(rtpbr_ctx* c, const rtpbr_camera* cam);

static thread_local char g_err[512];
int rt_fail(int code, const char* fmt, const char* a) {
    snprintf(g_err, sizeof g_err, fmt, a);
    return code;
}
int rt_fail_hip(const char* expr, hipError_t e) {
    snprintf(g_err, sizeof g_err, "%s failed: %s", expr, hipGetErrorString(e));
    return RTPBR_EHIP;
}
static int fail(int code, const char* fmt, const char* a = "") { return rt_fail(code, fmt, a); }
#define HIP_TRY(expr) RT_HIP_TRY(expr)

static int set_dev(rtpbr_ctx* c) {
    if (c->headless) return RTPBR_OK;       // (a context without a device: rtpbr_jit_prebuild derives launch constants on the host only)
    HIP_TRY(hipSetDevice(c->device));
    return RTPBR_OK;
}

// The lazy shading of the split's one-step launches (rt_ctx.hpp src_lazy): launch what the last such launch left for the next one's gen
// pass.  Called by everything that could tell the difference.
static int flush_shade(rtpbr_ctx* c) {
    if (!c->shade_pending) return RTPBR_OK;
    c->shade_pending = false;
    if (int r = set_dev(c)) return r;
    rtpbr_ctx::Params& P = c->P;
    const uint32_t keep = P.sample_base;
    P.sample_base = c->shade_pending_base;
    int rc = RTPBR_OK;
    if (c->jit_mod) rc = rt_jit_launch(c->jit_mod->src_shade, P, (unsigned)(((long long)P.np + 255) / 256), c->stream);
    else launch_src_shade(P, c->kind, c->stream);
    P.sample_base = keep;
    return rc;
}

extern "C" const char* rtpbr_last_error(void) { return g_err; }
extern "C" const char* rtpbr_backend(void) { return "hip-gfx950"; }

extern "C" int rtpbr_destroy(rtpbr_ctx* c);
static int create_resources(rtpbr_ctx* c) {
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipMalloc(&c->objfull, sizeof(ObjFull) * MAX_OBJ));
    HIP_TRY(hipMalloc(&c->team_counter, 1024 * 64));
    // the work counters of a launch and the claim counters of the complete-path kernels in one allocation: ONE fill per rtpbr_sample()
    // (a fill is a dispatch of its own: ~15 us between two small launches)
    constexpr size_t counters_bytes = (sizeof(Counters) + 64 + 255) / 256 * 256;
    HIP_TRY(hipMalloc(&c->counters_buf[0], 2 * counters_bytes));
    c->counters_buf[1] = reinterpret_cast<Counters*>(reinterpret_cast<char*>(c->counters_buf[0]) + counters_bytes);
    HIP_TRY(hipMemsetAsync(c->counters_buf[0], 0, 2 * counters_bytes, c->stream));
    c->counters_clean[0] = c->counters_clean[1] = true;
    c->counters = c->counters_buf[0];
    c->work_counter = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(c->counters) + sizeof(Counters));
    // timing events carry no data to the host (a read-back synchronises the stream itself): without the system-scope fence an event
    // costs the queue less (rocprofv3: ~10 us of idle queue per default event between two small kernels)
    HIP_TRY(hipEventCreateWithFlags(&c->ev_total1, hipEventDisableSystemFence));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, c->device));
    c->n_cu = prop.multiProcessorCount;
    return RTPBR_OK;
}

extern "C" int rtpbr_create(int device, rtpbr_ctx** out) {
    if (!out) return fail(RTPBR_EINVAL, "out is NULL");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(RTPBR_EINVAL, "no such HIP device");
    rtpbr_ctx* c = new rtpbr_ctx();
    c->device = device;
    const int r = create_resources(c);
    if (r != RTPBR_OK) {            // g_err already says which call failed; release whatever was created
        rtpbr_destroy(c);
        return r;
    }
    *out = c;
    return RTPBR_OK;
}

extern "C" int rtpbr_destroy(rtpbr_ctx* c) {
    if (!c) return RTPBR_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);      // (an asynchronous read-back may still be copying out of the buffers)
    (void)hipFree(c->image_buffer);
    (void)hipFree(c->image_pixels);
    (void)hipFree(c->ray_buffer);
    (void)hipFree(c->diff_buffer);
    (void)hipFree(c->diff_pixels);
    (void)hipFree(c->objfull);
    (void)hipFree(c->env);
    (void)hipFree(c->env8);
    (void)hipFree(c->env_lut);
    (void)hipFree(c->bunny);
    (void)hipFree(c->stage);
    (void)hipFree(c->primary);
    (void)hipFree(c->cost_buffer);
    (void)hipFree(c->march_out);
    (void)hipFree(c->order);
    (void)hipFree(c->plan);
    rt_rccl_release(c);
    rt_jit_release(c->jit_mod);
    c->jit_mod = nullptr;
    (void)hipFree(c->team_counter);
    for (void* hb : c->host_blocks) (void)hipHostFree(hb);
    c->host_blocks.clear();
    c->host_sizes.clear();
    (void)hipFree(c->counters_buf[0]);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->evp) (void)hipEventDestroy(e);
    if (c->ev_total1) (void)hipEventDestroy(c->ev_total1);
    if (c->stream2) {
        (void)hipStreamSynchronize(c->stream2);
        (void)hipStreamDestroy(c->stream2);
    }
    if (c->copy_stream) {
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamDestroy(c->copy_stream);
    }
    if (c->ev_read_ready) (void)hipEventDestroy(c->ev_read_ready);
    for (hipEvent_t e : c->ev_read_done)
        if (e) (void)hipEventDestroy(e);
    if (c->plan_rb_host) (void)hipHostFree(c->plan_rb_host);
    if (c->ev_plan_rb) (void)hipEventDestroy(c->ev_plan_rb);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RTPBR_OK;
}

static void update_tiles(rtpbr_ctx* c) {
    Params& P = c->P;
    const int W = c->cfg.width, H = c->cfg.height;
    int tw = c->tile_w, th = c->tile_h;
    if (c->world <= 1 || tw <= 0 || th <= 0) {
        if (tw <= 0 || th <= 0) {
            tw = W;
            th = H;
        }
    }
    P.tile_w = tw;
    P.tile_h = th;
    P.ntx = (W + tw - 1) / tw;
    P.nty = (H + th - 1) / th;
    P.rank = c->rank;
    P.world = c->world;
    int ntiles = P.ntx * P.nty;
    P.n_local_tiles = (ntiles + c->world - 1) / c->world;
    P.np = P.n_local_tiles * tw * th;
    // the src/ pool kernel's cost-ordered ownership lists LOCAL pixels: a new partition starts without a plan
    c->order_valid = false;
    c->cost_steps = 0;
    c->plan_np = 0;
}

// Work items (padded local pixels x samples per launch) are 32-bit: a rank's padded pixel count must
// leave room for at least one sample per launch plus the chunks the last waves claim past the end.
static int check_local_pixels(int W, int H, int tw, int th, int world) {
    if (world <= 1 || tw <= 0 || th <= 0) { tw = W; th = H; world = world < 1 ? 1 : world; }
    const long long ntx = (W + tw - 1) / tw, nty = (H + th - 1) / th;
    const long long local = (ntx * nty + world - 1) / world;
    if (local * tw * th > (1LL << 30)) return fail(RTPBR_EINVAL, "more than 2^30 (padded) pixels per rank: use more ranks or smaller frames");
    return RTPBR_OK;
}

extern "C" int rtpbr_set_config(rtpbr_ctx* c, const rtpbr_config* cfg) {
    if (!c || !cfg) return fail(RTPBR_EINVAL, "null argument");
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->width > 65535 || cfg->height > 65535)
        return fail(RTPBR_EINVAL, "resolution out of range (1..65535)");
    if (cfg->max_raymarch <= 0 || cfg->max_raytrace <= 0) return fail(RTPBR_EINVAL, "max_raymarch/max_raytrace must be > 0");
    if (int r = check_local_pixels(cfg->width, cfg->height, c->tile_w, c->tile_h, c->world)) return r;
    if (int r = set_dev(c)) return r;
    if (int r = flush_shade(c)) return r;
    bool realloc_buf = (!c->have_cfg || c->cfg.width != cfg->width || c->cfg.height != cfg->height) && !c->headless;
    c->cfg = *cfg;
    c->P.cfg = *cfg;
    c->have_cfg = true;
    if (realloc_buf) {
        size_t n = (size_t)cfg->width * cfg->height;
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));
        for (int& t : c->read_pending) t = -1;
        (void)hipFree(c->image_buffer);
        (void)hipFree(c->image_pixels);
        (void)hipFree(c->ray_buffer);
        (void)hipFree(c->diff_buffer);
        (void)hipFree(c->diff_pixels);
        c->image_buffer = nullptr;
        c->image_pixels = nullptr;
        c->ray_buffer = nullptr;
        c->diff_buffer = nullptr;
        c->diff_pixels = nullptr;
        HIP_TRY(hipMalloc(&c->image_buffer, n * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->image_pixels, n * 3 * sizeof(float)));
        HIP_TRY(hipMalloc(&c->ray_buffer, n * sizeof(rtpbr_ray)));
        HIP_TRY(hipMemsetAsync(c->image_buffer, 0, n * sizeof(float4), c->stream));
        HIP_TRY(hipMemsetAsync(c->image_pixels, 0, n * 3 * sizeof(float), c->stream));
        HIP_TRY(hipMemsetAsync(c->ray_buffer, 0, n * sizeof(rtpbr_ray), c->stream));
        HIP_TRY(hipMalloc(&c->diff_buffer, n * sizeof(float2)));
        HIP_TRY(hipMalloc(&c->diff_pixels, n * sizeof(float)));
        HIP_TRY(hipMemsetAsync(c->diff_buffer, 0, n * sizeof(float2), c->stream));
        HIP_TRY(hipMemsetAsync(c->diff_pixels, 0, n * sizeof(float), c->stream));
    }
    // bunny animation uniform: t = pi*frame/120 (bunny_sdf_glass.py:214)
    float t = PI * (float)cfg->frame / 120.0f;
    sincos_(t, &c->P.anim_s, &c->P.anim_c);
    c->P.anim_bz = cfg->anim_bob * c->P.anim_s;
    update_tiles(c);
    return RTPBR_OK;
}

// Euler -> matrix, src/util.py:36-42: M = Rz @ Ry @ Rx (row major)
static void m3_mul(const float* a, const float* b, float* o) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            o[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}
static void rotate(const float* rad, float* m) {
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

// exact sparsity pattern of a world->local matrix (see to_local): 0 entries must be +-0, the axis entry 1.0f
static int rotation_class(const float* m) {
    auto z = [&](int i) { return m[i] == 0.0f; };
    auto one = [&](int i) { return m[i] == 1.0f; };
    for (int i = 0; i < 9; i++)
        if (!std::isfinite(m[i])) return ROT_GENERAL;
    if (one(0) && one(4) && one(8) && z(1) && z(2) && z(3) && z(5) && z(6) && z(7)) return ROT_IDENT;
    if (one(0) && z(1) && z(2) && z(3) && z(6)) return ROT_X;
    if (one(4) && z(1) && z(3) && z(5) && z(7)) return ROT_Y;
    if (one(8) && z(2) && z(5) && z(6) && z(7)) return ROT_Z;
    return ROT_GENERAL;
}

// First listed signature (RT_BOX_SIGNATURES) whose every specialised class fits the object's matrix;
// an identity matrix fits every single-axis class.  0 = the general instance.
static uint32_t choose_signature(const ObjM* objm) {
    int cls[8];
    for (int i = 0; i < 8; i++) cls[i] = rotation_class(objm[i].m);
    const uint32_t sigs[] = {
#define RT_SIG_ITEM(sig, ...) sig,
        RT_BOX_SIGNATURES(RT_SIG_ITEM, 0)
#undef RT_SIG_ITEM
    };
    for (uint32_t sig : sigs) {
        bool ok = true;
        for (int i = 0; i < 8 && ok; i++) {
            const int want = sig_cls(sig, i);
            ok = want == ROT_GENERAL || want == cls[i] || cls[i] == ROT_IDENT;
        }
        if (ok) return sig;
    }
    return 0;
}

// Fill a march table: the general 64-byte blocks, or the signature's packed layout (rt_types.hpp:
// only the dwords each object's rotation class reads, wide-load friendly).
static void pack_table(const ObjM* src, int n, uint32_t sig, ObjM* dst_table) {
    memset(dst_table, 0, sizeof(ObjM) * MAX_OBJ);
    if (sig == 0) {
        memcpy(dst_table, src, sizeof(ObjM) * (size_t)n);
        return;
    }
    float* f = reinterpret_cast<float*>(dst_table);
    for (int i = 0; i < n; i++) {
        const ObjM& o = src[i];
        const int cls = sig_cls(sig, i);
        float* d = f + sig_offset(sig, i);
        if (cls == ROT_GENERAL) {
            memcpy(d, &o, sizeof o);
            continue;
        }
        d[0] = o.px, d[1] = o.py, d[2] = o.pz;
        if (cls == ROT_IDENT) {
            d[3] = o.sx, d[4] = o.sy, d[5] = o.sz;
            continue;
        }
        const int e[3][4] = {{4, 5, 7, 8}, {0, 2, 6, 8}, {0, 1, 3, 4}};   // X, Y, Z: the four entries that are not 0 / 1
        for (int k = 0; k < 4; k++) d[3 + k] = o.m[e[cls - ROT_X][k]];
        d[7] = o.sx, d[8] = o.sy, d[9] = o.sz;
    }
}

extern "C" int rtpbr_set_scene(rtpbr_ctx* c, const rtpbr_object* objs, int n, int scale10) {
    if (!c || !objs) return fail(RTPBR_EINVAL, "null argument");
    if (n <= 0 || n > MAX_OBJ) return fail(RTPBR_EINVAL, "object count must be 1..32");
    for (int i = 0; i < n; i++)   // validate before touching the context
        if (objs[i].type < RTPBR_SHAPE_NONE || objs[i].type > RTPBR_SHAPE_BUNNY) return fail(RTPBR_EINVAL, "unknown shape type");
    if (int r = set_dev(c)) return r;
    if (int r = flush_shade(c)) return r;
    ObjFull full[MAX_OBJ];
    memset(full, 0, sizeof full);
    bool all_box = true, all_bunny = true, any_bunny = false;
    for (int i = 0; i < n; i++) {
        c->obj[i] = objs[i];
        rtpbr_transform& t = c->obj[i].transform;
        if (scale10)
            for (int k = 0; k < 3; k++) {
                t.position[k] *= 10.0f;
                t.scale[k] *= 10.0f;
            }
        float rad[3] = {t.rotation[0] * DEG2RAD, t.rotation[1] * DEG2RAD, t.rotation[2] * DEG2RAD};
        rotate(rad, t.matrix);
        ObjM& m = c->objm[i];
        m.px = t.position[0]; m.py = t.position[1]; m.pz = t.position[2];
        memcpy(m.m, t.matrix, sizeof m.m);
        m.sx = t.scale[0]; m.sy = t.scale[1]; m.sz = t.scale[2];
        m.type = c->obj[i].type;
        ObjFull& f = full[i];
        memcpy(&f, &m, sizeof m);
        const rtpbr_material& mt = c->obj[i].material;
        memcpy(f.albedo, mt.albedo, 12);
        memcpy(f.emission, mt.emission, 12);
        f.roughness = mt.roughness; f.metallic = mt.metallic; f.transmission = mt.transmission; f.ior = mt.ior;
        if (m.type != RTPBR_SHAPE_BOX) all_box = false;
        if (m.type != RTPBR_SHAPE_BUNNY) all_bunny = false;
        else any_bunny = true;
    }
    c->scene_sig = (all_box && n == 8) ? choose_signature(c->objm) : 0;
    c->n_obj = n;
    c->P.n_obj = n;
    c->kind = all_box ? KIND_BOXES : (all_bunny && n == 1) ? KIND_BUNNY : any_bunny ? KIND_MIXED : KIND_GENERIC;
    if (!c->headless) {
        HIP_TRY(hipMemcpyAsync(c->objfull, full, sizeof(ObjFull) * n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // `full` is a stack buffer
    }
    c->have_scene = true;
    return RTPBR_OK;
}

// Test hook, HOST ONLY (no device needed): the signature an 8-box scene would be rendered with and
// its packed march table (128 floats), for tests/test_host_logic.py.
extern "C" int rtpbr_test_signature(const rtpbr_object* objs, int n, int scale10, uint32_t* sig, float* table) {
    if (!objs || !sig || n != 8) return RTPBR_EINVAL;
    ObjM objm[MAX_OBJ];
    memset(objm, 0, sizeof objm);
    bool all_box = true;
    for (int i = 0; i < n; i++) {
        rtpbr_transform t = objs[i].transform;
        if (scale10)
            for (int k = 0; k < 3; k++) {
                t.position[k] *= 10.0f;
                t.scale[k] *= 10.0f;
            }
        float rad[3] = {t.rotation[0] * DEG2RAD, t.rotation[1] * DEG2RAD, t.rotation[2] * DEG2RAD};
        rotate(rad, t.matrix);
        ObjM& m = objm[i];
        m.px = t.position[0]; m.py = t.position[1]; m.pz = t.position[2];
        memcpy(m.m, t.matrix, sizeof m.m);
        m.sx = t.scale[0]; m.sy = t.scale[1]; m.sz = t.scale[2];
        m.type = objs[i].type;
        if (m.type != RTPBR_SHAPE_BOX) all_box = false;
    }
    *sig = all_box ? choose_signature(objm) : 0;
    if (table) {
        ObjM packed[MAX_OBJ];
        pack_table(objm, n, *sig, packed);
        memcpy(table, packed, 128 * sizeof(float));
    }
    return RTPBR_OK;
}

// The key of the run-time instance of a scene (rt_jit.hip): everything the generated code depends on.  P carries the
// launch constants that get baked (box thresholds, tile geometry, scheduler thresholds, culling decision).
static void box_thresholds(float rho, int lazy_sqrt, Params& P) {
    P.box_lazy = (lazy_sqrt && rho >= 0.0f && rho <= 1e15f) ? 1 : 0;
    // the squared-distance keys of nearest_boxes_lazy are carried scaled by 4 (exact): thresholds likewise
    P.box_four_rho = 4.0f * rho;
    P.box_rho2m = (float)((double)rho * (double)rho * (1.0 + 1.0 / 524288.0));
    if (P.box_rho2m > 0.0f) P.box_rho2m = nextafterf(P.box_rho2m, INFINITY);
    P.box_4rho2m = nextafterf((float)(4.0 * (double)rho * (double)rho * (1.0 + 1.0 / 524288.0)), INFINITY);
    P.box_rho2m *= 4.0f;
    P.box_4rho2m *= 4.0f;
}
static RtJitKey make_jit_key(int kind, int n_obj, const ObjM* objm, const rtpbr_config& cfg, const Params& P, bool persistent, int jit_bake,
                             int jit_waves, bool jit_bunny, int fast = 0) {
    RtJitKey key{};
    key.fast = fast;
    key.kind = kind;
    key.n_obj = n_obj;
    for (int i = 0; i < n_obj; i++) {
        key.types |= (unsigned long long)(objm[i].type + 1) << (4 * i);
        key.sig |= (unsigned)rotation_class(objm[i].m) << (3 * i);
    }
    key.cull = P.cull_ok;
    key.form = persistent ? 1 : 0;
    if (jit_bunny) key.sig = 0;
    key.waves = kind == KIND_BOXES ? 6 : kind == KIND_BUNNY ? 4 : 5;      // as the ahead-of-time instances (RT_POOL_WAVES*)
    if (jit_waves > 0) key.waves = jit_waves;
    key.baked = jit_bake;
    key.table = reinterpret_cast<const unsigned*>(objm);
    if (key.baked) {
        rtpbr_config b = cfg;
        b.seed = 0;
        b.frame = 0;
        memcpy(key.cfg_words, &b, sizeof b);
        key.extra[0] = (unsigned)P.box_lazy;
        memcpy(&key.extra[1], &P.box_four_rho, 4);
        memcpy(&key.extra[2], &P.box_rho2m, 4);
        memcpy(&key.extra[3], &P.box_4rho2m, 4);
        const int ints[8] = {P.tile_w, P.tile_h, P.ntx, P.nty, P.world, P.shade_lanes, P.swap_lanes, P.mlp_mfma};
        memcpy(key.ints, ints, sizeof ints);
        if (key.baked == 2) memcpy(key.cam_words, &P.cam, sizeof key.cam_words);
    }
    return key;
}

// Test / tooling hook, HOST ONLY (no device needed): compile the BAKED run-time instance of a scene + configuration the
// way rtpbr_sample() would (single rank, default scheduler thresholds) and return the code object's path, so that the
// generated ISA can be inspected in a container without a GPU (tools/jit_offline.py).
extern "C" int rtpbr_test_jit_build_baked(const rtpbr_object* objs, int n, int scale10, const rtpbr_config* cfg, int waves, int fast, char* path_out, size_t cap) {
    if (!objs || !cfg || n <= 0 || n > 8) return fail(RTPBR_EINVAL, "bad arguments");
    ObjM objm[MAX_OBJ];
    memset(objm, 0, sizeof objm);
    bool all_box = true;
    for (int i = 0; i < n; i++) {
        rtpbr_transform t = objs[i].transform;
        if (scale10)
            for (int k = 0; k < 3; k++) {
                t.position[k] *= 10.0f;
                t.scale[k] *= 10.0f;
            }
        float rad[3] = {t.rotation[0] * DEG2RAD, t.rotation[1] * DEG2RAD, t.rotation[2] * DEG2RAD};
        rotate(rad, t.matrix);
        ObjM& m = objm[i];
        m.px = t.position[0]; m.py = t.position[1]; m.pz = t.position[2];
        memcpy(m.m, t.matrix, sizeof m.m);
        m.sx = t.scale[0]; m.sy = t.scale[1]; m.sz = t.scale[2];
        m.type = objs[i].type;
        if (m.type != RTPBR_SHAPE_BOX) all_box = false;
        if (m.type == RTPBR_SHAPE_BUNNY) return fail(RTPBR_EINVAL, "analytic shapes only");
    }
    rtpbr_ctx defaults;
    Params P{};
    P.cfg = *cfg;
    P.cull_ok = 1;
    P.tile_w = cfg->width, P.tile_h = cfg->height, P.ntx = 1, P.nty = 1, P.world = 1;
    P.shade_lanes = defaults.shade_lanes, P.swap_lanes = cfg->kernel_form == RTPBR_FORM_PERSISTENT_RAY ? 12 : 8;
    box_thresholds(cfg->box_round, 1, P);
    const RtJitKey key = make_jit_key(all_box ? KIND_BOXES : KIND_GENERIC, n, objm, *cfg, P, cfg->kernel_form == RTPBR_FORM_PERSISTENT_RAY, 1, waves, false, fast);
    std::string p;
    if (int r = rt_jit_build(key, &p, nullptr)) return r;
    if (path_out && cap) snprintf(path_out, cap, "%s", p.c_str());
    return RTPBR_OK;
}

// Ahead-of-time compilation of the run-time instance of ONE scene + configuration (+ camera, tiles, options) — NO DEVICE NEEDED:
// a headless context takes the same set_config / set_scene / set_camera / set_tiles / set_option calls a rendering host makes,
// derive_launch() forms the key rtpbr_sample() would ask for, and rt_jit_build() compiles it into $RTPBR_JIT_CACHE (or finds it
// there / in the catalog shipped next to the library, raytracingpbr_amd/data/jit).  __graft_entry__.build() fills that catalog for the
// BASELINE scenes this way, so that a target without hipcc still runs the scene-specialised kernels (rt_jit.hip looks there
// before it forks a compiler).  `options`: "key=value key=value ..." as for rtpbr_set_option (jit, jit_bake, precision, ...).
extern "C" int rtpbr_jit_prebuild(const rtpbr_object* objs, int n, int scale10, const rtpbr_config* cfg, const rtpbr_camera* cam,
                                  int tile_w, int tile_h, int world, const char* options, char* path_out, size_t cap) {
    if (!objs || !cfg || !cam) return fail(RTPBR_EINVAL, "rtpbr_jit_prebuild: scene, configuration and camera are required");
    rtpbr_ctx ctx;
    rtpbr_ctx* c = &ctx;
    c->headless = true;
    if (world > 1)
        if (int r = rtpbr_set_tiles(c, tile_w, tile_h, 0, world)) return r;
    if (int r = rtpbr_set_config(c, cfg)) return r;
    if (int r = rtpbr_set_scene(c, objs, n, scale10)) return r;
    if (int r = rtpbr_set_camera(c, cam)) return r;
    std::string opt = options ? options : "";
    for (size_t i = 0; i < opt.size();) {
        while (i < opt.size() && opt[i] == ' ') i++;
        size_t j = opt.find(' ', i);
        if (j == std::string::npos) j = opt.size();
        if (j > i) {
            const std::string kv = opt.substr(i, j - i);
            const size_t e = kv.find('=');
            if (e == std::string::npos) return fail(RTPBR_EINVAL, "rtpbr_jit_prebuild: options are key=value pairs (%s)", kv.c_str());
            if (int r = rtpbr_set_option(c, kv.substr(0, e).c_str(), atoll(kv.c_str() + e + 1))) return r;
        }
        i = j;
    }
    RtJitKey key{};
    bool want = false, strict_error = false;
    derive_launch(c, &key, &want, &strict_error);
    if (!want) return fail(RTPBR_ESTATE, "rtpbr_jit_prebuild: with these options rtpbr_sample() would not use a run-time instance for this scene "
                                         "(option jit, <= 8 analytic shapes or the neural shape with jit_bake, pool scheduler)");
    std::string p;
    if (int r = rt_jit_build(key, &p, nullptr)) return r;
    if (path_out && cap) snprintf(path_out, cap, "%s", p.c_str());
    return RTPBR_OK;
}

extern "C" int rtpbr_get_scene(rtpbr_ctx* c, rtpbr_object* objs, int n) {
    if (!c || !objs || n < 0 || n > c->n_obj) return fail(RTPBR_EINVAL, "bad get_scene arguments");
    memcpy(objs, c->obj, (size_t)n * sizeof *objs);
    return RTPBR_OK;
}

// thin-lens frame, src/camera.py:11-31 (same operation order as the oracle's camera_frame)
extern "C" int rtpbr_set_camera(rtpbr_ctx* c, const rtpbr_camera* cam) {
    if (!c || !cam) return fail(RTPBR_EINVAL, "null argument");
    if (int r = flush_shade(c)) return r;
    c->cam = *cam;
    vec3 lf = mk(cam->lookfrom[0], cam->lookfrom[1], cam->lookfrom[2]);
    vec3 la = mk(cam->lookat[0], cam->lookat[1], cam->lookat[2]);
    vec3 up = mk(cam->vup[0], cam->vup[1], cam->vup[2]);
    float theta = cam->vfov * DEG2RAD;
    float hh = tanf(theta * 0.5f);
    float hw = cam->aspect * hh;
    vec3 z = normalize(lf - la);
    vec3 x = normalize(cross(up, z));
    vec3 y = cross(z, x);
    vec3 hwfx = x * (hw * cam->focus);
    vec3 hhfy = y * (hh * cam->focus);
    vec3 llc = ((lf - hwfx) - hhfy) - z * cam->focus;
    vec3 hor = hwfx * 2.0f, ver = hhfy * 2.0f;
    CamFrame& f = c->P.cam;
    f.lf[0] = lf.x; f.lf[1] = lf.y; f.lf[2] = lf.z;
    f.x[0] = x.x; f.x[1] = x.y; f.x[2] = x.z;
    f.y[0] = y.x; f.y[1] = y.y; f.y[2] = y.z;
    f.llc[0] = llc.x; f.llc[1] = llc.y; f.llc[2] = llc.z;
    f.hor[0] = hor.x; f.hor[1] = hor.y; f.hor[2] = hor.z;
    f.ver[0] = ver.x; f.ver[1] = ver.y; f.ver[2] = ver.z;
    f.lens_radius = cam->aperture * 0.5f;
    c->have_cam = true;
    return RTPBR_OK;
}

// Image(path) + Image.process(exposure, gamma): src/ibl.py:14-23, postprocessor.adjust :17-21
extern "C" int rtpbr_set_env(rtpbr_ctx* c, const void* texels, int w, int h, int fmt, float exposure, float gamma) {
    if (!c || !texels) return fail(RTPBR_EINVAL, "null argument");
    if (w <= 0 || h <= 0) return fail(RTPBR_EINVAL, "bad env size");
    if (fmt != RTPBR_ENV_RGB8 && fmt != RTPBR_ENV_RGB32F) return fail(RTPBR_EINVAL, "bad env format");
    if (int r = set_dev(c)) return r;
    if (int r = flush_shade(c)) return r;
    size_t n = (size_t)w * h;
    std::vector<float4> host(n);
    if (fmt == RTPBR_ENV_RGB8) {
        const uint8_t* s = (const uint8_t*)texels;
        float lut[256];
        for (int i = 0; i < 256; i++) lut[i] = pow_(((float)i / 255.0f) * exposure, gamma);
        for (size_t i = 0; i < n; i++) host[i] = make_float4(lut[s[i * 3]], lut[s[i * 3 + 1]], lut[s[i * 3 + 2]], 0.0f);
    } else {
        const float* s = (const float*)texels;
        for (size_t i = 0; i < n; i++) host[i] = make_float4(s[i * 3], s[i * 3 + 1], s[i * 3 + 2], 0.0f);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(c->env);
    (void)hipFree(c->env8);
    (void)hipFree(c->env_lut);
    c->env = nullptr;
    c->env8 = nullptr;
    c->env_lut = nullptr;
    HIP_TRY(hipMalloc(&c->env, n * sizeof(float4)));
    HIP_TRY(hipMemcpy(c->env, host.data(), n * sizeof(float4), hipMemcpyHostToDevice));
    if (fmt == RTPBR_ENV_RGB8) {
        // ... and as the reference's image really is (T9): 8-bit texels + the table Image.process() amounts to
        const uint8_t* s8 = (const uint8_t*)texels;
        std::vector<uint32_t> packed(n);
        for (size_t i = 0; i < n; i++) packed[i] = (uint32_t)s8[i * 3] | ((uint32_t)s8[i * 3 + 1] << 8) | ((uint32_t)s8[i * 3 + 2] << 16);
        float lut[256];
        for (int i = 0; i < 256; i++) lut[i] = pow_(((float)i / 255.0f) * exposure, gamma);
        HIP_TRY(hipMalloc(&c->env8, n * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(&c->env_lut, sizeof lut));
        HIP_TRY(hipMemcpy(c->env8, packed.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->env_lut, lut, sizeof lut, hipMemcpyHostToDevice));
    }
    c->P.env = c->env;
    c->P.env_w = w;
    c->P.env_h = h;
    return RTPBR_OK;
}

extern "C" int rtpbr_set_shape_data(rtpbr_ctx* c, int shape, const float* data, int n) {
    if (!c || !data) return fail(RTPBR_EINVAL, "null argument");
    if (shape != RTPBR_SHAPE_BUNNY || n != 625) return fail(RTPBR_EINVAL, "only the bunny MLP (625 weights) takes shape data");
    if (int r = set_dev(c)) return r;
    if (int r = flush_shade(c)) return r;
    // device layout: the hidden layers' matrices in chain order (rt_device.hpp): [k][i][m][j] <- caller's [k][m][i][j]
    float dev[625];
    memcpy(dev, data, sizeof dev);
    for (int layer = 0; layer < 2; layer++)
        for (int k = 0; k < 4; k++)
            for (int m = 0; m < 4; m++)
                for (int i = 0; i < 4; i++)
                    for (int j = 0; j < 4; j++)
                        dev[64 + layer * 272 + k * 68 + i * 16 + m * 4 + j] = data[64 + layer * 272 + k * 68 + m * 16 + i * 4 + j];
    if (!c->bunny) HIP_TRY(hipMalloc(&c->bunny, 625 * sizeof(float)));
    HIP_TRY(hipMemcpy(c->bunny, dev, 625 * sizeof(float), hipMemcpyHostToDevice));
    c->P.bunny = c->bunny;
    return RTPBR_OK;
}

extern "C" int rtpbr_set_tiles(rtpbr_ctx* c, int tw, int th, int rank, int world) {
    if (!c) return fail(RTPBR_EINVAL, "null ctx");
    if (world < 1 || rank < 0 || rank >= world) return fail(RTPBR_EINVAL, "bad rank/world");
    if (world > 1 && (tw <= 0 || th <= 0)) return fail(RTPBR_EINVAL, "tile size must be > 0 when world > 1");
    if (int r = flush_shade(c)) return r;
    c->tile_w = tw;
    c->tile_h = th;
    c->rank = rank;
    c->world = world;
    if (c->have_cfg) {
        if (int r = check_local_pixels(c->cfg.width, c->cfg.height, tw, th, world)) return r;
        update_tiles(c);
    }
    return RTPBR_OK;
}

// A call that WRITES the buffers of `mask` (bit RTPBR_BUF_*) on the context's stream is ordered behind an asynchronous
// read-back that still copies out of them (rtpbr_read_buffer_async) — on the device: the host does not block.
int rt_order_after_reads(rtpbr_ctx* c, unsigned mask) {
    for (int b = 0; b < 5; b++)
        if (((mask >> b) & 1u) && c->read_pending[b] >= 0) {
            // (a copy that has landed already needs no ordering: a cross-stream wait is a barrier packet the command processor
            // resolves in ~20 us — per frame that is what separates a pipelined viewer from the device-only rate — a query is ~1 us)
            const hipError_t q = hipEventQuery(c->ev_read_done[c->read_pending[b] & 7]);
            if (q == hipErrorNotReady) {
                (void)hipGetLastError();
                HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_read_done[c->read_pending[b] & 7], 0));
            } else if (q != hipSuccess) {
                return rt_fail_hip("hipEventQuery(read-back)", q);
            }
            c->read_pending[b] = -1;
        }
    return RTPBR_OK;
}
enum : unsigned { W_IMAGE_BUFFER = 1u << RTPBR_BUF_IMAGE_BUFFER, W_IMAGE_PIXELS = 1u << RTPBR_BUF_IMAGE_PIXELS, W_RAY_BUFFER = 1u << RTPBR_BUF_RAY_BUFFER,
                W_DIFF_BUFFER = 1u << RTPBR_BUF_DIFF_BUFFER, W_DIFF_PIXELS = 1u << RTPBR_BUF_DIFF_PIXELS };

extern "C" int rtpbr_refresh(rtpbr_ctx* c) {
    if (!c || !c->have_cfg) return fail(RTPBR_ESTATE, "set_config first");
    if (int r = set_dev(c)) return r;
    if (int r = flush_shade(c)) return r;
    if (int r = rt_order_after_reads(c, W_IMAGE_BUFFER | W_RAY_BUFFER | W_DIFF_BUFFER | W_DIFF_PIXELS)) return r;
    size_t n = (size_t)c->cfg.width * c->cfg.height;
    launch_refresh(c->image_buffer, c->ray_buffer, c->diff_buffer, c->diff_pixels, c->cfg.adaptive_sampling, n, c->stream);
    HIP_TRY(hipGetLastError());
    return RTPBR_OK;
}

static void pack_objects(rtpbr_ctx* c, Params& P) { pack_table(c->objm, c->n_obj, P.box_sig, P.objm); }

// Staging (one 12-byte StageRec per item) and, for the primary split, the primary records (5 bytes per item: a float, then a byte).
// Grown on demand; hipMalloc of several GB takes 50..700 ms, so callers that time whole frames can
// reserve up front (option "reserve_spp").
// Returns RTPBR_ENOMEM (nothing allocated, no sticky HIP error) when the device has no room: the caller then renders
// with fewer samples per launch instead of failing.
constexpr size_t PRIMARY_REC_BYTES = 5;     // float t_eval + one byte of idx | state
// staging of `items` samples: the 12-byte records, then (dense staging) one fill count per chunk of >= 32 items and one byte per record
constexpr size_t STAGE_ITEM_BYTES = 14;     // what a sample is budgeted at (12 + 1 + 4 / 32, rounded up)
constexpr uint32_t DENSE_CHUNK_MIN = 32, DENSE_CHUNK_MAX = 256, DENSE_BATCH_MAX = 4096;
static size_t stage_bytes(size_t items) { return items * sizeof(StageRec) + (items / DENSE_CHUNK_MIN + 2) * sizeof(uint32_t) + items; }
static int staging_alloc(rtpbr_ctx* c, void** ptr, size_t* cap, size_t need) {
    if (need <= *cap) return RTPBR_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(*ptr);
    *ptr = nullptr;
    *cap = 0;
    const hipError_t e = hipMalloc(ptr, need);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        *ptr = nullptr;
        return RTPBR_ENOMEM;
    }
    if (e != hipSuccess) return rt_fail_hip("hipMalloc(staging)", e);
    *cap = need;
    return RTPBR_OK;
}
static int ensure_staging(rtpbr_ctx* c, size_t items, bool split, bool stage = true) {
    if (stage)
        if (int r = staging_alloc(c, (void**)&c->stage, &c->stage_cap, stage_bytes(items))) return r;
    if (split)
        if (int r = staging_alloc(c, (void**)&c->primary, &c->primary_cap, items * PRIMARY_REC_BYTES)) return r;
    return RTPBR_OK;
}

// next event of a reusable pool; a failed hipEventCreate is reported, never recorded
static int next_event_of(std::vector<hipEvent_t>& pool, int& used, hipEvent_t* out) {
    if (used == (int)pool.size()) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));      // (timing only: see create_resources)
        pool.push_back(e);
    }
    *out = pool[used++];
    return RTPBR_OK;
}
#define NEXT_EVENT(pool, used, var)                                   \
    hipEvent_t var = nullptr;                                         \
    if (int r_ = next_event_of(pool, used, &var)) return r_

static int trace_grid(rtpbr_ctx* c, uint32_t total_items) {
    int per_cu = c->jit_mod ? c->jit_mod->trace_blocks_per_cu
                            : trace_blocks_per_cu(c->kind, c->n_obj, c->P.box_sig, c->scheduler < 0 ? 1 : c->scheduler);
    if (per_cu <= 0) per_cu = 2;
    if (c->waves_per_cu > 0) per_cu = (c->waves_per_cu + 3) / 4;
    long long grid = (long long)per_cu * c->n_cu;
    long long need = ((long long)total_items + 255) / 256;
    if (grid > need) grid = need;
    if (grid < 1) grid = 1;
    return (int)grid;
}

// Room kept free below 2^32 for the claims the last waves make past the end of the work (one failing claim per wave):
// at most 32 waves per CU, at most 8192 items per claim (option "chunk" is clamped to that).
static long long work_margin(const rtpbr_ctx* c) { return (long long)(c->n_cu > 0 ? c->n_cu : 256) * 32LL * 8192LL; }

// Everything of a launch that is decided on the HOST from the context's state (configuration, scene, camera, tiles, options) and
// that the run-time instance's key depends on: shared by rtpbr_sample() and rtpbr_jit_prebuild() (which has no device), so that
// a code object compiled ahead of time for a catalog scene IS the one rtpbr_sample() will ask for.  *want = a run-time instance
// is to be used (key filled); *strict_error = option jit = 2 and no instance exists for this scene.
static void derive_launch(rtpbr_ctx* c, RtJitKey* key, bool* want, bool* strict_error) {
    Params& P = c->P;
    P.cfg = c->cfg;
    P.cam.inv_w = 1.0f / (float)c->cfg.width;
    P.cam.inv_h = 1.0f / (float)c->cfg.height;
    P.image_buffer = c->image_buffer;
    P.image_pixels = c->image_pixels;
    P.ray_buffer = c->ray_buffer;
    P.diff_buffer = c->diff_buffer;
    P.diff_pixels = c->diff_pixels;
    P.objfull = c->objfull;
    P.env8 = (c->env_packed && c->env8) ? c->env8 : nullptr;
    P.env_lut = c->env_lut;
    P.work_counter = c->work_counter;
    P.counters = c->counters;
    P.wait_lanes = c->wait_lanes;
    P.shade_lanes = c->shade_lanes;
    P.refill_lanes = c->refill_lanes;
    P.ready_low = c->ready_low;
    // lanes that must have finished before the march loop is left for a swap: the src/ form's swap moves 15-word contexts and
    // runs once per ~8 raycasts — 12 instead of 8 is +1.5 % at every frame size (profiles/r04); the complete-path kernel stays at 8
    P.swap_lanes = c->swap_lanes > 0 ? c->swap_lanes : (c->cfg.kernel_form == RTPBR_FORM_PERSISTENT_RAY ? 12 : 8);
    P.sparse_lanes = c->sparse_lanes;
    P.mlp_lanes = c->mlp_lanes;
    P.mlp_full = c->mlp_full;
    P.mlp_mfma = c->mlp_mfma;
    P.scheduler = c->scheduler < 0 ? 1 : c->scheduler;
    // the pool kernel's parked records hold the bounce number in 11 bits
    if (c->cfg.kernel_form == RTPBR_FORM_COMPLETE_PATH && c->cfg.max_raytrace > 2047) P.scheduler = 0;
    // signature instances exist for the complete-path kernels only
    P.box_sig = (c->specialize && c->cfg.kernel_form == RTPBR_FORM_COMPLETE_PATH) ? c->scene_sig : 0;
    {
        // nearest_culled needs |sdf| to be 1-Lipschitz: true for every analytic shape except a cone whose
        // slope vector (scale.x, scale.z) is longer than 1; the rounding allowance scales with the scene
        // (the camera position counts: camera rays start there)
        float ext = 16.0f + fabsf(P.cam.lf[0]) + fabsf(P.cam.lf[1]) + fabsf(P.cam.lf[2]);
        bool ok = c->n_obj <= 8 && c->kind != KIND_BUNNY && c->kind != KIND_MIXED && ext <= 1e12f;
        for (int i = 0; i < c->n_obj; i++) {
            const ObjM& o = c->objm[i];
            const float e = fabsf(o.px) + fabsf(o.py) + fabsf(o.pz) + fabsf(o.sx) + fabsf(o.sy) + fabsf(o.sz);
            if (!(e <= 1e12f)) ok = false;
            if (e > ext) ext = e;
            if (o.type == RTPBR_SHAPE_CONE && !(o.sx * o.sx + o.sz * o.sz <= 1.0f)) ok = false;
        }
        P.cull_ok = ok ? 1 : 0;
        P.cull_extent = 4.0f * ext;
    }
    box_thresholds(c->cfg.box_round, c->lazy_sqrt, P);
    *want = false;
    *strict_error = false;
    const bool jit_bunny = c->kind == KIND_BUNNY && c->jit_bake && c->jit >= 1;   // configuration baking only (incl. which units run the network)
    const bool persistent = c->cfg.kernel_form == RTPBR_FORM_PERSISTENT_RAY;
    if (c->jit != 0 && (persistent || P.scheduler == 1) && c->n_obj <= 8 && (c->kind == KIND_BOXES || c->kind == KIND_GENERIC || (jit_bunny && !persistent))) {
        const bool aot_special = c->kind == KIND_BOXES && c->n_obj == 8 && P.box_sig != 0;
        if (c->jit >= 1 || !aot_special || c->precision) {
            *key = make_jit_key(c->kind, c->n_obj, c->objm, c->cfg, P, persistent, c->jit_bake, c->jit_waves, jit_bunny, c->precision);
            key->dense = (c->stage_dense && !persistent && !c->precision) ? 1 : 0;
            *want = true;
        }
    } else if (c->jit == 2) {
        *strict_error = true;
    }
}

static int launch_split_steps(rtpbr_ctx* c, int steps);
// ------------------------------------------------------------------------------------------------------------------------
// THE HOST SCHEDULER'S CONSTANTS, in one place.  Everything below that is a number was MEASURED on one MI355X (256 CUs, 4 SIMDs
// per CU, 160 KB of LDS per CU); what follows from the device is read from hipDeviceProp (c->n_cu) or from the kernels'
// occupancy (hipOccupancyMaxActiveBlocksPerMultiprocessor / the run-time module's own query).  Option defaults — the numbers a
// caller can change — live in rt_ctx.hpp next to the option they belong to; rtpbr.h documents them.  On another SKU or a
// partitioned device (CPX) these are the lines to re-measure (tools/gpu_src_grid.py, tools/gpu_src_1step.py sweep them).
namespace tune {
constexpr int WAVES_PER_BLOCK = 4;              // 256 threads: one wave per SIMD of a CU
constexpr int CTX_PER_WAVE = 128;               // src/ pool kernel: 64 lanes + 64 LDS slots (rt_persistent.hpp)
constexpr int MAX_FUSED_STEPS = 256;            // bounce-steps per fused src/ launch (G_META holds 9 bits)
// --- src/ pool kernel (fused launches) -----------------------------------------------------------------------------------
constexpr int POOL_MIN_PIX_PER_WAVE = 64;       // a wave never owns fewer pixels than it has lanes (grid <= np / 256)
// beside the chain kernel a pool wave should own ~190 pixels, in whole blocks per CU, at least two (two waves of 162 contexts
// march with 54-57 of 64 lanes, three of 108 with 34; a wave issues one instruction per ~6.5 cycles whatever it carries):
// 1024x576 four -> three blocks per CU 30.8 -> 27.3 ms, 960x540 29.5 -> 24.2, 768x432 three -> two 22.7 -> 22.3 (round 5)
constexpr int CHAIN_POOL_PIX_PER_BLOCK = 760;
constexpr int CHAIN_POOL_MIN_BLOCKS_PER_CU = 2;
constexpr int CHAIN_GRID_BLOCKS = 512;          // chain kernel: 2048 waves = the largest chain set a plan can make (waves without an entry leave at once)
// --- src/ wavefront split (launches of <= src_split steps) -----------------------------------------------------------------
// waves per SIMD of the march kernel: 2 / 3 / 4 / 5 / 8 = 0.53 / 0.51 / 0.51 / 0.55 / 0.59 ms per 1080p launch (the arbiter
// serves the oldest wave first: with more the youngest starve and end last)
constexpr int MARCH_MAX_BLOCKS_PER_CU = 4;
// small frames are all tail: two waves per SIMD (640x360 0.230 / 0.236 / 0.240 ms at 2 / 3 / 4, 768x432 0.258 / 0.261 / 0.270,
// no difference from 1024x576 on) and the list's heavy head interleaved over the groups (768x432 0.2375 -> 0.224; 1080p loses)
constexpr long long MARCH_SMALL_FRAME_PIXELS = 600000;
constexpr int MARCH_SMALL_BLOCKS_PER_CU = 2;
constexpr int MARCH_MAX_TEAMS = 1024;           // team counters allocated (rtpbr_create)
constexpr int SPARSE_LANES_DEFAULT = 24;        // option sparse_lanes as rt_ctx.hpp sets it (fused pool kernel: tracked march when <= 24 lanes march)
constexpr int SPLIT_SPARSE_LANES = 8;           // ... the split march kernel with the object-parallel evaluation: from 8 lanes down
// --- complete-path form ----------------------------------------------------------------------------------------------------
constexpr long long PRIMARY_SPLIT_MIN_ITEMS = 1LL << 23;    // the separate primary kernel costs ~0.3 ms per launch: below ~8 M items the fused kernel wins
// work items a wave claims per atomic: total / (waves x 64) clamped to [256, 1024]; 128 when a launch has fewer than 256 per wave
// (1080p x 16 spp: 64 / 128 / 256 / 512 items = 10.6 / 9.6 / 9.2 / 9.3 ms; 4096 cost 2 % on the Cornell frame and 18 % on the
// glass bunny — a wave that claims the last chunk works it off 128 paths at a time; C1: 1407 -> 1492 Msamples/s with 128)
constexpr long long CHUNK_MIN = 256, CHUNK_MAX = 1024, CHUNK_TINY = 128;
constexpr int PRIMARY_BLOCKS_PER_CU = 8;        // the primary kernel needs ~57 VGPRs: 32 waves per CU
}  // namespace tune

// What the LAST plan decided about the chain set, learned asynchronously (a 4-byte copy behind the plan kernels): the pool
// grid makes room for the chain kernel only when a plan actually produced one (advisor, round 5: the grid used to shrink
// whenever the chain kernel was merely possible — uniform-cost scenes and the launches before the first plan then ran the
// pool kernel alone on the reduced grid, measured slower: 36 against 26 ms at 768x432).
static int poll_plan_readback(rtpbr_ctx* c) {
    if (!c->plan_rb_pending) return RTPBR_OK;
    const hipError_t q = hipEventQuery(c->ev_plan_rb);
    if (q == hipSuccess) {
        c->plan_chain_waves = (int)*c->plan_rb_host;
        c->plan_rb_pending = false;
    } else if (q != hipErrorNotReady) {
        return rt_fail_hip("hipEventQuery(plan read-back)", q);
    } else {
        (void)hipGetLastError();
    }
    return RTPBR_OK;
}

// src/ persistent-ray form: n launches of pathtrace() (src/renderer.py:29-30), each cfg.steps_per_launch bounce-steps.
// A pixel's steps are sequential and the RNG is keyed by the absolute step index, so k launches of s steps equal one launch of
// k*s steps bit for bit: fuse them (<= 256 steps per kernel) instead of paying a launch + an 80 B/pixel ray_buffer round
// trip per step.
static int sample_persistent(rtpbr_ctx* c, int n) {
    Params& P = c->P;
    long long left = (long long)n * c->cfg.steps_per_launch;
    while (left > 0) {
        int steps = (int)(left < tune::MAX_FUSED_STEPS ? left : tune::MAX_FUSED_STEPS);
        P.sample_base = c->sample_base;
        NEXT_EVENT(c->ev, c->ev_used, a);
        NEXT_EVENT(c->ev, c->ev_used, b);
        if (c->timed) HIP_TRY(hipEventRecord(a, c->stream));
        // pool scheduler unless asked otherwise: measured faster than one lane per pixel at every frame size from
        // 256x256 up (profiles/r03_src_*; round 2 switched at 2^20 pixels by a guess)
        const bool use_pool = c->scheduler != 0;
        if (use_pool) {
            // Static ownership (persistent_pool_impl): every resident wave owns np / waves pixels.  A wave holds 128 contexts:
            // when the frame fits (np <= 128 x resident waves) the grid is sized so that every wave owns <= 128 pixels and
            // keeps them for the whole launch, in whole multiples of the CU count (every CU the same number of blocks), but
            // never fewer than 64 pixels per wave; larger frames use every resident wave and walk their pixels in
            // residencies of `residency` bounce-steps.
            int per_cu = c->jit_mod ? c->jit_mod->persistent_blocks_per_cu : persistent_pool_blocks_per_cu(c->kind);
            if (per_cu <= 0) per_cu = 2;
            if (c->waves_per_cu > 0) per_cu = (c->waves_per_cu + 3) / 4;
            const long long max_blocks = (long long)per_cu * c->n_cu;
            const long long ctx_per_block = (long long)tune::CTX_PER_WAVE * tune::WAVES_PER_BLOCK;
            long long grid = ((long long)P.np + ctx_per_block - 1) / ctx_per_block;
            if (grid >= max_blocks) {
                grid = max_blocks;
            } else {
                grid = (grid + c->n_cu - 1) / c->n_cu * c->n_cu;
                if (grid > max_blocks) grid = max_blocks;
                const long long dense = (long long)P.np / (tune::POOL_MIN_PIX_PER_WAVE * tune::WAVES_PER_BLOCK);
                if (grid > dense) grid = dense;
            }
            // The chain kernel (rt_chain.hpp) runs BESIDE the pool kernel and needs wave slots of its own.  Frames up to about
            // 1080p may be chain-bound (plan_scan decides, on the device, against the grid the pool kernel has beside the chain kernel);
            // when the last plan made a chain set the pool grid makes room — whole multiples of the CU count: an uneven grid
            // costs more than it gives (1024x576 42.8 -> 30 ms, 1280x720 46 -> 40, 1600x900 54.7 -> 50.4, 1080p 58.6 -> 57.3) —
            // and is sized for ~190 pixels per wave.  Larger frames are throughput-bound and keep the whole device for the pool
            // kernel (2560x1440: 97.2 against 100.3 ms).
            if (int r = poll_plan_readback(c)) return r;
            const long long chain_blocks = (c->chain_waves + tune::WAVES_PER_BLOCK - 1) / tune::WAVES_PER_BLOCK;
            const bool chain_candidate = c->src_chain != 0 && c->grid_blocks == 0 && c->src_plan && (long long)P.np <= c->chain_np_max;
            const bool chain_room = chain_candidate && (c->plan_chain_waves > 0 || c->src_chain == 2);
            // (the grid WITH room is computed whether or not room is made: it is what the plan decides "chain-bound" against, so that
            // the decision is the same before and after — the host makes room only once a plan has said so)
            // Two separate things (round 6 took them apart: `src_chain = 0 / 1 / 2` at 1080p = 56.3 / 56.5 / 53.4 ms showed that what
            // helps there is the GRID, the plan never finds 1080p chain-bound): (a) a frame that may get a chain set leaves one block
            // per CU free — four blocks per CU instead of five are faster at these sizes with or without a chain kernel beside them
            // (1080p 56 -> 53.4 ms); (b) the ~190 pixels per wave of a small frame pay only BESIDE a chain kernel (without one the
            // heaviest pixels need the larger grid: 768x432 26 against 36 ms) — applied once a plan has produced a chain set.
            long long grid_room = grid;
            if (chain_candidate) {
                if (grid_room + chain_blocks > max_blocks && max_blocks - chain_blocks >= c->n_cu) grid_room = (max_blocks - chain_blocks) / c->n_cu * c->n_cu;
                if (grid_room < 1) grid_room = 1;
                grid = grid_room;                                                     // (a)
                long long want = ((long long)P.np / tune::CHAIN_POOL_PIX_PER_BLOCK + c->n_cu / 2) / c->n_cu * c->n_cu;
                if (want < (long long)tune::CHAIN_POOL_MIN_BLOCKS_PER_CU * c->n_cu) want = (long long)tune::CHAIN_POOL_MIN_BLOCKS_PER_CU * c->n_cu;
                if (grid_room > want) grid_room = want;                               // (b): what the plan decides "chain-bound" against
            }
            if (chain_room) grid = grid_room;
            if (c->grid_blocks > 0) grid = c->grid_blocks;
            if (grid < 1) grid = 1;
            P.total_items = (uint32_t)P.np;
            P.chunk = (uint32_t)c->residency;                               // bounce-steps per residency (a power of two)
            // Cost-ordered ownership: the kernel records every pixel's march steps; once plan_interval bounce-steps
            // are on record the pixels are re-ordered by them (three small kernels on the same stream) and the
            // heaviest get waves of their own.  The plan survives refresh(): what a pixel costs is a property of
            // the scene and the camera, not of the accumulated image.
            P.cost_buffer = nullptr;
            P.order = nullptr;
            P.plan = nullptr;
            P.heavy_own = c->heavy_own;
            P.heavy_prio = c->heavy_prio;
            P.src_track = c->src_track;
            P.src_op = c->src_op;
            P.leave_x8 = c->leave_x8;
            P.tiny_own = c->tiny_own;
            P.n_cu = c->n_cu;
            P.age_on = c->age_on;
            P.age_pack = 0;
            for (int k = 0; k < 8; k++) P.age_pack |= (uint32_t)(c->age_w[k] & 15) << (4 * k);
            // the chain kernel is launched only when the device has room for both: more resident waves than it holds just queue
            // the pool's last blocks behind the chain waves (measured -12 % at 720p)
            const bool chain_fits = chain_candidate && grid + chain_blocks <= max_blocks;
            const bool chain_eff = c->src_chain != 0 && (chain_fits || c->src_chain == 2);
            if (c->src_plan) {
                if (c->plan_np != (size_t)P.np) {
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    (void)hipFree(c->cost_buffer);
                    (void)hipFree(c->order);
                    c->cost_buffer = c->order = nullptr;
                    c->plan_np = 0;
                    c->order_valid = false;
                    c->cost_steps = 0;
                    c->plan_chain_waves = 0;
                    c->plan_rb_pending = false;
                    HIP_TRY(hipMalloc(&c->cost_buffer, (size_t)P.np * sizeof(uint32_t)));
                    HIP_TRY(hipMalloc(&c->order, (size_t)P.np * sizeof(uint32_t)));
                    if (!c->plan) {
                        HIP_TRY(hipMalloc(&c->plan, sizeof(PlanBuf)));
                        HIP_TRY(hipMemsetAsync(c->plan, 0, sizeof(PlanBuf), c->stream));
                    }
                    HIP_TRY(hipMemsetAsync(c->cost_buffer, 0, (size_t)P.np * sizeof(uint32_t), c->stream));
                    c->plan_np = (size_t)P.np;
                }
                if (c->cost_steps >= c->plan_interval) {
                    // (the plan sizes its heavy waves for THIS launch's grid and decides "chain-bound" against the grid the pool
                    // kernel has beside the chain kernel: the decision does not depend on whether room has been made already)
                    launch_plan(c->cost_buffer, c->order, c->plan, (uint32_t)P.np, (uint32_t)grid * 4u, c->heavy_own, c->heavy_mean_x16,
                                c->heavy_bulk_x16, c->tiny_waves, c->n_cu, (int)((grid + c->n_cu - 1) / c->n_cu), chain_candidate || c->src_chain == 2 ? c->chain_waves : 0,
                                (uint32_t)grid_room * 4u, c->stream);
                    c->order_valid = true;
                    c->cost_steps = 0;
                    if (!c->plan_rb_host) {
                        HIP_TRY(hipHostMalloc((void**)&c->plan_rb_host, 64, hipHostMallocDefault));
                        HIP_TRY(hipEventCreateWithFlags(&c->ev_plan_rb, hipEventDisableTiming));
                    }
                    HIP_TRY(hipMemcpyAsync(c->plan_rb_host, &c->plan->n_chain_waves, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
                    HIP_TRY(hipEventRecord(c->ev_plan_rb, c->stream));
                    c->plan_rb_pending = true;
                }
                P.cost_buffer = c->cost_buffer;
                P.order = c->order_valid ? c->order : nullptr;
                P.plan = c->plan;
                c->cost_steps += steps;
            }
            if (c->src_split > 0 && steps <= c->src_split) {
                if (int r = launch_split_steps(c, steps)) return r;
            } else {
                if (int r = flush_shade(c)) return r;
                // A chain-bound launch hands the head of the cost-ordered list to the chain kernel (rt_chain.hpp), which runs
                // BESIDE the pool kernel on a second stream: forked after the plan, joined before anything else touches the
                // buffers.  Its grid covers the largest chain set a plan can make; waves without an entry leave at once.
                P.chain_on = (chain_eff && P.order) ? 1 : 0;
                bool forked = false;
                int rc = RTPBR_OK;
                if (P.chain_on) {
                    if (!c->stream2) {
                        HIP_TRY(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
                        HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
                        HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
                    }
                    HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
                    HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
                    forked = true;
                    if (c->jit_mod)
                        rc = rt_jit_launch_steps(c->jit_mod->chain_steps, P, steps, (unsigned)tune::CHAIN_GRID_BLOCKS, c->stream2);
                    else
                        launch_chain_steps(P, c->kind, steps, tune::CHAIN_GRID_BLOCKS, c->stream2);
                }
                if (rc == RTPBR_OK) {
                    if (c->jit_mod)
                        rc = rt_jit_launch_steps(c->jit_mod->persistent_pool, P, steps, (unsigned)grid, c->stream);
                    else
                        launch_persistent_pool(P, c->kind, steps, (int)grid, c->stream);
                }
                // JOIN on every exit once the second stream has been forked (advisor, round 5): nothing that follows on the
                // context's stream may overtake a chain kernel that is still running — if the event cannot be recorded, wait
                // for the stream itself
                if (forked) {
                    if (hipEventRecord(c->ev_join, c->stream2) != hipSuccess || hipStreamWaitEvent(c->stream, c->ev_join, 0) != hipSuccess) {
                        (void)hipGetLastError();
                        (void)hipStreamSynchronize(c->stream2);
                    }
                }
                P.chain_on = 0;      // (persistent state of the context: only this launch ran with the chain set split off)
                if (rc != RTPBR_OK) return rc;
            }
        } else if (c->jit_mod) {
            if (int r = rt_jit_launch_steps(c->jit_mod->persistent_steps, P, steps, (unsigned)((P.np + 255) / 256), c->stream)) return r;
        } else {
            if (int r = flush_shade(c)) return r;
            launch_persistent(P, c->kind, steps, c->stream);
        }
        if (c->timed) HIP_TRY(hipEventRecord(b, c->stream));
        c->sample_base += (uint32_t)steps;
        left -= steps;
    }
    return RTPBR_OK;
}

// A launch of one (or a few) bounce-steps — the way the reference calls pathtrace(), src/renderer.py:29-30 — as the wavefront
// split of rt_split.hpp: per step gen (roulette / deposit / camera ray), march (the raycasts, heaviest first from the
// cost-ordered list), shade.  Same results, same counters as the fused kernels.
static int launch_split_steps(rtpbr_ctx* c, int steps) {
    Params& P = c->P;
    if (c->march_np != (size_t)P.np) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        (void)hipFree(c->march_out);
        c->march_out = nullptr;
        c->march_np = 0;
        HIP_TRY(hipMalloc(&c->march_out, (size_t)P.np * sizeof(uint32_t)));
        c->march_np = (size_t)P.np;
    }
    P.march_out = c->march_out;
    P.wait_lanes = c->split_wait;
    P.chain_on = 0;
    // with the object-parallel evaluation the tracked forms pay from 8 marching lanes down; between 9 and 24 a wave's lanes rarely
    // share an object and the three-smallest evaluation (2.7 kcycles per call in the tails) only replaces a plain step (1.1 k):
    // 768x432 0.2125 -> 0.205 ms per launch, 1080p 0.454 -> 0.450.  (An explicit sparse_lanes — the tests force 64 — is respected.)
    const int sparse_saved = P.sparse_lanes;
    if (c->sparse_lanes == tune::SPARSE_LANES_DEFAULT && (c->src_op & 1) && c->n_obj <= 8) P.sparse_lanes = tune::SPLIT_SPARSE_LANES;
    int mper = c->jit_mod ? c->jit_mod->march_blocks_per_cu : src_march_blocks_per_cu(c->kind);
    if (mper <= 0) mper = 2;
    if (mper > tune::MARCH_MAX_BLOCKS_PER_CU) mper = tune::MARCH_MAX_BLOCKS_PER_CU;
    const bool small = (long long)P.np <= tune::MARCH_SMALL_FRAME_PIXELS;
    if (small && mper > tune::MARCH_SMALL_BLOCKS_PER_CU) mper = tune::MARCH_SMALL_BLOCKS_PER_CU;
    P.split_head = c->split_head >= 0 ? c->split_head : (small ? 1 : 0);
    if (c->waves_per_cu > 0) mper = (c->waves_per_cu + 3) / 4;
    long long mgrid = (long long)mper * c->n_cu;
    const long long need = ((long long)P.np + 255) / 256;
    if (mgrid > need) mgrid = need;
    if (c->grid_blocks > 0) mgrid = c->grid_blocks;
    if (mgrid < 1) mgrid = 1;
    // teams of blocks that share a claim counter: the blocks resident on one CU (blocks are placed round-robin)
    P.team_counter = c->team_counter;
    P.n_teams = (int)(mgrid < c->n_cu ? mgrid : c->n_cu);
    if (P.n_teams > tune::MARCH_MAX_TEAMS) P.n_teams = tune::MARCH_MAX_TEAMS;
    int rc = RTPBR_OK;
    // lazy shading: a step's shading rides with the next step's gen pass (src_shade_gen) — flush_shade() launches it when something
    // else wants to see ray_buffer or the counters first
    const bool lazy = c->src_lazy != 0 && !c->ray_ptr_out;
    for (int i = 0; i < steps && rc == RTPBR_OK; i++) {
        P.sample_base = c->sample_base + (uint32_t)i;
        const bool shade_first = c->shade_pending, count = c->shade_pending_same_call;
        if (shade_first) {
            P.shade_base = c->shade_pending_base;
            c->shade_pending = false;
        }
        if (c->jit_mod) {
            rc = rt_jit_launch(!shade_first ? c->jit_mod->src_gen : count ? c->jit_mod->src_shade_gen_count : c->jit_mod->src_shade_gen, P, (unsigned)need, c->stream);
            if (rc == RTPBR_OK) rc = rt_jit_launch(c->jit_mod->src_march, P, (unsigned)mgrid, c->stream);
            if (rc == RTPBR_OK && !lazy) rc = rt_jit_launch(c->jit_mod->src_shade, P, (unsigned)need, c->stream);
        } else {
            if (shade_first) launch_src_shade_gen(P, c->kind, count, c->stream);
            else launch_src_gen(P, c->kind, c->stream);
            launch_src_march(P, c->kind, (int)mgrid, c->stream);
            if (!lazy) launch_src_shade(P, c->kind, c->stream);
        }
        if (lazy && rc == RTPBR_OK) {
            c->shade_pending = true;
            c->shade_pending_base = P.sample_base;
            c->shade_pending_same_call = true;
        }
    }
    P.sparse_lanes = sparse_saved;
    return rc;
}

// complete-path form: `n` samples per owned pixel (the spp loop of cornell_box_v3/renderer.py:31-36), in sub-launches of as
// many samples per pixel as the staging budget holds
static int sample_complete_path(rtpbr_ctx* c, int n) {
    Params& P = c->P;
    int left = n;
    if (int r = flush_shade(c)) return r;
    c->dense_launches = 0;
    while (left > 0) {
        const bool split_ok = c->primary_split && P.scheduler == 1 && c->kind != KIND_BUNNY && c->kind != KIND_MIXED;
        // the tolerance flavour accumulates in LDS and adds to image_buffer directly: no staging, no accumulate kernel
        const bool unstaged = c->precision != 0 && c->jit_mod != nullptr && P.scheduler == 1;
        long long per_spp = (long long)P.np * (long long)((unstaged ? 0 : STAGE_ITEM_BYTES) + (split_ok ? PRIMARY_REC_BYTES : 0));
        if (per_spp < 1) per_spp = 1;
        long long kmax = c->staging_bytes / per_spp;
        if (kmax < 1) kmax = 1;
        // keep total_items within 32 bits (minus the chunks the last waves claim past the end: the work counter must not wrap)
        long long k32 = (0xFFFFFFFFLL - work_margin(c)) / (long long)P.np;
        if (k32 < 1) k32 = 1;
        if (kmax > k32) kmax = k32;
        int K = (int)(left < kmax ? left : kmax);
        const bool split = split_ok && (c->primary_split == 2 || (long long)P.np * K >= tune::PRIMARY_SPLIT_MIN_ITEMS);
        if (int r = ensure_staging(c, (size_t)P.np * (size_t)K, split, !unstaged)) {
            // no room for K samples per launch: halve the budget and go round again (1 spp per launch must fit)
            if (r != RTPBR_ENOMEM || K == 1)
                return r == RTPBR_ENOMEM ? fail(RTPBR_ENOMEM, "no device memory for the staging of one sample per pixel") : r;
            c->staging_bytes = (long long)((K + 1) / 2) * per_spp;
            continue;
        }
        P.primary = c->primary;
        P.primary_code = reinterpret_cast<uint8_t*>(c->primary + (size_t)P.np * (size_t)K);
        P.primary_split = split ? 1 : 0;
        P.primary_lean = c->primary_lean;
        P.drain_lanes = c->drain_lanes;
        P.stage = c->stage;
        P.K = K;
        P.sample_base = c->sample_base;
        P.total_items = (uint32_t)((long long)P.np * K);
        int grid = trace_grid(c, P.total_items);
        long long waves = (long long)grid * tune::WAVES_PER_BLOCK;
        long long chunk = (long long)P.total_items / (waves * 64);
        if (chunk < tune::CHUNK_MIN) chunk = (long long)P.total_items < waves * tune::CHUNK_MIN ? tune::CHUNK_TINY : tune::CHUNK_MIN;
        if (chunk > tune::CHUNK_MAX) chunk = tune::CHUNK_MAX;
        if (c->chunk > 0) chunk = c->chunk;
        P.chunk = (uint32_t)chunk;
        // dense staging (rt_trace.hpp stage_sample, accumulate_dense): the claim must be whole pixels or a whole fraction of one, and
        // fit the record's one-byte offset; a launch that has no such claim size near the one wanted keeps the item-linear records
        P.stage_dense = 0;
        if (c->stage_dense && c->jit_mod != nullptr && !unstaged && P.scheduler == 1) {      // (compiled into run-time instances only: RtJitKey::dense)
            long long lo = DENSE_CHUNK_MIN, hi = chunk < 64 ? 64 : chunk > (long long)DENSE_CHUNK_MAX ? (long long)DENSE_CHUNK_MAX : chunk, cc = 0;
            if (c->chunk > 0) lo = hi = c->chunk;       // a claim size that was asked for is kept as it is (or the records stay item-linear)
            if (lo >= (long long)DENSE_CHUNK_MIN && hi <= (long long)DENSE_CHUNK_MAX)
                for (long long t = hi; t >= lo; t--)
                    if (t % K == 0 || K % t == 0) { cc = t; break; }
            const long long unit = cc > K ? cc : K;      // one divides the other: their least common multiple
            if (cc > 0 && unit <= (long long)DENSE_BATCH_MAX) {
                P.chunk = (uint32_t)cc;
                P.stage_dense = 1;
                c->dense_launches++;
                P.acc_batch = (uint32_t)(unit * ((long long)DENSE_BATCH_MAX / unit));
                P.acc_magic_k = K > 1 ? (uint32_t)(0x100000000ULL / (unsigned long long)K) + 1u : 0u;
                P.acc_magic_chunk = (uint32_t)(0x100000000ULL / (unsigned long long)cc) + 1u;
                const size_t n_fill = (size_t)P.total_items / (size_t)cc + 1;
                P.stage_fill = reinterpret_cast<uint32_t*>(c->stage + (size_t)P.total_items * 3u);
                P.stage_idx = reinterpret_cast<uint8_t*>(P.stage_fill + n_fill);
                HIP_TRY(hipMemsetAsync(P.stage_fill, 0, n_fill * sizeof(uint32_t), c->stream));
            }
        }
        // [0] trace items, [1] primary groups: zeroed with the work counters by rtpbr_sample(); again before every further sub-launch
        if (left != n) launch_zero(c->work_counter, 64, c->stream);
        if (split) {
            NEXT_EVENT(c->evp, c->evp_used, pa);
            NEXT_EVENT(c->evp, c->evp_used, pb);
            if (c->timed) HIP_TRY(hipEventRecord(pa, c->stream));
            if (c->jit_mod) {
                long long need = ((long long)P.total_items + 255) / 256, pg = (long long)c->n_cu * tune::PRIMARY_BLOCKS_PER_CU;
                if (int r = rt_jit_launch(c->jit_mod->primary, P, (unsigned)(pg < need ? pg : need), c->stream)) return r;
            } else
                launch_primary(P, c->kind, c->n_cu, c->stream);
            if (c->timed) HIP_TRY(hipEventRecord(pb, c->stream));
        }
        NEXT_EVENT(c->ev, c->ev_used, a);
        NEXT_EVENT(c->ev, c->ev_used, b);
        if (c->timed) HIP_TRY(hipEventRecord(a, c->stream));
        if (c->jit_mod) {
            if (int r = rt_jit_launch(c->jit_mod->trace, P, (unsigned)grid, c->stream)) return r;
        } else
            launch_trace(P, c->kind, grid, c->stream);
        if (c->timed) HIP_TRY(hipEventRecord(b, c->stream));
        if (!unstaged) launch_accumulate(P, c->n_cu, c->stream);
        c->sample_base += (uint32_t)K;
        left -= K;
    }
    return RTPBR_OK;
}

extern "C" int rtpbr_sample(rtpbr_ctx* c, int n) {
    if (!c) return fail(RTPBR_EINVAL, "null ctx");
    if (!c->have_cfg || !c->have_scene || !c->have_cam) return fail(RTPBR_ESTATE, "set_config, set_scene and set_camera first");
    if (n < 0) return fail(RTPBR_EINVAL, "n must be >= 0");
    if (int r = set_dev(c)) return r;
    // (diff_buffer: instrumented builds write their per-wave records over it)
    if (int r = rt_order_after_reads(c, W_IMAGE_BUFFER | W_RAY_BUFFER | W_DIFF_BUFFER)) return r;
    // A shading the previous call left pending (lazy shading of one-step launches) rides along only if this call is again a launch of the
    // wavefront split; otherwise it is launched now, while the previous call's work counters are still the current ones
    {
        const long long steps = (long long)n * c->cfg.steps_per_launch;
        const bool split_again = c->cfg.kernel_form == RTPBR_FORM_PERSISTENT_RAY && c->scheduler != 0 && c->src_split > 0 && steps >= 1 &&
                                 steps <= c->src_split && c->src_lazy != 0 && !c->ray_ptr_out;
        if (!split_again)
            if (int r = flush_shade(c)) return r;
        c->shade_pending_same_call = false;
    }
    Params& P = c->P;
    RtJitKey key{};
    bool want_jit = false, strict_error = false;
    derive_launch(c, &key, &want_jit, &strict_error);
    // A run-time compiled instance of THIS scene (rt_jit.hip), when no listed ahead-of-time signature serves it
    rt_jit_release(c->jit_mod);
    c->jit_mod = nullptr;
    if (want_jit) {
        RtJitModule* jm = nullptr;
        const int r = rt_jit_acquire(c, key, &jm);
        if (r == RTPBR_OK) {
            c->jit_mod = jm;
            P.box_sig = key.sig;
        } else if (c->jit == 2) {
            return r;
        }
    } else if (strict_error) {
        return fail(RTPBR_ESTATE, "option jit = 2 (strict): no run-time instance exists for this scene (needs <= 8 analytic shapes, "
                                  "or the neural shape with jit_bake, and the pool scheduler)");
    }
    if (c->precision && !c->jit_mod)
        return fail(RTPBR_ESTATE, "option precision = 1 (the tolerance flavour) exists as a run-time compiled instance only: needs hipcc and the kernel "
                                  "sources on this machine, option jit != 0, a scene of <= 8 analytic shapes (or the neural shape with jit_bake = 1, jit >= 1) "
                                  "and the pool scheduler");
    pack_objects(c, P);
    for (int i = 0; i < c->n_obj; i++)
        if (c->obj[i].type == RTPBR_SHAPE_BUNNY && !c->bunny) return fail(RTPBR_ESTATE, "bunny shape needs rtpbr_set_shape_data first");
    if (c->cfg.sky_kind == RTPBR_SKY_ENVMAP && !c->env) return fail(RTPBR_ESTATE, "sky_kind ENVMAP needs rtpbr_set_env first");
    static_assert((sizeof(Counters) + 64) % 16 == 0, "zero_next_counters / launch_zero fill 16-byte words");
    // This call's work counters (+ the claim counters behind them): the buffer whose turn it is, zeroed by the previous call's kernels
    // (or here, if that call launched none); this call's kernels zero the other one.
    const int turn = c->counters_turn;
    c->counters = c->counters_buf[turn];
    c->work_counter = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(c->counters) + sizeof(Counters));
    if (!c->counters_clean[turn]) launch_zero(c->counters, sizeof(Counters) + 64, c->stream);
    P.counters = c->counters;
    P.work_counter = c->work_counter;
    P.counters_next = c->cfg.kernel_form == RTPBR_FORM_PERSISTENT_RAY ? c->counters_buf[turn ^ 1] : nullptr;
    c->counters_clean[turn] = false;
    c->counters_clean[turn ^ 1] = false;
    c->counters_turn = turn ^ 1;
    c->ev_used = 0;
    c->evp_used = 0;
    c->timed = c->timing != 0;
    c->total1_recorded = false;
    if (c->cfg.kernel_form == RTPBR_FORM_PERSISTENT_RAY) {
        if (int r = sample_persistent(c, n)) return r;
    } else {
        if (int r = sample_complete_path(c, n)) return r;
    }
    // (the call's first event doubles as its start; an end event of its own only where work follows the last timed kernel: every event
    // is ~4 us of idle queue between two small kernels)
    if (c->timed && c->cfg.kernel_form != RTPBR_FORM_PERSISTENT_RAY) {
        HIP_TRY(hipEventRecord(c->ev_total1, c->stream));
        c->total1_recorded = true;
    }
    HIP_TRY(hipGetLastError());
    // (the first kernel of every src/ path zeroed the other buffer: persistent pool / persistent steps / src_gen)
    c->counters_clean[c->counters_turn] = n > 0 && c->cfg.kernel_form == RTPBR_FORM_PERSISTENT_RAY;
    return RTPBR_OK;
}

extern "C" int rtpbr_post_process(rtpbr_ctx* c) {
    if (!c || !c->have_cfg) return fail(RTPBR_ESTATE, "set_config first");
    if (int r = set_dev(c)) return r;
    if (int r = rt_order_after_reads(c, W_IMAGE_PIXELS | W_DIFF_BUFFER | W_DIFF_PIXELS)) return r;
    c->P.cfg = c->cfg;
    c->P.image_buffer = c->image_buffer;
    c->P.image_pixels = c->image_pixels;
    c->P.diff_buffer = c->diff_buffer;
    c->P.diff_pixels = c->diff_pixels;
    launch_post_process(c->P, c->stream);
    HIP_TRY(hipGetLastError());
    return RTPBR_OK;
}

extern "C" int rtpbr_sync(rtpbr_ctx* c) {
    if (!c) return fail(RTPBR_EINVAL, "null ctx");
    if (int r = set_dev(c)) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));      // asynchronous read-backs have landed too
    // a gather enqueued earlier has run by now: what did the communicator make of it (any rank)?
    return rt_rccl_check_async(c);
}

static int buf_ptr(rtpbr_ctx* c, int which, void** p, size_t* n) {
    if (!c || !c->have_cfg) return fail(RTPBR_ESTATE, "set_config first");
    size_t np = (size_t)c->cfg.width * c->cfg.height;
    switch (which) {
        case RTPBR_BUF_IMAGE_BUFFER: *p = c->image_buffer; *n = np * 16; return 0;
        case RTPBR_BUF_IMAGE_PIXELS: *p = c->image_pixels; *n = np * 12; return 0;
        case RTPBR_BUF_RAY_BUFFER: *p = c->ray_buffer; *n = np * sizeof(rtpbr_ray); return 0;
        case RTPBR_BUF_DIFF_BUFFER: *p = c->diff_buffer; *n = np * 8; return 0;
        case RTPBR_BUF_DIFF_PIXELS: *p = c->diff_pixels; *n = np * 4; return 0;
    }
    return fail(RTPBR_EINVAL, "unknown buffer id");
}

extern "C" int rtpbr_read_buffer(rtpbr_ctx* c, int which, void* dst, size_t nbytes) {
    void* p;
    size_t n;
    if (int r = buf_ptr(c, which, &p, &n)) return r;
    if (!dst || nbytes != n) return fail(RTPBR_EINVAL, "destination size does not match the buffer");
    if (int r = set_dev(c)) return r;
    if (which == RTPBR_BUF_RAY_BUFFER)
        if (int r = flush_shade(c)) return r;
    HIP_TRY(hipMemcpyAsync(dst, p, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTPBR_OK;
}

// canvas.set_image(image_pixels), src/main.py:64: the consumer takes the frame where it is
extern "C" int rtpbr_buffer_device_ptr(rtpbr_ctx* c, int which, void** device_ptr, size_t* nbytes) {
    void* p;
    size_t n;
    if (int r = buf_ptr(c, which, &p, &n)) return r;
    if (!device_ptr) return fail(RTPBR_EINVAL, "rtpbr_buffer_device_ptr: result pointer is required");
    if (which == RTPBR_BUF_RAY_BUFFER) {      // its holder reads ray_buffer unannounced: shade now, and with every launch from here on
        if (int r = flush_shade(c)) return r;
        c->ray_ptr_out = true;
    }
    *device_ptr = p;
    if (nbytes) *nbytes = n;
    return RTPBR_OK;
}

extern "C" int rtpbr_read_buffer_async(rtpbr_ctx* c, int which, void* dst, size_t nbytes, int* ticket) {
    void* p;
    size_t n;
    if (int r = buf_ptr(c, which, &p, &n)) return r;
    if (which == RTPBR_BUF_RAY_BUFFER)
        if (int r = flush_shade(c)) return r;
    if (!dst || !ticket || nbytes != n) return fail(RTPBR_EINVAL, "rtpbr_read_buffer_async: destination, ticket and the buffer's exact size are required");
    bool pinned = false;
    for (size_t i = 0; i < c->host_blocks.size() && !pinned; i++) {
        const char* b = (const char*)c->host_blocks[i];
        pinned = (const char*)dst >= b && (const char*)dst + n <= b + c->host_sizes[i];
    }
    if (!pinned) return fail(RTPBR_EINVAL, "rtpbr_read_buffer_async: the destination must lie inside a block of rtpbr_host_alloc (page-locked memory)");
    if (int r = set_dev(c)) return r;
    if (!c->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_read_ready, hipEventDisableTiming));
        for (hipEvent_t& e : c->ev_read_done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const int t = c->read_issued, slot = t & 7;
    if (t >= 8) HIP_TRY(hipEventSynchronize(c->ev_read_done[slot]));      // ticket t - 8 gives up its slot: its copy must have landed
    HIP_TRY(hipEventRecord(c->ev_read_ready, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_read_ready, 0));
    HIP_TRY(hipMemcpyAsync(dst, p, n, hipMemcpyDeviceToHost, c->copy_stream));
    HIP_TRY(hipEventRecord(c->ev_read_done[slot], c->copy_stream));
    c->read_pending[which] = t;
    c->read_issued = t + 1;
    *ticket = t;
    return RTPBR_OK;
}

extern "C" int rtpbr_read_wait(rtpbr_ctx* c, int ticket) {
    if (!c) return fail(RTPBR_EINVAL, "null ctx");
    if (ticket < 0 || ticket >= c->read_issued) return fail(RTPBR_EINVAL, "rtpbr_read_wait: no such ticket");
    if (ticket + 8 < c->read_issued) return RTPBR_OK;      // its slot was handed on, after its copy had landed
    if (int r = set_dev(c)) return r;
    HIP_TRY(hipEventSynchronize(c->ev_read_done[ticket & 7]));
    return RTPBR_OK;
}

extern "C" int rtpbr_host_alloc(rtpbr_ctx* c, size_t nbytes, void** ptr) {
    if (!c || !ptr || nbytes == 0) return fail(RTPBR_EINVAL, "rtpbr_host_alloc: context, size and result pointer are required");
    if (int r = set_dev(c)) return r;
    void* p = nullptr;
    HIP_TRY(hipHostMalloc(&p, nbytes, hipHostMallocDefault));
    c->host_blocks.push_back(p);
    c->host_sizes.push_back(nbytes);
    *ptr = p;
    return RTPBR_OK;
}

extern "C" int rtpbr_host_free(rtpbr_ctx* c, void* ptr) {
    if (!c || !ptr) return fail(RTPBR_EINVAL, "rtpbr_host_free: context and pointer are required");
    for (size_t i = 0; i < c->host_blocks.size(); i++)
        if (c->host_blocks[i] == ptr) {
            if (int r = set_dev(c)) return r;
            HIP_TRY(hipStreamSynchronize(c->stream));      // (a copy into it may be in flight)
            if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));
            c->host_blocks.erase(c->host_blocks.begin() + (long)i);
            c->host_sizes.erase(c->host_sizes.begin() + (long)i);
            HIP_TRY(hipHostFree(ptr));
            return RTPBR_OK;
        }
    return fail(RTPBR_EINVAL, "rtpbr_host_free: not a block of this context");
}

extern "C" int rtpbr_write_buffer(rtpbr_ctx* c, int which, const void* src, size_t nbytes) {
    void* p;
    size_t n;
    if (int r = buf_ptr(c, which, &p, &n)) return r;
    if (!src || nbytes != n) return fail(RTPBR_EINVAL, "source size does not match the buffer");
    if (int r = set_dev(c)) return r;
    if (int r = flush_shade(c)) return r;
    if (int r = rt_order_after_reads(c, 1u << which)) return r;
    HIP_TRY(hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return RTPBR_OK;
}

extern "C" int rtpbr_packed_bytes(rtpbr_ctx* c, size_t* nbytes) {
    if (!c || !nbytes || !c->have_cfg) return fail(RTPBR_ESTATE, "set_config first");
    *nbytes = (size_t)c->P.np * sizeof(float4);
    return RTPBR_OK;
}

extern "C" int rtpbr_pack_tiles(rtpbr_ctx* c, void* device_dst) {
    if (!c || !device_dst || !c->have_cfg) return fail(RTPBR_EINVAL, "bad pack_tiles arguments");
    if (int r = set_dev(c)) return r;
    c->P.cfg = c->cfg;
    c->P.image_buffer = c->image_buffer;
    launch_pack(c->P, (float4*)device_dst, c->stream);
    HIP_TRY(hipGetLastError());
    return RTPBR_OK;
}

extern "C" int rtpbr_unpack_tiles(rtpbr_ctx* c, const void* device_src, int src_rank) {
    if (!c || !device_src || !c->have_cfg) return fail(RTPBR_EINVAL, "bad unpack_tiles arguments");
    if (src_rank < 0 || src_rank >= c->world) return fail(RTPBR_EINVAL, "src_rank out of range");
    if (int r = set_dev(c)) return r;
    if (int r = rt_order_after_reads(c, W_IMAGE_BUFFER)) return r;
    Params P = c->P;
    P.cfg = c->cfg;
    P.image_buffer = c->image_buffer;
    P.rank = src_rank;
    launch_unpack(P, (const float4*)device_src, c->stream);
    HIP_TRY(hipGetLastError());
    return RTPBR_OK;
}

// the kernels add the six work counters up in 64 shards (rt_types.hpp Counters::shard): fold them into the fields
static void fold_shards(Counters& h) {
    unsigned long long* f[6] = {&h.march_steps, &h.raycasts, &h.hits, &h.sky_lookups, &h.samples, &h.deposits};
    for (int s = 0; s < 64; s++)
        for (int k = 0; k < 6; k++) *f[k] += h.shard[s][k];
}

extern "C" int rtpbr_get_counters(rtpbr_ctx* c, rtpbr_counters* out) {
    if (!c || !out) return fail(RTPBR_EINVAL, "null argument");
    if (int r = set_dev(c)) return r;
    if (int r = flush_shade(c)) return r;
    Counters h;
    HIP_TRY(hipMemcpyAsync(&h, c->counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    fold_shards(h);
    out->samples = h.samples;
    out->raycasts = h.raycasts;
    out->march_steps = h.march_steps;
    out->hits = h.hits;
    out->sky_lookups = h.sky_lookups;
    out->deposits = h.deposits;
    return RTPBR_OK;
}

// Named counters of the last rtpbr_sample() call: the six of rtpbr_counters plus "mlp_wave_evals" (32-ray half passes of the
// wave-cooperative neural-SDF MLP) and "mlp_lane_evals" (ray evaluations those passes were needed for).
extern "C" int rtpbr_get_counter(rtpbr_ctx* c, const char* name, unsigned long long* out) {
    if (!c || !name || !out) return fail(RTPBR_EINVAL, "null argument");
    if (int r = set_dev(c)) return r;
    if (int r = flush_shade(c)) return r;
    Counters h;
    HIP_TRY(hipMemcpyAsync(&h, c->counters, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    fold_shards(h);
    if (!strcmp(name, "samples")) *out = h.samples;
    else if (!strcmp(name, "raycasts")) *out = h.raycasts;
    else if (!strcmp(name, "march_steps")) *out = h.march_steps;
    else if (!strcmp(name, "hits")) *out = h.hits;
    else if (!strcmp(name, "sky_lookups")) *out = h.sky_lookups;
    else if (!strcmp(name, "deposits")) *out = h.deposits;
    else if (!strcmp(name, "mlp_wave_evals")) *out = h.mlp_wave_evals;
    else if (!strcmp(name, "mlp_lane_evals")) *out = h.mlp_lane_evals;
    else if (!strncmp(name, "dbg", 3) && name[3] >= '0' && name[3] <= '9' && !name[4]) *out = h.dbg[name[3] - '0'];
    else if (!strncmp(name, "dbg", 3) && name[3] >= 'a' && name[3] <= 'v' && !name[4]) *out = h.dbg[10 + name[3] - 'a'];
    else if (!strcmp(name, "dense_launches")) *out = c->dense_launches;      // complete-path launches of the last rtpbr_sample() that staged densely (option stage_dense)
    else if (!strncmp(name, "plan_ge:", 8)) {
        // pixels whose recorded cost was at least <n> march steps when the current plan was made (whole buckets)
        *out = 0;
        if (c->plan && c->order_valid) {
            const unsigned long long thr = strtoull(name + 8, nullptr, 10);
            PlanBuf pb;
            HIP_TRY(hipMemcpy(&pb, c->plan, sizeof pb, hipMemcpyDeviceToHost));
            for (unsigned b = 1; b < 256; b++) {
                const unsigned e = (b - 1u) >> 3, m = (b - 1u) & 7u;
                const unsigned long long fl = e >= 3u ? (unsigned long long)(8u + m) << (e - 3u) : (8u + m) >> (3u - e);
                if (fl >= thr) *out += pb.hist[b];
            }
        }
    }
    else if (!strcmp(name, "plan_heavy") || !strcmp(name, "plan_total")) {
        // the src/ pool kernel's current plan: pixels walked by heavy waves / march steps on record when it was made
        *out = 0;
        if (c->plan && c->order_valid) {
            PlanBuf pb;
            HIP_TRY(hipMemcpy(&pb, c->plan, sizeof pb, hipMemcpyDeviceToHost));
            *out = name[5] == 'h' ? pb.n_heavy : pb.total;
        }
    }
    else if (!strcmp(name, "jit_active")) *out = c->jit_mod ? 1 : 0;      // the last sample() ran a run-time compiled instance
    else return fail(RTPBR_EINVAL, "unknown counter %s", name);
    return RTPBR_OK;
}

extern "C" int rtpbr_last_sample_ms(rtpbr_ctx* c, float* trace_ms, float* total_ms, int* launches) {
    if (!c) return fail(RTPBR_EINVAL, "null ctx");
    if (!c->timed) return fail(RTPBR_ESTATE, "no timed rtpbr_sample() call yet (option timing = 0 records no events)");
    if (int r = set_dev(c)) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    float tr = 0.0f;
    for (int i = 0; i + 1 < c->ev_used; i += 2) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]));
        tr += ms;
    }
    float tot = 0.0f;
    if (c->ev_used >= 2) {
        hipEvent_t first = c->evp_used > 0 ? c->evp[0] : c->ev[0];
        HIP_TRY(hipEventElapsedTime(&tot, first, c->total1_recorded ? c->ev_total1 : c->ev[c->ev_used - 1]));
    }
    if (trace_ms) *trace_ms = tr;
    if (total_ms) *total_ms = tot;
    if (launches) *launches = c->ev_used / 2;
    return RTPBR_OK;
}

extern "C" int rtpbr_last_primary_ms(rtpbr_ctx* c, float* primary_ms, int* launches) {
    if (!c) return fail(RTPBR_EINVAL, "null ctx");
    if (!c->timed) return fail(RTPBR_ESTATE, "no timed rtpbr_sample() call yet (option timing = 0 records no events)");
    if (int r = set_dev(c)) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    float pr = 0.0f;
    for (int i = 0; i + 1 < c->evp_used; i += 2) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->evp[i], c->evp[i + 1]));
        pr += ms;
    }
    if (primary_ms) *primary_ms = pr;
    if (launches) *launches = c->evp_used / 2;
    return RTPBR_OK;
}

extern "C" int rtpbr_get_stream(rtpbr_ctx* c, void** stream) {
    if (!c || !stream) return fail(RTPBR_EINVAL, "null argument");
    *stream = (void*)c->stream;
    return RTPBR_OK;
}

extern "C" int rtpbr_set_option(rtpbr_ctx* c, const char* key, long long value) {
    if (!c || !key) return fail(RTPBR_EINVAL, "null argument");
    if (int r = flush_shade(c)) return r;
    if (!strcmp(key, "staging_bytes")) {
        if (value < (1 << 20)) return fail(RTPBR_EINVAL, "staging_bytes must be >= 1 MiB");
        c->staging_bytes = value;
    } else if (!strcmp(key, "wait_lanes")) {
        if (value < 1 || value > 64) return fail(RTPBR_EINVAL, "wait_lanes must be 1..64");
        c->wait_lanes = (int)value;
    } else if (!strcmp(key, "shade_lanes")) {
        if (value < 1 || value > 64) return fail(RTPBR_EINVAL, "shade_lanes must be 1..64");
        c->shade_lanes = (int)value;
    } else if (!strcmp(key, "jit_waves")) {
        if (value < 0 || value > 8) return fail(RTPBR_EINVAL, "jit_waves must be 0 (default) .. 8");
        c->jit_waves = (int)value;
    } else if (!strcmp(key, "chunk")) {
        // every wave makes one claim past the end, so the 32-bit work counter overshoots by waves x chunk: the margin kept
        // free below 2^32 (work_margin) covers 32 waves per CU x 8192
        if (value < 0 || value > 8192) return fail(RTPBR_EINVAL, "chunk must be 0 (automatic) .. 8192");
        c->chunk = (int)value;
    } else if (!strcmp(key, "sparse_lanes")) {
        if (value < 0 || value > 64) return fail(RTPBR_EINVAL, "sparse_lanes must be 0 (never) .. 64");
        c->sparse_lanes = (int)value;
    } else if (!strcmp(key, "src_plan")) {
        if (value < 0 || value > 1) return fail(RTPBR_EINVAL, "src_plan must be 0 or 1");
        c->src_plan = (int)value;
        c->order_valid = false;
        c->cost_steps = 0;
    } else if (!strcmp(key, "plan_interval")) {
        if (value < 1 || value > (1 << 20)) return fail(RTPBR_EINVAL, "plan_interval must be 1 .. 2^20 bounce-steps");
        c->plan_interval = (int)value;
    } else if (!strcmp(key, "heavy_own")) {
        if (value < 0 || value > 128) return fail(RTPBR_EINVAL, "heavy_own must be 0 (no heavy waves) .. 128");
        c->heavy_own = (int)value;
        c->order_valid = false;       // the plan's share of heavy pixels was sized for the old value
        c->cost_steps = 0;
    } else if (!strcmp(key, "src_chain")) {
        if (value < 0 || value > 2) return fail(RTPBR_EINVAL, "src_chain must be 0 (never), 1 (when the device has room beside the pool kernel) or 2 (always: tests)");
        c->src_chain = (int)value;
        c->order_valid = false;       // the plan carries the chain set
        c->cost_steps = 0;
        c->plan_chain_waves = 0;
        c->plan_rb_pending = false;
    } else if (!strcmp(key, "chain_np_max")) {
        if (value < 0) return fail(RTPBR_EINVAL, "chain_np_max must be >= 0");
        c->chain_np_max = value;
    } else if (!strcmp(key, "chain_waves")) {
        if (value < 1 || value > 2048) return fail(RTPBR_EINVAL, "chain_waves must be 1 .. 2048");
        c->chain_waves = (int)value;
        c->order_valid = false;
        c->cost_steps = 0;
    } else if (!strcmp(key, "src_split")) {
        if (value < 0 || value > 256) return fail(RTPBR_EINVAL, "src_split must be 0 (never) .. 256 (bounce-steps per launch up to which the wavefront split runs)");
        c->src_split = (int)value;
    } else if (!strcmp(key, "env_packed")) {
        if (value < 0 || value > 1) return fail(RTPBR_EINVAL, "env_packed must be 0 (float4 texels) or 1 (RGBA8 texels + 256-entry table: 8-bit sources, same values)");
        c->env_packed = (int)value;
    } else if (!strcmp(key, "src_lazy")) {
        if (value < 0 || value > 1) return fail(RTPBR_EINVAL, "src_lazy must be 0 (every one-step launch shades) or 1 (the shading rides with the next launch's gen pass)");
        c->src_lazy = (int)value;
    } else if (!strcmp(key, "split_head")) {
        if (value < -1 || value > 1) return fail(RTPBR_EINVAL, "split_head must be -1 (automatic: small frames), 0 (the heavy head fills the first groups) or 1 (interleaved: one head entry per group)");
        c->split_head = (int)value;
    } else if (!strcmp(key, "split_wait")) {
        if (value < 1 || value > 64) return fail(RTPBR_EINVAL, "split_wait must be 1..64");
        c->split_wait = (int)value;
    } else if (!strcmp(key, "src_track")) {
        if (value < 0 || value > 2) return fail(RTPBR_EINVAL, "src_track must be 0 (never), 1 (one-object bounds only) or 2 (one- and two-object bounds)");
        c->src_track = (int)value;
    } else if (!strcmp(key, "src_op")) {
        if (value < 0 || value > 7) return fail(RTPBR_EINVAL, "src_op must be 0 .. 7 (bit 0: object-parallel evaluation for sparse waves in the split march and chain kernels, bit 1: in the fused pool kernel, bit 2: the per-lane lean loop for lanes that track different objects)");
        c->src_op = (int)value;
    } else if (!strcmp(key, "age_weights")) {
        // one hex digit per residency slot, oldest first (0x88888 = equal shares); 0 switches the weighting off
        if (value < 0 || value > 0xffffffffLL) return fail(RTPBR_EINVAL, "age_weights must be 0 (off) or up to 8 hex digits, one per residency slot");
        c->age_on = value != 0 ? 2 : 0;
        int nd = 0;
        for (long long v = value; v; v >>= 4) nd++;
        for (int k = 0; k < 8; k++) c->age_w[k] = k < nd ? (int)((value >> (4 * (nd - 1 - k))) & 15) : 8;
        for (int k = 0; k < 8; k++)
            if (c->age_w[k] == 0) c->age_w[k] = 1;
    } else if (!strcmp(key, "age_tune")) {
        if (value < 0 || value > 1) return fail(RTPBR_EINVAL, "age_tune must be 0 (equal shares) or 1 (self-tuned age-weighted shares)");
        c->age_on = (int)value;
    } else if (!strcmp(key, "tiny_waves")) {
        if (value < 0 || value > 65535) return fail(RTPBR_EINVAL, "tiny_waves must be 0 .. 65535");
        c->tiny_waves = (int)value;
    } else if (!strcmp(key, "tiny_own")) {
        if (value < 0 || value > 128) return fail(RTPBR_EINVAL, "tiny_own must be 0 (no small heavy waves) .. 128");
        c->tiny_own = (int)value;
    } else if (!strcmp(key, "leave_x8")) {
        if (value < 1 || value > 4096) return fail(RTPBR_EINVAL, "leave_x8 must be 1 .. 4096 (eighths of a march iteration)");
        c->leave_x8 = (int)value;
    } else if (!strcmp(key, "heavy_prio")) {
        if (value < 0 || value > 1) return fail(RTPBR_EINVAL, "heavy_prio must be 0 or 1");
        c->heavy_prio = (int)value;
    } else if (!strcmp(key, "heavy_mean_x16")) {
        if (value < 0 || value > (1 << 20)) return fail(RTPBR_EINVAL, "heavy_mean_x16 must be 0 .. 2^20");
        c->heavy_mean_x16 = (int)value;
    } else if (!strcmp(key, "heavy_bulk_x16")) {
        if (value < 0 || value > (1 << 20)) return fail(RTPBR_EINVAL, "heavy_bulk_x16 must be 0 .. 2^20");
        c->heavy_bulk_x16 = (int)value;
    } else if (!strcmp(key, "grid_blocks")) {
        if (value < 0 || value > 65535) return fail(RTPBR_EINVAL, "grid_blocks must be 0 (automatic) .. 65535");
        c->grid_blocks = (int)value;
    } else if (!strcmp(key, "residency")) {
        if (value < 1 || value > 256 || (value & (value - 1))) return fail(RTPBR_EINVAL, "residency must be a power of two, 1..256");
        c->residency = (int)value;
    } else if (!strcmp(key, "ready_low")) {
        if (value < 0 || value > 63) return fail(RTPBR_EINVAL, "ready_low must be 0..63");
        c->ready_low = (int)value;
    } else if (!strcmp(key, "refill_lanes")) {
        if (value < 1 || value > 64) return fail(RTPBR_EINVAL, "refill_lanes must be 1..64");
        c->refill_lanes = (int)value;
    } else if (!strcmp(key, "timing")) {
        // 1 (default): rtpbr_sample() brackets its kernels with events (rtpbr_last_sample_ms / rtpbr_last_primary_ms); 0: none — an event
        // is a barrier packet with a completion signal, and four of them cost a launch of a few hundred microseconds a fifth of its time
        if (value < 0 || value > 1) return fail(RTPBR_EINVAL, "timing must be 0 or 1");
        c->timing = (int)value;
    } else if (!strcmp(key, "stage_dense")) {
        if (value < 0 || value > 1) return fail(RTPBR_EINVAL, "stage_dense must be 0 (item-linear staging) or 1 (records appended per claim in completion order)");
        c->stage_dense = (int)value;
    } else if (!strcmp(key, "primary_split")) {
        if (value < 0 || value > 2) return fail(RTPBR_EINVAL, "primary_split must be 0 (never), 1 (large launches) or 2 (always)");
        c->primary_split = (int)value;
    } else if (!strcmp(key, "drain_lanes")) {
        if (value < 0 || value > 64) return fail(RTPBR_EINVAL, "drain_lanes must be 0..64");
        c->drain_lanes = (int)value;
    } else if (!strcmp(key, "primary_lean")) {
        if (value < 0 || value > 1) return fail(RTPBR_EINVAL, "primary_lean must be 0 or 1");
        c->primary_lean = (int)value;
    } else if (!strcmp(key, "lazy_sqrt")) {
        if (value < 0 || value > 1) return fail(RTPBR_EINVAL, "lazy_sqrt must be 0 or 1");
        c->lazy_sqrt = (int)value;
    } else if (!strcmp(key, "jit")) {
        if (value < -1 || value > 2)
            return fail(RTPBR_EINVAL, "jit must be -1 (when no ahead-of-time specialisation fits), 0 (never), 1 (always, falling back to the "
                                      "ahead-of-time instance if compilation is impossible) or 2 (always, an error otherwise)");
        c->jit = (int)value;
    } else if (!strcmp(key, "precision")) {
        if (value < 0 || value > 1) return fail(RTPBR_EINVAL, "precision must be 0 (exactly rounded, the default) or 1 (tolerance flavour: hardware sqrt / rcp / sin / exp, contraction)");
        c->precision = (int)value;
    } else if (!strcmp(key, "jit_bake")) {
        if (value < 0 || value > 2) return fail(RTPBR_EINVAL, "jit_bake must be 0, 1 (scene + configuration) or 2 (+ the camera frame: fixed-camera offline renders)");
        c->jit_bake = (int)value;
    } else if (!strcmp(key, "specialize")) {
        if (value < 0 || value > 1) return fail(RTPBR_EINVAL, "specialize must be 0 or 1");
        c->specialize = (int)value;
    } else if (!strcmp(key, "mlp_mfma")) {
        if (value < 0 || value > 1) return fail(RTPBR_EINVAL, "mlp_mfma must be 0 or 1");
        c->mlp_mfma = (int)value;
    } else if (!strcmp(key, "mlp_lanes")) {
        if (value < 1 || value > 64) return fail(RTPBR_EINVAL, "mlp_lanes must be 1..64");
        c->mlp_lanes = (int)value;
    } else if (!strcmp(key, "mlp_full")) {
        if (value < 1 || value > 65) return fail(RTPBR_EINVAL, "mlp_full must be 1..65 (65 = never compute both halves at once)");
        c->mlp_full = (int)value;
    } else if (!strcmp(key, "swap_lanes")) {
        if (value < 0 || value > 64) return fail(RTPBR_EINVAL, "swap_lanes must be 0 (automatic: 8 in the complete-path pool kernel, 12 in the src/ one) .. 64");
        c->swap_lanes = (int)value;
    } else if (!strcmp(key, "scheduler")) {
        if (value < -1 || value > 1) return fail(RTPBR_EINVAL, "scheduler must be -1 (auto), 0 or 1");
        c->scheduler = (int)value;
    } else if (!strcmp(key, "waves_per_cu")) {
        if (value < 0 || value > 32) return fail(RTPBR_EINVAL, "waves_per_cu must be 0..32");
        c->waves_per_cu = (int)value;
    } else if (!strcmp(key, "reserve_spp")) {
        // allocate the staging of a `value`-spp rtpbr_sample() call now (complete-path form)
        if (!c->have_cfg || c->P.np <= 0) return fail(RTPBR_ESTATE, "set_config (and set_tiles) first");
        if (value < 1) return fail(RTPBR_EINVAL, "reserve_spp must be >= 1");
        if (int r = set_dev(c)) return r;
        const bool split_ok = c->primary_split && c->scheduler != 0 && !(c->have_scene && (c->kind == KIND_BUNNY || c->kind == KIND_MIXED));
        const bool unstaged = c->precision != 0 && c->scheduler != 0;      // the tolerance flavour has no staging (rtpbr_sample)
        long long per_spp = (long long)c->P.np * (long long)((unstaged ? 0 : STAGE_ITEM_BYTES) + (split_ok ? PRIMARY_REC_BYTES : 0));
        if (per_spp < 1) per_spp = 1;
        long long kmax = c->staging_bytes / per_spp;
        long long k32 = (0xFFFFFFFFLL - work_margin(c)) / (long long)c->P.np;
        if (k32 < 1) k32 = 1;
        if (kmax > k32) kmax = k32;
        if (kmax < 1) kmax = 1;
        const long long K = value < kmax ? value : kmax;
        if (int r = ensure_staging(c, (size_t)c->P.np * (size_t)K, split_ok, !unstaged))
            return r == RTPBR_ENOMEM ? fail(RTPBR_ENOMEM, "reserve_spp: no device memory for the staging of that many samples per pixel") : r;
    } else if (!strcmp(key, "sample_base")) {
        c->sample_base = (uint32_t)value;
    } else {
        return fail(RTPBR_EINVAL, "unknown option %s", key);
    }
    return RTPBR_OK;
}

// test hook: exact math functions evaluated on the device (not part of the drop-in surface)
extern "C" int rtpbr_test_math(rtpbr_ctx* c, int op, const float* a, const float* b, float* out, float* out2, int n) {
    if (!c || !a || !out || n <= 0) return fail(RTPBR_EINVAL, "bad test_math arguments");
    if (int r = set_dev(c)) return r;
    float *da = nullptr, *db = nullptr, *dout = nullptr, *dout2 = nullptr;
    size_t nb = (size_t)n * sizeof(float);
    HIP_TRY(hipMalloc(&da, nb));
    HIP_TRY(hipMalloc(&dout, nb));
    HIP_TRY(hipMalloc(&dout2, nb));
    HIP_TRY(hipMemcpy(da, a, nb, hipMemcpyHostToDevice));
    if (b) {
        HIP_TRY(hipMalloc(&db, nb));
        HIP_TRY(hipMemcpy(db, b, nb, hipMemcpyHostToDevice));
    }
    launch_math_probe(op, da, db, dout, dout2, n, c->stream);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, dout, nb, hipMemcpyDeviceToHost));
    if (out2) HIP_TRY(hipMemcpy(out2, dout2, nb, hipMemcpyDeviceToHost));
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    (void)hipFree(dout2);
    return RTPBR_OK;
}

// test hook: exhaustive check of the device sqrt_ against IEEE sqrt; *mismatches must come back 0
extern "C" int rtpbr_test_sqrt_exhaustive(rtpbr_ctx* c, unsigned long long* mismatches) {
    if (!c || !mismatches) return fail(RTPBR_EINVAL, "null argument");
    if (int r = set_dev(c)) return r;
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof *d));
    HIP_TRY(hipMemset(d, 0, sizeof *d));
    launch_sqrt_exhaustive(d, c->stream);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(mismatches, d, sizeof *d, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return RTPBR_OK;
}
