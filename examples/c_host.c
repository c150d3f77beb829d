/* c_host.c — the Cornell Box rendered through the C ABI from a plain C host (no Python, no PyTorch): what a maintainer of
 * the reference — or any C / Go (cgo) / Java (JNI) host — writes against include/rtpbr.h.  Mirrors the main loop of
 * examples/cornell_box/cornell_box_v3/main.py (scene table scene.py:6-27, constants config.py:3-25): refresh, N samples per
 * pixel, tone map, read the image back; writes a binary PPM and prints a checksum of image_pixels.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_host.c -Lraytracingpbr_amd/csrc -lrtpbr_hip -Wl,-rpath,$PWD/raytracingpbr_amd/csrc -lm -o c_host
 *   ./c_host 256 256 16 out.ppm
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rtpbr.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != RTPBR_OK) {                                                        \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, rtpbr_last_error());  \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

/* cornell_box_v3/scene.py:6-27 — type, {position, rotation (degrees), scale, matrix (filled by the library)},
 * {albedo, emission, roughness, metallic, transmission, ior}; unit scale, the script multiplies by 10 (scale10 below) */
static const rtpbr_object CORNELL[8] = {
    {RTPBR_SHAPE_BOX, {{0, 0, -1}, {0, 0, 0}, {1, 1, 0.2f}, {0}}, {{0.4f, 0.4f, 0.4f}, {1, 1, 1}, 1, 0, 0, 1.53f}},
    {RTPBR_SHAPE_BOX, {{0, 1, 0}, {90, 0, 0}, {1, 1, 0.2f}, {0}}, {{0.4f, 0.4f, 0.4f}, {1, 1, 1}, 1, 0, 0, 1.53f}},
    {RTPBR_SHAPE_BOX, {{0, -1, 0}, {90, 0, 0}, {1, 1, 0.2f}, {0}}, {{0.4f, 0.4f, 0.4f}, {1, 1, 1}, 1, 0, 0, 1.53f}},
    {RTPBR_SHAPE_BOX, {{-1, 0, 0}, {0, 90, 0}, {1, 1, 0.2f}, {0}}, {{0.5f, 0, 0}, {1, 1, 1}, 1, 0, 0, 1.53f}},
    {RTPBR_SHAPE_BOX, {{1, 0, 0}, {0, 90, 0}, {1, 1, 0.2f}, {0}}, {{0, 0.5f, 0}, {1, 1, 1}, 1, 0, 0, 1.53f}},
    {RTPBR_SHAPE_BOX, {{-0.275f, -0.3f, -0.2f}, {0, -253, 0}, {0.25f, 0.5f, 0.25f}, {0}}, {{0.4f, 0.4f, 0.4f}, {1, 1, 1}, 1, 0, 0, 1.53f}},
    {RTPBR_SHAPE_BOX, {{0.275f, -0.55f, 0.2f}, {0, -197, 0}, {0.25f, 0.25f, 0.25f}, {0}}, {{0.4f, 0.4f, 0.4f}, {1, 1, 1}, 1, 0, 0, 1.53f}},
    {RTPBR_SHAPE_BOX, {{0, 0.809f, 0}, {90, 0, 0}, {0.2f, 0.2f, 0.01f}, {0}}, {{1, 1, 1}, {100, 100, 100}, 1, 0, 0, 1}},
};

int main(int argc, char** argv) {
    const int W = argc > 1 ? atoi(argv[1]) : 256, H = argc > 2 ? atoi(argv[2]) : 256, SPP = argc > 3 ? atoi(argv[3]) : 16;
    const char* out = argc > 4 ? argv[4] : "cornell_c_host.ppm";
    if (W <= 0 || H <= 0 || SPP <= 0) return 2;

    rtpbr_config c;                                    /* cornell_box_v3/config.py:3-25 (every knob: SURVEY.md Appendix B) */
    memset(&c, 0, sizeof c);
    c.width = W; c.height = H; c.seed = 0;
    c.kernel_form = RTPBR_FORM_COMPLETE_PATH;
    c.max_raymarch = 512; c.max_raytrace = 4;
    c.march_kind = RTPBR_MARCH_RELAXED;
    c.min_dis = 0.05f; c.max_dis = 2000.0f;
    c.hit_eps = (float)(0.5 * (1.0 / (double)(W > H ? W : H)));  /* PIXEL_RADIUS = 0.5 * min(1/W, 1/H): Python doubles, stored as f32 */
    c.omega0 = 1.6f; c.omega_guard = 1; c.omega_fb_a = 1.0f; c.omega_fb_b = 0.0f;
    c.box_round = 0.01f; c.nearest_init = 0;
    c.normal_h = (float)(0.5773 * 0.005); c.normal_space = RTPBR_NORMAL_WORLD;
    c.rr_kind = 0; c.light_quality = 128.0f; c.quality_per_sample = 0.8f;
    c.surface_kind = 0; c.fresnel_kind = 0; c.fresnel_roughness_mix = 1;
    c.below_horizon = 0; c.origin_mode = 0; c.env_ior = 1.000277f;
    c.sky_kind = 0; c.primary_miss = 0;
    c.vis_lo = 0.000001f; c.vis_hi = 3.4028234663852886e38f;
    c.camera_kind = 0;
    c.tonemap_order = 0; c.aces_truncated = 0; c.exposure = 1.0f; c.gamma = 2.2f;
    c.frame = 0; c.steps_per_launch = 1; c.adaptive_sampling = 0; c.noise_threshold = 1e-4f; c.anim_bob = 0.0f;

    const rtpbr_camera cam = {{0, 0, 35.0f}, {0, 0, 1.0f}, {0, 1, 0}, 35.0f, (float)W / (float)H, 0.01f, 4.0f};

    rtpbr_ctx* ctx = NULL;
    CHECK(rtpbr_create(0, &ctx));
    CHECK(rtpbr_set_config(ctx, &c));
    CHECK(rtpbr_set_scene(ctx, CORNELL, 8, 1 /* scale10: the script's "* 10" */));
    CHECK(rtpbr_set_camera(ctx, &cam));
    CHECK(rtpbr_refresh(ctx));                         /* renderer.refresh()      */
    CHECK(rtpbr_sample(ctx, SPP));                     /* SPP x the sample kernel */
    CHECK(rtpbr_post_process(ctx));                    /* tone map                */

    const size_t n = (size_t)W * (size_t)H * 3;
    float* px = (float*)malloc(n * sizeof(float));
    if (!px) return 3;
    CHECK(rtpbr_read_buffer(ctx, RTPBR_BUF_IMAGE_PIXELS, px, n * sizeof(float)));
    rtpbr_counters k;
    CHECK(rtpbr_get_counters(ctx, &k));

    /* FNV-1a over the float bit patterns: the Python host must get the same number for the same call sequence */
    unsigned long long h = 1469598103934665603ull;
    const unsigned char* b = (const unsigned char*)px;
    for (size_t i = 0; i < n * sizeof(float); i++) h = (h ^ b[i]) * 1099511628211ull;

    FILE* f = fopen(out, "wb");                        /* field layout [x][y], y from the bottom -> image rows from the top */
    if (f) {
        fprintf(f, "P6\n%d %d\n255\n", W, H);
        for (int row = 0; row < H; row++)
            for (int x = 0; x < W; x++) {
                const float* p = px + ((size_t)x * (size_t)H + (size_t)(H - 1 - row)) * 3;
                unsigned char rgb[3];
                for (int ch = 0; ch < 3; ch++) {
                    float v = p[ch] < 0.0f ? 0.0f : (p[ch] > 1.0f ? 1.0f : p[ch]);
                    rgb[ch] = (unsigned char)(v * 255.0f + 0.5f);
                }
                fwrite(rgb, 1, 3, f);
            }
        fclose(f);
    }
    printf("backend %s  %dx%d x %d spp  samples %llu  raycasts %llu  march steps %llu  image_pixels fnv1a %016llx\n", rtpbr_backend(), W, H,
           SPP, (unsigned long long)k.samples, (unsigned long long)k.raycasts, (unsigned long long)k.march_steps, h);
    free(px);
    CHECK(rtpbr_destroy(ctx));
    return 0;
}
