/* Sanitizer driver of the CPU oracle (test infrastructure; built by `make san` with -fsanitize=address,undefined and
 * run by tests/test_oracle_sanitized.py).  Reads one case written by the test — rtpbr_config, n, scale10 flag, n x rtpbr_object,
 * rtpbr_camera, env width/height + RGB8 texels (0 x 0 = none), exposure, gamma, shape-data count + floats, tile
 * layout, rounds, samples per round — renders it through the oracle's public entry points and writes image_buffer,
 * image_pixels and the counters to the output file, so the test can also compare the instrumented -O1 build with the
 * -O2 checker build bit for bit. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rt_oracle.h"

#define CHECK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s failed: %d %s\n", #x, r_, rto_last_error()); return 2; } } while (0)

static int rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n ? 0 : -1; }

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: rt_oracle_san case.bin out.bin\n"); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    rtpbr_config cfg; rtpbr_camera cam; int n = 0, scale10 = 0, ew = 0, eh = 0, nshape = 0, tiles[4], rounds = 0, spp = 0;
    float env_exposure = 1.0f, env_gamma = 1.0f;
    if (rd(f, &cfg, sizeof cfg) || rd(f, &n, 4) || rd(f, &scale10, 4) || n < 1 || n > 64) return 1;
    rtpbr_object* objs = malloc(sizeof(rtpbr_object) * (size_t)n);
    if (rd(f, objs, sizeof(rtpbr_object) * (size_t)n) || rd(f, &cam, sizeof cam) || rd(f, &ew, 4) || rd(f, &eh, 4)) return 1;
    unsigned char* env = NULL;
    if (ew > 0 && eh > 0) {
        env = malloc((size_t)ew * (size_t)eh * 3);
        if (rd(f, env, (size_t)ew * (size_t)eh * 3)) return 1;
    }
    if (rd(f, &env_exposure, 4) || rd(f, &env_gamma, 4) || rd(f, &nshape, 4) || nshape < 0 || nshape > 4096) return 1;
    float* shape = nshape ? malloc(sizeof(float) * (size_t)nshape) : NULL;
    if (nshape && rd(f, shape, sizeof(float) * (size_t)nshape)) return 1;
    if (rd(f, tiles, sizeof tiles) || rd(f, &rounds, 4) || rd(f, &spp, 4)) return 1;
    fclose(f);

    struct rto_ctx* c = NULL;
    CHECK(rto_create(0, &c));
    CHECK(rto_set_threads(c, 2));
    CHECK(rto_set_config(c, &cfg));
    CHECK(rto_set_scene(c, objs, n, scale10));
    CHECK(rto_set_camera(c, &cam));
    if (env) CHECK(rto_set_env(c, env, ew, eh, 0, env_exposure, env_gamma));
    if (shape) CHECK(rto_set_shape_data(c, RTPBR_SHAPE_BUNNY, shape, nshape));
    if (tiles[3] > 1) CHECK(rto_set_tiles(c, tiles[0], tiles[1], tiles[2], tiles[3]));
    CHECK(rto_refresh(c));
    for (int r = 0; r < rounds; r++) CHECK(rto_sample(c, spp));
    CHECK(rto_post_process(c));
    const size_t px = (size_t)cfg.width * (size_t)cfg.height;
    float* t7 = malloc(px * 16);
    float* t8 = malloc(px * 12);
    rtpbr_counters ctr;
    CHECK(rto_read_buffer(c, RTPBR_BUF_IMAGE_BUFFER, t7, px * 16));
    CHECK(rto_read_buffer(c, RTPBR_BUF_IMAGE_PIXELS, t8, px * 12));
    CHECK(rto_get_counters(c, &ctr));
    /* error paths must be clean too */
    if (rto_read_buffer(c, 99, t7, 16) == 0 || rto_set_scene(c, objs, 0, 0) == 0) { fprintf(stderr, "bad arguments were accepted\n"); return 3; }
    CHECK(rto_destroy(c));
    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 1; }
    fwrite(t7, 16, px, o);
    fwrite(t8, 12, px, o);
    fwrite(&ctr, sizeof ctr, 1, o);
    fclose(o);
    free(t7); free(t8); free(objs); free(env); free(shape);
    return 0;
}
