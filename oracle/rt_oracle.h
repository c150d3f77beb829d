/*
 * rt_oracle.h — CPU ORACLE (test infrastructure only; see rt_oracle.c).
 * Same entry points as include/rtpbr.h with the prefix rto_, so the Python ctypes
 * wrapper used for the HIP library can drive the oracle from tests/ unchanged.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H
#include "../include/rtpbr.h"
#ifdef __cplusplus
extern "C" {
#endif
struct rto_ctx;
int rto_create(int device, struct rto_ctx** out);
int rto_destroy(struct rto_ctx* c);
const char* rto_last_error(void);
const char* rto_backend(void);
int rto_set_config(struct rto_ctx* c, const rtpbr_config* cfg);
int rto_set_scene(struct rto_ctx* c, const rtpbr_object* objs, int n, int scale10);
int rto_get_scene(struct rto_ctx* c, rtpbr_object* objs, int n);
int rto_set_camera(struct rto_ctx* c, const rtpbr_camera* cam);
int rto_set_env(struct rto_ctx* c, const void* texels, int w, int h, int fmt, float exposure, float gamma);
int rto_set_tiles(struct rto_ctx* c, int tw, int th, int rank, int world);
int rto_refresh(struct rto_ctx* c);
int rto_sample(struct rto_ctx* c, int n);
int rto_post_process(struct rto_ctx* c);
int rto_sync(struct rto_ctx* c);
int rto_read_buffer(struct rto_ctx* c, int which, void* dst, size_t nbytes);
int rto_write_buffer(struct rto_ctx* c, int which, const void* src, size_t nbytes);
int rto_get_counters(struct rto_ctx* c, rtpbr_counters* out);
int rto_get_counter(struct rto_ctx* c, const char* name, unsigned long long* out);
/* oracle-only controls */
int rto_set_threads(struct rto_ctx* c, int n);
int rto_set_sample_base(struct rto_ctx* c, uint32_t base);
int rto_set_bunny_weights(const float* w, int n);
int rto_set_shape_data(struct rto_ctx* c, int shape, const float* data, int n);
void rto_rotate(const float* rad, float* m);
#ifdef __cplusplus
}
#endif
#endif
