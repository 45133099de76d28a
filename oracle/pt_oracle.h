/*
 * pt_oracle.h — interface of the CPU restatement (see pt_oracle.c).  TEST INFRASTRUCTURE:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 * The POD scene types are the boundary's (include/ptrace.h); nothing else is shared with the product.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/ptrace.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pto_scene {
    pt_camera camera;
    const pt_object *objs;
    uint32_t n_objs;
    const pt_triangle *tris;
    uint32_t n_tris;
} pto_scene;

typedef struct pto_config {
    uint32_t width, height, spp;
    uint32_t _pad;
    uint64_t seed;
} pto_config;

typedef struct pto_counters {
    uint64_t ray_bounces; /* intersect_scene / radiance invocations (mod.rs:663) */
    uint64_t misses;
    uint64_t splits;
    uint64_t sphere_tests;
    uint64_t triangle_tests;
} pto_counters;

void pto_vec_ops(const float *a, const float *b, float s, float *out23);
float pto_sinf(float y);
float pto_cosf(float y);
void pto_sincos_vs_libm(uint32_t k_begin, uint32_t k_end, uint64_t *sin_mismatch, uint64_t *cos_mismatch);
void pto_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int rounds, uint32_t out[4]);
void pto_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void pto_philox4x32_7(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]); /* the contract's generator */
float pto_u32_to_unit(uint32_t u);
void pto_draw4(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t tag, float u[4]);
float pto_gamma_correction(float x);
uint32_t pto_to_int_with_gamma_correction(float x);
void pto_camera_basis(const pt_camera *cam, float lens_center[3], float su[3], float sv[3]);
void pto_mesh_bounding_sphere(const pt_triangle *tris, uint32_t n, float center[3], float *radius);
int pto_intersect_sphere(const float pos[3], float radius, const float o[3], const float d[3], float *t,
                         float x[3], float n[3]);
void pto_intersect_batch(const pto_scene *s, const float *o, const float *d, uint32_t n, float *t,
                         int32_t *object_id, int32_t *tri_id, float *x, float *nrm);
void pto_radiance_mean(const pto_scene *s, const float o[3], const float d[3], uint64_t seed, uint32_t pixel,
                       uint32_t n, float out[3], pto_counters *cnt);
void pto_radiance_mean_at(const pto_scene *s, const float o[3], const float d[3], uint32_t depth, uint64_t seed,
                          uint32_t pixel, uint32_t n, float out[3], pto_counters *cnt);
void pto_primary_ray(const pt_camera *cam, uint32_t width, uint32_t height, uint32_t pixel_index,
                     uint32_t sample, uint64_t seed, float o[3], float d[3]);
void pto_render_pixel(const pto_scene *s, const pto_config *cfg, uint32_t pixel_index, float out[3],
                      pto_counters *cnt);
uint64_t pto_dump_rays(const pto_scene *s, const pto_config *cfg, uint32_t idx_begin, uint32_t idx_end,
                       float *rays_od, uint64_t cap);
int pto_render(const pto_scene *s, const pto_config *cfg, uint32_t idx_begin, uint32_t idx_end, float *out_rgb,
               int threads, pto_counters *cnt_out, double *seconds);
/* render() with MOCK_RANDOM = true (mod.rs:31-51, 1017-1018): sequential pixels, one global cyclic 9-value table */
int pto_render_mock(const pto_scene *s, const pto_config *cfg, float *out_rgb, pto_counters *cnt_out, uint64_t *draws);
/* Mesh::new's bounding_box (mod.rs:452-476, 501-536), SceneObjectData::intersect_bounds (mod.rs:282-290) and
 * get_orbit_point (src/views/viewport_tab.rs:401-431); `boxes` = 12 object-local triangles per object */
void pto_mesh_bounding_box(const pt_triangle *tris, uint32_t n, pt_triangle out[12]);
void pto_intersect_bounds_batch(const pto_scene *s, const pt_triangle *boxes, uint32_t object, const float *o,
                                const float *d, uint32_t n, int32_t *hit, float *t, float *x, float *nrm);
void pto_orbit_point_batch(const pto_scene *s, const pt_triangle *boxes, const float *o, const float *d, uint32_t n,
                           int32_t *found, float *point, int32_t *object_id, float *t);
int pto_max_threads(void);
size_t pto_format_ppm(const float *rgb, uint32_t width, uint32_t height, uint32_t spp, const char *scene_id,
                      uint64_t seconds, char *buf, size_t cap);
uint64_t pto_siphash13(const uint8_t *data, size_t len);
uint64_t pto_image_hash(const float *rgb, size_t n_floats);

#ifdef __cplusplus
}
#endif
#endif
