/*
 * oracle.h — C API of the CPU oracle (liboracle.so).
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain C++17 restatement of the reference's
 * algorithms (giorgi-o/rust-raytracer, Rust, cannot be built here: no cargo/rustc, crates
 * not vendored — SURVEY.md §8c).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (librrt.so) never does.
 *
 * Parity status: PINNED ONLY WEAKLY.  The reference has no tests or golden vectors; the
 * oracle is checked against hand-derived known answers, render/depth.ppm
 * (all 255) and render/rgb.ppm (statistically: one unseeded photon-mapped run).  Third-party
 * arithmetic (`rand` 0.8.5 streams, `kd-tree` 0.5.1 traversal order) is "parity unpinned":
 * the oracle uses Philox4x32-10 streams and its own kd-tree with the same closed-box query.
 *
 * It shares include/rrt.h's description structs (data layout only) with the product.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "../include/rrt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

const char *orc_last_error(void);
orc_scene *orc_scene_create(int mode, const rrt_params *params);
void orc_scene_destroy(orc_scene *s);
int orc_scene_get_mode(const orc_scene *s);
const rrt_builder *orc_scene_builder(orc_scene *s);
int orc_object_count(const orc_scene *s);

int orc_camera_init(rrt_camera *cam, uint32_t width, uint32_t height, float fov_radians,
                    const float position[3], const float lookat[3], const float up[3]);
int orc_camera_ray(const rrt_camera *cam, uint32_t x, uint32_t y, float origin[3], float direction[3]);

/* Object::intersect of one top-level object: up to 6 hits in list order */
int orc_intersect(orc_scene *s, int object_id, const float origin[3], const float direction[3],
                  int *n_hits, float distance[6], int32_t entering[6], float position[18], float normal[18],
                  float uv[12], uint32_t prim[6]);

int orc_pre_render(orc_scene *s, uint64_t seed, int threads);
int orc_render(orc_scene *s, const rrt_camera *cam, uint32_t y0, uint32_t y1, float *rgb, float *depth,
               uint32_t *object_id, int threads);
int orc_raytrace(orc_scene *s, const float *origins, const float *directions, size_t n, float *rgb,
                 float *depth, uint32_t *object_id);
int orc_trace(orc_scene *s, const float *origins, const float *directions, size_t n, uint32_t *object_id,
              uint32_t *prim_id, float *distance, float *position, float *normal, float *uv);
int orc_shadowtrace(orc_scene *s, const float *origins, const float *directions, const float *limits,
                    size_t n, uint8_t *occluded);

/* photon pass pieces (for multi-rank tests): shoot [first, first+count) of (light, pass) into the
 * scene's local buffers; pass 1 requires orc_photon_set_caustic_seeds or a prior pass 0.        */
int orc_photon_begin(orc_scene *s, uint64_t seed);
int orc_photon_shoot(orc_scene *s, int light, int pass, uint64_t first, uint64_t count, int threads);
int orc_photon_build(orc_scene *s);
int orc_photon_count(orc_scene *s, int which, uint64_t *count);
int orc_photon_read(orc_scene *s, int which, float *position, float *incident, float *intensity,
                    uint8_t *type, uint64_t *key);
/* exchange points for multi-rank tests: [n][4] f32 host arrays (pos4.w = photon type bits), u64 keys */
int orc_photon_caustic_count(orc_scene *s, int light, uint64_t *count);
int orc_photon_caustic_export(orc_scene *s, int light, float *pos4, float *incident4, uint64_t *key);
int orc_photon_caustic_import(orc_scene *s, int light, const float *pos4, const float *incident4, const uint64_t *key, uint64_t count);
int orc_photon_export(orc_scene *s, int which, float *pos4, float *incident4, float *intensity4, uint64_t *meta);
int orc_photon_import(orc_scene *s, int which, const float *pos4, const float *incident4, const float *intensity4, const uint64_t *meta,
                      uint64_t count);
/* brute-force closed-box query used to validate the index: returns K and sums */
int orc_photon_query(orc_scene *s, int which, const float p[3], float half_width, uint64_t *k,
                     float dir_sum[3], float intensity_sum[3]);

/* PhotonTree::get_n_within_radius (photon_tree.rs:15-35, 41-50) for one point: photons inside the closed cube,
 * ascending by (squared distance, key), first k_max written; returns the count inside the cube in *count */
int orc_photon_query_list(orc_scene *s, int which, const float p[3], float half_width, uint32_t k_max, uint32_t *count, uint64_t *key,
                          float *dist2);

int orc_get_stats(const orc_scene *s, rrt_stats *out);

void orc_quantise(const float *values, size_t n, uint8_t *out);
int orc_write_rgb_ppm(const char *path, const float *rgb, uint32_t width, uint32_t height);
int orc_write_depth_pgm(const char *path, const float *depth, uint32_t width, uint32_t height);

/* Philox4x32-10 block, exposed so tests can pin it against the published known answers */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* Transform::inverse (transform.rs:46-177) for tests */
int orc_mat4_inverse(const float *m16, float *out16);

#ifdef __cplusplus
}
#endif
#endif
