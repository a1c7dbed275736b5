/*
 * rrt.h — C ABI of the MI355X render core (librrt.so).
 *
 * The reference (giorgi-o/rust-raytracer) has no FFI or plugin interface; the seam this
 * library replaces is the trait pair `Environment` (src/environments/environment.rs:25-75)
 * / `Object` (src/objects/object.rs:3-6) as driven by `Camera::render`
 * (src/cameras/camera.rs:7-43).  Per-pixel virtual calls cannot cross to a GPU, so the
 * seam sits one level up, at whole-pass granularity, with the same verbs:
 * add_object / add_light / pre_render / raytrace / render_rows.
 *
 * Conventions: plain pointers and sizes only; every function returns an rrt_status
 * (0 = ok, negative = error; the reference panics instead);
 * `rrt_last_error()` gives the message of the last failure on the calling thread.
 * Host buffers are owned by the caller, device memory by the library unless a function
 * name ends in `_device` (then the pointers are HBM addresses supplied by the caller).
 * All arithmetic on the path is IEEE f32 (dtype "f32").
 */
#ifndef RRT_H
#define RRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rrt_scene rrt_scene; /* opaque: one Environment (Scene or PhotonScene) */

enum rrt_status {
    RRT_OK = 0,
    RRT_ERR_INVALID = -1,  /* bad argument / bad handle                                  */
    RRT_ERR_DEVICE = -2,   /* no GPU, HIP failure, kernel fault                          */
    RRT_ERR_OVERFLOW = -3, /* reference would panic "HitVec overflow" (core/hit.rs:70-73) */
    RRT_ERR_PARSE = -4,    /* scene-file ParseError (scene_file.rs:29-35)                */
    RRT_ERR_IO = -5,       /* missing file (OBJ / PPM / scene)                           */
    RRT_ERR_STATE = -6,    /* call order violated (e.g. render before pre_render)        */
    RRT_ERR_LIMIT = -7     /* a documented capacity limit of this build was exceeded     */
};

/* scene_file.rs:278-285 — `scene Scene` | `scene PhotonScene` */
enum rrt_mode { RRT_MODE_SCENE = 0, RRT_MODE_PHOTON = 1 };

/* Compile-time constants of the reference, exposed as parameters whose defaults equal the
 * reference's values (SURVEY.md §5 "Config / flags").  rrt_params_default() fills them.  */
typedef struct rrt_params {
    uint32_t photons_per_light;         /* photon_scene.rs:22  PHOTONS_PER_LIGHT = 5_000_000        */
    uint32_t caustic_photons_per_light; /* photon_scene.rs:23  CAUSTIC_PHOTONS_PER_LIGHT = 10_000   */
    float gather_half_width;            /* photon_scene.rs:393 get_within_distance(.., 0.1): cube   */
    uint32_t max_depth;                 /* global_material.rs:72 `depth >= 5` -> black              */
    uint32_t max_photon_bounces;        /* reference: unbounded recursion (photon_scene.rs:171-179);
                                           this build stops a path after this many landings (64)    */
    uint32_t max_vueon_depth;           /* reference: unbounded (photon_scene.rs:329-352); deeper
                                           vueons return black here (default 12)                    */
    uint32_t chunk_pixels;              /* pixels per wavefront-queue chunk (0 = default 1<<23)     */
    uint32_t render_streams;            /* HIP streams the chunks of a frame are spread over (0 or 1 =
                                           the caller's stream only, the default; at most 8).  Measured
                                           on MI355X: 2 / 4 / 8 streams are 9 / 21 / 100 % slower on the
                                           1080p CSG scene, so this is off by default      */
    uint32_t bvh_builder;               /* RRT_BVH_AUTO (3, what rrt_params_default sets): RRT_BVH_DEVICE_SAH for meshes of
                                           32768 triangles or more, RRT_BVH_HOST_SAH below (same tree either way);
                                           RRT_BVH_HOST_SAH (0): binned-SAH build on the host, the faster trees;
                                           RRT_BVH_DEVICE_LBVH (1): Morton-order build on the GPU for meshes of more
                                           than 4 triangles (rebuilds in milliseconds; same frames, slower traversal);
                                           RRT_BVH_DEVICE_SAH (2): the host builder's algorithm level by level on the GPU:
                                           the host's tree (same frames, same traversal cost), built several times faster.
                                           The reference has no acceleration structure (polymesh_object.rs:219-236) */
    uint32_t gather_heavy_threshold;    /* photon mode: a radiance estimate whose two query boxes hold more candidate
                                           records than this is gathered by a whole wave in a second kernel instead of by
                                           its lane.  0 = default: derived from the photon maps at rrt_photon_build (4 x the
                                           candidate records of a typical query, at least 2048; vueons behind the first
                                           level, few and incoherent, hand over at an eighth of that, not below 256);
                                           a value given here holds for every level; 0xFFFFFFFF = never.  Same photons;
                                           the f32 sums are added in another (fixed) order                              */
    uint32_t gather_k;                  /* photon mode: 0 (default) = the reference's estimator, every photon inside the cube
                                           (photon_scene.rs:393 get_within_distance); k > 0 = PhotonTree::get_n_within_radius
                                           (photon_tree.rs:41-50, written but never called by the reference) as the estimator:
                                           the k photons of the cube closest to the hit (ascending squared distance, ties by
                                           record key), summed in that order.  At most RRT_GATHER_K_MAX; needs 8 k bytes of
                                           device workspace per ray in flight (chunks shrink to keep that under 2 GiB)      */
    uint32_t reserved[5];
} rrt_params;
#define RRT_GATHER_K_MAX 4096
enum rrt_bvh_builder { RRT_BVH_HOST_SAH = 0, RRT_BVH_DEVICE_LBVH = 1, RRT_BVH_DEVICE_SAH = 2, RRT_BVH_AUTO = 3 };

/* ---- materials (src/materials) ------------------------------------------------------ */
enum rrt_material_kind {
    RRT_MAT_MONOCHROME = 0, /* phong_material.rs:62-94                                  */
    RRT_MAT_TEXTURE = 1,    /* texture.rs:127-189                                       */
    RRT_MAT_GLOBAL = 2,     /* global_material.rs:12-137                                */
    RRT_MAT_FALSECOLOUR = 3,/* falsecolour_material.rs                                  */
    RRT_MAT_COMPOUND = 4    /* compound_material.rs: mean of parts; a Global part, if any,
                               must be the last part (true for every factory in
                               compound_material.rs:39-76); parts may not be compounds  */
};
#define RRT_MAX_COMPOUND_PARTS 4

typedef struct rrt_material_desc {
    int32_t kind;
    float colour[3];      /* Monochrome                                                 */
    float ambient;        /* Monochrome, Texture: ambient_strength                      */
    float shininess;      /* Monochrome, Texture                                        */
    int32_t diffuse_tex;  /* Texture: id from rrt_add_texture                           */
    int32_t normal_tex;   /* Texture: id or -1 (texture.rs:141 `.ok()`)                 */
    float scale;          /* Texture                                                    */
    float reflect_weight; /* Global                                                     */
    float refract_weight; /* Global                                                     */
    float ior;            /* Global                                                     */
    int32_t n_parts;      /* Compound                                                   */
    int32_t parts[RRT_MAX_COMPOUND_PARTS];
} rrt_material_desc;

/* ---- objects (src/objects) ----------------------------------------------------------- */
enum rrt_object_kind {
    RRT_OBJ_PLANE = 0,     /* v = point[3] up[3] normal[3]           plane_object.rs:26-50   */
    RRT_OBJ_SPHERE = 1,    /* v = centre[3] radius                   sphere_object.rs:24-32  */
    RRT_OBJ_CUBOID = 2,    /* v = corner[3] size[3]                  cuboid_object.rs:48-56  */
    RRT_OBJ_QUADRATIC = 3, /* v = a b c d e f g h i j                quadratic_object.rs:16-32 */
    RRT_OBJ_MESH = 4,      /* triangles below                        polymesh_object.rs:22-31 */
    RRT_OBJ_CSG = 5        /* csg_mode, left, right                  csg_object.rs:25-36     */
};
enum rrt_csg_mode { RRT_CSG_UNION = 0, RRT_CSG_INTERSECTION = 1, RRT_CSG_DIFFERENCE = 2 };

typedef struct rrt_object_desc {
    int32_t kind;
    int32_t material; /* id from rrt_add_material (ignored for CSG)                         */
    float v[10];
    /* mesh: what each `Triangle` holds (triangle_object.rs:24-40), corner-major           */
    const float *positions; /* [n_triangles][3 corners][xyz]                               */
    const float *normals;   /* [n_triangles][3 corners][xyz], or NULL when smooth == 0     */
    uint64_t n_triangles;
    int32_t smooth;
    /* csg: handles returned by rrt_object_define; both children are consumed (moved)      */
    int32_t csg_mode;
    int32_t left;
    int32_t right;
} rrt_object_desc;

/* ---- lights (src/lights) -------------------------------------------------------------- */
enum rrt_light_kind {
    RRT_LIGHT_DIRECTIONAL = 0, /* directional_light.rs       (no photon support: light.rs:16-18) */
    RRT_LIGHT_POINT = 1,       /* point_light.rs                                                 */
    RRT_LIGHT_DIRPOINT = 2     /* directional_point_light.rs                                     */
};
typedef struct rrt_light_desc {
    int32_t kind;
    float position[3];
    float direction[3];
    float colour[3]; /* `intensity` in the reference */
} rrt_light_desc;

/* ---- camera (src/cameras/full_camera.rs:12-20): the state AFTER FullCamera::new -------- */
typedef struct rrt_camera {
    uint32_t width, height;
    float fov; /* radians, used directly as focal length (full_camera.rs:68) */
    float position[3];
    float lookat[3];
    float up[3];
    float right[3];
} rrt_camera;

/* Counters of the last rrt_pre_render / rrt_render* call on a scene.                      */
typedef struct rrt_stats {
    uint64_t rays_primary;       /* camera rays traced                                      */
    uint64_t rays_secondary;     /* reflection + refraction rays traced (all levels)        */
    uint64_t rays_shadow;        /* shadow rays traced (scene.rs:65-79)                     */
    uint64_t bvh_nodes_visited;  /* ray-AABB pair tests                                     */
    uint64_t bvh_tris_tested;    /* ray-triangle tests (triangle_object.rs:129-160)         */
    uint64_t photons_emitted;    /* both passes, all lights                                 */
    uint64_t photons_stored_regular;
    uint64_t photons_stored_caustic;
    uint64_t photons_gathered;   /* photon records inside query cubes in the render pass    */
    uint64_t gather_queries;     /* box queries issued (2 per surface hit)                  */
    uint64_t gather_candidates;  /* photon records read by the gather (>= gathered)         */
    double ms_trace_kernels;     /* HIP-event time of the dominant kernel(s), last call     */
    double ms_total;             /* HIP-event time of the whole call on its stream          */
    double ms_photon_shoot;
    double ms_photon_build;
    uint64_t flops_algorithmic;  /* SURVEY.md 8d count: per-primitive flops x tests + 27/box + 61/triangle */
    uint64_t flops_per_trace;    /* analytic-primitive flops one trace() of this scene costs           */
    uint64_t frames_repeated;    /* render calls since rrt_pre_render that were run twice because a level
                                    queue outgrew the size predicted from the previous frame */
    double ms_bvh_build;         /* rrt_pre_render: time spent building mesh BVHs (host wall clock, whichever builder) */
    /* How often the caps on the reference's unbounded recursions bound (rrt_params.max_photon_bounces, max_vueon_depth;
     * photon_scene.rs:171-179, 329-352 recurse without limit): photon paths still alive after their last allowed landing
     * (all rrt_photon_shoot calls since rrt_photon_begin), and reflection / refraction vueons a frame did not trace because they
     * would have been deeper than the cap (last render call).  0 = the result is what the uncapped recursion gives.       */
    uint64_t paths_truncated;
    uint64_t vueons_truncated;
    uint64_t reserved[2];
} rrt_stats;

/* ---- builder vtable --------------------------------------------------------------------
 * The verbs scene_file.rs uses on an Environment while loading (scene_file.rs:86-91 and the
 * constructors in :287-551).  rrt_parse_scene_file drives any implementation of it; librrt
 * binds it to an rrt_scene, the CPU oracle binds it to its own scene type.                */
typedef struct rrt_builder {
    void *ctx;
    int (*set_mode)(void *ctx, int mode);
    int (*add_texture)(void *ctx, uint32_t width, uint32_t height, uint32_t maxval, const uint8_t *samples);
    int (*add_material)(void *ctx, const rrt_material_desc *desc);
    int (*object_define)(void *ctx, const rrt_object_desc *desc);
    int (*object_transform)(void *ctx, int handle, const float *m16);
    int (*add_object)(void *ctx, int handle);
    int (*add_light)(void *ctx, const rrt_light_desc *desc);
} rrt_builder;

/* ======================================================================================= */

const char *rrt_last_error(void);
const char *rrt_version(void);

/* replaces: the constants listed in rrt_params */
void rrt_params_default(rrt_params *p);

/* replaces: Scene::new (scene.rs:14-19) / PhotonScene::new (photon_scene.rs:33-40).
 * params == NULL means defaults.  Needs no GPU; device work starts at rrt_pre_render.     */
rrt_scene *rrt_scene_create(int mode, const rrt_params *params);
void rrt_scene_destroy(rrt_scene *s);
int rrt_scene_set_mode(rrt_scene *s, int mode);
int rrt_scene_get_mode(const rrt_scene *s);
int rrt_scene_set_params(rrt_scene *s, const rrt_params *params);

/* replaces: Image::from_ppm result (texture.rs:41-103).  samples = RGB interleaved, one byte per
 * sample if maxval <= 255 else two bytes big-endian as in the PPM body.  Returns texture id >= 0. */
int rrt_add_texture(rrt_scene *s, uint32_t width, uint32_t height, uint32_t maxval, const uint8_t *samples);
/* replaces: Image::from_ppm itself, reading the file (texture.rs:41-103).  Returns texture id.   */
int rrt_add_texture_ppm(rrt_scene *s, const char *path);

/* replaces: Monochrome::new / Texture::import / GlobalMaterial::new / FalseColour::new /
 * CompoundMaterial::add_material (src/materials).  Returns material id >= 0.              */
int rrt_add_material(rrt_scene *s, const rrt_material_desc *desc);

/* replaces: Plane::new / Sphere::new / Cuboid::new / Quadratic::new / Triangle::new x n /
 * Csg::new (src/objects).  Returns a handle to a detached object (a Box<dyn Object> not yet
 * owned by the Environment).                                                               */
int rrt_object_define(rrt_scene *s, const rrt_object_desc *desc);
/* replaces: PolyMesh::from_obj_file (polymesh_object.rs:34-194).  Returns a handle.        */
int rrt_object_define_obj(rrt_scene *s, const char *obj_path, int material, int smooth);
/* replaces: Object::apply_transform (object.rs:5) with each type's own rule
 * (plane_object.rs:115-124, sphere_object.rs:93-100, cuboid_object.rs:150-160,
 *  quadratic_object.rs:127-148, triangle_object.rs:162-173 + polymesh_object.rs:244-252,
 *  csg_object.rs:110-113).  m16 = row-major 4x4 as Transform.matrix (transform.rs:4-6).     */
int rrt_object_transform(rrt_scene *s, int handle, const float *m16);
/* replaces: Environment::add_object (scene.rs:132-134).  Returns the object_id (index into
 * Environment::objects(), i.e. insertion order — SURVEY.md §8a quirk 11).                  */
int rrt_add_object(rrt_scene *s, int handle);
/* replaces: Environment::add_light (scene.rs:136-138, photon_scene.rs:428-431)             */
int rrt_add_light(rrt_scene *s, const rrt_light_desc *desc);
int rrt_object_count(const rrt_scene *s);
int rrt_light_count(const rrt_scene *s);

/* builder vtable bound to `s` (valid until the scene is destroyed)                         */
const rrt_builder *rrt_scene_builder(rrt_scene *s);

/* replaces: FullCamera::new (full_camera.rs:23-58) incl. the y-negation, normalisation and
 * the perpendicularity check (error instead of panic).  fov in radians.                     */
int rrt_camera_init(rrt_camera *cam, uint32_t width, uint32_t height, float fov_radians,
                    const float position[3], const float lookat[3], const float up[3]);
/* replaces: FullCamera::get_ray_pixel (full_camera.rs:61-81), host side, for tests          */
int rrt_camera_ray(const rrt_camera *cam, uint32_t x, uint32_t y, float origin[3], float direction[3]);

/* replaces: SceneFile::from_path (scene_file.rs:59-122) against any builder.  asset_root is
 * the directory that holds `assets/` (the reference uses the cwd, main.rs:64-70).
 * mode_override: -1 = as written in the file, else rrt_mode.  On RRT_ERR_PARSE *err_line
 * holds the 1-based line (ParseError.line).                                                */
int rrt_parse_scene_file(const char *path, const char *asset_root, const rrt_builder *builder,
                         rrt_camera *camera_out, int mode_override, uint32_t *err_line);
/* same, from memory (SceneFile::from_contents, scene_file.rs:64)                            */
int rrt_parse_scene_text(const char *text, const char *asset_root, const rrt_builder *builder,
                         rrt_camera *camera_out, int mode_override, uint32_t *err_line);
/* convenience: create a scene and load a file into it                                       */
rrt_scene *rrt_scene_load(const char *path, const char *asset_root, const rrt_params *params,
                          int mode_override, rrt_camera *camera_out);

/* replaces: Environment::pre_render (scene.rs:140 no-op; photon_scene.rs:433-435 =
 * build_photon_maps :42-83).  Flattens the scene to SoA arrays in HBM, builds the mesh BVHs,
 * and in photon mode runs both emission passes for every light and builds the photon index.
 * `seed` keys the counter-based RNG (the reference uses unseeded thread_rng).              */
int rrt_pre_render(rrt_scene *s, uint64_t seed);

/* replaces: Camera::render_rows (full_camera.rs:93-146) for rows [y0, y1): the reference's
 * thread unit (camera.rs:22-40) and this library's multi-GPU shard unit.
 * rgb   [(y1-y0)*width*3] f32, depth [(y1-y0)*width] f32 (RaytraceResult, environment.rs:11-14),
 * object_id [(y1-y0)*width] u32 = index of the object whose intersect() produced the primary
 * hit, 0xFFFFFFFF on miss.  Any output pointer may be NULL.  Host buffers.                   */
int rrt_render(rrt_scene *s, const rrt_camera *cam, uint32_t y0, uint32_t y1, float *rgb,
               float *depth, uint32_t *object_id);
/* Threading (environment.rs:25 `Environment: Send + Sync`; scene.rs:83 raytrace(&self) runs on every render thread,
 * camera.rs:22-40): rrt_render, rrt_render_device and rrt_raytrace may be called from several threads on ONE handle
 * at the same time, for disjoint row bands of a frame or for different frames.  Each concurrent call renders through
 * queues and counters of its own; the scene and the photon maps are in HBM once.  Everything else on a handle (building,
 * rrt_pre_render, the photon calls, rrt_trace / rrt_shadowtrace) needs the handle to itself, like pre_render(&mut self).
 * Three frames in flight (three threads, a HIP stream each) trace 1.6x the rays per second of one.       */
/* same with outputs in HBM (caller-allocated device memory), launched on `hip_stream`
 * (a hipStream_t, NULL = default stream); returns after enqueueing + one sync for the
 * queue counters.  This is the entry bench.py times.                                       */
int rrt_render_device(rrt_scene *s, const rrt_camera *cam, uint32_t y0, uint32_t y1, float *rgb_dev,
                      float *depth_dev, uint32_t *object_id_dev, void *hip_stream);

/* Load-balanced variant of the row split for multi-GPU frames (SURVEY.md 8e "tiles round-robin"): renders the image rows
 * y with (y / group_rows) % group_stride == first_group (rank r of N passes first_group = r, group_stride = N), packed in
 * ascending order into the outputs ([rows][width] with rows = rrt_interleaved_row_count).  group_rows must be a multiple
 * of 8.  The reference's own split is contiguous bands (camera.rs:14-31), which leaves ranks with sky idle; pixels are
 * independent, so every split gives the same frame.  group_stride = 1 renders the whole frame.                     */
uint32_t rrt_interleaved_row_count(uint32_t height, uint32_t group_rows, uint32_t first_group, uint32_t group_stride);
int rrt_render_interleaved_device(rrt_scene *s, const rrt_camera *cam, uint32_t group_rows, uint32_t first_group,
                                  uint32_t group_stride, float *rgb_dev, float *depth_dev, uint32_t *object_id_dev,
                                  void *hip_stream);

/* Multi-GPU frames (the thread split of camera.rs:14-43 with ranks for threads): what a rank hands to rank 0 is
 * its share as the reference finally writes it, FrameBuffer::write_rgb_file / write_depth_file's `(c * 255.0) as u8`
 * (framebuffer.rs:96-130), 4 bytes r g b d per pixel instead of 20 bytes of f32 rgb + depth + id.  rrt_quantise_pack_device
 * packs a rendered share ([n_pixels] rgb f32 x 3 + depth f32, in HBM) into rgbd_dev [n_pixels][4]; rrt_assemble_frame_device
 * is FrameBuffer::combine_rows (framebuffer.rs:52-69) on rank 0: the gathered shares parts_dev [world][rows_max][width][4]
 * -> the P6 payload rgb_dev [height][width][3] and the P5 payload depth_dev [height][width] in image order.  row_group as in
 * rrt_render_interleaved_device (rank r rendered the groups r, r + world, ...); row_group = 0: contiguous bands, height /
 * world rows each and the remainder on the last rank (camera.rs:16-31).  Both run on `hip_stream` and return at once.     */
int rrt_quantise_pack_device(const float *rgb_dev, const float *depth_dev, size_t n_pixels, uint8_t *rgbd_dev, void *hip_stream);
int rrt_assemble_frame_device(const uint8_t *parts_dev, uint32_t world, uint32_t rows_max, uint32_t width, uint32_t height,
                              uint32_t row_group, uint8_t *rgb_dev, uint8_t *depth_dev, void *hip_stream);

/* replaces: Environment::raytrace (environment.rs:27), batched: n rays, origins/directions
 * [n][3] on the host.                                                                      */
int rrt_raytrace(rrt_scene *s, const float *origins, const float *directions, size_t n, float *rgb,
                 float *depth, uint32_t *object_id);
/* replaces: Environment::trace (environment.rs:55-74), batched, for parity tests: the first
 * entering hit with distance >= 0.  Outputs (each may be NULL): object_id [n] (0xFFFFFFFF =
 * miss), prim_id [n] (triangle index for meshes, leaf index in CSG post-order (mesh leaves too), else 0),
 * distance [n], position [n][3], normal [n][3], uv [n][2] (NaN when tex_coords is None).    */
int rrt_trace(rrt_scene *s, const float *origins, const float *directions, size_t n,
              uint32_t *object_id, uint32_t *prim_id, float *distance, float *position,
              float *normal, float *uv);
/* replaces: Scene::shadowtrace (scene.rs:65-79), batched: occluded[i] = 1 iff some object's
 * first entering hit lies in (1e-7, limit[i]).                                              */
int rrt_shadowtrace(rrt_scene *s, const float *origins, const float *directions, const float *limits,
                    size_t n, uint8_t *occluded);

/* ---- photon map access (photon mode) -------------------------------------
 * Photon record = core/photon.rs:15-20 {position, incident, intensity, photon_type} plus a
 * 64-bit key (emission index, landing, sub-index) that is unique within one (light, pass)
 * and makes the record set independent of GPU count and atomics order.
 * which: 0 = regular map, 1 = caustic map.                                                  */
enum rrt_photon_type { RRT_PHOTON_COLOUR = 0, RRT_PHOTON_SHADOW = 1, RRT_PHOTON_CAUSTIC = 2 };
int rrt_photon_count(rrt_scene *s, int which, uint64_t *count);
/* copy the map to host arrays (any may be NULL): position/incident/intensity [n][3], type [n],
 * key [n]; records are in canonical order (sorted by (light, pass, key)).                   */
int rrt_photon_read(rrt_scene *s, int which, float *position, float *incident, float *intensity,
                    uint8_t *type, uint64_t *key);

/* Multi-GPU photon pass, one process per GPU (replaces the thread fan-out of
 * PhotonLight::shoot_photons_mt, lights/light.rs:22-73, with ranks instead of threads):
 *   rrt_photon_begin(s, seed)                       flatten + BVH, empty buffers
 *   for each light l:
 *     rrt_photon_shoot(s, l, 0, first, count)       this rank's index range of pass 1
 *     rrt_photon_caustic_count / _export            this rank's pass-1 caustic photons
 *     (all-gather over RCCL by the host)            -> global caustic list
 *     rrt_photon_caustic_import(s, l, ...)          the same global list on every rank
 *     rrt_photon_shoot(s, l, 1, first, count)       this rank's range of pass 2
 *   rrt_photon_local_count / _export_device         this rank's records of both maps
 *   (all-gather over RCCL by the host)
 *   rrt_photon_import_device(s, which, ...)         global records, HBM pointers
 *   rrt_photon_build(s)                             index build; scene is now renderable
 * rrt_pre_render == this sequence with one rank.                                            */
int rrt_photon_begin(rrt_scene *s, uint64_t seed);
int rrt_photon_shoot(rrt_scene *s, int light, int pass, uint64_t first, uint64_t count);
int rrt_photon_caustic_count(rrt_scene *s, int light, uint64_t *count);
/* Exchange arrays are float4 records (16 bytes each), as the library keeps them in HBM: the collective moves them
 * without repacking.  Device pointers; the caller allocates count * 16 bytes per float4 array and count * 8 per key /
 * meta array, count from the matching *_count call.
 * Caustic seeds of one light (what pass 2 aims at: Point lights use `incident`, DirPoint lights `position`,
 * point_light.rs:117-121, directional_point_light.rs:110-115):
 *   pos4 [count][4] f32 = position.xyz, w = photon type as uint bits;  incident4 [count][4] f32 = incident.xyz, w unused;
 *   key [count] u64 = the seed photon's record key.                                              */
int rrt_photon_caustic_export_device(rrt_scene *s, int light, float *pos4_dev, float *incident4_dev,
                                     uint64_t *key_dev);
int rrt_photon_caustic_import_device(rrt_scene *s, int light, const float *pos4_dev,
                                     const float *incident4_dev, const uint64_t *key_dev, uint64_t count);
int rrt_photon_local_count(rrt_scene *s, int which, uint64_t *count);
/* SoA export of this rank's records of map `which`:
 *   pos4 [count][4] f32 = position.xyz, w = rrt_photon_type as uint bits;  incident4 [count][4] f32 = incident.xyz, w unused;
 *   intensity4 [count][4] f32 = intensity.rgb, w unused;
 *   meta [count] u64 = record key in the low 56 bits, (light * 2 + pass) in bits 56..62.
 * 56 bytes per record in all.                                                                     */
int rrt_photon_export_device(rrt_scene *s, int which, float *pos4_dev, float *incident4_dev,
                             float *intensity4_dev, uint64_t *meta_dev);
int rrt_photon_import_device(rrt_scene *s, int which, const float *pos4_dev, const float *incident4_dev,
                             const float *intensity4_dev, const uint64_t *meta_dev, uint64_t count);

/* The all-gathered map arrives as one segment per rank (a padded all-gather leaves gaps between them): reserve room for
 * the total once, import the first segment (rrt_photon_import_device, which replaces this rank's own records), append the
 * others.  Replaces the per-thread Vec concatenation of flatten_photons (photon_scene.rs:114-148).                    */
int rrt_photon_reserve(rrt_scene *s, int which, uint64_t total);
int rrt_photon_append_device(rrt_scene *s, int which, const float *pos4_dev, const float *incident4_dev,
                             const float *intensity4_dev, const uint64_t *meta_dev, uint64_t count);

/* PhotonTree::get_within_distance and get_n_within_radius (core/photon_tree.rs:15-35, 41-50), batched: for each of
 * the n query points the photons of map `which` (0 regular, 1 caustic) inside the closed cube of half width
 * `half_width` around it, ascending by squared distance (ties by key: the reference's sort is unstable there).
 * count[n] = photons inside the cube; key / dist2 [n][k_max] = the first min(count, k_max) of them (k_max = 0:
 * counts only; a k_max above count leaves the rest of the row untouched).  Host pointers.                       */
int rrt_photon_query(rrt_scene *s, int which, const float *positions, size_t n, float half_width, uint32_t k_max,
                     uint32_t *count, uint64_t *key, float *dist2);
/* PhotonTree::find_nearest (core/photon_tree.rs:37-39; kd-tree 0.5.1 `nearests`, never called by the reference), batched:
 * for each of the n query points the k photons of map `which` closest to it in Euclidean distance, ascending by squared
 * distance (ties by key).  count[n] = min(k, photons in the map); key / dist2 [n][k].  The hash grid is searched in cubes
 * of doubling half width until the k-th distance fits inside the cube; points whose k nearest photons are farther than
 * 64 index cells away are searched exhaustively on the host (maps of up to 2^26 records; RRT_ERR_LIMIT above).  Host
 * pointers.                                                                                                          */
int rrt_photon_nearest(rrt_scene *s, int which, const float *positions, size_t n, uint32_t k, uint32_t *count, uint64_t *key,
                       float *dist2);

/* Photon-map files: both maps of this rank (after rrt_photon_build's canonical sort order), so that cameras can be
 * re-rendered without re-shooting.  No counterpart in the reference (its maps live and die with the process);
 * SURVEY.md 8f rank 2.  Load needs rrt_photon_begin on the same scene first and rrt_photon_build after it.
 * Format: DESIGN.md "photon-map files".                                                          */
int rrt_photon_save(rrt_scene *s, const char *path);
int rrt_photon_load(rrt_scene *s, const char *path);
int rrt_photon_build(rrt_scene *s);

/* rrt_trace for callers that must not call hipMalloc / hipFree: tight serving loops, harnesses that forbid allocation.  (It
 * works on the null stream and synchronises it twice, so it cannot itself be captured into a HIP graph.)  No counterpart in the reference.  Same arguments, results and error behaviour as
 * rrt_trace (host pointers, any output may be NULL), but every byte of device memory the call needs beyond the scene
 * (staging of the ray and result arrays, the error word, the traversal counters) lives in one static device array of
 * 4.25 MiB that each GPU holds from library load to process exit: the call makes no hipMalloc / hipFree.  At most
 * RRT_NOALLOC_MAX_RAYS rays per call (RRT_ERR_LIMIT above, nothing is traced).  All calls on one GPU share that
 * GPU's array: calls that overlap in time on one GPU (from several threads or scene handles) race; one after the
 * other they are independent.  Needs rrt_pre_render first (that call does allocate: it puts the scene into HBM). */
#define RRT_NOALLOC_MAX_RAYS 65536
int rrt_trace_noalloc(rrt_scene *s, const float *origins, const float *directions, size_t n,
                      uint32_t *object_id, uint32_t *prim_id, float *distance, float *position,
                      float *normal, float *uv);

/* the render fields are those of the CALLING THREAD's last render call on this handle (concurrent callers each see
 * their own frame); after anything else changed the handle's statistics, the handle's latest values               */
int rrt_get_stats(const rrt_scene *s, rrt_stats *out);

/* replaces: FrameBuffer::write_rgb_file / write_depth_file (framebuffer.rs:96-130): the
 * saturating `(c * 255.0) as u8` quantiser and the P6 / P5 writers.                         */
void rrt_quantise(const float *values, size_t n, uint8_t *out);
int rrt_write_rgb_ppm(const char *path, const float *rgb, uint32_t width, uint32_t height);
int rrt_write_depth_pgm(const char *path, const float *depth, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif /* RRT_H */
