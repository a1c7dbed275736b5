//! Raw bindings to `librrt.so` — a transcription of `include/rrt.h`, struct for struct and function for function
//! (field order, widths and names as in the header; `bindgen include/rrt.h` produces the same).
//! NOT COMPILED in the build image (no Rust toolchain there); see integration/README.md.
#![allow(non_camel_case_types, non_upper_case_globals)]

use std::os::raw::{c_char, c_int, c_void};

/// opaque: one Environment (Scene or PhotonScene)
#[repr(C)]
pub struct rrt_scene {
    _private: [u8; 0],
}

// enum rrt_status
pub const RRT_OK: c_int = 0;
pub const RRT_ERR_INVALID: c_int = -1;
pub const RRT_ERR_DEVICE: c_int = -2;
pub const RRT_ERR_OVERFLOW: c_int = -3;
pub const RRT_ERR_PARSE: c_int = -4;
pub const RRT_ERR_IO: c_int = -5;
pub const RRT_ERR_STATE: c_int = -6;
pub const RRT_ERR_LIMIT: c_int = -7;

// enum rrt_mode
pub const RRT_MODE_SCENE: c_int = 0;
pub const RRT_MODE_PHOTON: c_int = 1;

// enum rrt_bvh_builder
pub const RRT_BVH_HOST_SAH: u32 = 0;
pub const RRT_BVH_DEVICE_LBVH: u32 = 1;
pub const RRT_BVH_DEVICE_SAH: u32 = 2;
pub const RRT_BVH_AUTO: u32 = 3;

pub const RRT_GATHER_K_MAX: u32 = 4096;
pub const RRT_NOALLOC_MAX_RAYS: usize = 65536;
pub const RRT_MAX_COMPOUND_PARTS: usize = 4;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_params {
    pub photons_per_light: u32,
    pub caustic_photons_per_light: u32,
    pub gather_half_width: f32,
    pub max_depth: u32,
    pub max_photon_bounces: u32,
    pub max_vueon_depth: u32,
    pub chunk_pixels: u32,
    pub render_streams: u32,
    pub bvh_builder: u32,
    pub gather_heavy_threshold: u32,
    pub gather_k: u32,
    pub reserved: [u32; 5],
}

// enum rrt_material_kind
pub const RRT_MAT_MONOCHROME: i32 = 0;
pub const RRT_MAT_TEXTURE: i32 = 1;
pub const RRT_MAT_GLOBAL: i32 = 2;
pub const RRT_MAT_FALSECOLOUR: i32 = 3;
pub const RRT_MAT_COMPOUND: i32 = 4;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_material_desc {
    pub kind: i32,
    pub colour: [f32; 3],
    pub ambient: f32,
    pub shininess: f32,
    pub diffuse_tex: i32,
    pub normal_tex: i32,
    pub scale: f32,
    pub reflect_weight: f32,
    pub refract_weight: f32,
    pub ior: f32,
    pub n_parts: i32,
    pub parts: [i32; RRT_MAX_COMPOUND_PARTS],
}

// enum rrt_object_kind
pub const RRT_OBJ_PLANE: i32 = 0;
pub const RRT_OBJ_SPHERE: i32 = 1;
pub const RRT_OBJ_CUBOID: i32 = 2;
pub const RRT_OBJ_QUADRATIC: i32 = 3;
pub const RRT_OBJ_MESH: i32 = 4;
pub const RRT_OBJ_CSG: i32 = 5;

// enum rrt_csg_mode
pub const RRT_CSG_UNION: i32 = 0;
pub const RRT_CSG_INTERSECTION: i32 = 1;
pub const RRT_CSG_DIFFERENCE: i32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_object_desc {
    pub kind: i32,
    pub material: i32,
    pub v: [f32; 10],
    pub positions: *const f32,
    pub normals: *const f32,
    pub n_triangles: u64,
    pub smooth: i32,
    pub csg_mode: i32,
    pub left: i32,
    pub right: i32,
}

// enum rrt_light_kind
pub const RRT_LIGHT_DIRECTIONAL: i32 = 0;
pub const RRT_LIGHT_POINT: i32 = 1;
pub const RRT_LIGHT_DIRPOINT: i32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_light_desc {
    pub kind: i32,
    pub position: [f32; 3],
    pub direction: [f32; 3],
    pub colour: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_camera {
    pub width: u32,
    pub height: u32,
    pub fov: f32,
    pub position: [f32; 3],
    pub lookat: [f32; 3],
    pub up: [f32; 3],
    pub right: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rrt_stats {
    pub rays_primary: u64,
    pub rays_secondary: u64,
    pub rays_shadow: u64,
    pub bvh_nodes_visited: u64,
    pub bvh_tris_tested: u64,
    pub photons_emitted: u64,
    pub photons_stored_regular: u64,
    pub photons_stored_caustic: u64,
    pub photons_gathered: u64,
    pub gather_queries: u64,
    pub gather_candidates: u64,
    pub ms_trace_kernels: f64,
    pub ms_total: f64,
    pub ms_photon_shoot: f64,
    pub ms_photon_build: f64,
    pub flops_algorithmic: u64,
    pub flops_per_trace: u64,
    pub frames_repeated: u64,
    pub ms_bvh_build: f64,
    pub paths_truncated: u64,
    pub vueons_truncated: u64,
    pub reserved: [u64; 2],
}

// enum rrt_photon_type
pub const RRT_PHOTON_COLOUR: i32 = 0;
pub const RRT_PHOTON_SHADOW: i32 = 1;
pub const RRT_PHOTON_CAUSTIC: i32 = 2;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct rrt_builder {
    pub ctx: *mut c_void,
    pub set_mode: Option<unsafe extern "C" fn(ctx: *mut c_void, mode: c_int) -> c_int>,
    pub add_texture: Option<unsafe extern "C" fn(ctx: *mut c_void, width: u32, height: u32, maxval: u32, samples: *const u8) -> c_int>,
    pub add_material: Option<unsafe extern "C" fn(ctx: *mut c_void, desc: *const rrt_material_desc) -> c_int>,
    pub object_define: Option<unsafe extern "C" fn(ctx: *mut c_void, desc: *const rrt_object_desc) -> c_int>,
    pub object_transform: Option<unsafe extern "C" fn(ctx: *mut c_void, handle: c_int, m16: *const f32) -> c_int>,
    pub add_object: Option<unsafe extern "C" fn(ctx: *mut c_void, handle: c_int) -> c_int>,
    pub add_light: Option<unsafe extern "C" fn(ctx: *mut c_void, desc: *const rrt_light_desc) -> c_int>,
}

#[link(name = "rrt")]
extern "C" {
    pub fn rrt_last_error() -> *const c_char;
    pub fn rrt_version() -> *const c_char;
    pub fn rrt_params_default(p: *mut rrt_params);

    pub fn rrt_scene_create(mode: c_int, params: *const rrt_params) -> *mut rrt_scene;
    pub fn rrt_scene_destroy(s: *mut rrt_scene);
    pub fn rrt_scene_set_mode(s: *mut rrt_scene, mode: c_int) -> c_int;
    pub fn rrt_scene_get_mode(s: *const rrt_scene) -> c_int;
    pub fn rrt_scene_set_params(s: *mut rrt_scene, params: *const rrt_params) -> c_int;

    pub fn rrt_add_texture(s: *mut rrt_scene, width: u32, height: u32, maxval: u32, samples: *const u8) -> c_int;
    pub fn rrt_add_texture_ppm(s: *mut rrt_scene, path: *const c_char) -> c_int;
    pub fn rrt_add_material(s: *mut rrt_scene, desc: *const rrt_material_desc) -> c_int;
    pub fn rrt_object_define(s: *mut rrt_scene, desc: *const rrt_object_desc) -> c_int;
    pub fn rrt_object_define_obj(s: *mut rrt_scene, obj_path: *const c_char, material: c_int, smooth: c_int) -> c_int;
    pub fn rrt_object_transform(s: *mut rrt_scene, handle: c_int, m16: *const f32) -> c_int;
    pub fn rrt_add_object(s: *mut rrt_scene, handle: c_int) -> c_int;
    pub fn rrt_add_light(s: *mut rrt_scene, desc: *const rrt_light_desc) -> c_int;
    pub fn rrt_object_count(s: *const rrt_scene) -> c_int;
    pub fn rrt_light_count(s: *const rrt_scene) -> c_int;
    pub fn rrt_scene_builder(s: *mut rrt_scene) -> *const rrt_builder;

    pub fn rrt_camera_init(cam: *mut rrt_camera, width: u32, height: u32, fov_radians: f32, position: *const f32, lookat: *const f32,
                           up: *const f32) -> c_int;
    pub fn rrt_camera_ray(cam: *const rrt_camera, x: u32, y: u32, origin: *mut f32, direction: *mut f32) -> c_int;

    pub fn rrt_parse_scene_file(path: *const c_char, asset_root: *const c_char, builder: *const rrt_builder, camera_out: *mut rrt_camera,
                                mode_override: c_int, err_line: *mut u32) -> c_int;
    pub fn rrt_parse_scene_text(text: *const c_char, asset_root: *const c_char, builder: *const rrt_builder, camera_out: *mut rrt_camera,
                                mode_override: c_int, err_line: *mut u32) -> c_int;
    pub fn rrt_scene_load(path: *const c_char, asset_root: *const c_char, params: *const rrt_params, mode_override: c_int,
                          camera_out: *mut rrt_camera) -> *mut rrt_scene;

    pub fn rrt_pre_render(s: *mut rrt_scene, seed: u64) -> c_int;
    pub fn rrt_render(s: *mut rrt_scene, cam: *const rrt_camera, y0: u32, y1: u32, rgb: *mut f32, depth: *mut f32, object_id: *mut u32) -> c_int;
    pub fn rrt_render_device(s: *mut rrt_scene, cam: *const rrt_camera, y0: u32, y1: u32, rgb_dev: *mut f32, depth_dev: *mut f32,
                             object_id_dev: *mut u32, hip_stream: *mut c_void) -> c_int;
    pub fn rrt_interleaved_row_count(height: u32, group_rows: u32, first_group: u32, group_stride: u32) -> u32;
    pub fn rrt_render_interleaved_device(s: *mut rrt_scene, cam: *const rrt_camera, group_rows: u32, first_group: u32, group_stride: u32,
                                         rgb_dev: *mut f32, depth_dev: *mut f32, object_id_dev: *mut u32, hip_stream: *mut c_void) -> c_int;
    pub fn rrt_quantise_pack_device(rgb_dev: *const f32, depth_dev: *const f32, n_pixels: usize, rgbd_dev: *mut u8, hip_stream: *mut c_void) -> c_int;
    pub fn rrt_assemble_frame_device(parts_dev: *const u8, world: u32, rows_max: u32, width: u32, height: u32, row_group: u32,
                                     rgb_dev: *mut u8, depth_dev: *mut u8, hip_stream: *mut c_void) -> c_int;
    pub fn rrt_raytrace(s: *mut rrt_scene, origins: *const f32, directions: *const f32, n: usize, rgb: *mut f32, depth: *mut f32,
                        object_id: *mut u32) -> c_int;
    pub fn rrt_trace(s: *mut rrt_scene, origins: *const f32, directions: *const f32, n: usize, object_id: *mut u32, prim_id: *mut u32,
                     distance: *mut f32, position: *mut f32, normal: *mut f32, uv: *mut f32) -> c_int;
    pub fn rrt_shadowtrace(s: *mut rrt_scene, origins: *const f32, directions: *const f32, limits: *const f32, n: usize,
                           occluded: *mut u8) -> c_int;

    pub fn rrt_photon_count(s: *mut rrt_scene, which: c_int, count: *mut u64) -> c_int;
    pub fn rrt_photon_read(s: *mut rrt_scene, which: c_int, position: *mut f32, incident: *mut f32, intensity: *mut f32, type_: *mut u8,
                           key: *mut u64) -> c_int;
    pub fn rrt_photon_begin(s: *mut rrt_scene, seed: u64) -> c_int;
    pub fn rrt_photon_shoot(s: *mut rrt_scene, light: c_int, pass: c_int, first: u64, count: u64) -> c_int;
    pub fn rrt_photon_caustic_count(s: *mut rrt_scene, light: c_int, count: *mut u64) -> c_int;
    pub fn rrt_photon_caustic_export_device(s: *mut rrt_scene, light: c_int, pos4_dev: *mut f32, incident4_dev: *mut f32,
                                            key_dev: *mut u64) -> c_int;
    pub fn rrt_photon_caustic_import_device(s: *mut rrt_scene, light: c_int, pos4_dev: *const f32, incident4_dev: *const f32,
                                            key_dev: *const u64, count: u64) -> c_int;
    pub fn rrt_photon_local_count(s: *mut rrt_scene, which: c_int, count: *mut u64) -> c_int;
    pub fn rrt_photon_export_device(s: *mut rrt_scene, which: c_int, pos4_dev: *mut f32, incident4_dev: *mut f32, intensity4_dev: *mut f32,
                                    meta_dev: *mut u64) -> c_int;
    pub fn rrt_photon_import_device(s: *mut rrt_scene, which: c_int, pos4_dev: *const f32, incident4_dev: *const f32,
                                    intensity4_dev: *const f32, meta_dev: *const u64, count: u64) -> c_int;
    pub fn rrt_photon_reserve(s: *mut rrt_scene, which: c_int, total: u64) -> c_int;
    pub fn rrt_photon_append_device(s: *mut rrt_scene, which: c_int, pos4_dev: *const f32, incident4_dev: *const f32,
                                    intensity4_dev: *const f32, meta_dev: *const u64, count: u64) -> c_int;
    pub fn rrt_photon_query(s: *mut rrt_scene, which: c_int, positions: *const f32, n: usize, half_width: f32, k_max: u32, count: *mut u32,
                            key: *mut u64, dist2: *mut f32) -> c_int;
    pub fn rrt_photon_nearest(s: *mut rrt_scene, which: c_int, positions: *const f32, n: usize, k: u32, count: *mut u32, key: *mut u64,
                              dist2: *mut f32) -> c_int;
    pub fn rrt_photon_save(s: *mut rrt_scene, path: *const c_char) -> c_int;
    pub fn rrt_photon_load(s: *mut rrt_scene, path: *const c_char) -> c_int;
    pub fn rrt_photon_build(s: *mut rrt_scene) -> c_int;

    pub fn rrt_trace_noalloc(s: *mut rrt_scene, origins: *const f32, directions: *const f32, n: usize, object_id: *mut u32,
                             prim_id: *mut u32, distance: *mut f32, position: *mut f32, normal: *mut f32, uv: *mut f32) -> c_int;

    pub fn rrt_get_stats(s: *const rrt_scene, out: *mut rrt_stats) -> c_int;

    pub fn rrt_quantise(values: *const f32, n: usize, out: *mut u8);
    pub fn rrt_write_rgb_ppm(path: *const c_char, rgb: *const f32, width: u32, height: u32) -> c_int;
    pub fn rrt_write_depth_pgm(path: *const c_char, depth: *const f32, width: u32, height: u32) -> c_int;
}

/// The message of the last failure on the calling thread.
pub fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(rrt_last_error()).to_string_lossy().into_owned() }
}
