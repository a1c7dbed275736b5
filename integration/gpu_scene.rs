//! src/environments/gpu_scene.rs — an `Environment` whose passes run on an MI355X through librrt.so.
//! NOT COMPILED in the build image (no Rust toolchain there); written against the reference's traits as they are at
//! src/environments/environment.rs:25-75, src/objects/object.rs:3-6, src/lights/light.rs:6-20,
//! src/materials/material.rs:10-40 and src/cameras/{camera.rs:7-43, full_camera.rs:12-146}.
//!
//! How it fits.  The reference's seam is per ray (`Environment::raytrace`, `Object::intersect`); a GPU needs whole
//! passes.  `GpuScene` therefore records what `add_object` / `add_light` are given — each object, material and light
//! describes itself once through the small `Describe*` traits below, which is the only change to the reference's types —
//! and forwards `pre_render` and whole row bands.  Object ids are insertion order on both sides.

use std::ffi::CString;
use std::sync::Arc;

use rrt_sys::*;

use crate::cameras::full_camera::FullCamera;
use crate::core::{colour::Colour, framebuffer::FrameBuffer, ray::Ray, transform::Transform, vector::Vector, vertex::Vertex};
use crate::environments::environment::{Environment, RaytraceResult};
use crate::lights::light::Light;
use crate::materials::material::Material;
use crate::objects::object::Object;

/// Seed of the counter-based photon streams (the reference draws from an unseeded thread_rng; any value is "a run").
pub const DEFAULT_SEED: u64 = 1;

fn check(status: i32) -> i32 {
    if status < 0 {
        panic!("librrt: {} ({status})", rrt_sys::last_error()); // the reference panics where the library reports
    }
    status
}

// ---- what the reference's types add --------------------------------------------------------------------------------

/// `impl DescribeMaterial for Monochrome | Texture | GlobalMaterial | FalseColour | CompoundMaterial`: registers the
/// material (and its textures / parts) with the scene and returns its id.  Ids are cached per `Arc` by the caller.
pub trait DescribeMaterial {
    fn describe(&self, scene: *mut rrt_scene) -> i32;
}
/// `impl DescribeObject for Plane | Sphere | Cuboid | Quadratic | PolyMesh | Csg`: defines the object as it is NOW (after
/// the `apply_transform` calls it has received) and returns a detached handle.
pub trait DescribeObject {
    fn describe(&self, env: &mut GpuScene) -> i32;
}
/// `impl DescribeLight for DirectionalLight | PointLight | DPLight`
pub trait DescribeLight {
    fn describe(&self) -> rrt_light_desc;
}

fn xyz(v: &Vector) -> [f32; 3] {
    [v.x, v.y, v.z]
}
fn pxyz(v: &Vertex) -> [f32; 3] {
    [v.x, v.y, v.z]
}
fn rgb(c: &Colour) -> [f32; 3] {
    [c.r, c.g, c.b]
}
fn object_desc(kind: i32, material: i32, v: [f32; 10]) -> rrt_object_desc {
    rrt_object_desc { kind, material, v, positions: std::ptr::null(), normals: std::ptr::null(), n_triangles: 0, smooth: 0, csg_mode: 0,
                      left: -1, right: -1 }
}

// The impls below live next to each type (they read private fields); shown here in one place.
//
// impl DescribeObject for Sphere {                                                    // sphere_object.rs:18-32
//     fn describe(&self, env: &mut GpuScene) -> i32 {
//         let m = env.material_id(&self.material);
//         let c = &self.centre;
//         env.define(&object_desc(RRT_OBJ_SPHERE, m, [c.x, c.y, c.z, self.radius, 0., 0., 0., 0., 0., 0.]))
//     }
// }
// impl DescribeObject for Plane {                                                     // plane_object.rs:18-50
//     fn describe(&self, env: &mut GpuScene) -> i32 {
//         let m = env.material_id(&self.material);
//         let (p, u, n) = (&self.centre, &self.up, &self.normal);
//         env.define(&object_desc(RRT_OBJ_PLANE, m, [p.x, p.y, p.z, u.x, u.y, u.z, n.x, n.y, n.z, 0.]))
//     }
// }
// impl DescribeObject for Cuboid {                                                    // cuboid_object.rs:40-56
//     fn describe(&self, env: &mut GpuScene) -> i32 {
//         let m = env.material_id(&self.material);
//         let (c, s) = (&self.corner, &self.size);
//         env.define(&object_desc(RRT_OBJ_CUBOID, m, [c.x, c.y, c.z, s.x, s.y, s.z, 0., 0., 0., 0.]))
//     }
// }
// impl DescribeObject for Quadratic {                                                 // quadratic_object.rs:16-32
//     fn describe(&self, env: &mut GpuScene) -> i32 {
//         let m = env.material_id(&self.material);
//         let (a, b, c, d, e, f, g, h, i, j) = self.variables;
//         env.define(&object_desc(RRT_OBJ_QUADRATIC, m, [a, b, c, d, e, f, g, h, i, j]))
//     }
// }
// impl DescribeObject for PolyMesh {                                                  // polymesh_object.rs:22-31
//     fn describe(&self, env: &mut GpuScene) -> i32 {
//         // what each Triangle holds (triangle_object.rs:24-40), corner-major
//         let mut positions = Vec::with_capacity(self.triangles.len() * 9);
//         let mut normals = Vec::with_capacity(if self.smooth { self.triangles.len() * 9 } else { 0 });
//         for t in &self.triangles {
//             for rv in [&t.a, &t.b, &t.c] {
//                 positions.extend_from_slice(&pxyz(&rv.vertex));
//                 if self.smooth {
//                     normals.extend_from_slice(&xyz(rv.normal.as_ref().expect("smooth mesh without vertex normals")));
//                 }
//             }
//         }
//         let mut d = object_desc(RRT_OBJ_MESH, env.material_id(&self.material), [0.; 10]);
//         d.positions = positions.as_ptr();
//         d.normals = if self.smooth { normals.as_ptr() } else { std::ptr::null() };
//         d.n_triangles = self.triangles.len() as u64;
//         d.smooth = self.smooth as i32;
//         env.define(&d)                                     // the library copies the arrays before returning
//     }
// }
// impl DescribeObject for Csg {                                                       // csg_object.rs:25-36
//     fn describe(&self, env: &mut GpuScene) -> i32 {
//         let (l, r) = (self.left.describe(env), self.right.describe(env));      // needs `Object: DescribeObject`
//         let mut d = object_desc(RRT_OBJ_CSG, -1, [0.; 10]);
//         d.csg_mode = match self.mode { CsgMode::Union => RRT_CSG_UNION, CsgMode::Intersection => RRT_CSG_INTERSECTION,
//                                        CsgMode::Difference => RRT_CSG_DIFFERENCE };
//         d.left = l;
//         d.right = r;
//         env.define(&d)
//     }
// }
//
// impl DescribeMaterial for Monochrome {                                              // phong_material.rs:62-94
//     fn describe(&self, s: *mut rrt_scene) -> i32 {
//         let mut d = blank_material(RRT_MAT_MONOCHROME);
//         d.colour = rgb(&self.colour); d.ambient = self.ambient_strength; d.shininess = self.shininess;
//         check(unsafe { rrt_add_material(s, &d) })
//     }
// }
// impl DescribeMaterial for Texture {                                                 // texture.rs:127-189
//     fn describe(&self, s: *mut rrt_scene) -> i32 {
//         let mut d = blank_material(RRT_MAT_TEXTURE);
//         d.diffuse_tex = add_image(s, &self.diffuse);
//         d.normal_tex = self.normal.as_ref().map_or(-1, |n| add_image(s, n));
//         d.scale = self.scale; d.ambient = self.ambient_strength; d.shininess = self.shininess;
//         check(unsafe { rrt_add_material(s, &d) })
//     }
// }
// impl DescribeMaterial for GlobalMaterial {                                          // global_material.rs:12-30
//     fn describe(&self, s: *mut rrt_scene) -> i32 {
//         let mut d = blank_material(RRT_MAT_GLOBAL);
//         d.reflect_weight = self.reflect_weight; d.refract_weight = self.refract_weight; d.ior = self.ior;
//         check(unsafe { rrt_add_material(s, &d) })
//     }
// }
// impl DescribeMaterial for FalseColour { fn describe(&self, s: *mut rrt_scene) -> i32 {
//     check(unsafe { rrt_add_material(s, &blank_material(RRT_MAT_FALSECOLOUR)) }) } }
// impl DescribeMaterial for CompoundMaterial {                                        // compound_material.rs:11-37
//     fn describe(&self, s: *mut rrt_scene) -> i32 {
//         assert!(self.materials.len() <= RRT_MAX_COMPOUND_PARTS, "librrt: a compound material holds at most 4 parts");
//         let mut d = blank_material(RRT_MAT_COMPOUND);
//         d.n_parts = self.materials.len() as i32;
//         for (i, m) in self.materials.iter().enumerate() { d.parts[i] = m.describe(s); }   // a Global part must come last
//         check(unsafe { rrt_add_material(s, &d) })
//     }
// }
//
// impl DescribeLight for PointLight {                                                 // point_light.rs:17-30
//     fn describe(&self) -> rrt_light_desc {
//         rrt_light_desc { kind: RRT_LIGHT_POINT, position: pxyz(&self.position), direction: [0.; 3], colour: rgb(&self.intensity) }
//     }
// }
// impl DescribeLight for DirectionalLight {                                           // directional_light.rs
//     fn describe(&self) -> rrt_light_desc {
//         rrt_light_desc { kind: RRT_LIGHT_DIRECTIONAL, position: [0.; 3], direction: xyz(&self.direction), colour: rgb(&self.intensity) }
//     }
// }
// impl DescribeLight for DPLight {                                                    // directional_point_light.rs:17-35
//     fn describe(&self) -> rrt_light_desc {
//         rrt_light_desc { kind: RRT_LIGHT_DIRPOINT, position: pxyz(&self.position), direction: xyz(&self.direction),
//                          colour: rgb(&self.intensity) }
//     }
// }

pub fn blank_material(kind: i32) -> rrt_material_desc {
    rrt_material_desc { kind, colour: [0.; 3], ambient: 0., shininess: 0., diffuse_tex: -1, normal_tex: -1, scale: 1., reflect_weight: 0.,
                        refract_weight: 0., ior: 1., n_parts: 0, parts: [-1; RRT_MAX_COMPOUND_PARTS] }
}

/// An `Image` (texture.rs:41-103) as 16-bit samples.  Image keeps `sample / maxval` as f32; handing the file itself over
/// (`rrt_add_texture_ppm(path)`) keeps the exact integers and is what `Texture::import` should do instead.
pub fn add_image_file(scene: *mut rrt_scene, path: &str) -> i32 {
    let c = CString::new(path).unwrap();
    check(unsafe { rrt_add_texture_ppm(scene, c.as_ptr()) })
}

// ---- the Environment -------------------------------------------------------------------------------------------------

pub struct GpuScene {
    handle: *mut rrt_scene,
    /// kept so that `objects()` still answers (the CPU `trace` default keeps working for debugging)
    objects: Vec<Box<dyn Object>>,
    /// material ids by `Arc` address: one registration per material however many objects share it
    materials: Vec<(*const (), i32)>,
    pub seed: u64,
}

// `Environment: Send + Sync` (environment.rs:25).  Sound: the calls reachable through `&self` (rrt_render,
// rrt_render_device, rrt_raytrace) may run concurrently on one handle (include/rrt.h, "Threading"); everything that needs
// the handle to itself is behind `&mut self`.
unsafe impl Send for GpuScene {}
unsafe impl Sync for GpuScene {}

impl GpuScene {
    /// `mode`: RRT_MODE_SCENE = Scene::new (scene.rs:14-19), RRT_MODE_PHOTON = PhotonScene::new (photon_scene.rs:33-40)
    pub fn new(mode: i32) -> Self {
        let handle = unsafe { rrt_scene_create(mode, std::ptr::null()) };
        assert!(!handle.is_null(), "librrt: {}", rrt_sys::last_error());
        Self { handle, objects: Vec::new(), materials: Vec::new(), seed: DEFAULT_SEED }
    }
    pub fn with_params(mode: i32, params: &rrt_params) -> Self {
        let handle = unsafe { rrt_scene_create(mode, params) };
        assert!(!handle.is_null(), "librrt: {}", rrt_sys::last_error());
        Self { handle, objects: Vec::new(), materials: Vec::new(), seed: DEFAULT_SEED }
    }
    pub fn handle(&self) -> *mut rrt_scene {
        self.handle
    }
    pub fn define(&mut self, d: &rrt_object_desc) -> i32 {
        check(unsafe { rrt_object_define(self.handle, d) })
    }
    pub fn material_id(&mut self, m: &Arc<dyn Material>) -> i32 {
        let key = Arc::as_ptr(m) as *const ();
        if let Some((_, id)) = self.materials.iter().find(|(k, _)| *k == key) {
            return *id;
        }
        let id = m.describe(self.handle); // needs `Material: DescribeMaterial`
        self.materials.push((key, id));
        id
    }
    /// Forward a transform to an already defined, still detached object instead of describing it afterwards: both
    /// routes give the same bits (the library applies each type's own rule, object.rs:5 and the per-type impls).
    pub fn transform(&mut self, handle: i32, t: &Transform) {
        let m: [f32; 16] = unsafe { std::mem::transmute(t.matrix) }; // row-major [[f32; 4]; 4], transform.rs:4-6
        check(unsafe { rrt_object_transform(self.handle, handle, m.as_ptr()) });
    }
    pub fn stats(&self) -> rrt_stats {
        let mut st = std::mem::MaybeUninit::<rrt_stats>::zeroed();
        check(unsafe { rrt_get_stats(self.handle, st.as_mut_ptr()) });
        unsafe { st.assume_init() }
    }
    /// rows [y0, y1) of the frame into host planes (rgb: 3 f32 per pixel, depth: 1)
    pub fn render_rows(&self, cam: &rrt_camera, y0: u32, y1: u32, rgb: &mut [f32], depth: &mut [f32]) {
        let n = ((y1 - y0) * cam.width) as usize;
        assert!(rgb.len() >= 3 * n && depth.len() >= n);
        check(unsafe { rrt_render(self.handle, cam, y0, y1, rgb.as_mut_ptr(), depth.as_mut_ptr(), std::ptr::null_mut()) });
    }
}

impl Drop for GpuScene {
    fn drop(&mut self) {
        unsafe { rrt_scene_destroy(self.handle) };
    }
}

impl Environment for GpuScene {
    fn pre_render(&mut self) {
        // scene.rs:140 (no-op) / photon_scene.rs:433-435 (build_photon_maps): flatten, BVHs, photon passes, index
        check(unsafe { rrt_pre_render(self.handle, self.seed) });
    }

    /// One ray through the batched entry: correct, and as slow as a kernel launch per ray — `Camera::render` must go
    /// through `render_rows` / `FullCamera::render_gpu`.  Kept for probes (main.rs prints single-ray results in places).
    fn raytrace(&self, ray: &Ray) -> RaytraceResult {
        let (o, d) = (pxyz(&ray.position), xyz(&ray.direction));
        let (mut c, mut depth) = ([0f32; 3], 0f32);
        check(unsafe { rrt_raytrace(self.handle, o.as_ptr(), d.as_ptr(), 1, c.as_mut_ptr(), &mut depth, std::ptr::null_mut()) });
        RaytraceResult { colour: Colour::new(c[0], c[1], c[2]), depth }
    }

    fn add_object(&mut self, object: Box<dyn Object + 'static>) {
        let handle = object.describe(self); // needs `Object: DescribeObject`
        check(unsafe { rrt_add_object(self.handle, handle) }); // returns the object id = self.objects.len()
        self.objects.push(object);
    }

    fn add_light(&mut self, light: Box<dyn Light + 'static>) {
        let d = light.describe(); // needs `Light: DescribeLight`
        check(unsafe { rrt_add_light(self.handle, &d) });
    }

    fn objects(&self) -> &[Box<dyn Object>] {
        &self.objects
    }
}

// ---- the camera ------------------------------------------------------------------------------------------------------

impl FullCamera {
    /// The state after `FullCamera::new` (full_camera.rs:23-58), field for field.
    pub fn as_rrt(&self) -> rrt_camera {
        rrt_camera { width: self.width, height: self.height, fov: self.fov, position: pxyz(&self.position), lookat: xyz(&self.lookat),
                     up: xyz(&self.up), right: xyz(&self.right) }
    }

    /// `Camera::render` (camera.rs:7-43) for a GpuScene: one call for the whole frame instead of the thread fan-out.
    /// (The fan-out itself keeps working — `render_rows` bands may run concurrently on one handle — but a band per CPU
    /// thread is the slow way to drive a GPU.)
    pub fn render_gpu(&self, env: &mut GpuScene) -> FrameBuffer {
        env.pre_render();
        let cam = self.as_rrt();
        let n = (self.width * self.height) as usize;
        let (mut rgb, mut depth) = (vec![0f32; 3 * n], vec![0f32; n]);
        env.render_rows(&cam, 0, self.height, &mut rgb, &mut depth);
        let mut fb = FrameBuffer::new(self.width, self.height);
        for y in 0..self.height {
            for x in 0..self.width {
                let i = (y * self.width + x) as usize;
                fb.plot_pixel(x, y, &Colour::new(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2])); // framebuffer.rs:76-84
                fb.plot_depth(x, y, depth[i]);
            }
        }
        fb
    }
}

// ---- scene files -----------------------------------------------------------------------------------------------------

/// `SceneFile::from_path` (scene_file.rs:59-122) without the Rust parser: the library parses the same grammar and fills a
/// scene directly.  `asset_root` = the directory that holds `assets/` (the reference uses the cwd, main.rs:64-70).
pub fn load_scene_file(path: &str, asset_root: &str) -> (GpuScene, rrt_camera) {
    let (p, r) = (CString::new(path).unwrap(), CString::new(asset_root).unwrap());
    let mut cam = std::mem::MaybeUninit::<rrt_camera>::zeroed();
    let handle = unsafe { rrt_scene_load(p.as_ptr(), r.as_ptr(), std::ptr::null(), -1, cam.as_mut_ptr()) };
    assert!(!handle.is_null(), "librrt: {}", rrt_sys::last_error()); // RRT_ERR_PARSE carries the reference's line number
    (GpuScene { handle, objects: Vec::new(), materials: Vec::new(), seed: DEFAULT_SEED }, unsafe { cam.assume_init() })
}
