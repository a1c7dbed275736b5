"""ctypes mirror of include/rrt.h (struct layouts and enums only)."""
import ctypes as C

RRT_OK = 0
ERR_INVALID, ERR_DEVICE, ERR_OVERFLOW, ERR_PARSE, ERR_IO, ERR_STATE, ERR_LIMIT = -1, -2, -3, -4, -5, -6, -7
MODE_SCENE, MODE_PHOTON = 0, 1
MAT_MONOCHROME, MAT_TEXTURE, MAT_GLOBAL, MAT_FALSECOLOUR, MAT_COMPOUND = range(5)
OBJ_PLANE, OBJ_SPHERE, OBJ_CUBOID, OBJ_QUADRATIC, OBJ_MESH, OBJ_CSG = range(6)
CSG_UNION, CSG_INTERSECTION, CSG_DIFFERENCE = range(3)
LIGHT_DIRECTIONAL, LIGHT_POINT, LIGHT_DIRPOINT = range(3)
MISS = 0xFFFFFFFF
NOALLOC_MAX_RAYS = 65536  # RRT_NOALLOC_MAX_RAYS
GATHER_K_MAX = 4096  # RRT_GATHER_K_MAX


BVH_HOST_SAH, BVH_DEVICE_LBVH, BVH_DEVICE_SAH, BVH_AUTO = 0, 1, 2, 3


class Params(C.Structure):
    _fields_ = [
        ("photons_per_light", C.c_uint32),
        ("caustic_photons_per_light", C.c_uint32),
        ("gather_half_width", C.c_float),
        ("max_depth", C.c_uint32),
        ("max_photon_bounces", C.c_uint32),
        ("max_vueon_depth", C.c_uint32),
        ("chunk_pixels", C.c_uint32),
        ("render_streams", C.c_uint32),
        ("bvh_builder", C.c_uint32),
        ("gather_heavy_threshold", C.c_uint32),
        ("gather_k", C.c_uint32),
        ("reserved", C.c_uint32 * 5),
    ]


class MaterialDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("colour", C.c_float * 3),
        ("ambient", C.c_float),
        ("shininess", C.c_float),
        ("diffuse_tex", C.c_int32),
        ("normal_tex", C.c_int32),
        ("scale", C.c_float),
        ("reflect_weight", C.c_float),
        ("refract_weight", C.c_float),
        ("ior", C.c_float),
        ("n_parts", C.c_int32),
        ("parts", C.c_int32 * 4),
    ]


class ObjectDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("material", C.c_int32),
        ("v", C.c_float * 10),
        ("positions", C.POINTER(C.c_float)),
        ("normals", C.POINTER(C.c_float)),
        ("n_triangles", C.c_uint64),
        ("smooth", C.c_int32),
        ("csg_mode", C.c_int32),
        ("left", C.c_int32),
        ("right", C.c_int32),
    ]


class LightDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("position", C.c_float * 3),
        ("direction", C.c_float * 3),
        ("colour", C.c_float * 3),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("fov", C.c_float),
        ("position", C.c_float * 3),
        ("lookat", C.c_float * 3),
        ("up", C.c_float * 3),
        ("right", C.c_float * 3),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64),
        ("rays_secondary", C.c_uint64),
        ("rays_shadow", C.c_uint64),
        ("bvh_nodes_visited", C.c_uint64),
        ("bvh_tris_tested", C.c_uint64),
        ("photons_emitted", C.c_uint64),
        ("photons_stored_regular", C.c_uint64),
        ("photons_stored_caustic", C.c_uint64),
        ("photons_gathered", C.c_uint64),
        ("gather_queries", C.c_uint64),
        ("gather_candidates", C.c_uint64),
        ("ms_trace_kernels", C.c_double),
        ("ms_total", C.c_double),
        ("ms_photon_shoot", C.c_double),
        ("ms_photon_build", C.c_double),
        ("flops_algorithmic", C.c_uint64),
        ("flops_per_trace", C.c_uint64),
        ("frames_repeated", C.c_uint64),
        ("ms_bvh_build", C.c_double),
        ("paths_truncated", C.c_uint64),
        ("vueons_truncated", C.c_uint64),
        ("reserved", C.c_uint64 * 2),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


class Builder(C.Structure):
    _fields_ = [
        ("ctx", C.c_void_p),
        ("set_mode", C.c_void_p),
        ("add_texture", C.c_void_p),
        ("add_material", C.c_void_p),
        ("object_define", C.c_void_p),
        ("object_transform", C.c_void_p),
        ("add_object", C.c_void_p),
        ("add_light", C.c_void_p),
    ]
