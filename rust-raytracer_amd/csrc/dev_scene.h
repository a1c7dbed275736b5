// dev_scene.h — the flattened scene as it lives in HBM (SoA per primitive type) and the
// per-ray records of the wavefront queues.  Shared by the host flattener and the gfx950 kernels.
//
// Layout rules: everything a ray touches for EVERY object (the "hot" arrays) is float4-packed so a
// wave reads it with uniform (scalar-cache) or 16-B vector loads; everything needed only for the one
// winning hit (uv frames, tangent frames, vertex normals) sits in separate "cold" arrays.
#pragma once
#include <stdint.h>

#include "../../include/rrt.h"
#include "rrt_math.h"

namespace rrt {

// ---- object programs ---------------------------------------------------------------------
// A top-level object (one entry of Environment::objects()) is a post-order program: leaves push a
// hit list, CSG ops pop two lists and push the merged one (csg_object.rs:39-108).  Plain objects
// are one-instruction programs.
enum Op : uint32_t {
    OP_PLANE = 0,
    OP_SPHERE = 1,
    OP_CUBOID = 2,
    OP_QUADRIC = 3,
    OP_MESH = 4,
    OP_CSG_UNION = 5,
    OP_CSG_INTERSECTION = 6,
    OP_CSG_DIFFERENCE = 7,
    OP_BOUND = 8  // conservative box around the object that follows; a wave whose rays all miss it skips `aux` records
};
// One 64-byte record per instruction, primitive parameters inline: the trace loop streams through ONE array with
// wave-uniform sequential addresses (scalar loads, no dependent index -> data chains).
//   plane   p = normal.xyz, d                       aux = plane_cold index
//   sphere  p = centre.xyz, radius
//   cuboid  p = lo.xyz, hi.xyz (with the +-1e-4 slack of cuboid_object.rs:107-112), d of its 6 planes
//                                                   aux = plane_cold index of the first of its 6 planes
//   quadric p = a b c d e f g h i j
//   mesh                                            aux = mesh index
//   bound   p = lo.xyz, hi.xyz (padded), p[6] = origin-relative pad    aux = records to skip on a miss
struct ProgRec {
    uint32_t op;      // Op in bits 0..7 | PF_* flags
    uint32_t object;  // index into Environment::objects()
    int32_t material;
    uint32_t aux;
    float p[12];
};
enum ProgFlags : uint32_t { PF_FIRST = 0x100, PF_LAST = 0x200, PF_CSG = 0x400 };

// winner-only plane data (uv frame, tangent frame); Plane::intersect itself needs only (normal, d)
struct PlaneCold {
    f3 normal;
    float d;
    f3 centre;
    f3 u_unit, v_unit;     // plane_object.rs:96-97, ray independent
    f3 tan_t, tan_b, tan_n;  // columns of to_tangent_space(right, normal), vector.rs:133-142
    int32_t material;
};
// Triangle as the traversal reads it: 64 B, one cache line pair per two triangles
struct TriHot {
    float4 a_nx;  // a.xyz, plane normal x
    float4 b_ny;  // b.xyz, plane normal y
    float4 c_nz;  // c.xyz, plane normal z
    float4 d_pad; // plane d, the triangle's index in its mesh as uint bits (= tri_index: ties go to the lowest, and the walk has it with the record), 0, 0
};
struct MeshInfo {
    uint32_t tri_first, tri_count;
    uint32_t node_first, node_count;
    int32_t material;
    uint32_t smooth;
    uint32_t has_normals;
    uint32_t nrm_first;  // first triangle of this mesh in tri_normals
    // grid of the quantised node boxes (BvhNodeQ): coordinate = qbase + q * qscale, q in 0..65535
    float qbase[3], qscale[3];
    float qmag;     // largest |coordinate| of the grid + its largest extent: scales the traversal's rounding pad
    uint32_t wild;  // 1: a coordinate is not finite or beyond 2^60: the boxes are not used, every node is visited
};
// BVH node as the traversal reads it, 64 B (four 16-byte loads per lane and visit): up to FOUR children's boxes on the
// mesh's 16-bit grid (lower faces rounded down, upper faces rounded up: the decoded box contains the builder's box).
// quantise_bvh collapses the builders' binary trees into these (a node adopts its largest inner children's children), so
// a ray takes about half as many dependent node fetches; leaves and their boxes are the binary tree's.
//   q[4a + c] = lo | hi << 16 of child c on axis a
//   child >= 0 -> inner node (mesh-local index); child < 0 -> leaf: -(child + 1) = first triangle slot << 4 | count;
//   RRT_BVH_EMPTY -> no child in this slot: a leaf of no triangles whose box is inside out (lower face 65535, upper
//   face 0), so it is never visited, and harmless when it is (wild meshes, whose boxes are not used)
#define RRT_BVH_EMPTY ((int32_t)-1)
struct BvhNodeQ {
    uint32_t q[12];
    int32_t child[4];
};
// BVH2 node as the builders emit it (host SAH, device LBVH), 64 B: both children's boxes in f32.
// child < 0  -> leaf: ~child = first triangle slot (mesh-local, in BVH order), count in cnt
struct BvhNode {
    float4 lo_x_hi_x;  // l.lo.x, r.lo.x, l.hi.x, r.hi.x
    float4 lo_y_hi_y;
    float4 lo_z_hi_z;
    int32_t child[2];
    uint32_t cnt[2];
};

// ---- materials ----------------------------------------------------------------------------
struct MatPart {
    int32_t kind;  // rrt_material_kind (never COMPOUND)
    float colour[3];
    float ambient, shininess;
    int32_t diffuse_tex, normal_tex;
    float scale;
    float wr, wt, ior;
};
struct DevMaterial {
    int32_t is_compound;
    int32_t n_parts;
    MatPart parts[RRT_MAX_COMPOUND_PARTS];
    int32_t global_part;  // index of the Global part (always last) or -1
    int32_t nm_part;      // first part with a normal map (compound_material.rs:104-116) or -1
    float w[4];           // behaviour_weight folded as the reference folds it
    float surface_weight; // w[Absorb] + w[Diffuse] + w[Specular]   (photon_scene.rs:318-320)
};
struct DevTexture {
    uint64_t offset;  // texel index into the u8x4 or u16x4 pool
    uint32_t width, height;
    uint32_t maxval;
    uint32_t wide;  // 0: uchar4 pool, 1: ushort4 pool
};
struct DevLight {
    int32_t kind;
    f3 position, direction, intensity;
};

struct DevScene {
    const ProgRec *prog;
    uint32_t n_prog;
    uint32_t n_objects;
    uint32_t n_lights;
    uint32_t has_csg;   // bit 0: some object is a CSG tree; bit 1: some CSG tree has a mesh leaf (the BVH stack then lies behind the hit lists in LDS)
    uint32_t has_bounds;
    uint32_t no_solids;  // 1: no record is a sphere, a quadric or a cuboid (planes and meshes only): see FEAT_NO_SOLIDS
    uint32_t n_plane_cold, n_meshes, n_materials;  // array sizes, for the device-side index guards
    const PlaneCold *plane_cold;
    const MeshInfo *meshes;
    const TriHot *tris;        // BVH order
    const uint32_t *tri_index; // BVH order -> original triangle index
    const float *tri_normals;  // [orig tri][3][3], only for smooth meshes
    const BvhNodeQ *nodes;
    // Meshes that are CSG leaves only: [2 m], [2 m + 1] = first, count of mesh m's entries behind the 2 * n_meshes header
    // words; an entry is the BVH-order slot of a triangle whose same-side test (triangle_object.rs:137-147) passes for
    // Vertex::zero(), the position of the plane's Hit::infinity (hit.rs:37-52): see leaf_list in dev_trace.h
    const uint32_t *mesh_inf;
    const DevMaterial *materials;
    const DevTexture *textures;
    const uchar4 *texels8;
    const ushort4 *texels16;
    const DevLight *lights;
};

// ---- wavefront queue records -----------------------------------------------------------------
// One queue per recursion level of Scene::raytrace (scene.rs:83-128, depth 0..5).
struct LevelBuffers {
    float4 *ray_o;   // origin.xyz, w = parent index (level 0: pixel index) as uint bits
    float4 *ray_d;   // direction.xyz, w = child slot (0 reflection, 1 refraction) as uint bits
    float4 *once;    // fold of the non-Global parts' compute_once, w = kr
    float4 *glob;    // wr, wt, depth (hit.distance; 0 for a miss), flags | compound divisor << NF_PARTS_SHIFT (n_parts, 0 = not compound)
    float4 *lights;  // [n_lights][capacity]: per-light term already multiplied by intensity; w = lit flag
    float4 *result;  // levels >= 1, written by k_combine for the parents: colour.rgb, depth
    int2 *child;     // queue index of the reflection / refraction child at level+1, or -1
    uint32_t *object_id;
    uint32_t capacity;
};
enum NodeFlags : uint32_t { NF_HIT = 1, NF_HAS_GLOBAL = 2, NF_GLOBAL_ACTIVE = 4 /* depth < max_depth */ };
#define NF_PARTS_SHIFT 8

}  // namespace rrt
