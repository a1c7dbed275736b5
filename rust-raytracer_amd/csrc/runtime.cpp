// runtime.cpp — scene flattening into HBM, the wavefront render loop, and the C ABI of include/rrt.h.
#include "runtime.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <thread>

using namespace rrt;

rrt_scene::~rrt_scene() {
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
    if (host_pinned) (void)hipHostFree(host_pinned);
    for (auto &L : lanes) {
        if (L->stream) (void)hipStreamDestroy(L->stream);
        if (L->ev_trace0) (void)hipEventDestroy(L->ev_trace0);
        if (L->ev_trace1) (void)hipEventDestroy(L->ev_trace1);
        if (L->ev_done) (void)hipEventDestroy(L->ev_done);
        if (L->host_pinned && L->owns_pinned) (void)hipHostFree(L->host_pinned);
        if (L->stripe_pinned && L->owns_stripe_pinned) (void)hipHostFree(L->stripe_pinned);
    }
}

namespace rrt {

int LevelStorage::ensure(uint32_t n, uint32_t n_lights, bool need_rays) {
    if (n == 0) n = 1;
    uint32_t cap = capacity;
    if (n > capacity) cap = n + n / 8 + 256;
    int rc = RRT_OK;
    // every array is checked on every call: the ray arrays are only needed by levels that read stored rays, and a
    // level first used with camera-generated rays may later be fed explicit ones (rrt_raytrace)
    if (need_rays) {
        if ((rc = ray_o.ensure(cap))) return rc;
        if ((rc = ray_d.ensure(cap))) return rc;
    }
    if ((rc = once.ensure(cap))) return rc;
    if ((rc = glob.ensure(cap))) return rc;
    if ((rc = lights.ensure((size_t)cap * std::max(1u, n_lights)))) return rc;
    if ((rc = result.ensure(cap))) return rc;
    if ((rc = child.ensure(cap))) return rc;
    if ((rc = object_id.ensure(cap))) return rc;
    capacity = cap;
    return RRT_OK;
}
LevelBuffers LevelStorage::view() const {
    LevelBuffers v;
    v.ray_o = ray_o.p;
    v.ray_d = ray_d.p;
    v.once = once.p;
    v.glob = glob.p;
    v.lights = lights.p;
    v.result = result.p;
    v.child = child.p;
    v.object_id = object_id.p;
    v.capacity = capacity;
    return v;
}
DevPhotonMap PhotonIndex::view() const {
    DevPhotonMap m;
    m.pos_cell = pos.p;
    m.incident = dir.p;
    m.intensity = rgb.p;
    m.keys = keys.p;
    m.ranges = ranges.p;
    m.table_mask = table_size ? table_size - 1 : 0;
    m.count = (uint32_t)count;
    m.inv_cell = inv_cell;
    m.dense_start = dense_start.p;
    for (int g = 0; g < RRT_DENSE_GRIDS; g++) m.grid[g] = grids[g];
    m.n_grids = n_grids;
    m.n_dense = n_dense;
    m.inv_dcell = inv_dcell;
    m.meta = meta.p;
    m.agg_dir = agg_dir.p;
    m.agg_rgb = agg_rgb.p;
    m.div = div;
    return m;
}

// ---- flattening ---------------------------------------------------------------------------------
namespace {

struct Flat {
    std::vector<ProgRec> prog;
    std::vector<PlaneCold> plane_cold;
    std::vector<MeshInfo> meshes;
    std::vector<TriHot> tris;
    std::vector<uint32_t> tri_index;
    std::vector<float> tri_normals;
    std::vector<BvhNodeQ> nodes;
    std::vector<std::vector<uint32_t>> mesh_inf;  // per mesh: triangle slots whose prism holds the world origin (CSG leaves only)
    // The scene's triangle / node / normal arrays are put together from segments at upload time: `tris`, `tri_index`, `nodes`
    // above hold the meshes that were flattened on the host, back to back; a mesh that was built on the device keeps its
    // arrays there (DeviceMesh) and never comes back; vertex normals go up straight from the objects' own arrays.
    struct Seg {
        size_t at, count;       // place in the scene's array
        size_t host_at;         // first element in the host vector (host segments)
        const DeviceMesh *dev;  // or the device mesh that holds it
        const float *normals;   // normals segments: the object's array (9 floats per triangle)
    };
    std::vector<Seg> tri_segs, node_segs, nrm_segs;
    std::vector<std::unique_ptr<DeviceMesh>> dev_meshes;
    size_t n_tris = 0, n_nodes = 0, n_nrm_tris = 0;  // totals over all segments
    uint32_t csg_nesting = 0;                     // > 0 while emit is inside a CSG tree
    uint32_t bvh_depth = 0;
    uint64_t flops = 0;
    uint32_t bvh_builder = RRT_BVH_HOST_SAH;
    bool have_device = false;  // RRT_BVH_AUTO builds large meshes on the device only where there is one (flattening for tools/cpu_emu needs none)
    int error = RRT_OK;       // first failure inside emit (device BVH build)
    double ms_bvh_build = 0;  // host wall clock
};

uint32_t add_plane(Flat &F, f3 point, f3 up, f3 normal, float d, int material) {
    uint32_t idx = (uint32_t)F.plane_cold.size();
    PlaneCold c;
    c.normal = normal;
    c.d = d;
    c.centre = point;
    c.v_unit = norm3(up);                         // plane_object.rs:96
    c.u_unit = norm3(cross3(c.v_unit, normal));   // plane_object.rs:97
    f3 right = norm3(cross3(normal, up));         // plane_object.rs:103
    c.tan_t = norm3(right);                       // vector.rs:134-136
    c.tan_n = norm3(normal);
    c.tan_b = cross3(c.tan_n, c.tan_t);
    c.material = material;
    F.plane_cold.push_back(c);
    return idx;
}

// Conservative box for the skip test of dev_trace.h "object bounds".  finite == false: never skipped.
struct Box {
    bool finite = false;
    float lo[3], hi[3];
    float area() const {
        float x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x * y + y * z + z * x;
    }
};
bool same_box(const Box &a, const Box &b) {
    if (a.finite != b.finite) return false;
    for (int k = 0; k < 3; k++)
        if (a.lo[k] != b.lo[k] || a.hi[k] != b.hi[k]) return false;
    return true;
}
Box bound_of(const HostObject &o) {
    Box b;
    switch (o.kind) {
        case RRT_OBJ_CUBOID: {  // hits must pass the inside test of cuboid_object.rs:107-112
            const float c[3] = {o.p0.x, o.p0.y, o.p0.z}, sz[3] = {o.p1.x, o.p1.y, o.p1.z};
            b.finite = true;
            for (int a = 0; a < 3; a++) {
                b.lo[a] = c[a] - 0.0001f;
                b.hi[a] = c[a] + sz[a] + 0.0001f;
                if (!std::isfinite(b.lo[a]) || !std::isfinite(b.hi[a])) b.finite = false;
            }
            break;
        }
        case RRT_OBJ_CSG: {
            Box l = bound_of(*o.left), r = bound_of(*o.right);
            if (o.csg_mode == RRT_CSG_UNION) {  // no hits unless one operand has hits
                if (l.finite && r.finite) {
                    b.finite = true;
                    for (int a = 0; a < 3; a++) {
                        b.lo[a] = std::min(l.lo[a], r.lo[a]);
                        b.hi[a] = std::max(l.hi[a], r.hi[a]);
                    }
                }
            } else if (o.csg_mode == RRT_CSG_INTERSECTION) {  // no hits unless BOTH operands have hits: either box will do
                if (l.finite && r.finite)
                    b = l.area() <= r.area() ? l : r;
                else
                    b = l.finite ? l : r;
            } else {  // difference: no hits unless the left operand has hits
                b = l;
            }
            break;
        }
        default: break;  // planes and quadrics: unbounded or not worth a test; meshes: their BVH root does this
    }
    return b;
}

ProgRec &push_rec(Flat &F, uint32_t op, int material, uint32_t aux) {
    ProgRec r;
    memset(&r, 0, sizeof(r));
    r.op = op;
    r.material = material;
    r.aux = aux;
    F.prog.push_back(r);
    return F.prog.back();
}

// OP_BOUND record for `bb`; the caller fills aux (records to skip) once they are emitted
int push_bound(Flat &F, const Box &bb, uint32_t flags) {
    int at = (int)F.prog.size();
    ProgRec &r = push_rec(F, OP_BOUND | flags, -1, 0);
    for (int a = 0; a < 3; a++) {
        // absolute pad: rounding of o + d*t at the box's own coordinate magnitude, with a wide margin
        float mag = std::max(std::fabs(bb.lo[a]), std::fabs(bb.hi[a])) + std::fabs(bb.hi[a] - bb.lo[a]);
        float pad = 1e-4f * mag + 1e-4f;
        r.p[a] = bb.lo[a] - pad;
        r.p[3 + a] = bb.hi[a] + pad;
    }
    r.p[6] = 3.2e-5f;  // pad per unit of |origin|: rounding of o + d*t for far origins (error <~ 1e-6 |o|)
    return at;
}

static bool g_bounds_enabled = true;

void emit(Flat &F, const HostObject &o, const Box &enclosing) {
    switch (o.kind) {
        case RRT_OBJ_PLANE: {
            uint32_t idx = add_plane(F, o.p0, o.p1, o.p2, o.d, o.material);
            ProgRec &r = push_rec(F, OP_PLANE, o.material, idx);
            r.p[0] = o.p2.x, r.p[1] = o.p2.y, r.p[2] = o.p2.z, r.p[3] = o.d;
            F.flops += 18;
            break;
        }
        case RRT_OBJ_SPHERE: {
            ProgRec &r = push_rec(F, OP_SPHERE, o.material, 0);
            r.p[0] = o.p0.x, r.p[1] = o.p0.y, r.p[2] = o.p0.z, r.p[3] = o.radius;
            F.flops += 35;
            break;
        }
        case RRT_OBJ_CUBOID: {  // cuboid_object.rs:58-90, plane order = CuboidPlanes::iter (:27-37)
            f3 corner = o.p0, size = o.p1;
            f3 fdl = corner;
            f3 ful = corner + mk3(0.0f, size.y, 0.0f);
            f3 bdl = corner + mk3(0.0f, 0.0f, size.z);
            f3 bdr = corner + mk3(size.x, 0.0f, size.z);
            f3 up = mk3(0, 1, 0), down = mk3(0, -1, 0), left = mk3(-1, 0, 0), right = mk3(1, 0, 0), fw = mk3(0, 0, 1), bw = mk3(0, 0, -1);
            struct {
                f3 pt, up, n;
            } pl[6] = {{fdl, up, left}, {bdr, up, right}, {ful, fw, up}, {fdl, fw, down}, {fdl, up, bw}, {bdl, down, fw}};
            uint32_t first = 0;
            float dd[6];
            for (int k = 0; k < 6; k++) {
                dd[k] = -dot3(pl[k].n, pl[k].pt);
                uint32_t idx = add_plane(F, pl[k].pt, pl[k].up, pl[k].n, dd[k], o.material);
                if (k == 0) first = idx;
            }
            ProgRec &r = push_rec(F, OP_CUBOID, o.material, first);
            r.p[0] = corner.x - 0.0001f, r.p[1] = corner.y - 0.0001f, r.p[2] = corner.z - 0.0001f;
            r.p[3] = corner.x + size.x + 0.0001f, r.p[4] = corner.y + size.y + 0.0001f, r.p[5] = corner.z + size.z + 0.0001f;
            for (int k = 0; k < 6; k++) r.p[6 + k] = dd[k];
            F.flops += 6 * 18 + 6 * 12;
            break;
        }
        case RRT_OBJ_QUADRATIC: {
            ProgRec &r = push_rec(F, OP_QUADRIC, o.material, 0);
            for (int k = 0; k < 10; k++) r.p[k] = o.q[k];
            F.flops += 95;
            break;
        }
        case RRT_OBJ_MESH: {
            // a leaf code holds its first triangle slot in 27 bits (BvhNodeQ, bvh_build.cpp), whichever builder made the tree
            if (o.tri_pos.size() / 9 >= (1u << 27)) {
                if (!F.error) F.error = fail(RRT_ERR_LIMIT, "a mesh with 2^27 or more triangles");
                break;
            }
            uint32_t n = (uint32_t)(o.tri_pos.size() / 9);
            MeshInfo mi;
            mi.tri_first = (uint32_t)F.n_tris;
            mi.tri_count = n;
            mi.node_first = (uint32_t)F.n_nodes;
            mi.material = o.material;
            mi.smooth = o.smooth ? 1u : 0u;
            mi.has_normals = o.tri_nrm.empty() ? 0u : 1u;
            mi.nrm_first = (uint32_t)F.n_nrm_tris;
            std::vector<uint32_t> order;
            std::vector<BvhNode> nodes;
            uint32_t depth = 0;
            auto t0 = std::chrono::steady_clock::now();
            bool built = false, on_device = false;
            int rc = RRT_OK;
            uint32_t stack_need = 0;  // traversal stack words a ray of this mesh can need (4-wide tree: more than its depth)
            const bool timing = getenv("RRT_LBVH_TIMING") != nullptr;  // debugging aid: the phases after the build
            if (F.bvh_builder == RRT_BVH_DEVICE_LBVH && n > 4) {
                rc = build_bvh_device(o.tri_pos.data(), n, order, nodes, depth, nullptr);
                built = true;
            } else if ((F.bvh_builder == RRT_BVH_DEVICE_SAH && n > 4) || (F.bvh_builder == RRT_BVH_AUTO && n >= 32768 && F.have_device)) {
                // The whole mesh on the device: tree, four-wide collapse + quantisation, triangle records (bvh_sah.hip).  A mesh
                // inside a CSG tree needs its triangles on the host once more (mesh_inf below): it takes the builder alone.
                std::unique_ptr<DeviceMesh> dm;
                if (!F.csg_nesting && !getenv("RRT_HOST_FLATTEN")) {
                    dm = std::make_unique<DeviceMesh>();
                    rc = build_mesh_device_sah(o.tri_pos.data(), n, mi, *dm, stack_need, nullptr);
                } else {
                    rc = build_bvh_device_sah(o.tri_pos.data(), n, order, nodes, depth, nullptr);
                }
                built = rc != 1;  // 1: a coordinate is not finite: the host builder and its "wild mesh" rules take the mesh
                // RRT_BVH_AUTO promises a tree, not a device build: whatever the device builder refuses (its workspace does not
                // fit, a tree deeper than its level limit) goes to the host builder, which makes the same tree
                // (RRT_ERR_LIMIT only: a HIP error or a failed self-check of the device builder is reported, not papered over)
                if (rc == RRT_ERR_LIMIT && F.bvh_builder == RRT_BVH_AUTO) built = false;
                if (!built) rc = RRT_OK;
                if (built && !rc && dm) {
                    on_device = true;
                    F.dev_meshes.push_back(std::move(dm));
                }
            }
            if (rc) {
                if (!F.error) F.error = rc;
                break;
            }
            if (!built) {
                build_bvh(o.tri_pos.data(), n, order, nodes, depth);
            }
            F.ms_bvh_build += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            auto t_phase = std::chrono::steady_clock::now();
            auto lap = [&](const char *what) {
                if (!timing) return;
                auto t = std::chrono::steady_clock::now();
                fprintf(stderr, "[flatten] %-26s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_phase).count());
                t_phase = t;
            };
            (void)depth;
            F.mesh_inf.emplace_back();
            if (on_device) {
                const DeviceMesh *dm = F.dev_meshes.back().get();
                mi.node_count = dm->n_nodes;
                F.node_segs.push_back({F.n_nodes, dm->n_nodes, 0, dm, nullptr});
                F.tri_segs.push_back({F.n_tris, n, 0, dm, nullptr});
            } else {
                std::vector<BvhNodeQ> qnodes;
                quantise_bvh(nodes, mi, qnodes, stack_need);
                lap("quantise + collapse");
                mi.node_count = (uint32_t)qnodes.size();
                F.node_segs.push_back({F.n_nodes, qnodes.size(), F.nodes.size(), nullptr, nullptr});
                F.nodes.insert(F.nodes.end(), qnodes.begin(), qnodes.end());
                const size_t tri_base = F.tris.size();
                F.tri_segs.push_back({F.n_tris, n, tri_base, nullptr, nullptr});
                F.tris.resize(tri_base + n);
                F.tri_index.resize(tri_base + n);
                auto tri_records = [&](uint32_t k0, uint32_t k1) {
                    for (uint32_t k = k0; k < k1; k++) {
                        const float *p = o.tri_pos.data() + (size_t)order[k] * 9;
                        f3 a = mk3(p[0], p[1], p[2]), b = mk3(p[3], p[4], p[5]), c = mk3(p[6], p[7], p[8]);
                        f3 ab = b - a, bc = c - b;                  // triangle_object.rs:51-52
                        f3 nrm = norm3(cross3(ab, bc));             // triangle_object.rs:85-89
                        float d = -dot3(nrm, a);                    // plane_object.rs:37
                        TriHot t;
                        t.a_nx = make_float4(a.x, a.y, a.z, nrm.x);
                        t.b_ny = make_float4(b.x, b.y, b.z, nrm.y);
                        t.c_nz = make_float4(c.x, c.y, c.z, nrm.z);
                        float index_bits;
                        memcpy(&index_bits, &order[k], sizeof(index_bits));
                        t.d_pad = make_float4(d, index_bits, 0.0f, 0.0f);
                        F.tris[tri_base + k] = t;
                        F.tri_index[tri_base + k] = order[k];
                    }
                };
                {  // independent records: slices on up to 16 host threads for large meshes (26 -> 3 ms for 1.07 M triangles)
                    const unsigned hw = std::thread::hardware_concurrency();
                    const uint32_t n_threads = n >= 65536 ? std::min<uint32_t>(hw ? hw : 4u, 16u) : 1u;
                    const uint32_t per = (n + n_threads - 1) / n_threads;
                    std::vector<std::thread> th;
                    for (uint32_t t = 1; t < n_threads; t++)
                        th.emplace_back(tri_records, std::min(n, t * per), std::min(n, (t + 1) * per));
                    tri_records(0, std::min(n, per));
                    for (auto &t : th) t.join();
                }
                lap("triangle records");
                if (F.csg_nesting) {
                    // A plane's Hit::infinity carries Vertex::zero() as its position (hit.rs:37-52) and Triangle::intersect puts
                    // it through the same-side test like any other hit (triangle_object.rs:134-150): whether it passes depends on
                    // the triangle alone.  Only a CSG merge can see such a hit (select_first_hit wants distance < f32::MAX).
                    for (uint32_t k = 0; k < n; k++) {
                        const TriHot &t = F.tris[tri_base + k];
                        f3 a = mk3(t.a_nx.x, t.a_nx.y, t.a_nx.z), b = mk3(t.b_ny.x, t.b_ny.y, t.b_ny.z), c = mk3(t.c_nz.x, t.c_nz.y, t.c_nz.z);
                        f3 z = mk3(0.0f, 0.0f, 0.0f);
                        f3 ai = z - a, bi = z - b, ci = z - c;
                        f3 ab = b - a, bc = c - b, ca = a - c;
                        f3 ab_n = cross3(ai, ab), bc_n = cross3(bi, bc), ca_n = cross3(ci, ca);
                        if (dot3(ab_n, bc_n) > 0.0f && dot3(bc_n, ca_n) > 0.0f) F.mesh_inf.back().push_back(k);
                    }
                }
            }
            F.bvh_depth = std::max(F.bvh_depth, stack_need);
            if (mi.node_count >= (1u << 26)) {  // the traversal addresses a mesh's nodes by a 32-bit byte offset
                if (!F.error) F.error = fail(RRT_ERR_LIMIT, "a mesh with 2^26 or more BVH nodes");
                break;
            }
            F.n_nodes += mi.node_count;
            F.n_tris += n;
            if (!o.tri_nrm.empty()) {
                F.nrm_segs.push_back({F.n_nrm_tris * 9, o.tri_nrm.size(), 0, nullptr, o.tri_nrm.data()});
                F.n_nrm_tris += o.tri_nrm.size() / 9;
            }
            push_rec(F, OP_MESH, o.material, (uint32_t)F.meshes.size());
            F.meshes.push_back(mi);
            break;
        }
        case RRT_OBJ_CSG: {
            // sub-trees that are CSG nodes themselves get their own bound when it is tighter than the enclosing one
            const HostObject *kids[2] = {o.left.get(), o.right.get()};
            F.csg_nesting++;
            for (const HostObject *kid : kids) {
                Box kb = bound_of(*kid);
                bool own = g_bounds_enabled && kid->kind == RRT_OBJ_CSG && kb.finite && !same_box(kb, enclosing);
                int at = own ? push_bound(F, kb, PF_CSG) : -1;
                uint32_t first = (uint32_t)F.prog.size();
                emit(F, *kid, own ? kb : enclosing);
                if (own) F.prog[at].aux = (uint32_t)F.prog.size() - first;
            }
            F.csg_nesting--;
            push_rec(F, (uint32_t)(OP_CSG_UNION + o.csg_mode), -1, 0);
            break;
        }
    }
}

int flatten_materials(rrt_scene *s, std::vector<DevMaterial> &out, bool &reflective) {
    reflective = false;
    auto part_of = [](const rrt_material_desc &d) {
        MatPart p;
        memset(&p, 0, sizeof(p));
        p.kind = d.kind;
        for (int i = 0; i < 3; i++) p.colour[i] = d.colour[i];
        p.ambient = d.ambient;
        p.shininess = d.shininess;
        p.diffuse_tex = d.diffuse_tex;
        p.normal_tex = d.normal_tex;
        p.scale = d.scale;
        p.wr = d.reflect_weight;
        p.wt = d.refract_weight;
        p.ior = d.ior;
        return p;
    };
    for (const rrt_material_desc &d : s->materials) {
        DevMaterial m;
        memset(&m, 0, sizeof(m));
        m.global_part = -1;
        m.nm_part = -1;
        if (d.kind == RRT_MAT_COMPOUND) {
            m.is_compound = 1;
            m.n_parts = d.n_parts;
            for (int i = 0; i < d.n_parts; i++) m.parts[i] = part_of(s->materials[d.parts[i]]);
        } else {
            m.is_compound = 0;
            m.n_parts = 1;
            m.parts[0] = part_of(d);
        }
        float w[4] = {0, 0, 0, 0};
        for (int i = 0; i < m.n_parts; i++) {
            const MatPart &p = m.parts[i];
            if (p.kind == RRT_MAT_GLOBAL) {
                m.global_part = i;
                if (p.wr > 0.0f || p.wt > 0.0f) reflective = true;
            }
            if (p.kind == RRT_MAT_TEXTURE && p.normal_tex >= 0 && m.nm_part < 0) m.nm_part = i;
            // material.rs:54-61 / global_material.rs:116-123
            float pw[4] = {0.6f, 0.3f, 0.1f, 0.0f};
            if (p.kind == RRT_MAT_GLOBAL) {
                pw[0] = pw[1] = pw[2] = 0.0f;
                pw[3] = 1.0f;
            }
            for (int b = 0; b < 4; b++) w[b] = m.is_compound ? w[b] + pw[b] : pw[b];  // compound_material.rs:124-128
        }
        for (int b = 0; b < 4; b++) m.w[b] = m.is_compound ? w[b] / (float)m.n_parts : w[b];
        m.surface_weight = m.w[0] + m.w[1] + m.w[2];  // photon_scene.rs:318-320
        out.push_back(m);
    }
    return RRT_OK;
}

}  // namespace

int flatten_scene(rrt_scene *s) {
    drop_render_workers(s);  // they point into the device arrays replaced below
    Flat F;
    if (s->params.bvh_builder > RRT_BVH_AUTO) return fail(RRT_ERR_INVALID, "unknown bvh_builder");
    F.bvh_builder = s->params.bvh_builder;
    {
        int ndev_probe = 0;
        F.have_device = hipGetDeviceCount(&ndev_probe) == hipSuccess && ndev_probe > 0;
    }
    g_bounds_enabled = getenv("RRT_NO_BOUNDS") == nullptr;  // debugging aid: the reference's plain all-objects loop
    uint32_t max_stack = 0;
    bool has_csg = false, has_bounds = false, csg_mesh = false;
    for (size_t oi = 0; oi < s->objects.size(); oi++) {
        const HostObject &ho = *s->objects[oi];
        Box bb = bound_of(ho);
        if (!g_bounds_enabled) bb.finite = false;
        int bound_rec = -1;
        if (bb.finite) {
            bound_rec = push_bound(F, bb, 0);
            has_bounds = true;
        }
        uint32_t first = (uint32_t)F.prog.size();
        emit(F, ho, bb);
        if (F.error) return F.error;  // rrt_last_error holds the message
        uint32_t count = (uint32_t)F.prog.size() - first;
        if (bound_rec >= 0) F.prog[bound_rec].aux = count;
        for (uint32_t k = 0; k < count; k++) F.prog[first + k].object = (uint32_t)oi;
        F.prog[first].op |= PF_FIRST;
        F.prog[first + count - 1].op |= PF_LAST;
        if (count > 1) {
            has_csg = true;
            if (ho.contains_mesh()) csg_mesh = true;  // a mesh leaf: its BVH is walked while hit lists are live
            uint32_t sp = 0;
            for (uint32_t k = 0; k < count; k++) {
                ProgRec &r = F.prog[first + k];
                r.op |= PF_CSG;
                if ((r.op & 0xFFu) == OP_BOUND) continue;
                if ((r.op & 0xFFu) < OP_CSG_UNION)
                    sp++;
                else
                    sp--;
                max_stack = std::max(max_stack, sp);
            }
        }
    }
    for (const ProgRec &r : F.prog)  // sub-tree bounds inside a CSG program count too, not only whole-object bounds
        if ((r.op & 0xFFu) == OP_BOUND) has_bounds = true;
    std::vector<uint32_t> mesh_inf(F.meshes.size() * 2, 0u);
    for (size_t m = 0; m < F.mesh_inf.size(); m++) {
        mesh_inf[2 * m] = (uint32_t)(mesh_inf.size() - 2 * F.meshes.size());
        mesh_inf[2 * m + 1] = (uint32_t)F.mesh_inf[m].size();
        mesh_inf.insert(mesh_inf.end(), F.mesh_inf[m].begin(), F.mesh_inf[m].end());
    }
    if (getenv("RRT_DUMP_PROGRAM")) {  // debugging aid: the record stream, without touching the device
        static const char *names[] = {"plane", "sphere", "cuboid", "quadric", "mesh", "union", "intersection", "difference", "bound"};
        for (size_t k = 0; k < F.prog.size(); k++) {
            const ProgRec &r = F.prog[k];
            fprintf(stderr, "%3zu obj %2u %-12s%s%s%s aux %u\n", k, r.object, names[r.op & 0xFFu], (r.op & PF_FIRST) ? " FIRST" : "",
                    (r.op & PF_LAST) ? " LAST" : "", (r.op & PF_CSG) ? " CSG" : "", r.aux);
        }
        fprintf(stderr, "max_stack %u\n", max_stack);
    }
    if (const char *path = getenv("RRT_DUMP_SCENE")) {  // debugging aid: geometry arrays for tools/cpu_emu (no device needed)
        if (FILE *f = fopen(path, "wb")) {
            auto put = [&](const void *p, size_t elem, size_t n) {
                uint64_t hdr[2] = {elem, n};
                fwrite(hdr, sizeof(hdr), 1, f);
                if (n) fwrite(p, elem, n, f);
            };
            uint32_t misc[4] = {has_csg ? max_stack + 1 : 0u, F.bvh_depth ? F.bvh_depth + 1 : 0u, has_bounds ? 1u : 0u, (has_csg ? 1u : 0u) | (csg_mesh ? 2u : 0u)};
            put(misc, 4, 4);
            put(F.prog.data(), sizeof(ProgRec), F.prog.size());
            put(F.plane_cold.data(), sizeof(PlaneCold), F.plane_cold.size());
            put(F.meshes.data(), sizeof(MeshInfo), F.meshes.size());
            // the arrays as the device will hold them: host segments copied, device segments (meshes flattened on the GPU) read back
            std::vector<TriHot> tris(F.n_tris);
            std::vector<uint32_t> tri_index(F.n_tris);
            std::vector<float> tri_normals(F.n_nrm_tris * 9);
            std::vector<BvhNodeQ> nodes(F.n_nodes);
            for (const Flat::Seg &g : F.tri_segs) {
                if (g.dev) {
                    (void)hipMemcpy(tris.data() + g.at, g.dev->tris.p, g.count * sizeof(TriHot), hipMemcpyDeviceToHost);
                    (void)hipMemcpy(tri_index.data() + g.at, g.dev->tri_index.p, g.count * sizeof(uint32_t), hipMemcpyDeviceToHost);
                } else {
                    std::copy(F.tris.begin() + g.host_at, F.tris.begin() + g.host_at + g.count, tris.begin() + g.at);
                    std::copy(F.tri_index.begin() + g.host_at, F.tri_index.begin() + g.host_at + g.count, tri_index.begin() + g.at);
                }
            }
            for (const Flat::Seg &g : F.node_segs) {
                if (g.dev)
                    (void)hipMemcpy(nodes.data() + g.at, g.dev->qnodes.p, g.count * sizeof(BvhNodeQ), hipMemcpyDeviceToHost);
                else
                    std::copy(F.nodes.begin() + g.host_at, F.nodes.begin() + g.host_at + g.count, nodes.begin() + g.at);
            }
            for (const Flat::Seg &g : F.nrm_segs) std::copy(g.normals, g.normals + g.count, tri_normals.begin() + g.at);
            put(tris.data(), sizeof(TriHot), tris.size());
            put(tri_index.data(), 4, tri_index.size());
            put(tri_normals.data(), 4, tri_normals.size());
            put(nodes.data(), sizeof(BvhNodeQ), nodes.size());
            put(mesh_inf.data(), 4, mesh_inf.size());
            fclose(f);
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(RRT_ERR_DEVICE, "no HIP device: librrt has no CPU fallback");
    RRT_HIP(hipGetDevice(&s->device));  // the caller's current device owns this scene from here on
    // the hit lists keep the record index of each hit in 16 bits (mk_meta in dev_trace.h)
    if (has_csg && F.prog.size() > 0xFFFF) return fail(RRT_ERR_LIMIT, "scene with CSG objects has more than 65535 primitive records");
    DeviceScene &D = s->dev;
    D.lds_slots = has_csg ? max_stack + 1 : 0;
    if (D.lds_slots > 8) return fail(RRT_ERR_LIMIT, "CSG tree needs more than 7 live hit lists");
    D.bvh_stack = F.bvh_depth ? F.bvh_depth + 1 : 0;
    size_t list_bytes = (size_t)D.lds_slots * 6 /* HitVec capacity */ * 256 * 8;
    size_t stack_bytes = (size_t)((D.bvh_stack + 1) / 2) * 256 * 8;  // two stack words per 8-byte column element (dev_trace.h Lds)
    // the stack aliases the hit lists (a lane never needs both at once) unless a CSG tree has a mesh leaf
    D.lds_bytes = csg_mesh ? list_bytes + stack_bytes : std::max(list_bytes, stack_bytes);
    if (D.lds_bytes > 160 * 1024) return fail(RRT_ERR_LIMIT, "per-block LDS need exceeds 160 KiB (BVH too deep)");
    D.flops_per_trace = F.flops;
    if (getenv("RRT_BVH_INFO"))
        fprintf(stderr, "rrt: %zu meshes, %zu triangles, %zu BVH nodes, depth %u, LDS %zu bytes per block\n", F.meshes.size(), F.n_tris,
                F.n_nodes, F.bvh_depth, D.lds_bytes);

    std::vector<DevMaterial> mats;
    int rc = flatten_materials(s, mats, D.reflective);
    if (rc) return rc;

    // textures: one uchar4 pool and one ushort4 pool
    std::vector<DevTexture> texs;
    std::vector<uchar4> t8;
    std::vector<ushort4> t16;
    for (const HostTexture &t : s->textures) {
        DevTexture d;
        d.width = t.width;
        d.height = t.height;
        d.maxval = t.maxval;
        size_t n = (size_t)t.width * t.height;
        if (t.maxval <= 255) {
            d.wide = 0;
            d.offset = t8.size();
            t8.resize(t8.size() + n);
            for (size_t i = 0; i < n; i++) t8[d.offset + i] = make_uchar4(t.samples[i * 3], t.samples[i * 3 + 1], t.samples[i * 3 + 2], 255);
        } else {
            d.wide = 1;
            d.offset = t16.size();
            t16.resize(t16.size() + n);
            for (size_t i = 0; i < n; i++) {
                const uint8_t *p = &t.samples[i * 6];
                t16[d.offset + i] = make_ushort4((uint16_t)((p[0] << 8) | p[1]), (uint16_t)((p[2] << 8) | p[3]), (uint16_t)((p[4] << 8) | p[5]), 0);
            }
        }
        texs.push_back(d);
    }
    std::vector<DevLight> lights;
    for (const rrt_light_desc &l : s->lights) {
        DevLight d;
        d.kind = l.kind;
        d.position = mk3(l.position[0], l.position[1], l.position[2]);
        d.direction = mk3(l.direction[0], l.direction[1], l.direction[2]);
        d.intensity = mk3(l.colour[0], l.colour[1], l.colour[2]);
        lights.push_back(d);
    }

#define UP(field, vec)                      \
    if ((rc = D.field.upload(vec))) return rc;
    UP(prog, F.prog)
    UP(plane_cold, F.plane_cold)
    UP(meshes, F.meshes)
    // triangles, nodes, normals: segment by segment — from the host vectors, from the meshes that were flattened on the device
    // (device to device), from the objects' own normal arrays
    if ((rc = D.tris.ensure(std::max<size_t>(F.n_tris, 1))) || (rc = D.tri_index.ensure(std::max<size_t>(F.n_tris, 1))) ||
        (rc = D.tri_normals.ensure(std::max<size_t>(F.n_nrm_tris * 9, 1))) || (rc = D.nodes.ensure(std::max<size_t>(F.n_nodes, 1))))
        return rc;
    for (const Flat::Seg &g : F.tri_segs) {
        if (!g.count) continue;
        if (g.dev) {
            RRT_HIP(hipMemcpyAsync(D.tris.p + g.at, g.dev->tris.p, g.count * sizeof(TriHot), hipMemcpyDeviceToDevice, nullptr));
            RRT_HIP(hipMemcpyAsync(D.tri_index.p + g.at, g.dev->tri_index.p, g.count * sizeof(uint32_t), hipMemcpyDeviceToDevice, nullptr));
        } else {
            RRT_HIP(hipMemcpy(D.tris.p + g.at, F.tris.data() + g.host_at, g.count * sizeof(TriHot), hipMemcpyHostToDevice));
            RRT_HIP(hipMemcpy(D.tri_index.p + g.at, F.tri_index.data() + g.host_at, g.count * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }
    for (const Flat::Seg &g : F.node_segs) {
        if (!g.count) continue;
        if (g.dev)
            RRT_HIP(hipMemcpyAsync(D.nodes.p + g.at, g.dev->qnodes.p, g.count * sizeof(BvhNodeQ), hipMemcpyDeviceToDevice, nullptr));
        else
            RRT_HIP(hipMemcpy(D.nodes.p + g.at, F.nodes.data() + g.host_at, g.count * sizeof(BvhNodeQ), hipMemcpyHostToDevice));
    }
    for (const Flat::Seg &g : F.nrm_segs)
        if (g.count) RRT_HIP(hipMemcpy(D.tri_normals.p + g.at, g.normals, g.count * sizeof(float), hipMemcpyHostToDevice));
    RRT_HIP(hipDeviceSynchronize());  // the device meshes are released when F goes
    UP(mesh_inf, mesh_inf)
    UP(materials, mats)
    UP(textures, texs)
    UP(texels8, t8)
    UP(texels16, t16)
    UP(lights, lights)
#undef UP
    DevScene &V = D.view;
    V.prog = D.prog.p;
    V.n_prog = (uint32_t)F.prog.size();
    V.n_objects = (uint32_t)s->objects.size();
    V.n_lights = (uint32_t)lights.size();
    V.has_csg = (has_csg ? 1u : 0u) | (csg_mesh ? 2u : 0u);
    V.has_bounds = has_bounds ? 1 : 0;
    V.no_solids = 1;
    for (const ProgRec &r : F.prog) {
        const uint32_t op = r.op & 0xFFu;
        if (op == OP_SPHERE || op == OP_QUADRIC || op == OP_CUBOID) V.no_solids = 0;
    }
    V.n_plane_cold = (uint32_t)F.plane_cold.size();
    V.n_meshes = (uint32_t)F.meshes.size();
    V.n_materials = (uint32_t)mats.size();
    V.plane_cold = D.plane_cold.p;
    V.meshes = D.meshes.p;
    V.tris = D.tris.p;
    V.tri_index = D.tri_index.p;
    V.tri_normals = D.tri_normals.p;
    V.nodes = D.nodes.p;
    V.mesh_inf = D.mesh_inf.p;
    V.materials = D.materials.p;
    V.textures = D.textures.p;
    V.texels8 = D.texels8.p;
    V.texels16 = D.texels16.p;
    V.lights = D.lights.p;
    s->flattened = true;
    s->stats.ms_bvh_build = F.ms_bvh_build;
    return ensure_workspace(s);
}

#define RRT_MAX_LEVELS 64u  // levels a lane's stripe counters are laid out for (render_rays_impl refuses more than 58)
constexpr size_t kStripeWordsPerLevel = (size_t)RRT_STRIPES * RRT_STRIPE_STRIDE;       // 32-bit words
constexpr size_t kStripeWordsAll = (size_t)RRT_MAX_LEVELS * kStripeWordsPerLevel;

int ensure_workspace(rrt_scene *s) {
    int rc;
    if ((rc = s->counters.ensure(64))) return rc;
    // lane 0's 64 counters and the stripe counters of its level queues live right behind the statistics (device and pinned
    // host copies alike), so that one memset clears all of them at the start of a frame and one copy brings them back at its end
    if ((rc = s->stats_dev.ensure(RRT_STAT_WORDS + 32 + kStripeWordsAll / 2))) return rc;
    if (!s->host_pinned) RRT_HIP(hipHostMalloc((void **)&s->host_pinned, (RRT_STAT_WORDS + 32 + kStripeWordsAll / 2) * sizeof(unsigned long long)));
    while (s->events.size() < 4) {
        hipEvent_t e;
        RRT_HIP(hipEventCreate(&e));
        s->events.push_back(e);
    }
    return RRT_OK;
}

int read_stats(rrt_scene *s, unsigned long long out[ST_COUNT]) {
    std::vector<unsigned long long> all(RRT_STAT_WORDS);
    RRT_HIP(hipMemcpy(all.data(), s->stats_dev.p, RRT_STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < ST_COUNT; k++) {
        out[k] = 0;
        for (int c = 0; c < RRT_STAT_COPIES; c++) out[k] += all[c * RRT_STAT_STRIDE + k];
    }
    return RRT_OK;
}

// ---- the render loop ------------------------------------------------------------------------------
// A frame (or ray batch) is cut into chunks; each chunk goes through one queue per recursion level.
//
// Level sizes.  The size of level k+1 is only known once level k has run.  The first frame of a given size finds the
// sizes out level by level (a 4-byte readback + stream sync per level).  Later frames launch every level for the sizes
// the previous frame had (+12.5 % + 4096) without waiting, read all counters back once at the end, and fall back to
// the level-by-level path if a queue outgrew its prediction (RRT_ERR_BIT_QUEUE; nothing is written past a launch's
// reach).
//
// Lanes.  The deep levels of a chunk hold few rays: their launches cannot fill the GPU and cost one wave's latency
// each.  Chunks can be spread over `render_streams` lanes (own queues, counters and stream) so that the deep levels of
// one chunk may run beside the wide shallow levels of another.  On MI355X this measured SLOWER (the smaller launches
// pay the per-level latency four times and the queues overlap little), so one lane is the default.
namespace {

int ensure_lanes(rrt_scene *s, uint32_t n) {
    while (s->lanes.size() < n) {
        s->lanes.emplace_back(new RenderLane());
        RenderLane &L = *s->lanes.back();
        int rc;
        if (s->lanes.size() == 1) {  // lane 0 borrows the tail of the statistics buffers (see ensure_workspace)
            L.ctr = reinterpret_cast<uint32_t *>(s->stats_dev.p + RRT_STAT_WORDS);
            L.host_pinned = s->host_pinned + 2 * RRT_STAT_WORDS;
            L.stripe = reinterpret_cast<uint32_t *>(s->stats_dev.p + RRT_STAT_WORDS + 32);
            L.stripe_pinned = s->host_pinned + 2 * (RRT_STAT_WORDS + 32);
        } else {
            if ((rc = L.stripe_own.ensure(kStripeWordsAll))) return rc;
            L.stripe = L.stripe_own.p;
            RRT_HIP(hipHostMalloc((void **)&L.stripe_pinned, kStripeWordsAll * sizeof(uint32_t)));
            L.owns_stripe_pinned = true;
            if ((rc = L.counters.ensure(64))) return rc;
            L.ctr = L.counters.p;
            RRT_HIP(hipHostMalloc((void **)&L.host_pinned, 64 * sizeof(uint32_t)));
            L.owns_pinned = true;
            RRT_HIP(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        }
        RRT_HIP(hipEventCreate(&L.ev_trace0));
        RRT_HIP(hipEventCreate(&L.ev_trace1));
        RRT_HIP(hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming));
    }
    return RRT_OK;
}

struct ChunkPlan {
    uint32_t base, n;
    std::vector<uint32_t> counts;  // rays per level, filled in when known
    std::vector<uint32_t> most;    // rays in the fullest stripe of each level >= 1
};

void fill_args(rrt_scene *s, TraceArgs &A, const DevCamera *cam, uint32_t y0, uint32_t n_pixels, uint32_t base, uint32_t depth_cap,
               RenderLane &L, float *rgb_dev, float *depth_dev, uint32_t *id_dev) {
    DeviceScene &D = s->dev;
    memset(&A, 0, sizeof(A));
    A.S = D.view;
    if (cam) A.cam = *cam;
    A.maps[0] = s->map_view[0];
    A.maps[1] = s->map_view[1];
    A.max_depth = depth_cap;
    A.mode = (uint32_t)s->mode;
    A.y0 = y0;
    A.pixel_base = base;
    A.band_rows = cam && cam->width ? n_pixels / cam->width : 0;
    A.lds_slots = D.lds_slots;
    A.fuse = (s->mode == RRT_MODE_SCENE && !D.reflective) ? 1u : 0u;  // single-level frames need no combine pass
    A.tile_order = 1u;  // (every trace launch is made of one-wave blocks: launch_trace_shade)
    A.gather_half_width = s->params.gather_half_width;
    A.err = L.ctr + 63;
    A.stats = s->stats_dev.p;
    A.heavy = L.heavy.p;  // null unless render_rays_impl decided to use the queue; heavy_capacity is set per level
    A.heavy_threshold = s->params.gather_heavy_threshold ? s->params.gather_heavy_threshold : s->heavy_auto;
    A.out_rgb = rgb_dev;
    A.out_depth = depth_dev;
    A.out_id = id_dev;
}

// The table of tile places for a camera frame's band (TraceArgs::tile_table): rebuilt on the lane's stream when the
// frame width, the band height or the tile order changes.
int set_tile_table(RenderLane &L, TraceArgs &A, hipStream_t st) {
    A.tile_table = nullptr;
    const uint32_t n_tiles = (A.cam.width >> 3) * (A.band_rows >> 3);
    if (n_tiles == 0 || A.cam.width > (0xFFFFu << 3) || A.band_rows > (0xFFFFu << 3)) return RRT_OK;
    if (L.tt_width != A.cam.width || L.tt_rows != A.band_rows || L.tt_order != A.tile_order + 1u) {
        int rc;
        L.tt_width = 0;
        if ((rc = L.tile_table.ensure(n_tiles))) return rc;
        launch_tile_table(L.tile_table.p, n_tiles, A.cam.width, A.band_rows, A.tile_order, st);
        L.tt_width = A.cam.width, L.tt_rows = A.band_rows, L.tt_order = A.tile_order + 1u;
    }
    A.tile_table = L.tile_table.p;
    return RRT_OK;
}

}  // namespace

// What one render call carries from round to round.
struct FrameRun {
    rrt_scene *s;
    const DevCamera *cam;
    uint32_t y0, n_pixels;
    const float *origins_dev, *directions_dev;
    float *rgb_dev, *depth_dev;
    uint32_t *id_dev;
    hipStream_t stream;
    uint32_t n_lights, depth_cap, n_levels, knn_k;
    bool heavy_queue;
    // results so far
    bool lane0_clean = true;   // lane 0's counters were zeroed together with the statistics
    bool tail_done = false;    // the frame's end event and statistics copy are already on the stream
    uint64_t secondary = 0;
    float ms_trace = 0.0f;
    uint32_t err = 0;
};

// Level k of a chunk.  `cover`: level 0: its rays; a striped level: rays per stripe the launch covers.  scap / scap_next:
// nodes per stripe the arrays of this level and of the next one hold (scap_next 0: nothing may be appended).
static void set_level_args(FrameRun &R, TraceArgs &A, RenderLane &L, uint32_t k, uint32_t cover, uint32_t scap, uint32_t scap_next) {
    A.cur = L.levels[k].view();
    A.next = L.levels[k + 1].view();
    A.count = cover;
    A.striped = k ? 1u : 0u;
    {
        // Small levels of a raytraced mesh scene in quarter-filled waves (TraceArgs::lane_shift); `cover` rays per stripe are
        // what the launch is sized for.  Levels 4 and 5 of the 4K frame with glass teapots (24 680 and 6 760 rays, a lone wave's
        // latency each): kernels 2.560 -> 2.537 ms, 7.18 -> 7.26 Grays/s; a threshold of 150 000 rays already costs more in idle
        // lanes than it saves (7.08), so only launches below 32 768 rays take it.  Nodes keep their places: results do not
        // depend on it (the parity suite passes with every level forced sparse, RRT_SPARSE_BELOW=100000000; 0 switches it off).
        static const uint32_t sparse_below = getenv("RRT_SPARSE_BELOW") ? (uint32_t)atoi(getenv("RRT_SPARSE_BELOW")) : 32768u;
        A.lane_shift = (k && R.s->mode == RRT_MODE_SCENE && R.s->dev.view.n_meshes && (uint64_t)RRT_STRIPES * cover < sparse_below) ? 2u : 0u;
    }
    A.stripe_cap = scap;
    A.count_ptr = k ? L.stripe + k * kStripeWordsPerLevel : nullptr;
    A.next_stripe_cap = scap_next;
    A.level = k;
    A.next_count = L.stripe + (k + 1) * kStripeWordsPerLevel;
    A.heavy_count = L.ctr + RRT_HEAVY_COUNTER + k;
    A.heavy = L.heavy.p;
    {
        // Photon mode, threshold derived from the maps (rrt_params.gather_heavy_threshold 0): the levels behind the first hold
        // few, incoherent rays -- their launch does not fill the GPU and lasts as long as its slowest LANE, which gathers up
        // to `threshold` candidate records four loads at a time.  They hand their queries over to the wave-per-query kernel
        // at an eighth of the threshold (not below 256): level 1 of the photon-mapped scene2 frame (114 k rays) 0.55 -> 0.32 ms.
        // A function of the ray's level and the maps only, so frames stay independent of bands and chunks.
        rrt_scene *s = R.s;
        const uint32_t base = s->params.gather_heavy_threshold ? s->params.gather_heavy_threshold : s->heavy_auto;
        const uint32_t deep = std::max(256u, base / 8u);
        A.heavy_threshold = (k && s->params.gather_heavy_threshold == 0) ? std::min(base, deep) : base;
    }
    const uint32_t nodes = k ? RRT_STRIPES * scap : cover;  // node indices of the level run below this
    A.heavy_capacity = nodes;
    A.knn_k = R.knn_k;
    A.knn_ws = L.knn.p;
    A.knn_stride = nodes;
}
// most child rays one stripe of level 1 can receive from a level 0 of n rays (blocks of 64 or 256 rays, dealt round-robin)
static uint32_t level1_stripe_bound(uint32_t n) { return 2u * (n / RRT_STRIPES + 512u); }
// ... and from a striped level none of whose stripes holds more than `most` rays: the blocks that append to one stripe
// (those with the same index modulo 8) work on every eighth block of every stripe (node_slot), two children per ray
static uint32_t stripe_children_bound(uint32_t most) { return 2u * (most + 8u * 256u); }
// sum and maximum over the stripes of one level's counters (host copy)
static void stripe_counts(const uint32_t *level_words, uint32_t &total, uint32_t &most) {
    total = most = 0;
    for (uint32_t q = 0; q < RRT_STRIPES; q++) {
        const uint32_t c = level_words[q * RRT_STRIPE_STRIDE];
        total += c;
        most = std::max(most, c);
    }
}

// One round of chunks launched for the level sizes of the previous frame: no readback between launches, one sync at the
// end.  overflow: a queue outgrew its launch, nothing was written past it, the caller repeats the frame level by level.
static int run_round_predicted(FrameRun &R, std::vector<ChunkPlan> &plan, uint32_t round, bool last_round, bool &overflow) {
    rrt_scene *s = R.s;
    DeviceScene &D = s->dev;
    hipStream_t stream = R.stream;
    const uint32_t n_levels = R.n_levels;
    int rc;
    overflow = false;
    uint32_t copy_levels = 0;  // levels whose stripe counters ride on the frame's one copy
    if (plan.size() > 1) {  // fork: the other lanes start after what the caller's stream holds so far
        RRT_HIP(hipEventRecord(s->events[2], stream));
        for (size_t j = 1; j < plan.size(); j++) RRT_HIP(hipStreamWaitEvent(s->lanes[j]->stream, s->events[2], 0));
    }
    for (size_t j = 0; j < plan.size(); j++) {
        RenderLane &L = *s->lanes[j];
        hipStream_t st = j == 0 ? stream : L.stream;
        const std::vector<uint32_t> &pred = L.prediction[round];  // [0] chunk size, [1 + k] rays of level k, [2 + n_levels + k] its fullest stripe
        // scap[k]: nodes per stripe the arrays of level k >= 1 hold, and the launch covers: what the fullest stripe held in
        // the previous frame + an eighth + 256, never more than the level before could fill
        std::vector<uint32_t> scap(n_levels + 2, 0);
        // Levels the previous frame left empty are not launched at all (a photon-mapped frame of scene2 has rays on 2
        // of its 13 levels): the last launched level may append nothing (next_stripe_cap 0), so a ray that does want to go
        // deeper raises RRT_ERR_BIT_QUEUE and the frame is repeated level by level, like any other outgrown prediction.
        uint32_t n_run = n_levels;
        for (uint32_t k = 1; k < n_levels; k++)
            if (pred[1 + k] == 0) {
                n_run = k;
                break;
            }
        if ((rc = L.levels[0].ensure(plan[j].n, R.n_lights, R.cam == nullptr))) return rc;
        uint32_t widest = plan[j].n;
        for (uint32_t k = 1; k < n_run; k++) {
            const uint32_t most = pred[2 + n_levels + k];
            const uint64_t want = (uint64_t)most + most / 8 + 256;
            scap[k] = (uint32_t)std::min<uint64_t>(want, k == 1 ? level1_stripe_bound(plan[j].n) : stripe_children_bound(scap[k - 1]));
            if ((rc = L.levels[k].ensure(RRT_STRIPES * scap[k], R.n_lights, true))) return rc;
            widest = std::max(widest, RRT_STRIPES * scap[k]);
        }
        if ((rc = L.levels[n_run].ensure(1, R.n_lights, true))) return rc;
        if (R.heavy_queue && (rc = L.heavy.ensure(widest))) return rc;
        if (R.knn_k && (rc = L.knn.ensure((size_t)R.knn_k * widest))) return rc;
        if (!(j == 0 && R.lane0_clean)) {  // (the frame's first memset has cleared lane 0's counters with the statistics)
            RRT_HIP(hipMemsetAsync(L.ctr, 0, 64 * sizeof(uint32_t), st));
            if (n_levels > 1) RRT_HIP(hipMemsetAsync(L.stripe, 0, (n_run + 1) * kStripeWordsPerLevel * sizeof(uint32_t), st));
        }
        if (j == 0) R.lane0_clean = false;
        if (!R.cam)
            launch_load_rays(R.origins_dev + (size_t)plan[j].base * 3, R.directions_dev + (size_t)plan[j].base * 3, plan[j].n, L.levels[0].view(), st);
        TraceArgs A;
        fill_args(s, A, R.cam, R.y0, R.n_pixels, plan[j].base, R.depth_cap, L, R.rgb_dev, R.depth_dev, R.id_dev);
        if (R.cam && (rc = set_tile_table(L, A, st))) return rc;
        RRT_HIP(hipEventRecord(L.ev_trace0, st));
        for (uint32_t k = 0; k < n_run; k++) {
            set_level_args(R, A, L, k, k ? scap[k] : plan[j].n, scap[k], k + 1 < n_run ? scap[k + 1] : 0u);
            launch_trace_shade(A, D.lds_bytes, st);
            launch_gather_heavy(A, st);
        }
        RRT_HIP(hipEventRecord(L.ev_trace1, st));
        for (int k = (int)n_run - 1; k >= 0 && !A.fuse; k--) {
            set_level_args(R, A, L, (uint32_t)k, k ? scap[k] : plan[j].n, scap[k], k + 1 < (int)n_run ? scap[k + 1] : 0u);
            launch_combine(A, st);
        }
        const bool one_copy = plan.size() == 1 && last_round;  // statistics + counters in one go below
        if (!one_copy) {
            RRT_HIP(hipMemcpyAsync(L.host_pinned, L.ctr, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            if (n_run > 1)
                RRT_HIP(hipMemcpyAsync(L.stripe_pinned, L.stripe, n_run * kStripeWordsPerLevel * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        }
        copy_levels = std::max(copy_levels, n_run > 1 ? n_run : 0u);
        if (j > 0) {  // join
            RRT_HIP(hipEventRecord(L.ev_done, st));
            RRT_HIP(hipStreamWaitEvent(stream, L.ev_done, 0));
        }
    }
    if (last_round) {  // the frame's end event and the stats ride on the same sync
        unsigned long long *st_all = reinterpret_cast<unsigned long long *>(s->host_pinned);
        RRT_HIP(hipEventRecord(s->events[1], stream));
        RRT_HIP(hipMemcpyAsync(st_all, s->stats_dev.p,
                               (RRT_STAT_WORDS + (plan.size() == 1 ? 32 : 0)) * sizeof(unsigned long long) +
                                   (plan.size() == 1 ? copy_levels * kStripeWordsPerLevel * sizeof(uint32_t) : 0),
                               hipMemcpyDeviceToHost, stream));
        R.tail_done = true;
    }
    RRT_HIP(hipStreamSynchronize(stream));
    for (size_t j = 0; j < plan.size(); j++) overflow |= (s->lanes[j]->host_pinned[63] & RRT_ERR_BIT_QUEUE) != 0;
    if (overflow) return RRT_OK;
    float first = 0.0f, last = 0.0f;  // wall-clock span of the trace launches of this round
    for (size_t j = 0; j < plan.size(); j++) {
        RenderLane &L = *s->lanes[j];
        R.err |= L.host_pinned[63];
        uint32_t n_run = n_levels;  // (as above: the levels that were launched)
        for (uint32_t k = 1; k < n_levels; k++)
            if (L.prediction[round][1 + k] == 0) {
                n_run = k;
                break;
            }
        for (uint32_t k = 1; k < n_run; k++) {
            stripe_counts(L.stripe_pinned + k * kStripeWordsPerLevel, plan[j].counts[k], plan[j].most[k]);
            R.secondary += plan[j].counts[k];
        }
        float t0 = 0.0f, t1 = 0.0f;
        RRT_HIP(hipEventElapsedTime(&t0, s->events[0], L.ev_trace0));
        RRT_HIP(hipEventElapsedTime(&t1, s->events[0], L.ev_trace1));
        first = j == 0 ? t0 : std::min(first, t0);
        last = j == 0 ? t1 : std::max(last, t1);
    }
    R.ms_trace += last - first;
    return RRT_OK;
}

// One round of chunks level by level, one chunk after the other, on the caller's stream: every level's size is read
// back before the next one is launched (the first frame of a size, and frames that outgrew their prediction).
static int run_round_by_level(FrameRun &R, std::vector<ChunkPlan> &plan) {
    rrt_scene *s = R.s;
    DeviceScene &D = s->dev;
    hipStream_t stream = R.stream;
    const uint32_t n_levels = R.n_levels;
    int rc;
    for (size_t j = 0; j < plan.size(); j++) {
        RenderLane &L = *s->lanes[j];
        std::vector<uint32_t> &counts = plan[j].counts, &most = plan[j].most;
        if ((rc = L.levels[0].ensure(plan[j].n, R.n_lights, R.cam == nullptr))) return rc;
        if (!(j == 0 && R.lane0_clean)) {
            RRT_HIP(hipMemsetAsync(L.ctr, 0, 64 * sizeof(uint32_t), stream));
            if (n_levels > 1) RRT_HIP(hipMemsetAsync(L.stripe, 0, (n_levels + 1) * kStripeWordsPerLevel * sizeof(uint32_t), stream));
        }
        if (j == 0) R.lane0_clean = false;
        if (!R.cam)
            launch_load_rays(R.origins_dev + (size_t)plan[j].base * 3, R.directions_dev + (size_t)plan[j].base * 3, plan[j].n, L.levels[0].view(), stream);
        TraceArgs A;
        fill_args(s, A, R.cam, R.y0, R.n_pixels, plan[j].base, R.depth_cap, L, R.rgb_dev, R.depth_dev, R.id_dev);
        if (R.cam && (rc = set_tile_table(L, A, stream))) return rc;
        uint32_t last = 0;
        std::vector<uint32_t> scap(n_levels + 2, 0);  // nodes per stripe of each level's arrays
        for (uint32_t k = 0; k < n_levels; k++) {
            bool spawns = k + 1 < n_levels;
            scap[k + 1] = spawns ? (k == 0 ? level1_stripe_bound(plan[j].n) : stripe_children_bound(most[k])) : 0u;
            if (spawns && (rc = L.levels[k + 1].ensure(RRT_STRIPES * scap[k + 1], R.n_lights, true))) return rc;
            // the stream is idle here (synchronised after the previous level): growing the per-level buffers is safe
            const uint32_t nodes = k ? RRT_STRIPES * scap[k] : counts[0];
            if (R.heavy_queue && (rc = L.heavy.ensure(nodes))) return rc;
            if (R.knn_k && (rc = L.knn.ensure((size_t)R.knn_k * nodes))) return rc;
            set_level_args(R, A, L, k, k ? most[k] : counts[0], scap[k], scap[k + 1]);
            RRT_HIP(hipEventRecord(L.ev_trace0, stream));
            launch_trace_shade(A, D.lds_bytes, stream);
            launch_gather_heavy(A, stream);
            RRT_HIP(hipEventRecord(L.ev_trace1, stream));
            last = k;
            uint32_t next_n = 0;
            if (spawns) {
                RRT_HIP(hipMemcpyAsync(L.stripe_pinned, L.stripe + (k + 1) * kStripeWordsPerLevel, kStripeWordsPerLevel * sizeof(uint32_t),
                                       hipMemcpyDeviceToHost, stream));
                RRT_HIP(hipStreamSynchronize(stream));
                stripe_counts(L.stripe_pinned, next_n, most[k + 1]);
            } else {
                RRT_HIP(hipStreamSynchronize(stream));
            }
            float ms = 0.0f;
            RRT_HIP(hipEventElapsedTime(&ms, L.ev_trace0, L.ev_trace1));
            R.ms_trace += ms;
            static const bool level_stats = getenv("RRT_LEVEL_STATS") != nullptr;  // debugging aid (with RRT_LEVEL_SYNC=1)
            if (level_stats) {
                static unsigned long long seen[ST_COUNT];
                unsigned long long all[RRT_STAT_WORDS], now[ST_COUNT] = {0};
                RRT_HIP(hipMemcpy(all, s->stats_dev.p, sizeof(all), hipMemcpyDeviceToHost));
                for (int q = 0; q < ST_COUNT; q++)
                    for (int c = 0; c < RRT_STAT_COPIES; c++) now[q] += all[c * RRT_STAT_STRIDE + q];
                if (k == 0) memset(seen, 0, sizeof(seen));
                fprintf(stderr, "[level %u] %u rays, %.3f ms: box tests %llu, shadow rays %llu, gather queries %llu, candidates %llu, gathered %llu\n", k,
                        counts[k], ms, now[ST_NODES] - seen[ST_NODES], now[ST_SHADOW] - seen[ST_SHADOW], now[ST_QUERIES] - seen[ST_QUERIES],
                        now[ST_CANDIDATES] - seen[ST_CANDIDATES], now[ST_GATHERED] - seen[ST_GATHERED]);
                memcpy(seen, now, sizeof(seen));
            }
            counts[k + 1] = next_n;
            R.secondary += next_n;
            if (next_n == 0) break;
        }
        for (int k = (int)last; k >= 0 && !A.fuse; k--) {
            set_level_args(R, A, L, (uint32_t)k, k ? most[k] : counts[0], scap[k], scap[k + 1]);
            launch_combine(A, stream);
        }
        RRT_HIP(hipMemcpyAsync(L.host_pinned, L.ctr + 63, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        RRT_HIP(hipStreamSynchronize(stream));
        R.err |= L.host_pinned[0];
    }
    return RRT_OK;
}

static int render_rays_impl(rrt_scene *s, const DevCamera *cam, uint32_t y0, uint32_t n_pixels, const float *origins_dev,
                            const float *directions_dev, float *rgb_dev, float *depth_dev, uint32_t *id_dev, hipStream_t stream,
                            bool allow_predicted, bool &retry) {
    retry = false;
    DeviceScene &D = s->dev;
    FrameRun R{s, cam, y0, n_pixels, origins_dev, directions_dev, rgb_dev, depth_dev, id_dev, stream};
    R.n_lights = D.view.n_lights;
    R.depth_cap = s->mode == RRT_MODE_SCENE ? s->params.max_depth : s->params.max_vueon_depth;
    R.n_levels = D.reflective ? R.depth_cap + 1 : 1;
    const uint32_t n_levels = R.n_levels;
    if (n_levels + 2 > 60) return fail(RRT_ERR_LIMIT, "recursion depth above 57");
    int rc;
    uint32_t want_lanes = s->params.render_streams ? std::min(s->params.render_streams, 8u) : 1u;
    if (n_levels == 1) want_lanes = 1;  // no deep levels to hide
    uint32_t chunk = s->params.chunk_pixels ? s->params.chunk_pixels : (1u << 23);  // a 3840x2160 frame is one chunk
    // photon mode with the k-nearest estimator (get_n_within_radius): every ray in flight owns gather_k heap entries of 8
    // bytes; chunks shrink so that a level of chunk size stays under 2 GiB of them
    R.knn_k = s->mode == RRT_MODE_PHOTON ? s->params.gather_k : 0u;
    if (R.knn_k > RRT_GATHER_K_MAX) return fail(RRT_ERR_LIMIT, "gather_k above RRT_GATHER_K_MAX (4096)");
    if (R.knn_k) chunk = std::min(chunk, std::max(1u << 16, ((1u << 28) / R.knn_k) & ~255u));
    if (want_lanes > 1 && n_pixels >= (1u << 18)) {  // big enough to be worth cutting: one chunk per lane, whole waves
        uint32_t per_lane = (n_pixels + want_lanes - 1) / want_lanes;
        per_lane = (per_lane + 255u) & ~255u;
        chunk = std::min(chunk, per_lane);
    } else {
        want_lanes = 1;
    }
    if ((rc = ensure_lanes(s, want_lanes))) return rc;
    for (uint32_t j = 0; j < want_lanes; j++)
        if (s->lanes[j]->levels.size() < n_levels + 1) s->lanes[j]->levels.resize(n_levels + 1);
    // photon mode: heavy radiance estimates are filed per level (lane counters [32 + level]) and gathered by whole waves.
    // The queue of a level holds one record per ray of that level (every query may be heavy), so it is sized from the
    // level sizes, never from the chunk: a level behind glass can hold more rays than the chunk has pixels, and a lane
    // that found the queue full would gather for itself in another summation order (frames would then depend on atomics
    // order).
    R.heavy_queue = s->mode == RRT_MODE_PHOTON && s->params.gather_heavy_threshold != 0xFFFFFFFFu && n_levels <= 30 && !R.knn_k;
    for (uint32_t j = 0; j < want_lanes; j++) {
        if (!R.heavy_queue) s->lanes[j]->heavy.release();
        if (!R.knn_k) s->lanes[j]->knn.release();
    }

    RRT_HIP(hipMemsetAsync(s->stats_dev.p, 0,
                           (RRT_STAT_WORDS + 32) * sizeof(unsigned long long) + (n_levels > 1 ? (n_levels + 1) * kStripeWordsPerLevel * sizeof(uint32_t) : 0),
                           stream));  // + lane 0's counters and the stripe counters of its levels
    RRT_HIP(hipEventRecord(s->events[0], stream));
    unsigned long long *st_all = reinterpret_cast<unsigned long long *>(s->host_pinned);  // RRT_STAT_WORDS slots

    uint32_t round = 0;
    static const std::vector<uint32_t> no_prediction;
    for (uint32_t round_base = 0; round_base < n_pixels; round++) {
        // the chunks of this round, one per lane
        std::vector<ChunkPlan> plan;
        for (uint32_t j = 0; j < want_lanes && round_base < n_pixels; j++) {
            ChunkPlan c;
            c.base = round_base;
            c.n = std::min(chunk, n_pixels - round_base);
            c.counts.assign(n_levels + 1, 0);
            c.most.assign(n_levels + 1, 0);
            c.counts[0] = c.n;
            plan.push_back(c);
            round_base += c.n;
        }
        bool predicted = allow_predicted;  // single-level frames too: no readback between launches, one sync per round
        for (size_t j = 0; j < plan.size(); j++) {
            const auto &all = s->lanes[j]->prediction;
            const std::vector<uint32_t> &pred = round < all.size() ? all[round] : no_prediction;
            if (!(pred.size() == 2 * n_levels + 3 && pred[0] == plan[j].n)) predicted = false;
        }
        if (predicted) {
            bool overflow = false;
            if ((rc = run_round_predicted(R, plan, round, round_base >= n_pixels, overflow))) return rc;
            if (overflow) {
                for (auto &L : s->lanes) L->prediction.clear();
                retry = true;
                return RRT_OK;
            }
        } else {
            if ((rc = run_round_by_level(R, plan))) return rc;
        }
        // what the next frame's chunks of these sizes will be launched for
        for (size_t j = 0; j < plan.size(); j++) {
            if (s->lanes[j]->prediction.size() <= round) s->lanes[j]->prediction.resize(round + 1);
            std::vector<uint32_t> &pred = s->lanes[j]->prediction[round];
            pred.assign(2 * n_levels + 3, 0);  // [0] chunk size, [1 + k] rays of level k, [2 + n_levels + k] rays in its fullest stripe
            pred[0] = plan[j].n;
            for (uint32_t k = 0; k <= n_levels; k++) pred[1 + k] = plan[j].counts[k], pred[2 + n_levels + k] = plan[j].most[k];
        }
    }
    if (!R.tail_done) {
        RRT_HIP(hipEventRecord(s->events[1], stream));
        RRT_HIP(hipMemcpyAsync(st_all, s->stats_dev.p, RRT_STAT_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        RRT_HIP(hipStreamSynchronize(stream));
    }
    unsigned long long st[ST_COUNT];
    for (int k = 0; k < ST_COUNT; k++) {
        st[k] = 0;
        for (int c = 0; c < RRT_STAT_COPIES; c++) st[k] += st_all[c * RRT_STAT_STRIDE + k];
    }
    float ms_total = 0.0f;
    RRT_HIP(hipEventElapsedTime(&ms_total, s->events[0], s->events[1]));
    rrt_stats &O = s->render_stats;
    O.rays_primary = n_pixels;
    O.rays_secondary = R.secondary;
    O.rays_shadow = st[ST_SHADOW];
    O.bvh_nodes_visited = st[ST_NODES];
    O.bvh_tris_tested = st[ST_TRIS];
    O.photons_gathered = st[ST_GATHERED];
    O.gather_queries = st[ST_QUERIES];
    O.gather_candidates = st[ST_CANDIDATES];
    O.vueons_truncated = s->mode == RRT_MODE_PHOTON ? st[ST_TRUNCATED] : 0;
    O.flops_per_trace = D.flops_per_trace;
    O.flops_algorithmic = (O.rays_primary + O.rays_secondary + O.rays_shadow) * D.flops_per_trace + 27ull * st[ST_NODES] + 61ull * st[ST_TRIS];
    O.ms_trace_kernels = R.ms_trace;
    O.ms_total = ms_total;
    if (R.err & RRT_ERR_BIT_INTERNAL) return fail(RRT_ERR_DEVICE, "device-side index guard tripped (internal error)");
    if (R.err & 1u) return fail(RRT_ERR_OVERFLOW, "HitVec overflow");
    if (R.err & 2u) return fail(RRT_ERR_INVALID, "No texture coordinates");
    return RRT_OK;
}

static int render_rays_on(rrt_scene *s, const DevCamera *cam, uint32_t y0, uint32_t n_pixels, const float *origins_dev,
                          const float *directions_dev, float *rgb_dev, float *depth_dev, uint32_t *id_dev, hipStream_t stream) {
    static const bool level_sync = getenv("RRT_LEVEL_SYNC") != nullptr;  // debugging aid: always size levels by readback
    bool retry = false;
    int rc = render_rays_impl(s, cam, y0, n_pixels, origins_dev, directions_dev, rgb_dev, depth_dev, id_dev, stream, !level_sync, retry);
    if (rc == RRT_OK && retry) s->render_stats.frames_repeated++;
    if (rc == RRT_OK && retry)
        rc = render_rays_impl(s, cam, y0, n_pixels, origins_dev, directions_dev, rgb_dev, depth_dev, id_dev, stream, false, retry);
    return rc;
}

int use_scene_device(const rrt_scene *s) {
    if (s->device < 0) return RRT_OK;
    int cur = -1;
    RRT_HIP(hipGetDevice(&cur));
    if (cur != s->device) RRT_HIP(hipSetDevice(s->device));
    return RRT_OK;
}

void drop_render_workers(rrt_scene *s) {
    std::lock_guard<std::mutex> lock(s->render_mutex);
    s->workers.clear();
    s->epoch++;
}

namespace {

// statistics of the calling thread's last render call, per handle: concurrent callers each see their own frame
struct ThreadStats {
    const rrt_scene *scene = nullptr;
    uint64_t epoch = 0;
    rrt_stats stats{};
};
thread_local ThreadStats t_stats;

std::unique_ptr<rrt_scene> make_worker(const rrt_scene *s) {
    std::unique_ptr<rrt_scene> w(new rrt_scene());
    w->is_worker = true;
    w->mode = s->mode;
    w->params = s->params;
    w->flattened = true;
    w->maps_built = s->maps_built;
    w->seed = s->seed;
    w->device = s->device;
    // the scene by pointer: w->dev owns no buffers
    w->dev.view = s->dev.view;
    w->dev.lds_slots = s->dev.lds_slots;
    w->dev.bvh_stack = s->dev.bvh_stack;
    w->dev.lds_bytes = s->dev.lds_bytes;
    w->dev.reflective = s->dev.reflective;
    w->dev.flops_per_trace = s->dev.flops_per_trace;
    w->map_view[0] = s->map_view[0];
    w->map_view[1] = s->map_view[1];
    w->heavy_auto = s->heavy_auto;
    return w;
}

void copy_render_stats(rrt_stats &to, const rrt_stats &from) {
    to.rays_primary = from.rays_primary;
    to.rays_secondary = from.rays_secondary;
    to.rays_shadow = from.rays_shadow;
    to.bvh_nodes_visited = from.bvh_nodes_visited;
    to.bvh_tris_tested = from.bvh_tris_tested;
    to.photons_gathered = from.photons_gathered;
    to.gather_queries = from.gather_queries;
    to.gather_candidates = from.gather_candidates;
    to.vueons_truncated = from.vueons_truncated;
    to.flops_per_trace = from.flops_per_trace;
    to.flops_algorithmic = from.flops_algorithmic;
    to.ms_trace_kernels = from.ms_trace_kernels;
    to.ms_total = from.ms_total;
}

}  // namespace

int render_rays(rrt_scene *s, const DevCamera *cam, uint32_t y0, uint32_t n_pixels, const float *origins_dev,
                const float *directions_dev, float *rgb_dev, float *depth_dev, uint32_t *id_dev, hipStream_t stream) {
    if (!s->flattened) return fail(RRT_ERR_STATE, "rrt_pre_render has not been called");
    if (s->mode == RRT_MODE_PHOTON && !s->maps_built) return fail(RRT_ERR_STATE, "Photon map not built");
    {
        int rc_dev = use_scene_device(s);  // render threads other than the one that built the scene start on device 0
        if (rc_dev) return rc_dev;
    }
    rrt_scene *w = nullptr;
    {
        std::lock_guard<std::mutex> lock(s->render_mutex);
        if (!s->busy) {
            w = s;
        } else {
            for (auto &c : s->workers)
                if (!c->busy) {
                    w = c.get();
                    break;
                }
            if (!w) {
                if (s->workers.size() >= 64) return fail(RRT_ERR_LIMIT, "more than 65 concurrent render calls on one scene handle");
                s->workers.push_back(make_worker(s));
                w = s->workers.back().get();
            }
        }
        w->busy = true;
    }
    int rc = w == s ? RRT_OK : ensure_workspace(w);
    const uint64_t repeated_before = w->render_stats.frames_repeated;
    if (rc == RRT_OK) rc = render_rays_on(w, cam, y0, n_pixels, origins_dev, directions_dev, rgb_dev, depth_dev, id_dev, stream);
    {
        std::lock_guard<std::mutex> lock(s->render_mutex);
        copy_render_stats(s->stats, w->render_stats);
        s->stats.frames_repeated += w->render_stats.frames_repeated - repeated_before;
        t_stats.scene = s;
        t_stats.epoch = s->epoch;
        t_stats.stats = s->stats;
        w->busy = false;
    }
    return rc;
}

}  // namespace rrt

// =================================================================================================
// C ABI
// =================================================================================================
namespace {

int b_set_mode(void *ctx, int mode) { return rrt_scene_set_mode((rrt_scene *)ctx, mode); }
int b_add_texture(void *ctx, uint32_t w, uint32_t h, uint32_t mv, const uint8_t *p) { return rrt_add_texture((rrt_scene *)ctx, w, h, mv, p); }
int b_add_material(void *ctx, const rrt_material_desc *d) { return rrt_add_material((rrt_scene *)ctx, d); }
int b_object_define(void *ctx, const rrt_object_desc *d) { return rrt_object_define((rrt_scene *)ctx, d); }
int b_object_transform(void *ctx, int h, const float *m) { return rrt_object_transform((rrt_scene *)ctx, h, m); }
int b_add_object(void *ctx, int h) { return rrt_add_object((rrt_scene *)ctx, h); }
int b_add_light(void *ctx, const rrt_light_desc *d) { return rrt_add_light((rrt_scene *)ctx, d); }

DevCamera to_dev(const rrt_camera *c) {
    DevCamera d;
    d.position = mk3(c->position[0], c->position[1], c->position[2]);
    d.right = mk3(c->right[0], c->right[1], c->right[2]);
    d.up = mk3(c->up[0], c->up[1], c->up[2]);
    d.lookat = mk3(c->lookat[0], c->lookat[1], c->lookat[2]);
    d.fov = c->fov;
    d.width = c->width;
    d.height = c->height;
    d.row_group = 0;
    d.row_stride = 1;
    d.row_first = 0;
    return d;
}

void invalidate(rrt_scene *s) {
    drop_render_workers(s);
    s->flattened = false;
    s->maps_built = false;
}

}  // namespace

extern "C" {

extern "C++" {
namespace rrt {
void preload_kernels();
void preload_photon();
void preload_bvh_sah();
void preload_bvh_lbvh();
// First use of the device by this process, paid where a scene is made instead of inside its first rrt_pre_render: the
// library's code objects (loaded by HIP on first use, one per translation unit) and HIP's host-to-device copy path (its first
// large copy sets staging up: 80 ms measured).  Once per process; nothing happens without a device.
static void warm_up_device() {
    static std::once_flag once;
    std::call_once(once, [] {
        int n = 0;
        if (getenv("RRT_NO_WARMUP") || hipGetDeviceCount(&n) != hipSuccess || n == 0) return;
        preload_kernels();
        preload_photon();
        preload_bvh_sah();
        preload_bvh_lbvh();
        const size_t bytes = 8u << 20;
        void *d = nullptr;
        if (hipMalloc(&d, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        std::vector<char> h(bytes, 0);
        (void)hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        (void)hipMemcpy(h.data(), d, 64, hipMemcpyDeviceToHost);
        (void)hipFree(d);
        (void)hipGetLastError();
    });
}
}  // namespace rrt
}  // extern "C++"

rrt_scene *rrt_scene_create(int mode, const rrt_params *params) {
    if (mode != RRT_MODE_SCENE && mode != RRT_MODE_PHOTON) {
        set_error("bad mode");
        return nullptr;
    }
    rrt::warm_up_device();
    rrt_scene *s = new rrt_scene();
    s->mode = mode;
    if (params)
        s->params = *params;
    else
        rrt_params_default(&s->params);
    s->builder = rrt_builder{s, b_set_mode, b_add_texture, b_add_material, b_object_define, b_object_transform, b_add_object, b_add_light};
    return s;
}
void rrt_scene_destroy(rrt_scene *s) { delete s; }
int rrt_scene_set_mode(rrt_scene *s, int mode) {
    if (!s || (mode != RRT_MODE_SCENE && mode != RRT_MODE_PHOTON)) return fail(RRT_ERR_INVALID, "bad mode");
    drop_render_workers(s);  // workers copy mode and params when they are made: none may outlive a change
    s->mode = mode;
    return RRT_OK;
}
int rrt_scene_get_mode(const rrt_scene *s) { return s ? s->mode : RRT_ERR_INVALID; }
int rrt_scene_set_params(rrt_scene *s, const rrt_params *params) {
    if (!s || !params) return fail(RRT_ERR_INVALID, "null argument");
    if (params->max_depth > 40 || params->max_vueon_depth > 40 || params->max_photon_bounces > 64)
        return fail(RRT_ERR_LIMIT, "max_depth/max_vueon_depth <= 40, max_photon_bounces <= 64");
    if (params->gather_k > RRT_GATHER_K_MAX) return fail(RRT_ERR_LIMIT, "gather_k above RRT_GATHER_K_MAX (4096)");
    drop_render_workers(s);  // see rrt_scene_set_mode
    s->params = *params;
    s->maps_built = false;
    return RRT_OK;
}
const rrt_builder *rrt_scene_builder(rrt_scene *s) { return s ? &s->builder : nullptr; }
int rrt_object_count(const rrt_scene *s) { return s ? (int)s->objects.size() : RRT_ERR_INVALID; }
int rrt_light_count(const rrt_scene *s) { return s ? (int)s->lights.size() : RRT_ERR_INVALID; }

int rrt_add_texture(rrt_scene *s, uint32_t width, uint32_t height, uint32_t maxval, const uint8_t *samples) {
    if (!s || !samples || width == 0 || height == 0 || maxval == 0 || maxval > 65535) return fail(RRT_ERR_INVALID, "bad texture");
    HostTexture t;
    t.width = width;
    t.height = height;
    t.maxval = maxval;
    size_t bytes = (size_t)width * height * 3 * (maxval <= 255 ? 1 : 2);
    t.samples.assign(samples, samples + bytes);
    s->textures.push_back(std::move(t));
    invalidate(s);
    return (int)s->textures.size() - 1;
}
int rrt_add_texture_ppm(rrt_scene *s, const char *path) {
    if (!s || !path) return fail(RRT_ERR_INVALID, "null argument");
    HostTexture t;
    int rc = read_ppm(path, t);
    if (rc) return rc;
    s->textures.push_back(std::move(t));
    invalidate(s);
    return (int)s->textures.size() - 1;
}

int rrt_add_material(rrt_scene *s, const rrt_material_desc *d) {
    if (!s || !d) return fail(RRT_ERR_INVALID, "null argument");
    int nt = (int)s->textures.size();
    switch (d->kind) {
        case RRT_MAT_MONOCHROME:
        case RRT_MAT_GLOBAL:
        case RRT_MAT_FALSECOLOUR: break;
        case RRT_MAT_TEXTURE:
            if (d->diffuse_tex < 0 || d->diffuse_tex >= nt || d->normal_tex >= nt) return fail(RRT_ERR_INVALID, "bad texture id");
            break;
        case RRT_MAT_COMPOUND:
            if (d->n_parts < 1 || d->n_parts > RRT_MAX_COMPOUND_PARTS) return fail(RRT_ERR_LIMIT, "compound needs 1..4 parts");
            for (int i = 0; i < d->n_parts; i++) {
                if (d->parts[i] < 0 || d->parts[i] >= (int)s->materials.size()) return fail(RRT_ERR_INVALID, "bad compound part id");
                int k = s->materials[d->parts[i]].kind;
                if (k == RRT_MAT_COMPOUND) return fail(RRT_ERR_LIMIT, "compound parts may not be compounds");
                if (k == RRT_MAT_GLOBAL && i != d->n_parts - 1) return fail(RRT_ERR_LIMIT, "a Global part must be the last part of a compound");
            }
            break;
        default: return fail(RRT_ERR_INVALID, "bad material kind");
    }
    s->materials.push_back(*d);
    invalidate(s);
    return (int)s->materials.size() - 1;
}

int rrt_object_define(rrt_scene *s, const rrt_object_desc *d) {
    if (!s || !d) return fail(RRT_ERR_INVALID, "null argument");
    auto o = std::make_unique<HostObject>();
    o->kind = d->kind;
    o->material = d->material;
    if (d->kind != RRT_OBJ_CSG && (d->material < 0 || d->material >= (int)s->materials.size())) return fail(RRT_ERR_INVALID, "bad material id");
    const float *v = d->v;
    switch (d->kind) {
        case RRT_OBJ_PLANE:
            o->p0 = mk3(v[0], v[1], v[2]);
            o->p1 = mk3(v[3], v[4], v[5]);
            o->p2 = mk3(v[6], v[7], v[8]);
            o->d = -dot3(o->p2, o->p0);  // plane_object.rs:37
            break;
        case RRT_OBJ_SPHERE:
            o->p0 = mk3(v[0], v[1], v[2]);
            o->radius = v[3];
            break;
        case RRT_OBJ_CUBOID:
            o->p0 = mk3(v[0], v[1], v[2]);
            o->p1 = mk3(v[3], v[4], v[5]);
            break;
        case RRT_OBJ_QUADRATIC:
            for (int i = 0; i < 10; i++) o->q[i] = v[i];
            break;
        case RRT_OBJ_MESH:
            if (!d->positions && d->n_triangles) return fail(RRT_ERR_INVALID, "mesh without positions");
            if (d->smooth && !d->normals) return fail(RRT_ERR_INVALID, "Vertex normals not set in smoothen_hit()");
            if (d->n_triangles >= (1u << 27)) return fail(RRT_ERR_LIMIT, "mesh larger than 2^27 triangles");
            o->tri_pos.assign(d->positions, d->positions + d->n_triangles * 9);
            if (d->normals) o->tri_nrm.assign(d->normals, d->normals + d->n_triangles * 9);
            o->smooth = d->smooth != 0;
            break;
        case RRT_OBJ_CSG: {
            int n = (int)s->detached.size();
            if (d->csg_mode < 0 || d->csg_mode > 2) return fail(RRT_ERR_INVALID, "bad csg mode");
            if (d->left < 0 || d->left >= n || d->right < 0 || d->right >= n || d->left == d->right || !s->detached[d->left] ||
                !s->detached[d->right])
                return fail(RRT_ERR_INVALID, "bad csg child handle");
            o->csg_mode = d->csg_mode;
            o->left = std::move(s->detached[d->left]);
            o->right = std::move(s->detached[d->right]);
            break;
        }
        default: return fail(RRT_ERR_INVALID, "bad object kind");
    }
    s->detached.push_back(std::move(o));
    return (int)s->detached.size() - 1;
}

int rrt_object_define_obj(rrt_scene *s, const char *obj_path, int material, int smooth) {
    if (!s || !obj_path) return fail(RRT_ERR_INVALID, "null argument");
    std::vector<float> pos, nrm;
    int rc = load_obj(obj_path, smooth != 0, pos, nrm);
    if (rc) return rc;
    rrt_object_desc d;
    memset(&d, 0, sizeof(d));
    d.kind = RRT_OBJ_MESH;
    d.material = material;
    d.positions = pos.data();
    d.normals = nrm.empty() ? nullptr : nrm.data();
    d.n_triangles = pos.size() / 9;
    d.smooth = smooth;
    return rrt_object_define(s, &d);
}

int rrt_object_transform(rrt_scene *s, int handle, const float *m16) {
    if (!s || !m16 || handle < 0 || handle >= (int)s->detached.size() || !s->detached[handle]) return fail(RRT_ERR_INVALID, "bad handle");
    float m[4][4];
    memcpy(m, m16, sizeof(m));
    try {
        s->detached[handle]->apply_transform(m);
    } catch (int) {
        return RRT_ERR_INVALID;  // message set by the thrower ("Matrix is not invertible")
    }
    return RRT_OK;
}

int rrt_add_object(rrt_scene *s, int handle) {
    if (!s || handle < 0 || handle >= (int)s->detached.size() || !s->detached[handle]) return fail(RRT_ERR_INVALID, "bad handle");
    s->objects.push_back(std::move(s->detached[handle]));
    invalidate(s);
    return (int)s->objects.size() - 1;
}

int rrt_add_light(rrt_scene *s, const rrt_light_desc *d) {
    if (!s || !d || d->kind < 0 || d->kind > 2) return fail(RRT_ERR_INVALID, "bad light");
    s->lights.push_back(*d);
    invalidate(s);
    return (int)s->lights.size() - 1;
}

rrt_scene *rrt_scene_load(const char *path, const char *asset_root, const rrt_params *params, int mode_override, rrt_camera *camera_out) {
    rrt_scene *s = rrt_scene_create(RRT_MODE_SCENE, params);
    if (!s) return nullptr;
    uint32_t line = 0;
    int rc = rrt_parse_scene_file(path, asset_root, &s->builder, camera_out, mode_override, &line);
    if (rc) {
        rrt_scene_destroy(s);
        return nullptr;
    }
    return s;
}

int rrt_pre_render(rrt_scene *s, uint64_t seed) {
    if (!s) return fail(RRT_ERR_INVALID, "null scene");
    if (s->mode == RRT_MODE_PHOTON) {
        int rc = rrt_photon_begin(s, seed);
        if (rc) return rc;
        // Big passes go in index ranges (streams are keyed by photon index and the maps are sorted into canonical
        // order before the index build, so the split changes no bit): bounds the staging buffers to a few GB at the
        // 100 M photons of BASELINE configs[4].
        uint64_t step = 1ull << 24;
        if (const char *e = getenv("RRT_SHOOT_CHUNK")) step = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
        for (int l = 0; l < (int)s->lights.size(); l++)
            for (int pass = 0; pass < 2; pass++) {
                const uint64_t total = pass == 0 ? s->params.photons_per_light : s->params.caustic_photons_per_light;
                for (uint64_t first = 0; first < total; first += step)
                    if ((rc = rrt_photon_shoot(s, l, pass, first, std::min(step, total - first)))) return rc;
            }
        return rrt_photon_build(s);
    }
    s->seed = seed;
    return flatten_scene(s);
}

int rrt_render_device(rrt_scene *s, const rrt_camera *cam, uint32_t y0, uint32_t y1, float *rgb_dev, float *depth_dev,
                      uint32_t *object_id_dev, void *hip_stream) {
    if (!s || !cam) return fail(RRT_ERR_INVALID, "null argument");
    if (y0 > y1 || y1 > cam->height) return fail(RRT_ERR_INVALID, "bad row range");
    uint64_t n = (uint64_t)(y1 - y0) * cam->width;
    if (n >= (1ull << 31)) return fail(RRT_ERR_LIMIT, "more than 2^31 pixels in one band");
    DevCamera dc = to_dev(cam);
    return render_rays(s, &dc, y0, (uint32_t)n, nullptr, nullptr, rgb_dev, depth_dev, object_id_dev, (hipStream_t)hip_stream);
}

uint32_t rrt_interleaved_row_count(uint32_t height, uint32_t group_rows, uint32_t first_group, uint32_t group_stride) {
    if (group_rows == 0 || group_stride == 0 || first_group >= group_stride) return 0;
    uint32_t rows = 0;
    for (uint64_t y = (uint64_t)first_group * group_rows; y < height; y += (uint64_t)group_stride * group_rows)
        rows += (uint32_t)std::min<uint64_t>(group_rows, height - y);
    return rows;
}

int rrt_render_interleaved_device(rrt_scene *s, const rrt_camera *cam, uint32_t group_rows, uint32_t first_group, uint32_t group_stride,
                                  float *rgb_dev, float *depth_dev, uint32_t *object_id_dev, void *hip_stream) {
    if (!s || !cam) return fail(RRT_ERR_INVALID, "null argument");
    if (group_rows == 0 || group_rows % 8 != 0) return fail(RRT_ERR_INVALID, "group_rows must be a positive multiple of 8 (the frame is walked in 8x8 tiles)");
    if (group_stride == 0 || first_group >= group_stride) return fail(RRT_ERR_INVALID, "bad group stride / first group");
    const uint32_t rows = rrt_interleaved_row_count(cam->height, group_rows, first_group, group_stride);
    uint64_t n = (uint64_t)rows * cam->width;
    if (n >= (1ull << 31)) return fail(RRT_ERR_LIMIT, "more than 2^31 pixels in one share");
    if (n == 0) return RRT_OK;
    DevCamera dc = to_dev(cam);
    dc.row_group = group_rows;
    dc.row_stride = group_stride;
    dc.row_first = first_group;
    return render_rays(s, &dc, 0, (uint32_t)n, nullptr, nullptr, rgb_dev, depth_dev, object_id_dev, (hipStream_t)hip_stream);
}

int rrt_quantise_pack_device(const float *rgb_dev, const float *depth_dev, size_t n_pixels, uint8_t *rgbd_dev, void *hip_stream) {
    if (n_pixels == 0) return RRT_OK;
    if (!rgb_dev || !depth_dev || !rgbd_dev) return fail(RRT_ERR_INVALID, "null argument");
    if (n_pixels >= (1ull << 31)) return fail(RRT_ERR_LIMIT, "more than 2^31 pixels in one share");
    launch_quantise_pack(rgb_dev, depth_dev, (uint32_t)n_pixels, rgbd_dev, (hipStream_t)hip_stream);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

int rrt_assemble_frame_device(const uint8_t *parts_dev, uint32_t world, uint32_t rows_max, uint32_t width, uint32_t height, uint32_t row_group,
                              uint8_t *rgb_dev, uint8_t *depth_dev, void *hip_stream) {
    if (!parts_dev || !rgb_dev || !depth_dev) return fail(RRT_ERR_INVALID, "null argument");
    if (world == 0 || height >= 65536u) return fail(RRT_ERR_INVALID, "bad world size or frame height");
    // every rank's share must fit the rows_max rows its segment holds
    for (uint32_t r = 0; r < world; r++) {
        uint32_t rows;
        if (row_group) {
            rows = rrt_interleaved_row_count(height, row_group, r, world);
        } else {
            const uint32_t per = height / world;
            rows = r + 1 == world ? height - per * (world - 1) : per;
        }
        if (rows > rows_max) return fail(RRT_ERR_INVALID, "rows_max is smaller than a rank's share of the frame");
    }
    launch_assemble_frame(parts_dev, world, rows_max, width, height, row_group, rgb_dev, depth_dev, (hipStream_t)hip_stream);
    RRT_HIP(hipGetLastError());
    return RRT_OK;
}

int rrt_render(rrt_scene *s, const rrt_camera *cam, uint32_t y0, uint32_t y1, float *rgb, float *depth, uint32_t *object_id) {
    if (!s || !cam) return fail(RRT_ERR_INVALID, "null argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (y0 > y1 || y1 > cam->height) return fail(RRT_ERR_INVALID, "bad row range");
    if (!s->flattened) return fail(RRT_ERR_STATE, "rrt_pre_render has not been called");
    size_t n = (size_t)(y1 - y0) * cam->width;
    DBuf<float> d_rgb, d_depth;
    DBuf<uint32_t> d_id;
    int rc;
    if ((rc = d_rgb.ensure(n * 3))) return rc;
    if ((rc = d_depth.ensure(n))) return rc;
    if ((rc = d_id.ensure(n))) return rc;
    rc = rrt_render_device(s, cam, y0, y1, d_rgb.p, d_depth.p, d_id.p, nullptr);
    if (rc) return rc;
    if (rgb) RRT_HIP(hipMemcpy(rgb, d_rgb.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (depth) RRT_HIP(hipMemcpy(depth, d_depth.p, n * sizeof(float), hipMemcpyDeviceToHost));
    if (object_id) RRT_HIP(hipMemcpy(object_id, d_id.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RRT_OK;
}

int rrt_raytrace(rrt_scene *s, const float *origins, const float *directions, size_t n, float *rgb, float *depth, uint32_t *object_id) {
    if (!s || !origins || !directions) return fail(RRT_ERR_INVALID, "null argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (!s->flattened) return fail(RRT_ERR_STATE, "rrt_pre_render has not been called");
    if (n >= (1ull << 31)) return fail(RRT_ERR_LIMIT, "more than 2^31 rays");
    DBuf<float> d_o, d_d, d_rgb, d_depth;
    DBuf<uint32_t> d_id;
    int rc;
    if ((rc = d_o.ensure(n * 3)) || (rc = d_d.ensure(n * 3)) || (rc = d_rgb.ensure(n * 3)) || (rc = d_depth.ensure(n)) || (rc = d_id.ensure(n)))
        return rc;
    RRT_HIP(hipMemcpy(d_o.p, origins, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    RRT_HIP(hipMemcpy(d_d.p, directions, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    rc = render_rays(s, nullptr, 0, (uint32_t)n, d_o.p, d_d.p, d_rgb.p, d_depth.p, d_id.p, nullptr);
    if (rc) return rc;
    if (rgb) RRT_HIP(hipMemcpy(rgb, d_rgb.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (depth) RRT_HIP(hipMemcpy(depth, d_depth.p, n * sizeof(float), hipMemcpyDeviceToHost));
    if (object_id) RRT_HIP(hipMemcpy(object_id, d_id.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RRT_OK;
}

static int batch_common(rrt_scene *s, const float *origins, const float *directions, const float *limits, size_t n, uint32_t *object_id,
                        uint32_t *prim_id, float *distance, float *position, float *normal, float *uv, uint8_t *occluded, bool shadow) {
    if (!s || !origins || !directions) return fail(RRT_ERR_INVALID, "null argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (!s->flattened) return fail(RRT_ERR_STATE, "rrt_pre_render has not been called");
    if (n >= (1ull << 31)) return fail(RRT_ERR_LIMIT, "more than 2^31 rays");
    DBuf<float> d_o, d_d, d_lim, d_dist, d_pos, d_nrm, d_uv;
    DBuf<uint32_t> d_id, d_prim;
    DBuf<uint8_t> d_occ;
    int rc;
    if ((rc = d_o.ensure(n * 3)) || (rc = d_d.ensure(n * 3))) return rc;
    RRT_HIP(hipMemcpy(d_o.p, origins, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    RRT_HIP(hipMemcpy(d_d.p, directions, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    BatchTraceArgs A;
    memset(&A, 0, sizeof(A));
    A.S = s->dev.view;
    A.origins = d_o.p;
    A.directions = d_d.p;
    A.n = (uint32_t)n;
    A.lds_slots = s->dev.lds_slots;
    A.err = s->counters.p + 63;
    A.stats = s->stats_dev.p;
    RRT_HIP(hipMemset(s->counters.p, 0, 64 * sizeof(uint32_t)));
    RRT_HIP(hipMemset(s->stats_dev.p, 0, RRT_STAT_WORDS * sizeof(unsigned long long)));
    if (shadow) {
        if (!limits || !occluded) return fail(RRT_ERR_INVALID, "null argument");
        if ((rc = d_lim.ensure(n)) || (rc = d_occ.ensure(n))) return rc;
        RRT_HIP(hipMemcpy(d_lim.p, limits, n * sizeof(float), hipMemcpyHostToDevice));
        A.limits = d_lim.p;
        A.occluded = d_occ.p;
        launch_batch_shadow(A, s->dev.lds_bytes, nullptr);
    } else {
        if ((rc = d_id.ensure(n)) || (rc = d_prim.ensure(n)) || (rc = d_dist.ensure(n)) || (rc = d_pos.ensure(n * 3)) || (rc = d_nrm.ensure(n * 3)) ||
            (rc = d_uv.ensure(n * 2)))
            return rc;
        A.object_id = d_id.p;
        A.prim_id = d_prim.p;
        A.distance = d_dist.p;
        A.position = d_pos.p;
        A.normal = d_nrm.p;
        A.uv = d_uv.p;
        launch_batch_trace(A, s->dev.lds_bytes, nullptr);
    }
    RRT_HIP(hipGetLastError());
    RRT_HIP(hipDeviceSynchronize());
    uint32_t err = 0;
    RRT_HIP(hipMemcpy(&err, s->counters.p + 63, sizeof(err), hipMemcpyDeviceToHost));
    unsigned long long st[ST_COUNT];
    if ((rc = read_stats(s, st))) return rc;
    s->stats.bvh_nodes_visited = st[ST_NODES];
    s->stats.bvh_tris_tested = st[ST_TRIS];
    if (err & RRT_ERR_BIT_INTERNAL) return fail(RRT_ERR_DEVICE, "device-side index guard tripped (internal error)");
    if (err & 1u) return fail(RRT_ERR_OVERFLOW, "HitVec overflow");
    if (shadow) {
        RRT_HIP(hipMemcpy(occluded, d_occ.p, n, hipMemcpyDeviceToHost));
    } else {
        if (object_id) RRT_HIP(hipMemcpy(object_id, d_id.p, n * 4, hipMemcpyDeviceToHost));
        if (prim_id) RRT_HIP(hipMemcpy(prim_id, d_prim.p, n * 4, hipMemcpyDeviceToHost));
        if (distance) RRT_HIP(hipMemcpy(distance, d_dist.p, n * 4, hipMemcpyDeviceToHost));
        if (position) RRT_HIP(hipMemcpy(position, d_pos.p, n * 12, hipMemcpyDeviceToHost));
        if (normal) RRT_HIP(hipMemcpy(normal, d_nrm.p, n * 12, hipMemcpyDeviceToHost));
        if (uv) RRT_HIP(hipMemcpy(uv, d_uv.p, n * 8, hipMemcpyDeviceToHost));
    }
    return RRT_OK;
}

int rrt_trace(rrt_scene *s, const float *origins, const float *directions, size_t n, uint32_t *object_id, uint32_t *prim_id, float *distance,
              float *position, float *normal, float *uv) {
    return batch_common(s, origins, directions, nullptr, n, object_id, prim_id, distance, position, normal, uv, nullptr, false);
}
int rrt_shadowtrace(rrt_scene *s, const float *origins, const float *directions, const float *limits, size_t n, uint8_t *occluded) {
    return batch_common(s, origins, directions, limits, n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, occluded, true);
}

int rrt_get_stats(const rrt_scene *s, rrt_stats *out) {
    if (!s || !out) return fail(RRT_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lock(const_cast<rrt_scene *>(s)->render_mutex);
    // the calling thread's own last frame, if it rendered on this handle since the scene last changed
    *out = (t_stats.scene == s && t_stats.epoch == s->epoch) ? t_stats.stats : s->stats;
    return RRT_OK;
}

}  // extern "C"
