// Records the step sequence of every BVH walk (node visits and leaves) of a set of camera rays and of their shadow rays, by
// running librrt's device trace code on the CPU (one ray at a time: the scene must have no bound records).  The sequences
// feed tools/walk_schedule_sim.py, which replays them under different wave schedules (which lanes step when).
//   walk_trace scene.bin rays.f32 out.bin lx ly lz        (scene.bin: RRT_DUMP_SCENE, rays: [n][6] f32, l: point light)
// out.bin: per ray  u32 n_primary, u32 n_shadow, then the steps as bytes (0 = node visit, k = leaf of k triangles)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
static std::vector<unsigned char> *g_log = nullptr;
#define RRT_WALK_LOG(step) do { if (g_log) g_log->push_back((unsigned char)(step)); } while (0)
#include "dev_trace.h"
using namespace rrt;
bool __any(bool x) { return x; }
unsigned long long __ballot(bool x) { return x ? 1ull : 0ull; }
template <class T> static std::vector<T> get(FILE *f) {
    uint64_t hdr[2];
    if (fread(hdr, sizeof(hdr), 1, f) != 1) { fprintf(stderr, "short file\n"); exit(2); }
    if (hdr[0] != sizeof(T)) { fprintf(stderr, "element size\n"); exit(2); }
    std::vector<T> v(hdr[1]);
    if (hdr[1] && fread(v.data(), sizeof(T), hdr[1], f) != hdr[1]) { fprintf(stderr, "short array\n"); exit(2); }
    return v;
}
int main(int argc, char **argv) {
    if (argc < 7) return 2;
    FILE *f = fopen(argv[1], "rb");
    auto misc = get<uint32_t>(f);
    auto prog = get<ProgRec>(f);
    auto plane_cold = get<PlaneCold>(f);
    auto meshes = get<MeshInfo>(f);
    auto tris = get<TriHot>(f);
    auto tri_index = get<uint32_t>(f);
    auto tri_normals = get<float>(f);
    auto nodes = get<BvhNodeQ>(f);
    auto mesh_inf = get<uint32_t>(f);
    fclose(f);
    DevScene S;
    memset(&S, 0, sizeof(S));
    S.prog = prog.data(); S.n_prog = (uint32_t)prog.size(); S.has_bounds = misc[2]; S.has_csg = misc[3];
    S.plane_cold = plane_cold.data(); S.meshes = meshes.data(); S.tris = tris.data(); S.tri_index = tri_index.data();
    S.tri_normals = tri_normals.data(); S.nodes = nodes.data(); S.mesh_inf = mesh_inf.data();
    S.n_plane_cold = (uint32_t)plane_cold.size(); S.n_meshes = (uint32_t)meshes.size();
    if (S.has_bounds) { fprintf(stderr, "scene has bound records\n"); return 2; }
    FILE *r = fopen(argv[2], "rb");
    fseek(r, 0, SEEK_END); long bytes = ftell(r); fseek(r, 0, SEEK_SET);
    std::vector<float> rays(bytes / 4);
    if (fread(rays.data(), 4, rays.size(), r) != rays.size()) return 2;
    fclose(r);
    const f3 light = mk3((float)atof(argv[4]), (float)atof(argv[5]), (float)atof(argv[6]));
    std::vector<int> stack(256);
    Lds lds; lds.le = nullptr; lds.stack = stack.data(); lds.nthreads = 1; lds.tid = 0;
    FILE *out = fopen(argv[3], "wb");
    size_t n = rays.size() / 6;
    for (size_t i = 0; i < n; i++) {
        f3 o = mk3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), d = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
        std::vector<unsigned char> a, b;
        Counters cnt = {0, 0, 0, 0};
        TraceResult tr;
        g_log = &a;
        trace<FEAT_MESH>(S, o, d, lds, cnt, tr);
        g_log = nullptr;
        if (tr.object != 0xFFFFFFFFu) {
            HitAttr h;
            eval_hit<FEAT_MESH>(S, o, d, tr.instr, tr.sub, tr.t, h);
            f3 ldir = norm3(h.position - light);
            if (!(dot3(ldir, h.normal) > 0.0f)) {
                f3 sd = -ldir, so = h.position + sd * 0.0001f;
                g_log = &b;
                shadowtrace<FEAT_MESH>(S, so, sd, len3(ldir), lds, cnt);
                g_log = nullptr;
            }
        }
        if (getenv("WALK_TRACE_MIRROR")) {  // third column instead of the shadow rays: the mirror ray of the hit (global_material.rs:76-85)
            b.clear();
            if (tr.object != 0xFFFFFFFFu) {
                HitAttr h;
                eval_hit<FEAT_MESH>(S, o, d, tr.instr, tr.sub, tr.t, h);
                f3 ad = norm3(reflect3(h.normal, d)), ao = h.position + ad * 0.0001f;
                TraceResult t2;
                g_log = &b;
                trace<FEAT_MESH>(S, ao, ad, lds, cnt, t2);
                g_log = nullptr;
            }
        }
        uint32_t hdr[2] = {(uint32_t)a.size(), (uint32_t)b.size()};
        fwrite(hdr, 4, 2, out);
        fwrite(a.data(), 1, a.size(), out);
        fwrite(b.data(), 1, b.size(), out);
    }
    fclose(out);
    return 0;
}
