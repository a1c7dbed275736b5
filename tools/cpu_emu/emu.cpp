// Runs librrt's DEVICE trace code on the CPU in waves of 64 lanes (one thread per lane; __any is a real collective
// over the lanes that have not left the current call, so a lane sequence that is not wave-uniform deadlocks or
// is reported), under AddressSanitizer, on a scene dumped by
// RRT_DUMP_SCENE=<file> (runtime.cpp) and rays from a raw f32 file [n][6]; writes object / distance per ray.
// Used to find indexing faults before a kernel goes to the GPU box.  Build: tools/cpu_emu/build.sh
#include <stdio.h>
#include <stdlib.h>
#include <vector>
// every LDS access must stay inside the block's allocation (g_lds_elems 8-byte elements, the host's lds_bytes / 8)
// and inside the accessing lane's own column: element index % block size == lane
static size_t g_lds_elems = 0;
#define RRT_LDS_CHECK(lds, elem)                                                                                         \
    do {                                                                                                                 \
        size_t e_ = (elem);                                                                                              \
        if (e_ >= g_lds_elems || e_ % (lds).nthreads != (lds).tid) {                                                    \
            fprintf(stderr, "LDS CHECK FAILED: element %zu (allocation %zu), lane %u of %u\n", e_, g_lds_elems, (lds).tid, (lds).nthreads); \
            abort();                                                                                                     \
        }                                                                                                                \
    } while (0)
#define RRT_LDS_CHECK_WORD(lds, word)                                                                                    \
    do {                                                                                                                 \
        size_t w_ = (word);                                                                                              \
        if (w_ >= 2 * g_lds_elems || w_ % (lds).nthreads != (lds).tid) {                                                \
            fprintf(stderr, "LDS CHECK FAILED: word %zu (allocation %zu), lane %u of %u\n", w_, 2 * g_lds_elems, (lds).tid, (lds).nthreads); \
            abort();                                                                                                     \
        }                                                                                                                \
    } while (0)
#include "dev_trace.h"
#include <atomic>
#include <barrier>
#include <condition_variable>
#include <mutex>
#include <thread>
using namespace rrt;

struct Wave {
    std::mutex m;
    std::condition_variable cv;
    int active = 0, arrived = 0;
    bool acc = false, result = false;
    unsigned long long bits = 0, bits_result = 0;
    uint64_t gen = 0;
    void complete() {
        result = acc;
        bits_result = bits;
        bits = 0;
        acc = false;
        arrived = 0;
        gen++;
        cv.notify_all();
    }
    bool any(bool x) {
        std::unique_lock<std::mutex> l(m);
        acc |= x;
        if (++arrived == active) {
            complete();
            return result;
        }
        uint64_t g = gen;
        if (!cv.wait_for(l, std::chrono::seconds(20), [&] { return gen != g; })) {
            fprintf(stderr, "DEADLOCK: lanes disagree on the sequence of wave collectives\n");
            abort();
        }
        return result;
    }
    unsigned long long ballot(bool x, int lane) {
        std::unique_lock<std::mutex> l(m);
        if (x) bits |= 1ULL << lane;
        if (++arrived == active) {
            complete();
            return bits_result;
        }
        uint64_t g = gen;
        if (!cv.wait_for(l, std::chrono::seconds(20), [&] { return gen != g; })) {
            fprintf(stderr, "DEADLOCK: lanes disagree on the sequence of wave collectives\n");
            abort();
        }
        return bits_result;
    }
    void leave() {  // lane returns from the call
        std::unique_lock<std::mutex> l(m);
        active--;
        if (active > 0 && arrived == active) complete();
    }
    void join_all(int n) {
        std::unique_lock<std::mutex> l(m);
        active = n;
        arrived = 0;
        acc = false;
    }
};
static thread_local Wave *tl_wave = nullptr;
static thread_local int tl_lane = 0;
bool __any(bool x) { return tl_wave->any(x); }
unsigned long long __ballot(bool x) { return tl_wave->ballot(x, tl_lane); }
void emu_wave_sync() { (void)tl_wave->any(false); }
static std::atomic<long long> g_cull_visited{0}, g_cull_total{0};
void emu_cull_stat(int visited, int total) { g_cull_visited += visited; g_cull_total += total; }
template <class T> static std::vector<T> get(FILE *f) {
    uint64_t hdr[2];
    if (fread(hdr, sizeof(hdr), 1, f) != 1) { fprintf(stderr, "short file\n"); exit(2); }
    if (hdr[0] != sizeof(T)) { fprintf(stderr, "element size %llu != %zu\n", (unsigned long long)hdr[0], sizeof(T)); exit(2); }
    std::vector<T> v(hdr[1]);
    if (hdr[1] && fread(v.data(), sizeof(T), hdr[1], f) != hdr[1]) { fprintf(stderr, "short array\n"); exit(2); }
    return v;
}
int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: emu scene.bin rays.f32 out.bin\n"); return 2; }
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
    FILE *r = fopen(argv[2], "rb");
    fseek(r, 0, SEEK_END); long bytes = ftell(r); fseek(r, 0, SEEK_SET);
    std::vector<float> rays(bytes / 4);
    if (fread(rays.data(), 4, rays.size(), r) != rays.size()) return 2;
    fclose(r);
    size_t n = rays.size() / 6;
    std::vector<uint32_t> results(n * 4);
    uint32_t overflow = 0;
    std::mutex om;
    // wave width: 64 as on the device; 1 makes every ray its own wave, so each bound decides for that ray alone (the
    // strictest check of the skip logic)
    const int W = getenv("EMU_WAVE") ? atoi(getenv("EMU_WAVE")) : 64;
    // 64 lane threads live for the whole run; a std::barrier separates waves and phases (trace, then shadowtrace)
    const uint32_t BLOCK = 256;
    size_t list_bytes = (size_t)misc[0] * 6 * BLOCK * 8, stack_bytes = (size_t)((misc[1] + 1) / 2) * BLOCK * 8;
    const bool csg_mesh = (misc[3] & 2u) != 0;  // a mesh leaf in a CSG tree: the stack lies behind the hit lists
    g_lds_elems = (csg_mesh ? list_bytes + stack_bytes : std::max(list_bytes, stack_bytes)) / 8;
    // ONE buffer, laid out and sized as runtime.cpp does for a 256-thread block (lists and BVH stack alias); the wave
    // emulated here plays lanes 64..127 of that block.  Refilled with garbage for every wave, as on the device.
    std::vector<float2> lists(std::max<size_t>(g_lds_elems, 1));
    const float shadow_limit = getenv("EMU_SHADOW_LIMIT") ? (float)atof(getenv("EMU_SHADOW_LIMIT")) : 6.0f;
    Wave wave;
    std::barrier<> phase_sync(W);
    std::vector<std::thread> th;
    for (int lane = 0; lane < W; lane++)
        th.emplace_back([&, lane] {
            tl_wave = &wave;
            tl_lane = lane;
            for (size_t base = 0; base < n; base += W) {
                int lanes = (int)std::min<size_t>(W, n - base);
                for (int phase = 0; phase < 2; phase++) {
                    if (lane == 0) {
                        wave.join_all(lanes);
                        if (phase == 0) std::fill(lists.begin(), lists.end(), float2{1.5f, __uint_as_float(0xFFFFFFFFu)});
                    }
                    phase_sync.arrive_and_wait();
                    if (lane < lanes) {
                        size_t i = base + lane;
                        Lds lds; lds.le = lists.data(); lds.stack = reinterpret_cast<int *>(lists.data()); lds.nthreads = BLOCK; lds.tid = (64u + (uint32_t)lane) % BLOCK;
                        lds.stack_base = csg_mesh ? misc[0] * RRT_LIST_CAP : 0u;
                        Counters cnt = {0, 0, 0, 0};
                        f3 o = mk3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), d = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
                        if (phase == 0) {
                            TraceResult tr;
                            // the instance the launcher would pick: scenes without CSG use the plain stack layout
                            if (S.has_csg) trace<FEAT_ALL>(S, o, d, lds, cnt, tr);
                            else trace<FEAT_MESH>(S, o, d, lds, cnt, tr);
                            results[i * 4] = tr.object;
                            results[i * 4 + 1] = __float_as_uint(tr.object == 0xFFFFFFFFu ? 0.0f : tr.t);
                            uint32_t prim = 0;  // as k_batch_trace computes it
                            if (tr.object != 0xFFFFFFFFu) {
                                const ProgRec &in = S.prog[tr.instr];
                                uint32_t op = in.op & 0xFFu;
                                if (in.op & PF_CSG) {
                                    if (op == OP_MESH && tr.sub >= S.meshes[in.aux].tri_count) abort();  // the winner's triangle
                                    for (uint32_t k = tr.instr; !(S.prog[k].op & PF_FIRST);) {
                                        k--;
                                        if ((S.prog[k].op & 0xFFu) < OP_CSG_UNION) prim++;
                                    }
                                } else if (op == OP_MESH) {
                                    prim = S.tri_index[S.meshes[in.aux].tri_first + tr.sub];
                                } else if (op == OP_CUBOID) {
                                    prim = tr.sub;
                                    if (in.aux + (tr.sub >> 1) >= plane_cold.size()) abort();
                                }
                            }
                            results[i * 4 + 2] = prim;
                        } else {
                            results[i * 4 + 3] = (S.has_csg ? shadowtrace<FEAT_ALL>(S, o, d, shadow_limit, lds, cnt) : shadowtrace<FEAT_MESH>(S, o, d, shadow_limit, lds, cnt)) ? 1u : 0u;
                        }
                        wave.leave();
                        if (cnt.overflow) { std::lock_guard<std::mutex> g(om); overflow |= cnt.overflow; }
                    }
                    phase_sync.arrive_and_wait();
                }
            }
        });
    for (auto &t : th) t.join();
    FILE *out = fopen(argv[3], "wb");
    fwrite(results.data(), 4, results.size(), out);
    fclose(out);
    fprintf(stderr, "%zu rays, overflow %u\n", n, overflow);
    if (g_cull_total) fprintf(stderr, "object visits after the wave cull: %.1f %% of lane-objects\n", 100.0 * (double)g_cull_visited / (double)std::max<long long>(1, g_cull_total));
    return 0;
}
