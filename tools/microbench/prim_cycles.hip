// Cycles per evaluation of each primitive of the trace loop, measured in a lone wave (latency) and with 12 waves
// per CU (throughput): tools/microbench/build.sh && ./prim_cycles on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "dev_trace.h"
using namespace rrt;
extern __shared__ float4 g_smem[];
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ Lds mk_lds() {
    Lds l; l.nthreads = blockDim.x; l.tid = threadIdx.x; l.le = (float2 *)g_smem; l.stack = (int *)g_smem; return l;
}
// every kernel: `iters` evaluations, ray perturbed per iteration so nothing is hoisted; returns cycles of lane 0
template <int WHAT>
__global__ void k_prim(const ProgRec *recs, uint32_t n_rec, const float *rays, int iters, float *sink, long long *cycles) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    f3 o = mk3(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), d = mk3(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
    Lds lds = mk_lds();
    Counters cnt = {0, 0, 0, 0};
    float acc = 0.0f;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        f3 oo = o + mk3(acc * 1e-30f, 0.0f, 0.0f);
        if (WHAT == 0) {  // cuboid
            bool hf, hb; float tf, tb; uint32_t sf, sb;
            cuboid_hits(recs[0], oo, d, hf, tf, sf, hb, tb, sb);
            acc += (hf ? tf : 0.0f) + (hb ? tb : 0.0f);
        } else if (WHAT == 1) {  // quadric
            float t0_, t1_;
            if (quadric_hits(recs[1].p, oo, d, t0_, t1_)) acc += t0_ + t1_;
        } else if (WHAT == 2) {  // sphere
            float t0_, t1_;
            if (sphere_hits(rec4(recs[2]), oo, d, t0_, t1_)) acc += t0_ + t1_;
        } else if (WHAT == 3) {  // plane
            float t0_, t1_; bool e0, e1;
            if (plane_hits(rec4(recs[3]), oo, d, t0_, e0, t1_, e1)) acc += t0_;
        } else if (WHAT == 4) {  // bound test
            RayBox rb = make_raybox(oo, d);
            acc += bound_hit(recs[4], rb, oo) ? 1.0f : 0.0f;
        } else if (WHAT == 5) {  // whole object program recs[5 .. n_rec) through the real scan loop
            DevScene S; memset(&S, 0, sizeof(S));
            S.prog = recs + 5; S.n_prog = n_rec - 5; S.has_bounds = 1;
            TraceResult r;
            trace(S, oo, d, lds, cnt, r);
            acc += r.object != 0xFFFFFFFFu ? r.t : 0.0f;
        } else if (WHAT == 6) {  // one leaf-leaf merge: cuboid list + sphere list, intersection
            uint32_t n0 = leaf_list(recs[0], 0, oo, d, lds, 0);
            uint32_t n1 = leaf_list(recs[2], 2, oo, d, lds, 1);
            uint32_t no;
            csg_merge(OP_CSG_INTERSECTION, lds, 0, n0, 1, n1, 2, no, cnt);
            acc += (float)no;
        } else if (WHAT == 7) {  // the two leaves of case 6 alone
            uint32_t n0 = leaf_list(recs[0], 0, oo, d, lds, 0);
            uint32_t n1 = leaf_list(recs[2], 2, oo, d, lds, 1);
            acc += (float)(n0 + n1);
        }
    }
    long long t1 = clock64();
    sink[i] = acc + (float)cnt.overflow;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main(int argc, char **argv) {
    // records: 0 cuboid, 1 quadric (unit sphere as quadric), 2 sphere, 3 plane, 4 bound, 5.. = program file (argv[1], raw ProgRec array)
    std::vector<ProgRec> recs(5);
    memset(recs.data(), 0, sizeof(ProgRec) * 5);
    recs[0].op = OP_CUBOID; float cub[12] = {-1.0001f, -1.0001f, -1.0001f, 1.0001f, 1.0001f, 1.0001f, -1, -1, -1, -1, -1, -1};
    memcpy(recs[0].p, cub, sizeof(cub));
    recs[1].op = OP_QUADRIC; recs[1].p[0] = 1; recs[1].p[4] = 1; recs[1].p[7] = 1; recs[1].p[9] = -1;
    recs[2].op = OP_SPHERE; recs[2].p[3] = 1.0f;
    recs[3].op = OP_PLANE; recs[3].p[1] = 1.0f; recs[3].p[3] = 1.5f;
    recs[4].op = OP_BOUND; float bb[7] = {-1, -1, -1, 1, 1, 1, 3.2e-5f}; memcpy(recs[4].p, bb, sizeof(bb));
    if (argc > 1) {
        FILE *f = fopen(argv[1], "rb");
        if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
        ProgRec r;
        while (fread(&r, sizeof(r), 1, f) == 1) recs.push_back(r);
        fclose(f);
    }
    const int BLOCK = 64, MAXBLOCKS = 256 * 12, iters = 200;
    std::vector<float> rays((size_t)BLOCK * MAXBLOCKS * 6);
    uint32_t seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (float)(seed >> 8) / 16777216.0f; };
    for (size_t i = 0; i < rays.size() / 6; i++) {
        float o[3] = {(rnd() - 0.5f) * 2.0f, (rnd() - 0.5f) * 2.0f, -5.0f};
        float t[3] = {(rnd() - 0.5f) * 2.4f, (rnd() - 0.5f) * 2.4f, (rnd() - 0.5f) * 2.0f};
        float dx = t[0] - o[0], dy = t[1] - o[1], dz = t[2] - o[2], l = sqrtf(dx * dx + dy * dy + dz * dz);
        float v[6] = {o[0], o[1], o[2], dx / l, dy / l, dz / l};
        memcpy(&rays[i * 6], v, sizeof(v));
    }
    ProgRec *d_recs; float *d_rays, *d_sink; long long *d_cyc;
    HIPCHECK(hipMalloc(&d_recs, recs.size() * sizeof(ProgRec)));
    HIPCHECK(hipMalloc(&d_rays, rays.size() * 4));
    HIPCHECK(hipMalloc(&d_sink, (size_t)BLOCK * MAXBLOCKS * 4));
    HIPCHECK(hipMalloc(&d_cyc, MAXBLOCKS * 8));
    HIPCHECK(hipMemcpy(d_recs, recs.data(), recs.size() * sizeof(ProgRec), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_rays, rays.data(), rays.size() * 4, hipMemcpyHostToDevice));
    const char *names[8] = {"cuboid", "quadric", "sphere", "plane", "bound", "program(trace)", "2 leaves+merge", "2 leaves"};
    size_t lds = 4 * 6 * BLOCK * 8;
    for (int what = 0; what < 8; what++) {
        if (what == 5 && recs.size() <= 5) continue;
        for (int blocks : {1, 256 * 4, 256 * 12}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; rep++) {
                if (rep == 1) hipEventRecord(e0, 0);
#define LAUNCH(W) hipLaunchKernelGGL(k_prim<W>, dim3(blocks), dim3(BLOCK), lds, 0, d_recs, (uint32_t)recs.size(), d_rays, iters, d_sink, d_cyc)
                switch (what) { case 0: LAUNCH(0); break; case 1: LAUNCH(1); break; case 2: LAUNCH(2); break; case 3: LAUNCH(3); break;
                                case 4: LAUNCH(4); break; case 5: LAUNCH(5); break; case 6: LAUNCH(6); break; default: LAUNCH(7); }
                if (rep == 1) hipEventRecord(e1, 0);
                HIPCHECK(hipDeviceSynchronize());
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> cyc(blocks);
            HIPCHECK(hipMemcpy(cyc.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost));
            double s = 0; for (long long c : cyc) s += (double)c;
            printf("%-16s waves/CU %2d: %8.1f clock64 ticks, %8.1f ns per evaluation (kernel %.1f us)\n", names[what], blocks == 1 ? 0 : blocks / 256,
                   s / blocks / iters, ms * 1e6 / iters, ms * 1e3);
        }
    }
    return 0;
}
