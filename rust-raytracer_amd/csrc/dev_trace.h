// dev_trace.h — gfx950 device functions: ray/object intersection as per-ray interval lists,
// the nearest-entering-hit search, BVH traversal with per-lane LDS stacks, and lazy evaluation
// of the winning hit's attributes.
//
// Every lane owns one ray; the loop over objects and each object's program is wave-uniform, so
// scene data is fetched through the scalar cache and divergence is limited to data-dependent
// branches (discriminant < 0, hit-list lengths, BVH descent).
//
// Exactness: distances, ids and hit attributes must equal the reference's f32 results bit for bit
// (SURVEY.md §8a), so the formulas below keep the reference's operation order and nothing here may
// be contracted into FMAs (the file is compiled with -ffp-contract=off).  The BVH only prunes:
// its box test is deliberately conservative, the triangle test is the reference's.
#pragma once
#include "dev_scene.h"

namespace rrt {

#define RRT_LIST_CAP 6  // HitVec capacity, core/hit.rs:56
#define RRT_F32_MAX 3.40282347e+38f
#define RRT_INF __builtin_huge_valf()

// hit-list entry meta word: bit 31 = entering (after CSG relabelling), bits 8..23 = instruction
// index inside the object's program, bits 0..7 = sub-hit selector of that primitive
#define META_ENTER 0x80000000u
__device__ __forceinline__ uint32_t mk_meta(bool entering, uint32_t instr, uint32_t sub) {
    return (entering ? META_ENTER : 0u) | (instr << 8) | sub;
}

// Per-block LDS views.  lists: [slot][entry][thread] of 8-byte entries; the BVH stack aliases the lists (a lane never
// uses both at once; scenes with a mesh leaf inside a CSG tree put it behind them: stack_base) and therefore MUST stay inside the lane's own 8-byte
// column: the other waves of the block are at other records, and a stack laid out as [entry][thread] of 4-byte words
// would put lane t's words into the list columns of lanes t/2 of another wave.  Two stack words per column element.
// tools/cpu_emu defines this to assert that an access stays inside the block's LDS allocation and inside the
// accessing lane's own 8-byte column (argument: index of the 8-byte element)
#ifndef RRT_LDS_CHECK
#define RRT_LDS_CHECK(lds, elem)
#endif
#ifndef RRT_LDS_CHECK_WORD  // same for the plain stack layout (argument: index of the 4-byte word)
#define RRT_LDS_CHECK_WORD(lds, word)
#endif
struct Lds {
    float2 *le;  // hit-list entries: (distance, meta word) as one 8-byte element -> ds_read_b64 / ds_write_b64
    int *stack;
    uint32_t nthreads, tid;
    uint32_t stack_base = 0;  // 8-byte elements in front of the BVH stack: 0 (the stack aliases the hit lists: a lane never
                              // needs both) unless a CSG tree has a mesh leaf, which is walked while hit lists are live
    __device__ __forceinline__ void get(uint32_t slot, uint32_t e, float &t, uint32_t &m) const {
        RRT_LDS_CHECK(*this, (slot * RRT_LIST_CAP + e) * nthreads + tid);
        float2 v = le[(slot * RRT_LIST_CAP + e) * nthreads + tid];
        t = v.x;
        m = __float_as_uint(v.y);
    }
    __device__ __forceinline__ void put(uint32_t slot, uint32_t e, float t, uint32_t m) {
        RRT_LDS_CHECK(*this, (slot * RRT_LIST_CAP + e) * nthreads + tid);
        le[(slot * RRT_LIST_CAP + e) * nthreads + tid] = make_float2(t, __uint_as_float(m));
    }
    __device__ __forceinline__ int &st(uint32_t e) {
        RRT_LDS_CHECK(*this, (stack_base + (e >> 1)) * nthreads + tid);
        return stack[(((stack_base + (e >> 1)) * nthreads + tid) << 1) + (e & 1u)];
    }
    // Scenes without CSG keep nothing but traversal stacks in LDS, so the stack can be plain 4-byte words [word][lane]:
    // the next word is one constant stride away (one add per push or pop instead of the column arithmetic above).
    __device__ __forceinline__ int &word(uint32_t e) {
        RRT_LDS_CHECK_WORD(*this, e * nthreads + tid);
        return stack[e * nthreads + tid];
    }
};
// the traversal stack of one lane in either layout
template <bool PLAIN>
struct LaneStack {
    Lds &lds;
    uint32_t sp;
    __device__ __forceinline__ void push(int v) {
        if (PLAIN)
            lds.word(sp) = v;
        else
            lds.st(sp) = v;
        sp++;
    }
    __device__ __forceinline__ int pop() {
        sp--;
        return PLAIN ? lds.word(sp) : lds.st(sp);
    }
    __device__ __forceinline__ bool empty() const { return sp == 0; }
};

struct Counters {
    uint32_t nodes, tris, shadow, overflow;
    uint32_t truncated = 0;  // rays or photon paths the recursion caps cut off (rrt_stats.paths_truncated / vueons_truncated)
#ifdef RRT_WALK_STATS  // instrumentation builds only (tools/walk_stats.sh): three 64-bit sums that travel in the photon statistics' slots
    unsigned long long ws[3] = {0, 0, 0};
#endif
};
#ifdef RRT_WALK_STATS
// one count per WAVE for the code path that executes this (the first active lane counts)
#define RRT_WS_WAVE_COUNT(slot)                                                                  \
    do {                                                                                         \
        const unsigned long long m_ = __ballot(1);                                               \
        if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m_) - 1u) cnt.ws[slot] += 1ull;  \
    } while (0)
// wave-level time stamps between the phases of k_trace_shade (RRT_WALK_STATS 3, 4): the first active lane adds the difference
#define RRT_WS_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define RRT_WS_ADD(slot, dt)                                                                     \
    do {                                                                                         \
        const unsigned long long m_ = __ballot(1);                                               \
        if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m_) - 1u) cnt.ws[slot] += (dt);  \
    } while (0)
#else
#define RRT_WS_WAVE_COUNT(slot)
#define RRT_WS_STAMP(var)
#define RRT_WS_ADD(slot, dt)
#endif
#ifndef RRT_WALK_LOG  // tools/cpu_emu/walk_trace.cpp records every step of a walk: 0 = node visit, n = leaf of n triangles
#define RRT_WALK_LOG(step)
#endif
#if RRT_WALK_STATS == 1
#define RRT_WS_TICK(w) (w).ws_last = __builtin_amdgcn_s_memtime()
#else
#define RRT_WS_TICK(w)
#endif

struct TraceResult {
    float t;
    uint32_t object;  // 0xFFFFFFFF = miss
    uint32_t instr;   // global instruction index of the leaf that produced the hit
    uint32_t sub;     // sub-hit selector (triangle BVH slot for meshes)
};

// ---- primitives: distances only ------------------------------------------------------------------

// Plane::intersect (plane_object.rs:54-113), distances and facing only.  Returns the hit count (0/2).
__device__ __forceinline__ int plane_hits(float4 nd, f3 o, f3 d, float &t0, bool &e0, float &t1, bool &e1) {
    f3 n = mk3(nd.x, nd.y, nd.z);
    float U = dot3(n, o) + nd.w;
    float V = dot3(n, d);
    if (V == 0.0f) {
        if (U >= 0.0f) {
            t0 = -RRT_INF; e0 = true;
            t1 = RRT_INF; e1 = false;
            return 2;
        }
        return 0;
    }
    float t = U / -V;
    if (V > 0.0f) {
        t0 = -RRT_INF; e0 = true;
        t1 = t; e1 = false;
    } else {
        t0 = t; e0 = true;
        t1 = RRT_INF; e1 = false;
    }
    return 2;
}

// Sphere::intersect (sphere_object.rs:35-53): note the division by 2.0, not 2a
__device__ __forceinline__ bool sphere_hits(float4 cr, f3 o, f3 d, float &t0, float &t1) {
    f3 ro = o - mk3(cr.x, cr.y, cr.z);
    float a = dot3(d, d);
    float b = 2.0f * dot3(d, ro);
    float c = dot3(ro, ro) - cr.w * cr.w;
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return false;
    float ds = sqrtf(disc);
    t0 = (-b - ds) / 2.0f;
    t1 = (-b + ds) / 2.0f;
    return true;
}

// Quadratic::intersect (quadratic_object.rs:43-89)
__device__ __forceinline__ bool quadric_hits(const float *q, f3 P, f3 D, float &t0, float &t1) {
    float a = q[0], b = q[1], c = q[2], d = q[3];
    float e = q[4], f = q[5], g = q[6], h = q[7];
    float i = q[8], j = q[9];
    float Aq = a * (D.x * D.x) + 2.0f * b * D.x * D.y + 2.0f * c * D.x * D.z + e * (D.y * D.y) + 2.0f * f * D.y * D.z +
               h * (D.z * D.z);
    float Bq = 2.0f * (a * P.x * D.x + b * (P.x * D.y + P.y * D.x) + c * (P.x * D.z + D.x * P.z) + d * D.x + e * P.y * D.y +
                       f * (P.y * D.z + D.y * P.z) + g * D.y + h * P.z * D.z + i * D.z);
    float Cq = a * (P.x * P.x) + 2.0f * b * P.x * P.y + 2.0f * c * P.x * P.z + 2.0f * d * P.x + e * (P.y * P.y) +
               2.0f * f * P.y * P.z + 2.0f * g * P.y + h * (P.z * P.z) + 2.0f * i * P.z + j;
    if (fabsf(Aq) == 0.0f) return false;
    float disc = Bq * Bq - 4.0f * Aq * Cq;
    if (disc < 0.0f) return false;
    disc = sqrtf(disc);
    float ta = (-Bq - disc) / (2.0f * Aq);
    float tb = (-Bq + disc) / (2.0f * Aq);
    if (ta > tb) {
        t0 = tb;
        t1 = ta;
    } else {
        t0 = ta;
        t1 = tb;
    }
    return true;
}

// Cuboid::intersect (cuboid_object.rs:94-148): six half-space planes; the entering candidate with
// the LARGEST distance and the exiting candidate with the SMALLEST distance survive (later planes
// replace on ties).  sub = plane * 2 + position of the hit in that plane's list.
__device__ __forceinline__ void cuboid_hits(const ProgRec &C, f3 o, f3 d, bool &have_first, float &t_first, uint32_t &sub_first,
                                            bool &have_back, float &t_back, uint32_t &sub_back) {
    // Branch-free restatement of the reference's loop: "no hit yet" is the sentinel -1 for the entering side (every
    // accepted t is >= 0, so `!(have && t_first > t)` == `t_first <= t`) and +inf for the exit side (accepted t's are
    // finite, so `!(have && t_back < t)` == `t_back >= t`).  NaN hit points fail the inside test as in the reference.
    float tf = -1.0f, tb = RRT_INF;
    uint32_t sf = 0, sb = 0;
    // plane order = CuboidPlanes::iter (cuboid_object.rs:27-37): left, right, up, down, front, back
    const f3 normals[6] = {{-1.0f, 0.0f, 0.0f}, {1.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, -1.0f, 0.0f}, {0.0f, 0.0f, -1.0f}, {0.0f, 0.0f, 1.0f}};
#pragma unroll
    for (uint32_t p = 0; p < 6; p++) {
        f3 n = normals[p];
        float U = dot3(n, o) + C.p[6 + p];
        float V = dot3(n, d);
        float t = U / -V;
        f3 hp = o + d * t;
        bool inside = hp.x >= C.p[0] && hp.x <= C.p[3] && hp.y >= C.p[1] && hp.y <= C.p[4] && hp.z >= C.p[2] && hp.z <= C.p[5];
        // V == 0: both hits infinite, skipped at cuboid_object.rs:120-122; then t < 0 and infinite t
        bool valid = inside & (V != 0.0f) & !(t < 0.0f) & !isinf(t);
        bool entering = !(V > 0.0f);
        bool uf = valid & entering & (tf <= t);
        bool ub = valid & !entering & (tb >= t);
        tf = uf ? t : tf;
        sf = uf ? p * 2 : sf;          // list position 0 when V < 0
        tb = ub ? t : tb;
        sb = ub ? p * 2 + 1 : sb;      // list position 1 when V > 0
    }
    have_first = tf >= 0.0f;
    have_back = tb < RRT_INF;
    t_first = have_first ? tf : 0.0f;
    t_back = have_back ? tb : 0.0f;
    sub_first = sf;
    sub_back = sb;
}

// ---- mesh: BVH2 traversal, per-lane stack in LDS ---------------------------------------------------

// Triangle::intersect (triangle_object.rs:129-160) on the finite plane hit only (the infinite ones
// can never be selected: see DESIGN.md "mesh semantics").  Returns true when the hit point passes
// the same-side test; `entering` is the plane hit's facing.
__device__ __forceinline__ uint32_t tri_index_of(const TriHot &T) {
    uint32_t u;
    memcpy(&u, &T.d_pad.y, sizeof(u));
    return u;
}
template <bool EAGER = false>
__device__ __forceinline__ bool tri_test(const TriHot &T, f3 o, f3 d, float &t, bool &entering) {
    f3 n = mk3(T.a_nx.w, T.b_ny.w, T.c_nz.w);
    float U = dot3(n, o) + T.d_pad.x;
    float V = dot3(n, d);
    // EAGER: no early exit for a ray parallel to the plane.  With the exit the compiler loads the plane, tests V and only then
    // asks for the corners, so every triangle waits for the memory twice; without it the record comes in one round trip
    // (the quotient of a parallel ray is computed and thrown away).  Mirror / glass teapots at 4K: kernels 3.26 / 2.72 ->
    // 3.09 / 2.61 ms.  The fused single-level instance keeps the exit: it sits at its 128-register bound and the wider
    // live range costs it more scalar spills than the round trip (0.741 -> 0.743 ms).
    const bool crosses = V != 0.0f;
    if (!EAGER && !crosses) return false;
    t = U / -V;
    entering = !(V > 0.0f);
    f3 a = mk3(T.a_nx.x, T.a_nx.y, T.a_nx.z), b = mk3(T.b_ny.x, T.b_ny.y, T.b_ny.z), c = mk3(T.c_nz.x, T.c_nz.y, T.c_nz.z);
    f3 p = o + d * t;
    f3 ai = p - a, bi = p - b, ci = p - c;
    f3 ab = b - a, bc = c - b, ca = a - c;  // Triangle::new, triangle_object.rs:51-53
    f3 ab_n = cross3(ai, ab), bc_n = cross3(bi, bc), ca_n = cross3(ci, ca);
    if (EAGER) return ((int)crosses & (int)(dot3(ab_n, bc_n) > 0.0f) & (int)(dot3(bc_n, ca_n) > 0.0f)) != 0;
    return dot3(ab_n, bc_n) > 0.0f && dot3(bc_n, ca_n) > 0.0f;
}

// Conservative ray/box slab test against both children of a node.  The BVH may only prune what the reference's exact
// triangle test would reject anyway; three margins see to that:
//  * the builders pad every box by 2^-16 of its coordinate magnitude (bvh_build.cpp, bvh_lbvh.h) and the quantiser
//    rounds lower faces down and upper faces up on the mesh's 16-bit grid (quantise_bvh);
//  * the box is grown by e = 2^-20 of (the ray origin's magnitude + the grid's magnitude): lower faces are tested
//    against origin + e, upper faces against origin - e.  That covers the rounding of o + d*t in the reference's hit
//    point and of the slab distances here, each of which is ONE fma on the quantised coordinate:
//        t = q * (qscale / d) + (qbase - (o +- e)) / d
//    (which of a pair of faces the ray meets first follows from the sign of 1/d alone: make_slabs)
//  * the far end of the [near, far] interval is widened by a relative 2^-18 (1/d is a 1-ulp reciprocal).
// Reciprocals are clamped to +-2^60 and rays or meshes with coordinates beyond 2^60 (or not finite) skip the pruning
// altogether (every node is visited), so no inf - inf can turn a slab distance into a NaN that min/max would drop on the
// wrong side.  None of this arithmetic is part of the parity contract (it only prunes): it may fuse and approximate.
struct RaySlabs {
    f3 a, bn, bf;             // per axis: slope, offset of the face the ray meets first, offset of the one it leaves through
    uint32_t sx, sy, sz;      // per axis: 0, or 16 when the ray runs towards smaller coordinates (it then enters through the upper face)
};
__device__ __forceinline__ RaySlabs make_slabs(const MeshInfo &M, f3 o, f3 d) {
    const float lim = 1152921504606846976.0f;  // 2^60
    const float omax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
    RaySlabs r;
    r.sx = r.sy = r.sz = 0u;
    if (M.wild || !(omax < lim)) {  // no pruning: every slab distance is 0, every box "hit"
        r.a = r.bn = r.bf = mk3(0.0f, 0.0f, 0.0f);
        return r;
    }
    f3 inv = mk3(__builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.x), -lim, lim), __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.y), -lim, lim),
                 __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d.z), -lim, lim));
    const float e = (omax + M.qmag) * 9.5367431640625e-7f;  // 2^-20
    r.a = mk3(M.qscale[0] * inv.x, M.qscale[1] * inv.y, M.qscale[2] * inv.z);
    const f3 blo = mk3((M.qbase[0] - (o.x + e)) * inv.x, (M.qbase[1] - (o.y + e)) * inv.y, (M.qbase[2] - (o.z + e)) * inv.z);
    const f3 bhi = mk3((M.qbase[0] - (o.x - e)) * inv.x, (M.qbase[1] - (o.y - e)) * inv.y, (M.qbase[2] - (o.z - e)) * inv.z);
    // With a positive reciprocal the lower face (grid word's low half, offset blo) has the smaller slab distance, with a
    // negative one the upper face (high half, bhi): q_lo <= q_hi, blo <= bhi resp. >=, and an fma is monotone.  Which is
    // which is settled here once per ray instead of by a min and a max per face pair and node.
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    r.sx = nx ? 16u : 0u, r.sy = ny ? 16u : 0u, r.sz = nz ? 16u : 0u;
    r.bn = mk3(nx ? bhi.x : blo.x, ny ? bhi.y : blo.y, nz ? bhi.z : blo.z);
    r.bf = mk3(nx ? blo.x : bhi.x, ny ? blo.y : bhi.y, nz ? blo.z : bhi.z);
    return r;
}
// one child of a node: slab distances of its box, entry (near) and whether [max(near, 0), min(far, tmax)] is not empty
// once its far end is widened.  wx / wy / wz: the child's grid words (lo | hi << 16), rotated by the ray's shift their
// LOW half is the face the ray meets first.
__device__ __forceinline__ bool box_test(uint32_t wx, uint32_t wy, uint32_t wz, const RaySlabs &R, float tmax, float &near) {
#define RRT_ROT(w, s) __builtin_amdgcn_alignbit((w), (w), (s))
#define RRT_QN(w) ((float)((w) & 0xFFFFu))
#define RRT_QF(w) ((float)((w) >> 16))
    const uint32_t x = RRT_ROT(wx, R.sx), y = RRT_ROT(wy, R.sy), z = RRT_ROT(wz, R.sz);
    const float n = fmaxf(fmaxf(__builtin_fmaf(RRT_QN(x), R.a.x, R.bn.x), __builtin_fmaf(RRT_QN(y), R.a.y, R.bn.y)), __builtin_fmaf(RRT_QN(z), R.a.z, R.bn.z));
    const float f = fminf(fminf(__builtin_fmaf(RRT_QF(x), R.a.x, R.bf.x), __builtin_fmaf(RRT_QF(y), R.a.y, R.bf.y)), __builtin_fmaf(RRT_QF(z), R.a.z, R.bf.z));
#undef RRT_ROT
#undef RRT_QN
#undef RRT_QF
    const float K = 1.000003814697265625f;  // 1 + 2^-18
    near = n;
    return f >= 0.0f && n <= fminf(f, tmax) * K;
}
// compare-exchange of two (distance, child) pairs: afterwards a is the nearer one
__device__ __forceinline__ void order2(float &ka, int &ca, float &kb, int &cb) {
    const bool swap = kb < ka;
    const float k = swap ? ka : kb;
    const int c = swap ? ca : cb;
    ka = swap ? kb : ka;
    ca = swap ? cb : ca;
    kb = k;
    cb = c;
}

// One lane's walk through a mesh's BVH, as a state that can be advanced step by step: mesh_closest runs one walk to its
// end (the pooled-walk experiment, tools/experiments/pool_kernel.patch, kept 64 of them going and handed a lane whose
// walk was over the next ray; tools/cpu_emu/walk_trace.cpp records the steps).
// "While-while" order: walk_nodes takes a lane down through inner nodes until it holds a leaf (or is done); the wave
// leaves that loop when no lane has a node left, and only then walk_leaf tests the leaves' triangles, for all lanes that
// hold one.
#define RRT_WALK_DONE ((int)0x80000000)  // as a leaf code its first triangle is beyond every mesh (meshes hold < 2^27 triangles)
struct MeshView {  // what the loops read of a MeshInfo, in registers (M is memory the loops would re-read)
    const BvhNodeQ *nodes;
    const TriHot *tris;
    const uint32_t *orig;
    uint32_t node_count, tri_count;
};
__device__ __forceinline__ MeshView mesh_view(const DevScene &S, const MeshInfo &M) {
    return MeshView{S.nodes + M.node_first, S.tris + M.tri_first, S.tri_index + M.tri_first, M.node_count, M.tri_count};
}
struct Walk {
    f3 o, d;
    RaySlabs slabs;
    float best;  // distance of the nearest hit accepted so far; the bound the walk started with until there is one
    uint32_t best_orig, best_slot;
    bool best_entering, found;
    bool ties;  // a hit AT the starting bound is accepted too (for a bound that is the hit of a LATER object record, which loses ties)
    int cur;    // >= 0: inner node, < 0: leaf code, RRT_WALK_DONE: the walk is over
#if RRT_WALK_STATS == 1
    unsigned long long ws_last;  // s_memtime of the lane's last step
#endif
};
__device__ __forceinline__ void walk_begin(Walk &w, const MeshInfo &M, f3 o, f3 d, float tmax, bool ties) {
    w.o = o;
    w.d = d;
    w.slabs = make_slabs(M, o, d);
    w.best = tmax;
    w.best_orig = 0xFFFFFFFFu;
    w.best_slot = 0;
    w.best_entering = w.found = false;
    w.ties = ties;
    w.cur = M.tri_count ? 0 : RRT_WALK_DONE;  // root
}
// the four children of a fetched node tested, the nearest hit one next, the others stacked
template <bool PLAIN_STACK>
__device__ __forceinline__ void walk_node_visit(const BvhNodeQ &N, Walk &w, LaneStack<PLAIN_STACK> &stack, Counters &cnt) {
    int c0 = N.child[0], c1 = N.child[1], c2 = N.child[2], c3 = N.child[3];
    float k0, k1, k2, k3;
    const bool h0 = box_test(N.q[0], N.q[4], N.q[8], w.slabs, w.best, k0);
    const bool h1 = box_test(N.q[1], N.q[5], N.q[9], w.slabs, w.best, k1);
    const bool h2 = box_test(N.q[2], N.q[6], N.q[10], w.slabs, w.best, k2);
    const bool h3 = box_test(N.q[3], N.q[7], N.q[11], w.slabs, w.best, k3);
    cnt.nodes += 4;
    k0 = h0 ? k0 : RRT_INF, k1 = h1 ? k1 : RRT_INF, k2 = h2 ? k2 : RRT_INF, k3 = h3 ? k3 : RRT_INF;
    // the nearest child first; the other three go on the stack as they come (ordering them as well costs two more
    // exchanges per visit and buys nothing measurable: 0.998 -> 0.989 ms on the 4K mesh frame without)
    order2(k0, c0, k1, c1);
    order2(k2, c2, k3, c3);
    order2(k0, c0, k2, c2);
    if (k3 < RRT_INF) stack.push(c3);
    if (k2 < RRT_INF) stack.push(c2);
    if (k1 < RRT_INF) stack.push(c1);
    if (k0 < RRT_INF)
        w.cur = c0;
    else if (stack.empty())
        w.cur = RRT_WALK_DONE;
    else
        w.cur = stack.pop();
}
// one node visit of a lane that holds an inner node: four children tested, the nearest hit one next, the others stacked
// UNIFORM_MESH: every lane of the wave walks the SAME mesh (the mesh comes from the wave-uniform record loop, not from a
// lane's own winning record)
template <bool PLAIN_STACK, bool UNIFORM_MESH>
__device__ __forceinline__ void walk_node_step(const MeshView &V, Walk &w, LaneStack<PLAIN_STACK> &stack, Counters &cnt) {
    if ((uint32_t)w.cur >= V.node_count) {
        cnt.overflow |= 16u;
        w.cur = RRT_WALK_DONE;
        return;
    }
#if RRT_WALK_STATS == 2
    RRT_WS_WAVE_COUNT(0);
#elif RRT_WALK_STATS == 8  // how often do all the lanes that visit a node visit the SAME node
    {
        const unsigned long long m_ = __ballot(true), u_ = __ballot(w.cur == __builtin_amdgcn_readfirstlane(w.cur));
        if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)m_) - 1u) {
            cnt.ws[0] += 1ull;
            cnt.ws[1] += m_ == u_ ? 1ull : 0ull;
            cnt.ws[2] += (unsigned long long)__popcll(u_);  // lanes on the first lane's node
        }
    }
#endif
    RRT_WS_TICK(w);  // (inside the loop: a stamp taken on the way out would be sunk behind the loop's exit)
    RRT_WALK_LOG(0u);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RRT_NO_UNIFORM_NODE)
    // All lanes that visit a node often visit the SAME one (55 % of the node iterations of the 4K mesh frame: a tile's rays
    // agree on the upper levels; mirror / glass teapots 44 / 41 %).  Its 64 bytes then come through the scalar cache (one
    // s_load_dwordx16) instead of 64 x 64 bytes through the vector memory pipeline, and the box tests read them as scalar
    // operands: 0.937 -> 0.921 ms on the 4K mesh frame (the vector memory pipeline was busy 81 % of the launch).
    if (UNIFORM_MESH) {
        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane(w.cur);
        if (__ballot((uint32_t)w.cur == first) == __ballot(true)) {
            typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
            // (UNIFORM_MESH: V.nodes is the same in every lane; first < node_count was checked above;
            //  readfirstlane returns int: widened through uint32_t, or bit 31 of the low half would smear over the high half)
            const uint64_t base = (uint64_t)reinterpret_cast<uintptr_t>(V.nodes);
            const uint32_t b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base), b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
            const uint64_t addr = (((uint64_t)b_hi << 32) | (uint64_t)b_lo) + ((uint64_t)first << 6);
            u32x16 v;
            // (no "memory" clobber, not volatile: the tree is read-only while the kernel runs, and a clobber would turn every
            //  later scalar load of scene data into a vector load)
            asm("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(addr));
            BvhNodeQ N;
#pragma unroll
            for (int k = 0; k < 12; k++) N.q[k] = v[k];
#pragma unroll
            for (int k = 0; k < 4; k++) N.child[k] = (int32_t)v[12 + k];
            walk_node_visit(N, w, stack, cnt);
            return;
        }
    }
#endif
    const BvhNodeQ N = *reinterpret_cast<const BvhNodeQ *>(reinterpret_cast<const char *>(V.nodes) + ((uint32_t)w.cur << 6));
    walk_node_visit(N, w, stack, cnt);
}
template <bool PLAIN_STACK, bool UNIFORM_MESH>
__device__ __forceinline__ void walk_nodes(const MeshView &V, Walk &w, LaneStack<PLAIN_STACK> &stack, Counters &cnt) {
    while (w.cur >= 0) walk_node_step<PLAIN_STACK, UNIFORM_MESH>(V, w, stack, cnt);
}
// the triangles of the leaf the lane holds (Triangle::intersect for each, the reference's test bit for bit), then the next
// node off the stack
template <bool PLAIN_STACK, bool EAGER = false>
__device__ __forceinline__ void walk_leaf(const MeshView &V, Walk &w, LaneStack<PLAIN_STACK> &stack, Counters &cnt) {
#if RRT_WALK_STATS == 2
    RRT_WS_WAVE_COUNT(2);
#endif
    const uint32_t code = (uint32_t)(-(w.cur + 1));
    const uint32_t first = code >> 4, n = code & 15u;
    if (first + n > V.tri_count) {
        cnt.overflow |= 16u;
        w.cur = RRT_WALK_DONE;  // index guard, as in walk_nodes: give up the walk
        return;
    }
    RRT_WALK_LOG(n);
    for (uint32_t k = 0; k < n; k++) {
        const TriHot T = V.tris[first + k];
        float t;
        bool ent;
        cnt.tris++;
#if RRT_WALK_STATS == 2
        RRT_WS_WAVE_COUNT(1);
#endif
        if (!tri_test<EAGER>(T, w.o, w.d, t, ent)) continue;
        if (t < 0.0f) continue;
        const uint32_t oi = tri_index_of(T);  // (= V.orig[first + k], without another load behind the test)
        if (t < w.best || ((w.found || w.ties) && t == w.best && oi < w.best_orig)) {
            w.found = true;
            w.best = t;
            w.best_orig = oi;
            w.best_slot = first + k;
            w.best_entering = ent;
        }
    }
    RRT_WS_TICK(w);
    w.cur = stack.empty() ? RRT_WALK_DONE : stack.pop();
}

// PolyMesh::intersect (polymesh_object.rs:216-242): the closest hit of EITHER facing with t >= 0,
// the lowest triangle index winning exact ties; only hits with t < tmax are reported (callers pass
// the best distance so far: a farther mesh hit could not win anyway).  Returns the BVH slot.
template <bool PLAIN_STACK, bool UNIFORM_MESH, bool EAGER = false>
__device__ __forceinline__ bool mesh_closest(const DevScene &S, const MeshInfo &M, f3 o, f3 d, float tmax, Lds &lds, Counters &cnt,
                                             float &best_t, uint32_t &best_slot, bool &best_entering) {
    const MeshView V = mesh_view(S, M);
    LaneStack<PLAIN_STACK> stack{lds, 0u};
    Walk w;
    walk_begin(w, M, o, d, tmax, false);
#if RRT_WALK_STATS == 1  // time a lane spends in the walk against the time its wave does (primary rays, or shadow rays with RRT_WS_SHADOW)
#ifdef RRT_WS_SHADOW
    const bool ws_on = !(tmax > 1e30f);
#else
    const bool ws_on = tmax > 1e30f;
#endif
    const unsigned long long ws_t0 = __builtin_amdgcn_s_memtime();
    w.ws_last = ws_t0;
#endif
    while (w.cur != RRT_WALK_DONE) {
        walk_nodes<PLAIN_STACK, UNIFORM_MESH>(V, w, stack, cnt);
        if (w.cur == RRT_WALK_DONE) break;
        walk_leaf<PLAIN_STACK, EAGER>(V, w, stack, cnt);
    }
#if RRT_WALK_STATS == 1
    if (ws_on) {
        const unsigned long long ws_t1 = __builtin_amdgcn_s_memtime();
        cnt.ws[0] += w.ws_last - ws_t0;
        cnt.ws[1] += ws_t1 - ws_t0;
        cnt.ws[2] += 1ull;
    }
#endif
    best_t = w.best;
    best_slot = w.best_slot;
    best_entering = w.best_entering;
    return w.found;
}

// ---- CSG: per-ray interval lists (csg_object.rs:39-108) ---------------------------------------------

__device__ __forceinline__ void csg_merge(uint32_t mode, Lds &lds, uint32_t sl, uint32_t nl, uint32_t sr, uint32_t nr, uint32_t so,
                                          uint32_t &no, Counters &cnt) {
    // action tables indexed by 4*L.entering + 2*R.entering + (L.d > R.d); 2 bits side/kind:
    // 0 = ADrop, 1 = AEnter, 2 = AExit, 4 = BDrop, 5 = BEnter, 6 = BExit  (packed 4 bits per state)
    //                        state:      7      6      5      4      3      2      1      0
    const uint32_t tab_union = 0x51604240u;         // [ADrop,BDrop,AExit,BDrop,ADrop,BExit,AEnter,BEnter]
    const uint32_t tab_intersection = 0x40415062u;  // [AExit,BExit,ADrop,BEnter,AEnter,BDrop,ADrop,BDrop]
    const uint32_t tab_difference = 0x41406250u;    // [ADrop,BEnter,AExit,BExit,ADrop,BDrop,AEnter,BDrop]
    uint32_t tab = mode == OP_CSG_UNION ? tab_union : (mode == OP_CSG_INTERSECTION ? tab_intersection : tab_difference);
    uint32_t li = 0, ri = 0;
    no = 0;
    // heads of both lists live in registers; only the side that advanced is re-read (index clamped: the value read
    // past the end of a list is never used)
    float tl = 0.0f, tr = 0.0f;
    uint32_t ml = 0, mr = 0;
    lds.get(sl, 0, tl, ml);
    lds.get(sr, 0, tr, mr);
    nl = min(nl, (uint32_t)RRT_LIST_CAP);  // guards: lengths are <= 6 by construction, so the loop ends within 12 steps
    nr = min(nr, (uint32_t)RRT_LIST_CAP);
    while (li < nl && ri < nr) {
        uint32_t state = ((ml & META_ENTER) ? 4u : 0u) + ((mr & META_ENTER) ? 2u : 0u) + ((tl > tr) ? 1u : 0u);
        uint32_t act = (tab >> (state * 4)) & 15u;
        bool is_b = (act & 4u) != 0;
        uint32_t kind = act & 3u;  // 0 drop, 1 enter, 2 exit
        if (kind != 0) {
            if (no >= RRT_LIST_CAP) {
                cnt.overflow = 1;  // reference: panic "HitVec overflow"
                return;
            }
            uint32_t mm = (is_b ? mr : ml) & ~META_ENTER;
            lds.put(so, no, is_b ? tr : tl, mm | (kind == 1 ? META_ENTER : 0u));
            no++;
        }
        li += is_b ? 0u : 1u;
        ri += is_b ? 1u : 0u;
        float tn;
        uint32_t mn;
        lds.get(is_b ? sr : sl, min(is_b ? ri : li, (uint32_t)(RRT_LIST_CAP - 1)), tn, mn);
        tl = is_b ? tl : tn;
        ml = is_b ? ml : mn;
        tr = is_b ? tn : tr;
        mr = is_b ? mn : mr;
    }
    if (mode != OP_CSG_INTERSECTION) {
        for (; li < nl; li++) {
            if (no >= RRT_LIST_CAP) {
                cnt.overflow = 1;
                return;
            }
            float t;
            uint32_t m;
            lds.get(sl, li, t, m);
            lds.put(so, no, t, m);
            no++;
        }
    }
    if (mode == OP_CSG_UNION) {
        for (; ri < nr; ri++) {
            if (no >= RRT_LIST_CAP) {
                cnt.overflow = 1;
                return;
            }
            float t;
            uint32_t m;
            lds.get(sr, ri, t, m);
            lds.put(so, no, t, m);
            no++;
        }
    }
}

__device__ __forceinline__ float4 rec4(const ProgRec &r) { return make_float4(r.p[0], r.p[1], r.p[2], r.p[3]); }

// Leaf of a CSG program: write the primitive's hit list into LDS slot `slot`, return its length.
// `k` = index of the record in the program (kept in the meta word so the winner can be re-evaluated).
// A mesh leaf (csg_object.rs:25-36 takes any Box<dyn Object>): PolyMesh::intersect returns at most ONE hit, the nearest of
// either facing with d >= 0 (polymesh_object.rs:216-242).  Its triangle does not fit the meta word; object_step looks it
// up again for the winner.
template <bool WITH_MESH>
__device__ __forceinline__ uint32_t leaf_list(const DevScene &S, const ProgRec &r, uint32_t k, f3 o, f3 d, Lds &lds, uint32_t slot, Counters &cnt) {
    switch (r.op & 0xFFu) {
        case OP_MESH: {
            if (!WITH_MESH || r.aux >= S.n_meshes) return 0;
            float t;
            uint32_t tri;
            bool ent;
            const MeshInfo &M = S.meshes[r.aux];
            if (mesh_closest<false, true>(S, M, o, d, RRT_INF, lds, cnt, t, tri, ent)) {  // (a leaf of the wave-uniform program)
                lds.put(slot, 0, t, mk_meta(ent, k, 0));
                return 1;
            }
            // No finite hit.  A triangle's plane still reports Hit::infinity (position Vertex::zero()) as its exit when the
            // ray runs against the normal or lies inside the half-space (plane_object.rs:60-67, 90-111); the triangle lets
            // it pass when the world origin is inside its prism (mesh_inf lists those), and PolyMesh::intersect, having
            // nothing nearer, returns it: an exit at +inf.
            const uint32_t first = S.mesh_inf[2 * r.aux], n = S.mesh_inf[2 * r.aux + 1];
            const uint32_t *slots = S.mesh_inf + 2 * S.n_meshes + first;
            for (uint32_t j = 0; j < n; j++) {
                const uint32_t ts = slots[j];
                if (ts >= M.tri_count) break;  // index guard
                const TriHot T = S.tris[M.tri_first + ts];
                const f3 nrm = mk3(T.a_nx.w, T.b_ny.w, T.c_nz.w);
                const float U = dot3(nrm, o) + T.d_pad.x, V = dot3(nrm, d);
                const bool exits_at_inf = V == 0.0f ? U >= 0.0f : !(V > 0.0f);
                if (exits_at_inf) {
                    lds.put(slot, 0, RRT_INF, mk_meta(false, k, 0));
                    return 1;
                }
            }
            return 0;
        }
        case OP_PLANE: {
            float t0, t1;
            bool e0, e1;
            if (plane_hits(rec4(r), o, d, t0, e0, t1, e1) == 0) return 0;
            lds.put(slot, 0, t0, mk_meta(e0, k, 0));
            lds.put(slot, 1, t1, mk_meta(e1, k, 1));
            return 2;
        }
        case OP_SPHERE: {
            float t0, t1;
            if (!sphere_hits(rec4(r), o, d, t0, t1)) return 0;
            lds.put(slot, 0, t0, mk_meta(true, k, 0));
            lds.put(slot, 1, t1, mk_meta(false, k, 1));
            return 2;
        }
        case OP_QUADRIC: {
            float t0, t1;
            if (!quadric_hits(r.p, o, d, t0, t1)) return 0;
            lds.put(slot, 0, t0, mk_meta(true, k, 0));
            lds.put(slot, 1, t1, mk_meta(false, k, 1));
            return 2;
        }
        case OP_CUBOID: {
            bool hf, hb;
            float tf, tb;
            uint32_t sf, sb;
            cuboid_hits(r, o, d, hf, tf, sf, hb, tb, sb);
            uint32_t n = 0;
            if (hf) {
                lds.put(slot, n, tf, mk_meta(true, k, sf));
                n++;
            }
            if (hb) {
                lds.put(slot, n, tb, mk_meta(false, k, sb));
                n++;
            }
            return n;
        }
        default: return 0;
    }
}

// State of one CSG program evaluation.  The program is wave-uniform, so the slot stack lives in scalar registers;
// only the list lengths are per lane.
struct CsgState {
    uint32_t stack_slots = 0, sp = 0;  // slot ids of the live lists, 4 bits each
    uint32_t free_mask = 0xFFu;        // free LDS slots
    uint32_t len_l = 0;                // per-lane lengths of the live lists, 4 bits each by stack position
    template <bool WITH_MESH = true>
    __device__ __forceinline__ void step(const DevScene &S, const ProgRec &r, uint32_t k, f3 o, f3 d, Lds &lds, Counters &cnt) {
        uint32_t op = r.op & 0xFFu;
        if (op < OP_CSG_UNION) {
            uint32_t slot = __builtin_ctz(free_mask);
            free_mask &= ~(1u << slot);
            uint32_t n = leaf_list<WITH_MESH>(S, r, k, o, d, lds, slot, cnt);
            stack_slots = (stack_slots & ~(15u << (sp * 4))) | (slot << (sp * 4));
            len_l = (len_l & ~(15u << (sp * 4))) | (n << (sp * 4));
            sp++;
        } else {
            uint32_t sr = (stack_slots >> ((sp - 1) * 4)) & 15u, sl = (stack_slots >> ((sp - 2) * 4)) & 15u;
            uint32_t nr = (len_l >> ((sp - 1) * 4)) & 15u, nl = (len_l >> ((sp - 2) * 4)) & 15u;
            // Wave-uniform shortcuts (rays of a wave mostly miss the same operands).  With an empty operand the
            // merge loop of csg_object.rs:52-91 does not run and only the tails are copied: the result IS the other
            // operand's list (union, difference with empty right) or empty, so the slot is handed over as it is.
            bool l_empty = !__any(nl != 0), r_empty = !__any(nr != 0);
            uint32_t so, no;
            if (r_empty && op != OP_CSG_INTERSECTION) {
                so = sl, no = nl;
                free_mask |= 1u << sr;
            } else if (l_empty && op == OP_CSG_UNION) {
                so = sr, no = nr;
                free_mask |= 1u << sl;
            } else if (l_empty || (r_empty && op == OP_CSG_INTERSECTION)) {
                so = sl, no = 0;
                free_mask |= 1u << sr;
            } else {
                so = __builtin_ctz(free_mask);
                csg_merge(op, lds, sl, nl, sr, nr, so, no, cnt);
                free_mask &= ~(1u << so);
                free_mask |= (1u << sl) | (1u << sr);
            }
            sp -= 2;
            stack_slots = (stack_slots & ~(0xFFu << (sp * 4))) | (so << (sp * 4));
            len_l = (len_l & ~(0xFFu << (sp * 4))) | (no << (sp * 4));
            sp++;
        }
    }
    // a sub-tree no ray of the wave can hit (inner OP_BOUND): its list is empty
    __device__ __forceinline__ void push_empty() {
        uint32_t slot = __builtin_ctz(free_mask);
        free_mask &= ~(1u << slot);
        stack_slots = (stack_slots & ~(15u << (sp * 4))) | (slot << (sp * 4));
        len_l = len_l & ~(15u << (sp * 4));
        sp++;
    }
    __device__ __forceinline__ uint32_t out_slot() const { return stack_slots & 15u; }
    __device__ __forceinline__ uint32_t out_len() const { return len_l & 15u; }
    __device__ __forceinline__ void reset() {
        stack_slots = sp = len_l = 0;
        free_mask = 0xFFu;
    }
};

// ---- Environment::trace (environment.rs:34-74) -----------------------------------------------------
// First entering hit with distance >= 0 over all objects; strict `<` everywhere, so the first hit in
// list order wins inside an object and the first object wins across objects.

// Scene features a kernel instance is compiled for.  The render kernel is instantiated per feature set (the flags are
// wave-uniform facts of the scene), so a scene without meshes does not carry the BVH traversal's registers and code, and
// a scene without CSG does not carry the hit-list machinery.
// FEAT_NO_SOLIDS: the scene has no sphere, quadric or cuboid record (DevScene::no_solids; planes and meshes only, as the
// 1 M-triangle frame).  Their cases are then not compiled at all: the compiler hoists the ray-dependent halves of their
// arithmetic (squares and products of the direction, the cuboid's six dot products) in front of the record loop, where a
// scene without such records paid them for every ray and every shadow ray (some 200 vector instructions per wave).
enum SceneFeatures : int { FEAT_CSG = 1, FEAT_MESH = 2, FEAT_ALL = 3, FEAT_NO_SOLIDS = 4, FEAT_TRI_EAGER = 8 };  // FEAT_TRI_EAGER: tri_test<true>

// The one entering hit a single-primitive object can produce (list position 0 of a plane, t0 of a
// sphere / quadric, the cuboid's first_hit, the mesh's closest hit when it faces the ray), with the
// reference's `distance < 0 -> skip` applied.  `mesh_bound`: only mesh hits below it are searched.
template <int FEAT = FEAT_ALL>
__device__ __forceinline__ bool single_entering_hit(const DevScene &S, const ProgRec &r, f3 o, f3 d, float mesh_bound, Lds &lds,
                                                    Counters &cnt, float &t_out, uint32_t &sub_out) {
    sub_out = 0;
    switch (r.op & 0xFFu) {
        case OP_PLANE: {
            float t0, t1;
            bool e0, e1;
            if (plane_hits(rec4(r), o, d, t0, e0, t1, e1) == 0) return false;
            if (t0 < 0.0f) return false;  // also the -inf entering hit of a back-facing plane
            t_out = t0;
            return true;
        }
        case OP_SPHERE: {
            if (FEAT & FEAT_NO_SOLIDS) return false;
            float t0, t1;
            if (!sphere_hits(rec4(r), o, d, t0, t1)) return false;
            if (t0 < 0.0f) return false;
            t_out = t0;
            return true;
        }
        case OP_QUADRIC: {
            if (FEAT & FEAT_NO_SOLIDS) return false;
            float t0, t1;
            if (!quadric_hits(r.p, o, d, t0, t1)) return false;
            if (t0 < 0.0f) return false;
            t_out = t0;
            return true;
        }
        case OP_CUBOID: {
            if (FEAT & FEAT_NO_SOLIDS) return false;
            bool hf, hb;
            float tf, tb;
            uint32_t sf, sb;
            cuboid_hits(r, o, d, hf, tf, sf, hb, tb, sb);
            if (!hf) return false;
            t_out = tf;
            sub_out = sf;
            return true;
        }
        case OP_MESH: {
            if (!(FEAT & FEAT_MESH)) return false;  // this instance serves scenes without meshes
            float t;
            uint32_t slot;
            bool ent;
            if (!mesh_closest<!(FEAT & FEAT_CSG), true, (FEAT & FEAT_TRI_EAGER) != 0>(S, S.meshes[r.aux], o, d, mesh_bound, lds, cnt, t, slot, ent)) return false;
            if (!ent) return false;  // a nearer back face hides farther front faces
            t_out = t;
            sub_out = slot;
            return true;
        }
        default: return false;
    }
}

// Environment::select_first_hit (environment.rs:34-53) on the final list of a CSG object
__device__ __forceinline__ bool csg_first_hit(Lds &lds, uint32_t slot, uint32_t n, float &t_out, uint32_t &k_out, uint32_t &sub_out) {
    bool have = false;
    float best = RRT_F32_MAX;
    for (uint32_t e = 0; e < n; e++) {
        float t;
        uint32_t m;
        lds.get(slot, e, t, m);
        if (t < 0.0f) continue;
        if (!(m & META_ENTER)) continue;
        if (t < best) {
            best = t;
            have = true;
            k_out = (m >> 8) & 0xFFFFu;
            sub_out = m & 0xFFu;
        }
    }
    t_out = best;
    return have;
}

// ---- object bounds ----------------------------------------------------------------------------------
// The reference tests every ray against every object.  A cuboid can only report a hit whose point lies inside its
// (slack-expanded) box, and a CSG list is empty whenever the list of the operand that gates it is empty
// (csg_object.rs:39-108: intersection and difference emit nothing without hits of the left / both operands), so a
// ray that misses a conservatively padded box around those cuboids cannot produce a hit: skipping the object's
// records leaves every result bit-identical.  The skip is wave-uniform (all lanes must miss).
struct RayBox {
    f3 inv;
    float omax;
};
__device__ __forceinline__ RayBox make_raybox(f3 o, f3 d) {
    RayBox b;
    // 1-ulp reciprocals: their error is inside the 2^-20 widening of bound_hit
    b.inv = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    b.omax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
    return b;
}
__device__ __forceinline__ bool bound_hit(const ProgRec &r, const RayBox &rb, f3 o) {
    float e = rb.omax * r.p[6];
    float x0 = (r.p[0] - (o.x + e)) * rb.inv.x, x1 = (r.p[3] - (o.x - e)) * rb.inv.x;
    float y0 = (r.p[1] - (o.y + e)) * rb.inv.y, y1 = (r.p[4] - (o.y - e)) * rb.inv.y;
    float z0 = (r.p[2] - (o.z + e)) * rb.inv.z, z1 = (r.p[5] - (o.z - e)) * rb.inv.z;
    // fminf/fmaxf drop NaNs (0 * inf when the origin lies on a slab plane of a parallel ray)
    float n = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    float f = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    const float k = 9.5367431640625e-7f;  // 2^-20
    float nc = n - fabsf(n) * k, fc = f + fabsf(f) * k;
    return nc <= fc && fc >= 0.0f;
}

// One step of the object loop shared by trace and shadowtrace: feeds record k, and reports the object's first
// entering hit once the object's last record has been consumed.
template <int FEAT = FEAT_ALL>
__device__ __forceinline__ bool object_step(const DevScene &S, uint32_t k, const ProgRec &r, CsgState &csg, f3 o, f3 d, float bound, Lds &lds,
                                            Counters &cnt, float &t_out, uint32_t &k_out, uint32_t &sub_out) {
    if (!(FEAT & FEAT_CSG) || !(r.op & PF_CSG)) {
        k_out = k;
        if (!single_entering_hit<FEAT>(S, r, o, d, bound, lds, cnt, t_out, sub_out)) return false;
        return t_out < RRT_F32_MAX;  // select_first_hit starts from f32::MAX with a strict `<`
    }
    if (r.op & PF_FIRST) csg.reset();
    csg.template step<(FEAT & FEAT_MESH) != 0>(S, r, k, o, d, lds, cnt);
    if (!(r.op & PF_LAST)) return false;
    if (!csg_first_hit(lds, csg.out_slot(), csg.out_len(), t_out, k_out, sub_out)) return false;
    if ((FEAT & FEAT_MESH) && (S.has_csg & 2u) && k_out < S.n_prog && (S.prog[k_out].op & 0xFFu) == OP_MESH) {
        // the winner is a mesh leaf: which triangle it was (the hit lists are done with, the walk may use the LDS again)
        const ProgRec &leaf = S.prog[k_out];
        float t;
        bool ent;
        if (leaf.aux >= S.n_meshes || !mesh_closest<false, false, (FEAT & FEAT_TRI_EAGER) != 0>(S, S.meshes[leaf.aux], o, d, RRT_INF, lds, cnt, t, sub_out, ent)) return false;  // (the lane's own winner)
    }
    return true;
}

// The ray-dependent halves of the primitives' arithmetic (squares and products of the direction, the cuboid's dot
// products, ...) do not depend on the record, so the compiler computes them once in front of the record loop -- and
// they then stay live across every BVH walk inside the loop.  An empty asm that "rewrites" the ray at the top of every
// iteration makes it a value of the iteration: nothing is hoisted, no instruction is emitted.
// Measured (RRT_NO_HOIST builds): the photon-mode kernels of scenes with meshes drop from 161 to 126 registers (four waves
// per SIMD instead of three) and the raytraced CSG kernel from 141 + scratch to 121, but the C3 render does not move
// (49.0 against 49.9 G gathered records/s) and the C2 frame gets slower (kernels 0.447 -> 0.483 ms: every quadric and cuboid
// record now recomputes what the records shared, and the hit lists in LDS cap that kernel at three waves anyway).  Off by
// default; scenes without such primitives have instances of their own (FEAT_NO_SOLIDS).
#if defined(__HIP_DEVICE_COMPILE__) && defined(RRT_NO_HOIST)
#define RRT_KEEP_IN_LOOP(o, d)                                                                                                      \
    do {                                                                                                                            \
        if (!(FEAT & FEAT_NO_SOLIDS)) asm volatile("" : "+v"((o).x), "+v"((o).y), "+v"((o).z), "+v"((d).x), "+v"((d).y), "+v"((d).z)); \
    } while (0)
#else
#define RRT_KEEP_IN_LOOP(o, d)
#endif
// Record k of the wave-uniform program, taken as a whole with ONE scalar load.  Left to itself the compiler loads the op
// word, waits, branches on it and only then asks for the parameters (two or three dependent round trips per record; a lone
// wave of a deep recursion level spends most of its time in them).  The record then lives in 16 scalar registers instead of
// being picked apart through vector ones: the raytraced CSG kernel drops from 141 registers + scratch to 122 (four waves per
// SIMD), C2 10.06 -> 10.37 Grays/s one frame at a time, 15.5 -> 17.2 with three in flight; mirror / glass teapots + 1-2 %.
// The load is scalar: k must be the same in every active lane -- it is in the record loops of trace / shadowtrace, whose
// only jumps (`k += rec.aux` behind a bound that every lane misses) are decided by __any.
// WHOLE = false: the fused instance for planes and meshes (two or three records per ray, and 48 more scalar spills at its
// 128-register bound: 0.747 -> 0.755 ms on the 4K mesh frame) keeps the compiler's loads.
template <bool WHOLE>
struct RecordLoad {  // WHOLE = false: the record where it lies
    static __device__ __forceinline__ const ProgRec &get(const ProgRec *prog, uint32_t k) { return prog[k]; }
};
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RRT_NO_REC_X16)
template <>
struct RecordLoad<true> {
    static __device__ __forceinline__ ProgRec get(const ProgRec *prog, uint32_t k) {
        typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
        const uint64_t base = (uint64_t)reinterpret_cast<uintptr_t>(prog + k);
        const uint32_t b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base), b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
        const uint64_t addr = ((uint64_t)b_hi << 32) | (uint64_t)b_lo;
        u32x16 v;
        // (no "memory" clobber, not volatile: the program is read-only while the kernel runs)
        asm("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(addr));
        ProgRec r;
        r.op = v[0], r.object = v[1], r.material = (int32_t)v[2], r.aux = v[3];
#pragma unroll
        for (int j = 0; j < 12; j++) r.p[j] = __uint_as_float(v[4 + j]);
        return r;
    }
};
#endif
template <int FEAT>
constexpr bool whole_records() { return !((FEAT & FEAT_NO_SOLIDS) && !(FEAT & FEAT_TRI_EAGER)); }
template <int FEAT = FEAT_ALL>
__device__ __forceinline__ void trace(const DevScene &S, f3 o, f3 d, Lds &lds, Counters &cnt, TraceResult &r) {
    r.object = 0xFFFFFFFFu;
    r.t = RRT_F32_MAX;
    r.instr = 0;
    r.sub = 0;
    CsgState csg;
    RayBox rb;
    if (S.has_bounds) rb = make_raybox(o, d);
    for (uint32_t k = 0; k < S.n_prog; k++) {
        RRT_KEEP_IN_LOOP(o, d);
        const ProgRec &rec = RecordLoad<whole_records<FEAT>()>::get(S.prog, k);  // (a temporary lives as long as the reference)
        if ((rec.op & 0xFFu) == OP_BOUND) {
            if ((FEAT & FEAT_CSG) && (rec.op & PF_FIRST)) csg.reset();
            if (!__any(bound_hit(rec, rb, o))) {
                k += rec.aux;
                if ((FEAT & FEAT_CSG) && (rec.op & PF_CSG)) csg.push_empty();  // bound of a sub-tree inside a CSG program
            }
            continue;
        }
        float t;
        uint32_t kk, sub;
        if (!object_step<FEAT>(S, k, rec, csg, o, d, r.t, lds, cnt, t, kk, sub)) continue;
        if (t < r.t) {
            r.t = t;
            r.object = rec.object;
            r.instr = kk;
            r.sub = sub;
        }
    }
}

// Scene::shadowtrace (scene.rs:65-79)
template <int FEAT = FEAT_ALL>
__device__ __forceinline__ bool shadowtrace(const DevScene &S, f3 o, f3 d, float limit, Lds &lds, Counters &cnt) {
    cnt.shadow++;
    CsgState csg;
    RayBox rb;
    if (S.has_bounds) rb = make_raybox(o, d);
    for (uint32_t k = 0; k < S.n_prog; k++) {
        RRT_KEEP_IN_LOOP(o, d);
        const ProgRec &rec = RecordLoad<whole_records<FEAT>()>::get(S.prog, k);  // (a temporary lives as long as the reference)
        if ((rec.op & 0xFFu) == OP_BOUND) {
            if ((FEAT & FEAT_CSG) && (rec.op & PF_FIRST)) csg.reset();
            if (!__any(bound_hit(rec, rb, o))) {
                k += rec.aux;
                if ((FEAT & FEAT_CSG) && (rec.op & PF_CSG)) csg.push_empty();  // bound of a sub-tree inside a CSG program
            }
            continue;
        }
        float t;
        uint32_t kk, sub;
        // for meshes only hits below `limit` matter: the closest hit is either below it (and found)
        // or cannot occlude
        if (!object_step<FEAT>(S, k, rec, csg, o, d, limit, lds, cnt, t, kk, sub)) continue;
        if (t > 0.0000001f && t < limit) return true;
    }
    return false;
}

// ---- lazy hit attributes -------------------------------------------------------------------------

struct HitAttr {
    f3 position, normal;
    float u, v;
    bool has_tex;
    int32_t material;
};

// Image::get_uv / get_xy (texture.rs:105-119); texel = sample / maxval (texture.rs:89-93)
__device__ __forceinline__ f3 tex_fetch(const DevScene &S, int32_t id, float u, float v) {
    const DevTexture T = S.textures[id];
    uint32_t x = sat_u32(floorf(rem_euclid_f(-u, 1.0f) * (float)(T.width - 1)));
    float max_y = (float)T.width / (float)T.height;
    uint32_t y = sat_u32(floorf(rem_euclid_f(-v, 1.0f) * max_y * (float)(T.height - 1)));
    uint64_t idx = T.offset + (uint64_t)(y % T.height) * T.width + (x % T.width);
    float mv = (float)T.maxval;
    if (T.wide) {
        ushort4 c = S.texels16[idx];
        return mk3((float)c.x / mv, (float)c.y / mv, (float)c.z / mv);
    }
    uchar4 c = S.texels8[idx];
    return mk3((float)c.x / mv, (float)c.y / mv, (float)c.z / mv);
}

// Material::normal (texture.rs:173-184 through compound_material.rs:104-116)
__device__ __forceinline__ bool material_normal(const DevScene &S, int32_t material, float u, float v, f3 &out) {
    if ((uint32_t)material >= S.n_materials) return false;
    const DevMaterial &M = S.materials[material];
    if (M.nm_part < 0) return false;
    const MatPart &P = M.parts[M.nm_part];
    f3 c = tex_fetch(S, P.normal_tex, u * P.scale, v * P.scale);
    out = norm3(c * 2.0f - mk3(1.0f, 1.0f, 1.0f));
    return true;
}

// finite plane hit attributes (plane_object.rs:74-112); `list_pos` = position in the plane's hit list
__device__ __forceinline__ void plane_attr(const DevScene &S, uint32_t plane, uint32_t list_pos, f3 o, f3 d, float t, HitAttr &h) {
    const PlaneCold &C = S.plane_cold[plane];
    f3 n = C.normal;
    h.material = C.material;
    float V = dot3(n, d);
    bool finite_exit = (V > 0.0f) && list_pos == 1;
    bool finite_enter = !(V == 0.0f) && !(V > 0.0f) && list_pos == 0;
    if (finite_exit) {
        h.position = o + d * t;
        h.normal = -n;
        h.has_tex = false;
        h.u = h.v = 0.0f;
    } else if (finite_enter) {
        h.position = o + d * t;
        f3 fc = h.position - C.centre;
        h.u = dot3(fc, C.u_unit);
        h.v = dot3(fc, C.v_unit);
        h.has_tex = true;
        h.normal = n;
        f3 nm;
        if (material_normal(S, C.material, h.u, h.v, nm)) h.normal = norm3(tbn_apply(C.tan_t, C.tan_b, C.tan_n, nm));
    } else {  // Hit::infinity, core/hit.rs:37-52
        h.position = mk3(0.0f, 0.0f, 0.0f);
        h.normal = mk3(0.0f, 0.0f, 0.0f);
        h.has_tex = false;
        h.u = h.v = 0.0f;
    }
}

template <int FEAT = FEAT_ALL>
__device__ __forceinline__ void eval_hit(const DevScene &S, f3 o, f3 d, uint32_t instr_index, uint32_t sub, float t, HitAttr &h) {
    // Index guards: a winner record that is out of range means a broken invariant upstream.  Refuse it (the caller
    // sees material -1 and raises the internal-error bit) rather than fault the GPU on a wild address.
    h.position = h.normal = mk3(0.0f, 0.0f, 0.0f);
    h.u = h.v = 0.0f;
    h.has_tex = false;
    h.material = -1;
    if (instr_index >= S.n_prog) return;
    const ProgRec &in = S.prog[instr_index];
    const float PI_F = 3.14159265358979323846f;
    switch (in.op & 0xFFu) {
        case OP_PLANE:
            if (in.aux >= S.n_plane_cold) return;
            plane_attr(S, in.aux, sub, o, d, t, h);
            break;
        case OP_CUBOID:
            if ((FEAT & FEAT_NO_SOLIDS) || sub >= 12u || in.aux + 6u > S.n_plane_cold) return;
            plane_attr(S, in.aux + (sub >> 1), sub & 1u, o, d, t, h);
            break;
        case OP_SPHERE: {  // sphere_object.rs:55-88
            if (FEAT & FEAT_NO_SOLIDS) return;
            f3 c = mk3(in.p[0], in.p[1], in.p[2]);
            h.material = in.material;
            h.position = o + d * t;
            f3 n = norm3(h.position - c);
            if (dot3(n, d) > 0.0f) n = -n;
            // libm calls go through f64 and are rounded to f32 once (= a correctly rounded atan2f / acosf up to
            // double-rounding ties): x86 glibc and the device's ocml then agree, which their f32 versions do not
            float theta = (float)atan2((double)n.x, (double)n.z) + PI_F;
            float phi = (float)acos((double)-n.y);
            h.u = theta / (2.0f * PI_F);
            h.v = (PI_F - phi) / PI_F;
            h.has_tex = true;
            f3 nm;
            if (material_normal(S, h.material, h.u, h.v, nm)) {
                f3 tangent = norm3(cross3(mk3(1.0f, 0.0f, 0.0f), h.position - c));
                nm = to_tangent_space3(nm, tangent, n);
                n = norm3(nm);
            }
            h.normal = n;
            break;
        }
        case OP_QUADRIC: {  // quadratic_object.rs:91-118
            if (FEAT & FEAT_NO_SOLIDS) return;
            const float *q = in.p;  // a b c d e f g h i j
            h.material = in.material;
            f3 p = o + d * t;
            h.position = p;
            f3 n = norm3(mk3(q[0] * p.x + q[1] * p.y + q[2] * p.z + q[3], q[1] * p.x + q[4] * p.y + q[5] * p.z + q[6],
                             q[2] * p.x + q[5] * p.y + q[7] * p.z + q[8]));
            if (dot3(n, d) > 0.0f) n = -n;
            h.normal = n;
            h.has_tex = false;
            h.u = h.v = 0.0f;
            break;
        }
        case OP_MESH: {  // triangle_object.rs:85-160 for the winning triangle only
            if (!(FEAT & FEAT_MESH) || in.aux >= S.n_meshes) return;
            const MeshInfo &M = S.meshes[in.aux];
            if (sub >= M.tri_count) return;
            const TriHot T = S.tris[M.tri_first + sub];
            h.material = M.material;
            f3 n = mk3(T.a_nx.w, T.b_ny.w, T.c_nz.w);
            f3 a = mk3(T.a_nx.x, T.a_nx.y, T.a_nx.z), b = mk3(T.b_ny.x, T.b_ny.y, T.b_ny.z), c = mk3(T.c_nz.x, T.c_nz.y, T.c_nz.z);
            f3 ab = b - a, bc = c - b, ca = a - c;
            float V = dot3(n, d);
            h.position = o + d * t;
            if (V > 0.0f) {
                h.normal = -n;
                h.has_tex = false;
                h.u = h.v = 0.0f;
            } else {  // Plane::new_raw(a, up = ab, normal): plane_object.rs:92-105
                f3 fc = h.position - a;
                f3 v_unit = norm3(ab);
                f3 u_unit = norm3(cross3(v_unit, n));
                h.u = dot3(fc, u_unit);
                h.v = dot3(fc, v_unit);
                h.has_tex = true;
                h.normal = n;
                f3 nm;
                if (material_normal(S, M.material, h.u, h.v, nm)) {
                    f3 right = norm3(cross3(n, ab));
                    h.normal = to_tangent_space3(nm, right, n);
                }
            }
            if (M.smooth) {  // smoothen_hit, triangle_object.rs:98-125 (not flipped toward the ray)
                f3 ai = h.position - a, bi = h.position - b, ci = h.position - c;
                float gamma = len3(cross3(ai, ab));
                float alpha = len3(cross3(bi, bc));
                float beta = len3(cross3(ci, ca));
                const float *vn = S.tri_normals + ((uint64_t)M.nrm_first + S.tri_index[M.tri_first + sub]) * 9;
                f3 an = mk3(vn[0], vn[1], vn[2]), bn = mk3(vn[3], vn[4], vn[5]), cn = mk3(vn[6], vn[7], vn[8]);
                h.normal = norm3(an * alpha + bn * beta + cn * gamma);
            }
            break;
        }
        default: break;
    }
}

}  // namespace rrt
