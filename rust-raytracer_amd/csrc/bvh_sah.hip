// bvh_sah.hip — the host's binned-SAH BVH2 build (bvh_build.cpp) on the device, level by level.
// Selected with rrt_params.bvh_builder = RRT_BVH_DEVICE_SAH.  Same bins (16 per axis over the node's centroid bounds), same
// cost expression, same tie rule, same leaf size, evaluated with the same f32 operations: the tree is the one build_bvh
// makes (node numbering aside, which quantise_bvh does not look at), so frames cost what they cost with the host's tree,
// and the build runs where the triangles already are.
//
// One level of the tree per round, all of its nodes at once:
//   bin      every triangle of a node with more than kLeafMax triangles grows the bin of its centroid on each axis
//            (atomics on order-preserving words; runs of 64 or more positions of one node are collected in LDS first)
//   decide   one thread per node: sweep the 3 x 15 candidate planes as the host does, create the two children
//   flag     which side each triangle goes to; exclusive scan of the flags (rocPRIM)
//   scatter  stable partition of every node's range at once (position = range start + rank on its side), the children's
//            centroid bounds collected on the way
// The host reads one word per level (how many nodes the level created) and stops when it is zero.  Nodes of at most 64
// triangles leave the rounds early: k_sah_small finishes each such subtree with one wave, bins and partition in LDS.
//
// Memory is spent freely (MI355X: 288 GB): 4n node records of 72 bytes, a 1344-byte bin block for each of n / 16 level nodes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>

#include <rocprim/rocprim.hpp>

#include "bvh_lbvh.h"
#include "bvh_quant.h"
#include "runtime.h"

namespace rrt {

namespace {

using lbvh::ordered;
using lbvh::unordered;

constexpr int kBlock = 256;
constexpr uint32_t kLeafMax = 4;  // bvh_build.cpp
constexpr int kBins = 16;         // bvh_build.cpp
constexpr uint32_t kChunk = 512;  // positions per wave in the binning and scatter kernels
constexpr uint32_t kBinWords = 3 * kBins * 7;  // per node: [axis][bin] lo xyz, hi xyz (ordered words), count
inline dim3 grid_for(uint64_t n) { return dim3((uint32_t)((n + kBlock - 1) / kBlock)); }

struct Node {
    uint32_t lo, hi;      // positions [lo, hi) of `order`
    int32_t left;         // index of the left child (right = left + 1), -1: leaf, -2: not decided yet
    uint32_t n_left;      // triangles that go left
    int32_t axis, split;  // axis < 0: no plane found (all centroids coincide), split by position at lo + n_left
    uint32_t box[6];      // ordered words: lo xyz, hi xyz of the triangles' boxes
    uint32_t cb[6];       // same for their centroids
    uint32_t depth;       // root = 1 (bvh_build.cpp's Work::depth)
    uint32_t pad;
};
constexpr uint32_t kSmall = 64;  // a node of at most this many triangles leaves the level rounds: k_sah_small finishes its subtree, one wave, in LDS
constexpr int32_t kPendingSmall = -3;

// empty-aware decode: 0xFFFFFFFF / 0 are the "nothing seen" values of a lower / upper word
__device__ __forceinline__ float dec_lo(uint32_t u) { return u == 0xFFFFFFFFu ? __builtin_huge_valf() : unordered(u); }
__device__ __forceinline__ float dec_hi(uint32_t u) { return u == 0u ? -__builtin_huge_valf() : unordered(u); }

// 0. triangle boxes and centroids (bvh_build.cpp:75-81), root box and centroid bounds; flags non-finite input
__global__ __launch_bounds__(kBlock) void k_sah_prepare(const float *positions, uint32_t n, float *tb, float *cen, uint32_t *order, uint32_t *node_of,
                                                        Node *nodes, uint32_t *err) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t c0 = (blockIdx.x * (kBlock / 64) + wave) * kChunk;  // one wave per kChunk triangles: 12 atomics per wave on the root
    const uint32_t c1 = min(c0 + kChunk, n);
    uint32_t w[12];
    for (int k = 0; k < 3; k++) w[k] = w[6 + k] = 0xFFFFFFFFu, w[3 + k] = w[9 + k] = 0u;
    bool bad = false;
    for (uint32_t i = c0 + lane; i < c1; i += 64) {
        const float *p = positions + (size_t)i * 9;
        for (int a = 0; a < 3; a++) {
            float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
            for (int c = 0; c < 3; c++) {
                lo = fminf(lo, p[c * 3 + a]);
                hi = fmaxf(hi, p[c * 3 + a]);
                if (!(fabsf(p[c * 3 + a]) < __builtin_huge_valf())) bad = true;
            }
            const float ce = 0.5f * (lo + hi);
            if (!(fabsf(ce) < __builtin_huge_valf())) bad = true;
            tb[(size_t)i * 6 + a] = lo;
            tb[(size_t)i * 6 + 3 + a] = hi;
            cen[(size_t)i * 3 + a] = ce;
            w[a] = min(w[a], ordered(lo)), w[3 + a] = max(w[3 + a], ordered(hi));
            w[6 + a] = min(w[6 + a], ordered(ce)), w[9 + a] = max(w[9 + a], ordered(ce));
        }
        order[i] = i;
        node_of[i] = 0;
    }
    if (bad) atomicOr(err, 1u);
    if (c0 >= n) return;  // whole waves only: the butterflies below have every lane
    for (int k = 0; k < 12; k++) {
        const bool is_lo = (k % 6) < 3;
        uint32_t v = w[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
            v = is_lo ? min(v, o) : max(v, o);
        }
        if (lane == 0) {
            uint32_t *dst = k < 6 ? &nodes[0].box[k] : &nodes[0].cb[k - 6];
            if (is_lo)
                atomicMin(dst, v);
            else
                atomicMax(dst, v);
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_sah_clear_bins(uint32_t *bins, uint64_t words) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < words) bins[i] = (i % 7) < 3 ? 0xFFFFFFFFu : 0u;  // lower words, then upper words and the count
}

__device__ __forceinline__ int bin_of(float c, float lo, float scale) {
    int b = (int)((c - lo) * scale);  // bvh_build.cpp:116
    return min(max(b, 0), kBins - 1);
}

// 1. binning.  One wave per kChunk positions.  The part of a node that lies in the chunk is a run of positions; a run of
// kLongRun or more is collected in the wave's LDS bins first and only the bins it touched go to memory (a top-level node
// would otherwise put 21 atomics per triangle on the same eleven cache lines); shorter runs use the atomics directly.
constexpr uint32_t kLongRun = 64;
__device__ __forceinline__ bool splits(const Node &N) { return N.left == -2 && N.hi - N.lo > kSmall; }
__device__ __forceinline__ void bin_triangle(uint32_t *bins_of_node, const Node &N, const float *tb, const float *cen, uint32_t t) {
    uint32_t bw[6];
    for (int k = 0; k < 6; k++) bw[k] = ordered(tb[(size_t)t * 6 + k]);
    for (int a = 0; a < 3; a++) {
        const float clo = dec_lo(N.cb[a]);
        const float ext = dec_hi(N.cb[3 + a]) - clo;
        if (!(ext > 0.0f)) continue;
        uint32_t *d = bins_of_node + (a * kBins + bin_of(cen[(size_t)t * 3 + a], clo, (float)kBins / ext)) * 7;
        for (int k = 0; k < 3; k++) atomicMin(d + k, bw[k]);
        for (int k = 3; k < 6; k++) atomicMax(d + k, bw[k]);
        atomicAdd(d + 6, 1u);
    }
}
__global__ __launch_bounds__(kBlock) void k_sah_bin(const uint32_t *order, const uint32_t *node_of, const Node *nodes, const float *tb, const float *cen,
                                                    uint32_t n, uint32_t level_first, uint32_t *bins) {
    __shared__ uint32_t s_bins[kBlock / 64][kBinWords];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t c0 = (blockIdx.x * (kBlock / 64) + wave) * kChunk;
    if (c0 >= n) return;  // no block-wide barrier below: waves are on their own
    const uint32_t c1 = min(c0 + kChunk, n);
    // long runs: any run of kLongRun positions holds one of the sampled positions
    bool long_here = false;
    for (uint32_t i = c0 + lane * (kLongRun / 2); i < c1; i += 64 * (kLongRun / 2)) {
        const Node N = nodes[node_of[i]];
        if (splits(N) && min(N.hi, c1) - max(N.lo, c0) >= kLongRun) long_here = true;
    }
    if (__any(long_here)) {
        uint32_t *sb = s_bins[wave];
        for (uint32_t pos = c0; pos < c1;) {  // wave-uniform walk over the runs
            const uint32_t node = node_of[pos];
            const Node N = nodes[node];
            const uint32_t end = min(N.hi, c1);
            if (end <= pos) break;  // cannot be (pos lies in N's range); never loop for ever
            if (splits(N) && end - max(N.lo, c0) >= kLongRun) {
                for (uint32_t k = lane; k < kBinWords; k += 64) sb[k] = (k % 7) < 3 ? 0xFFFFFFFFu : 0u;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = pos + lane; i < end; i += 64) bin_triangle(sb, N, tb, cen, order[i]);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                uint32_t *g = bins + (size_t)(node - level_first) * kBinWords;
                for (uint32_t k = lane; k < kBinWords; k += 64) {
                    const uint32_t f = k % 7, v = sb[k];
                    if (f < 3) {
                        if (v != 0xFFFFFFFFu) atomicMin(g + k, v);
                    } else if (v != 0u) {
                        if (f < 6)
                            atomicMax(g + k, v);
                        else
                            atomicAdd(g + k, v);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            pos = end;
        }
    }
    for (uint32_t i = c0 + lane; i < c1; i += 64) {  // the short runs
        const uint32_t node = node_of[i];
        const Node N = nodes[node];
        if (!splits(N) || min(N.hi, c1) - max(N.lo, c0) >= kLongRun) continue;
        bin_triangle(bins + (size_t)(node - level_first) * kBinWords, N, tb, cen, order[i]);
    }
}

struct FBox {
    float lo[3], hi[3];
    __device__ void reset() {
        for (int a = 0; a < 3; a++) lo[a] = __builtin_huge_valf(), hi[a] = -__builtin_huge_valf();
    }
    __device__ void grow(const FBox &b) {
        for (int a = 0; a < 3; a++) lo[a] = fminf(lo[a], b.lo[a]), hi[a] = fmaxf(hi[a], b.hi[a]);
    }
    __device__ float half_area() const {  // bvh_build.cpp:42-46
        const float e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
        if (e0 < 0 || e1 < 0 || e2 < 0) return 0.0f;
        return e0 * e1 + e1 * e2 + e2 * e0;
    }
};
__device__ __forceinline__ FBox load_bin(const uint32_t *d) {
    FBox b;
    for (int k = 0; k < 3; k++) b.lo[k] = dec_lo(d[k]), b.hi[k] = dec_hi(d[3 + k]);
    return b;
}

// 2. the split of every node of the level (bvh_build.cpp:94-170), and its two children
__global__ __launch_bounds__(kBlock) void k_sah_decide(Node *nodes, uint32_t level_first, uint32_t level_count, const uint32_t *bins, uint32_t next_first,
                                                       uint32_t capacity, uint32_t *created, uint32_t *err) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= level_count) return;
    Node &N = nodes[level_first + k];
    const uint32_t cnt = N.hi - N.lo;
    if (cnt <= kLeafMax) {
        N.left = -1;
        return;
    }
    if (cnt <= kSmall) {
        N.left = kPendingSmall;
        return;
    }
    const uint32_t *B = bins + (size_t)k * kBinWords;
    int best_axis = -1, best_split = -1;
    float best_cost = __builtin_huge_valf();
    for (int a = 0; a < 3; a++) {
        const float ext = dec_hi(N.cb[3 + a]) - dec_lo(N.cb[a]);
        if (!(ext > 0.0f)) continue;
        const uint32_t *A = B + a * kBins * 7;
        float right_area[kBins];
        uint32_t right_cnt[kBins];
        FBox acc;
        acc.reset();
        uint32_t c = 0;
        for (int b = kBins - 1; b > 0; b--) {
            acc.grow(load_bin(A + b * 7));
            c += A[b * 7 + 6];
            right_area[b] = acc.half_area();
            right_cnt[b] = c;
        }
        acc.reset();
        c = 0;
        for (int b = 0; b < kBins - 1; b++) {
            acc.grow(load_bin(A + b * 7));
            c += A[b * 7 + 6];
            if (c == 0 || right_cnt[b + 1] == 0) continue;
            const float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
            if (cost < best_cost) {
                best_cost = cost;
                best_axis = a;
                best_split = b;
            }
        }
    }
    const uint32_t pair = atomicAdd(created, 2u);
    if (next_first + pair + 2u > capacity) {  // cannot happen (a binary tree over n leaves has fewer than 2n nodes); refuse rather than overrun
        atomicOr(err, 2u);
        N.left = -1;
        return;
    }
    Node L, R;
    for (int q = 0; q < 3; q++) L.box[q] = R.box[q] = L.cb[q] = R.cb[q] = 0xFFFFFFFFu, L.box[3 + q] = R.box[3 + q] = L.cb[3 + q] = R.cb[3 + q] = 0u;
    uint32_t n_left;
    if (best_axis < 0) {
        n_left = cnt / 2;  // bvh_build.cpp:148; the children's boxes are collected by the scatter
    } else {
        const uint32_t *A = B + best_axis * kBins * 7;
        n_left = 0;
        for (int b = 0; b < kBins; b++) {
            Node &D = b <= best_split ? L : R;
            for (int q = 0; q < 3; q++) D.box[q] = min(D.box[q], A[b * 7 + q]), D.box[3 + q] = max(D.box[3 + q], A[b * 7 + 3 + q]);
            if (b <= best_split) n_left += A[b * 7 + 6];
        }
        if (n_left == 0 || n_left >= cnt) {  // the bins and the range disagree: internal error
            atomicOr(err, 4u);
            n_left = cnt / 2;
        }
    }
    L.lo = N.lo, L.hi = R.lo = N.lo + n_left, R.hi = N.hi;
    L.left = R.left = -2;
    L.depth = R.depth = N.depth + 1u;
    L.pad = R.pad = 0u;
    L.n_left = R.n_left = 0;
    L.axis = R.axis = -1, L.split = R.split = 0;
    N.left = (int32_t)(next_first + pair);
    N.n_left = n_left;
    N.axis = best_axis;
    N.split = best_split;
    nodes[next_first + pair] = L;
    nodes[next_first + pair + 1] = R;
}

// 3. side of every triangle of a node that splits: 1 = left
__global__ __launch_bounds__(kBlock) void k_sah_flag(const uint32_t *order, const uint32_t *node_of, const Node *nodes, const float *cen, uint32_t n,
                                                     uint32_t *flag) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Node N = nodes[node_of[i]];
    uint32_t f = 0;
    if (N.left >= 0) {
        if (N.axis < 0) {
            f = i < N.lo + N.n_left ? 1u : 0u;
        } else {
            const float clo = dec_lo(N.cb[N.axis]);
            const float ext = dec_hi(N.cb[3 + N.axis]) - clo;
            f = bin_of(cen[(size_t)order[i] * 3 + N.axis], clo, (float)kBins / ext) <= N.split ? 1u : 0u;  // bvh_build.cpp:153-157
        }
    }
    flag[i] = f;
}

// 4. stable partition of all ranges; the children's centroid bounds (and, for position splits, their boxes).  Runs as in
// k_sah_bin: a long run reduces its centroids over the wave (12 atomics per run), short runs use the atomics directly.
struct ScatterArgs {
    const uint32_t *order, *node_of;
    Node *nodes;
    const float *tb, *cen;
    const uint32_t *flag, *rank;
    uint32_t *order_out, *node_out;
};
// moves position i of node N (which splits); returns the side (1 = left) and the centroid's ordered words
__device__ __forceinline__ uint32_t scatter_one(const ScatterArgs &A, const Node &N, uint32_t i, uint32_t cw[3]) {
    const uint32_t t = A.order[i];
    const uint32_t f = A.flag[i];
    const uint32_t lb = A.rank[i] - A.rank[N.lo];  // left-going triangles of this node in front of i
    const uint32_t pos = f ? N.lo + lb : N.lo + N.n_left + (i - N.lo - lb);
    const int32_t child = N.left + (f ? 0 : 1);
    A.order_out[pos] = t;
    A.node_out[pos] = (uint32_t)child;
    for (int a = 0; a < 3; a++) cw[a] = ordered(A.cen[(size_t)t * 3 + a]);
    if (N.axis < 0) {  // split by position: nobody has the children's boxes yet
        Node &C = A.nodes[child];
        for (int a = 0; a < 3; a++) {
            atomicMin(&C.box[a], ordered(A.tb[(size_t)t * 6 + a]));
            atomicMax(&C.box[3 + a], ordered(A.tb[(size_t)t * 6 + 3 + a]));
        }
    }
    return f;
}
__global__ __launch_bounds__(kBlock) void k_sah_scatter(ScatterArgs A, uint32_t n) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t c0 = (blockIdx.x * (kBlock / 64) + wave) * kChunk;
    if (c0 >= n) return;
    const uint32_t c1 = min(c0 + kChunk, n);
    bool long_here = false;
    for (uint32_t i = c0 + lane * (kLongRun / 2); i < c1; i += 64 * (kLongRun / 2)) {
        const Node N = A.nodes[A.node_of[i]];
        if (N.left >= 0 && min(N.hi, c1) - max(N.lo, c0) >= kLongRun) long_here = true;
    }
    if (__any(long_here)) {
        for (uint32_t pos = c0; pos < c1;) {  // wave-uniform walk over the runs
            const Node N = A.nodes[A.node_of[pos]];
            const uint32_t end = min(N.hi, c1);
            if (end <= pos) break;
            if (N.left >= 0 && end - max(N.lo, c0) >= kLongRun) {
                uint32_t acc[12];  // [side][lo xyz, hi xyz]
                for (int k = 0; k < 12; k++) acc[k] = (k % 6) < 3 ? 0xFFFFFFFFu : 0u;
                for (uint32_t i = pos + lane; i < end; i += 64) {
                    uint32_t cw[3];
                    uint32_t *d = acc + (scatter_one(A, N, i, cw) ? 0 : 6);
                    for (int a = 0; a < 3; a++) d[a] = min(d[a], cw[a]), d[3 + a] = max(d[3 + a], cw[a]);
                }
                for (int k = 0; k < 12; k++) {
                    const bool is_lo = (k % 6) < 3;
                    uint32_t v = acc[k];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) {
                        const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
                        v = is_lo ? min(v, o) : max(v, o);
                    }
                    if (lane == 0) {
                        Node &C = A.nodes[N.left + k / 6];
                        if (is_lo) {
                            if (v != 0xFFFFFFFFu) atomicMin(&C.cb[k % 6], v);
                        } else if (v != 0u) {
                            atomicMax(&C.cb[k % 6], v);
                        }
                    }
                }
            }
            pos = end;
        }
    }
    for (uint32_t i = c0 + lane; i < c1; i += 64) {  // leaves and the short runs
        const uint32_t node = A.node_of[i];
        const Node N = A.nodes[node];
        if (N.left < 0) {  // a leaf keeps its place
            A.order_out[i] = A.order[i];
            A.node_out[i] = node;
            continue;
        }
        if (min(N.hi, c1) - max(N.lo, c0) >= kLongRun) continue;
        uint32_t cw[3];
        Node &C = A.nodes[N.left + (scatter_one(A, N, i, cw) ? 0 : 1)];
        for (int a = 0; a < 3; a++) {
            atomicMin(&C.cb[a], cw[a]);
            atomicMax(&C.cb[3 + a], cw[a]);
        }
    }
}

// 4b. The subtree of a node of at most kSmall triangles, by one wave: lane l holds the triangle at position lo + l of the
// node at hand (box, centroid, index in registers), bins live in LDS, the candidate planes are swept by 45 lanes at once,
// and a split is a stable partition of the lanes (ballots for the ranks, one trip through LDS for the registers).  Same
// bins, costs and tie rule as k_sah_decide and bvh_build.cpp.  Replaces the deepest ten or so level rounds, which paid
// 21 global atomics per triangle and round.
struct SmallLds {
    uint32_t bins[kBinWords];
    uint32_t move[10][64];
    uint32_t stack[kSmall][4];  // node, first lane, end lane, depth
};
__device__ __forceinline__ uint32_t wave_min_u(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off, 64));
    return v;
}
#define SAH_LDS_SYNC()                                      \
    do {                                                    \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                    \
    } while (0)
__global__ __launch_bounds__(kBlock) void k_sah_small(Node *nodes, uint32_t total, uint32_t capacity, uint32_t *order, const float *tb, const float *cen,
                                                      uint32_t *alloc, uint32_t *max_depth, uint32_t *err) {
    __shared__ SmallLds s_lds[kBlock / 64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t root = blockIdx.x * (kBlock / 64) + wave;
    if (root >= total) return;  // whole waves leave; no block-wide barrier anywhere below
    if (nodes[root].left != kPendingSmall) return;
    SmallLds &S = s_lds[wave];
    const uint32_t lo = nodes[root].lo, cnt = nodes[root].hi - lo;
    if (cnt > kSmall || cnt <= kLeafMax) {  // cannot be: k_sah_decide marks exactly these
        if (lane == 0) atomicOr(err, 64u);
        return;
    }
    // this lane's triangle (lanes >= cnt hold nothing and never take part)
    uint32_t t = 0, bw[6] = {0, 0, 0, 0, 0, 0};
    float ce[3] = {0.0f, 0.0f, 0.0f};
    if (lane < cnt) {
        t = order[lo + lane];
        for (int k = 0; k < 6; k++) bw[k] = ordered(tb[(size_t)t * 6 + k]);
        for (int a = 0; a < 3; a++) ce[a] = cen[(size_t)t * 3 + a];
    }
    // the subtree's node records: one allocation of the most it can need (a binary tree over cnt leaves: 2 cnt - 2 nodes below
    // its root), handed out from a wave-local count; what is left over is marked unused at the end.  (One atomic per wave:
    // an atomic per split, 400 k of them on one word, serialise in the L2 and took 3 of this kernel's 3.7 ms.)
    const uint32_t block_nodes = 2u * cnt - 2u;
    uint32_t first_node = 0, used = 0;
    if (lane == 0) first_node = atomicAdd(alloc, block_nodes);
    first_node = total + (uint32_t)__shfl((int)first_node, 0, 64);
    if (first_node + block_nodes > capacity) {
        if (lane == 0) atomicOr(err, 2u);
        return;
    }
    uint32_t sp = 0, deepest = 0;
    if (lane == 0) S.stack[0][0] = root, S.stack[0][1] = 0u, S.stack[0][2] = cnt, S.stack[0][3] = nodes[root].depth;
    sp = 1;
    SAH_LDS_SYNC();
    while (sp) {  // wave-uniform
        sp--;
        const uint32_t node = S.stack[sp][0], a = S.stack[sp][1], b = S.stack[sp][2], dep = S.stack[sp][3];
        SAH_LDS_SYNC();
        const uint32_t c = b - a;
        deepest = max(deepest, dep);
        if (c <= kLeafMax) {
            if (lane == 0) nodes[node].left = -1;
            continue;
        }
        const bool in = lane >= a && lane < b;
        // centroid bounds of the node (bvh_build.cpp:102-104)
        float clo[3], ext[3];
        for (int x = 0; x < 3; x++) {
            const uint32_t o = ordered(ce[x]);
            const float l = unordered(wave_min_u(in ? o : 0xFFFFFFFFu)), h = unordered(wave_max_u(in ? o : 0u));
            clo[x] = l;
            ext[x] = h - l;
        }
        for (uint32_t k = lane; k < kBinWords; k += 64) S.bins[k] = (k % 7) < 3 ? 0xFFFFFFFFu : 0u;
        SAH_LDS_SYNC();
        int my_bin[3] = {0, 0, 0};
        if (in) {
            for (int x = 0; x < 3; x++) {
                if (!(ext[x] > 0.0f)) continue;
                my_bin[x] = bin_of(ce[x], clo[x], (float)kBins / ext[x]);
                uint32_t *d = S.bins + (x * kBins + my_bin[x]) * 7;
                for (int k = 0; k < 3; k++) atomicMin(d + k, bw[k]);
                for (int k = 3; k < 6; k++) atomicMax(d + k, bw[k]);
                atomicAdd(d + 6, 1u);
            }
        }
        SAH_LDS_SYNC();
        // candidate plane (axis x, after bin s) on lane x * 15 + s: the host's sweep, both sides accumulated afresh
        float cost = __builtin_huge_valf();
        if (lane < 45) {
            const int x = (int)lane / 15, sb = (int)lane % 15;
            if (ext[x] > 0.0f) {
                const uint32_t *A = S.bins + x * kBins * 7;
                FBox L, R;
                L.reset();
                R.reset();
                uint32_t cl = 0, cr = 0;
                for (int q = 0; q < kBins; q++) {
                    const FBox B = load_bin(A + q * 7);
                    if (q <= sb)
                        L.grow(B), cl += A[q * 7 + 6];
                    else
                        R.grow(B), cr += A[q * 7 + 6];
                }
                if (cl != 0 && cr != 0) {
                    const float v = L.half_area() * (float)cl + R.half_area() * (float)cr;
                    if (v < __builtin_huge_valf()) cost = v;  // as `cost < best_cost` from +inf: NaN and +inf never win
                }
            }
        }
        // the first strictly lowest cost in (axis, bin) order = the lowest lane among the minima
        float best = cost;
        uint32_t who = lane;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const uint32_t ow = (uint32_t)__shfl_xor((int)who, off, 64);
            if (ob < best || (ob == best && ow < who)) best = ob, who = ow;
        }
        const bool by_plane = best < __builtin_huge_valf();
        const int axis = by_plane ? (int)who / 15 : -1, split = by_plane ? (int)who % 15 : 0;
        bool left = false;
        if (in) left = by_plane ? (axis == 0 ? my_bin[0] : (axis == 1 ? my_bin[1] : my_bin[2])) <= split : (lane - a) < c / 2;
        const unsigned long long ml = __ballot(in && left), mr = __ballot(in && !left);
        const uint32_t n_left = (uint32_t)__popcll(ml);
        if (n_left == 0 || n_left >= c) {  // the bins and the lanes disagree: internal error; leave a valid (if poor) tree
            if (lane == 0) {
                atomicOr(err, 4u);
                nodes[node].left = -1;
            }
            continue;
        }
        // the children's boxes: unions of the triangles' boxes on either side (what the bins' unions are too)
        uint32_t cbx[12];
        for (int k = 0; k < 6; k++) {
            const bool is_lo = k < 3;
            const uint32_t none = is_lo ? 0xFFFFFFFFu : 0u;
            const uint32_t vl = (in && left) ? bw[k] : none, vr = (in && !left) ? bw[k] : none;
            cbx[k] = is_lo ? wave_min_u(vl) : wave_max_u(vl);
            cbx[6 + k] = is_lo ? wave_min_u(vr) : wave_max_u(vr);
        }
        const uint32_t child = first_node + used;
        used += 2u;
        if (lane < 2) {
            Node C;
            C.lo = lo + (lane == 0 ? a : a + n_left);
            C.hi = lo + (lane == 0 ? a + n_left : b);
            C.left = -2;
            C.n_left = 0;
            C.axis = -1, C.split = 0;
            for (int k = 0; k < 6; k++) C.box[k] = cbx[lane * 6 + k], C.cb[k] = (k < 3) ? 0xFFFFFFFFu : 0u;
            C.depth = dep + 1u;
            C.pad = 0u;
            nodes[child + lane] = C;
        }
        if (lane == 0) {
            nodes[node].left = (int32_t)child;
            nodes[node].n_left = n_left;
            nodes[node].axis = axis;
            nodes[node].split = split;
            S.stack[sp][0] = child + 1u, S.stack[sp][1] = a + n_left, S.stack[sp][2] = b, S.stack[sp][3] = dep + 1u;
            S.stack[sp + 1][0] = child, S.stack[sp + 1][1] = a, S.stack[sp + 1][2] = a + n_left, S.stack[sp + 1][3] = dep + 1u;
        }
        sp += 2;
        // stable partition of lanes [a, b): ranks from the ballots, registers through LDS
        if (in) {
            const unsigned long long below = lane == 0 ? 0ULL : (~0ULL >> (64 - lane));
            const uint32_t to = left ? a + (uint32_t)__popcll(ml & below) : a + n_left + (uint32_t)__popcll(mr & below);
            S.move[0][to] = t;
            for (int k = 0; k < 6; k++) S.move[1 + k][to] = bw[k];
            for (int x = 0; x < 3; x++) S.move[7 + x][to] = __float_as_uint(ce[x]);
        }
        SAH_LDS_SYNC();
        if (in) {
            t = S.move[0][lane];
            for (int k = 0; k < 6; k++) bw[k] = S.move[1 + k][lane];
            for (int x = 0; x < 3; x++) ce[x] = __uint_as_float(S.move[7 + x][lane]);
        }
        SAH_LDS_SYNC();
    }
    if (lane < cnt) order[lo + lane] = t;
    for (uint32_t k = used + lane; k < block_nodes; k += 64) {  // the unused tail: leaves of nothing that nobody points to
        Node U;
        U.lo = U.hi = 0u;
        U.left = -1;
        U.n_left = 0u;
        U.axis = -1, U.split = 0;
        for (int q = 0; q < 6; q++) U.box[q] = U.cb[q] = (q < 3) ? 0xFFFFFFFFu : 0u;
        U.depth = 0u, U.pad = 0u;
        nodes[first_node + k] = U;
    }
    if (lane == 0 && deepest > *max_depth) atomicMax(max_depth, deepest);
}

// 5. inner nodes -> BvhNode (bvh_build.cpp:276-290: padded boxes of both children)
__global__ __launch_bounds__(kBlock) void k_sah_mark_inner(const Node *nodes, uint32_t total, uint32_t *inner) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < total) inner[i] = nodes[i].left >= 0 ? 1u : 0u;
}
__device__ __forceinline__ void padded_axis(uint32_t ulo, uint32_t uhi, float &lo, float &hi) {  // bvh_build.cpp:49-58
    const float l = dec_lo(ulo), h = dec_hi(uhi);
    const float mag = fmaxf(fmaxf(fabsf(l), fabsf(h)), h - l);
    const float pad = mag * 1.52587890625e-5f + 1e-30f;
    lo = l - pad;
    hi = h + pad;
}
__global__ __launch_bounds__(kBlock) void k_sah_emit(const Node *nodes, uint32_t total, const uint32_t *inner, const uint32_t *index, uint32_t n,
                                                     BvhNode *out, uint32_t *err) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= total || !inner[i]) return;
    const Node N = nodes[i];
    if ((uint32_t)N.left + 1u >= total) {
        atomicOr(err, 8u);
        return;
    }
    BvhNode nd;
    float lo[2][3], hi[2][3];
    for (int c = 0; c < 2; c++) {
        const Node C = nodes[N.left + c];
        for (int a = 0; a < 3; a++) padded_axis(C.box[a], C.box[3 + a], lo[c][a], hi[c][a]);
        const bool child_inner = C.left >= 0;
        if (C.lo >= C.hi || C.hi > n || (!child_inner && C.hi - C.lo > kLeafMax)) atomicOr(err, 16u);
        nd.child[c] = child_inner ? (int32_t)index[N.left + c] : ~(int32_t)C.lo;
        nd.cnt[c] = child_inner ? 0u : C.hi - C.lo;
    }
    nd.lo_x_hi_x = make_float4(lo[0][0], lo[1][0], hi[0][0], hi[1][0]);
    nd.lo_y_hi_y = make_float4(lo[0][1], lo[1][1], hi[0][1], hi[1][1]);
    nd.lo_z_hi_z = make_float4(lo[0][2], lo[1][2], hi[0][2], hi[1][2]);
    out[index[i]] = nd;
}

}  // namespace

// positions: host [n][3][3].  Outputs as build_bvh (bvh_build.cpp).  n > kLeafMax.  RRT_ERR_UNSUPPORTED_INPUT (returned as
// +1, not an rrt error code) = a coordinate is not finite: the caller uses the host builder, whose "wild mesh" rules apply.
namespace {
// the builder's device arrays: they outlive the build itself so that whoever called it can go on from them on the device
struct SahBuffers {
    DBuf<float> d_pos, d_tb, d_cen;
    DBuf<uint32_t> d_order, d_order2, d_nodeof, d_nodeof2, d_flag, d_rank, d_bins, d_word, d_inner, d_index;
    DBuf<Node> d_nodes;
    DBuf<BvhNode> d_out;
    DBuf<char> d_temp;
    // results
    const uint32_t *order = nullptr;  // triangle order (one of d_order / d_order2)
    uint32_t n_inner = 0, levels = 0;
};
int sah_build(const float *positions, uint32_t n, SahBuffers &B, float *device_ms);
}  // namespace

int build_bvh_device_sah(const float *positions, uint32_t n, std::vector<uint32_t> &order, std::vector<BvhNode> &nodes, uint32_t &max_depth,
                         float *device_ms) {
    SahBuffers B;
    int rc = sah_build(positions, n, B, device_ms);
    if (rc) return rc;
    const bool timing = getenv("RRT_LBVH_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    nodes.resize(B.n_inner);
    order.resize(n);
    RRT_HIP(hipMemcpy(nodes.data(), B.d_out.p, (size_t)B.n_inner * sizeof(BvhNode), hipMemcpyDeviceToHost));
    RRT_HIP(hipMemcpy(order.data(), B.order, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    max_depth = B.levels;
    if (timing) fprintf(stderr, "[sah] %-28s %8.2f ms\n", "download", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count());
    return RRT_OK;
}

namespace {
int sah_build(const float *positions, uint32_t n, SahBuffers &B, float *device_ms) {
    DBuf<float> &d_pos = B.d_pos, &d_tb = B.d_tb, &d_cen = B.d_cen;
    DBuf<uint32_t> &d_order = B.d_order, &d_order2 = B.d_order2, &d_nodeof = B.d_nodeof, &d_nodeof2 = B.d_nodeof2, &d_flag = B.d_flag, &d_rank = B.d_rank,
                   &d_bins = B.d_bins, &d_word = B.d_word, &d_inner = B.d_inner, &d_index = B.d_index;
    DBuf<Node> &d_nodes = B.d_nodes;
    DBuf<BvhNode> &d_out = B.d_out;
    DBuf<char> &d_temp = B.d_temp;
    if (n <= kLeafMax) return fail(RRT_ERR_INVALID, "device BVH build needs more than 4 triangles");
    if (n >= (1u << 27)) return fail(RRT_ERR_LIMIT, "more than 2^27 triangles in one mesh");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(RRT_ERR_DEVICE, "no HIP device: the device BVH builder has no CPU fallback");
    const bool timing = getenv("RRT_LBVH_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[sah] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count());
        T0 = t;
    };
    const uint32_t capacity = 4 * n + 64;      // node records: fewer than 2n in the tree; k_sah_small reserves 2 cnt - 2 per subtree and leaves some unused
    const uint32_t bin_slots = n / 16 + 64;    // nodes of one level: the children of nodes of more than kSmall triangles, so fewer than
                                               // 2 n / 65; twice that is reserved (a wider level is refused, not overrun)
    int rc;
    if ((rc = d_pos.ensure((size_t)n * 9)) || (rc = d_tb.ensure((size_t)n * 6)) || (rc = d_cen.ensure((size_t)n * 3)) || (rc = d_order.ensure(n)) ||
        (rc = d_order2.ensure(n)) || (rc = d_nodeof.ensure(n)) || (rc = d_nodeof2.ensure(n)) || (rc = d_flag.ensure(n + 1)) || (rc = d_rank.ensure(n + 1)) ||
        (rc = d_bins.ensure((size_t)bin_slots * kBinWords)) || (rc = d_word.ensure(4)) || (rc = d_inner.ensure(capacity)) ||
        (rc = d_index.ensure(capacity)) || (rc = d_nodes.ensure(capacity)) || (rc = d_out.ensure(n)))
        return rc;
    size_t scan_bytes = 0, scan2_bytes = 0;
    RRT_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, d_flag.p, d_rank.p, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), nullptr));
    RRT_HIP(rocprim::exclusive_scan(nullptr, scan2_bytes, d_inner.p, d_index.p, 0u, (size_t)capacity, rocprim::plus<uint32_t>(), nullptr));
    if ((rc = d_temp.ensure(std::max<size_t>(std::max(scan_bytes, scan2_bytes), 1)))) return rc;
    lap("allocations");
    RRT_HIP(hipMemcpy(d_pos.p, positions, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice));
    Node root;
    memset(&root, 0, sizeof(root));
    root.lo = 0, root.hi = n, root.left = -2, root.axis = -1, root.depth = 1;
    for (int k = 0; k < 3; k++) root.box[k] = root.cb[k] = 0xFFFFFFFFu, root.box[3 + k] = root.cb[3 + k] = 0u;
    RRT_HIP(hipMemcpy(d_nodes.p, &root, sizeof(root), hipMemcpyHostToDevice));
    RRT_HIP(hipMemset(d_word.p, 0, 4 * sizeof(uint32_t)));  // [0] nodes a level created, [1] error bits, [2] nodes k_sah_small created, [3] its deepest node
    RRT_HIP(hipMemset(d_flag.p + n, 0, sizeof(uint32_t)));  // the scan runs over n + 1 entries: rank[n] = all left-going triangles
    lap("upload");
    hipEvent_t e0, e1;
    RRT_HIP(hipEventCreate(&e0));
    RRT_HIP(hipEventCreate(&e1));
    RRT_HIP(hipEventRecord(e0, nullptr));
    const dim3 chunk_grid((uint32_t)(((uint64_t)n + kChunk * (kBlock / 64) - 1) / (kChunk * (kBlock / 64))));
    hipLaunchKernelGGL(k_sah_prepare, chunk_grid, dim3(kBlock), 0, nullptr, d_pos.p, n, d_tb.p, d_cen.p, d_order.p, d_nodeof.p, d_nodes.p, d_word.p + 1);
    uint32_t level_first = 0, level_count = 1, levels = 0, total = 1;
    uint32_t *order_a = d_order.p, *order_b = d_order2.p, *nodeof_a = d_nodeof.p, *nodeof_b = d_nodeof2.p;
    hipError_t se = hipGetLastError();
    uint32_t word[2] = {0, 0};  // [0] nodes the level created, [1] error bits
    while (se == hipSuccess) {
        levels++;
        if (level_count > bin_slots || levels > 4096) {  // a level wider than n / 2 nodes cannot be (see bin_slots)
            word[1] |= 32u;
            break;
        }
        hipLaunchKernelGGL(k_sah_clear_bins, grid_for((uint64_t)level_count * kBinWords), dim3(kBlock), 0, nullptr, d_bins.p, (uint64_t)level_count * kBinWords);
        (void)hipMemsetAsync(d_word.p, 0, sizeof(uint32_t), nullptr);
        hipLaunchKernelGGL(k_sah_bin, chunk_grid, dim3(kBlock), 0, nullptr, order_a, nodeof_a, d_nodes.p, d_tb.p, d_cen.p, n, level_first, d_bins.p);
        hipLaunchKernelGGL(k_sah_decide, grid_for(level_count), dim3(kBlock), 0, nullptr, d_nodes.p, level_first, level_count, d_bins.p,
                           level_first + level_count, capacity, d_word.p, d_word.p + 1);
        if ((se = hipGetLastError()) != hipSuccess) break;
        if ((se = hipMemcpy(word, d_word.p, sizeof(word), hipMemcpyDeviceToHost)) != hipSuccess) break;
        if (word[1] || word[0] == 0) break;  // an error, or every node of this level is a leaf
        hipLaunchKernelGGL(k_sah_flag, grid_for(n), dim3(kBlock), 0, nullptr, order_a, nodeof_a, d_nodes.p, d_cen.p, n, d_flag.p);
        se = rocprim::exclusive_scan(d_temp.p, scan_bytes, d_flag.p, d_rank.p, 0u, (size_t)n + 1, rocprim::plus<uint32_t>(), nullptr);
        if (se != hipSuccess) break;
        const ScatterArgs sa = {order_a, nodeof_a, d_nodes.p, d_tb.p, d_cen.p, d_flag.p, d_rank.p, order_b, nodeof_b};
        hipLaunchKernelGGL(k_sah_scatter, chunk_grid, dim3(kBlock), 0, nullptr, sa, n);
        std::swap(order_a, order_b);
        std::swap(nodeof_a, nodeof_b);
        level_first += level_count;
        level_count = word[0];
        total += word[0];
        se = hipGetLastError();
    }
    uint32_t n_inner = 0;
    if (se == hipSuccess && word[1] == 0) {  // the subtrees of the nodes the level rounds left at kSmall triangles or fewer
        hipLaunchKernelGGL(k_sah_small, dim3((total + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, nullptr, d_nodes.p, total, capacity, order_a, d_tb.p,
                           d_cen.p, d_word.p + 2, d_word.p + 3, d_word.p + 1);
        uint32_t w4[4] = {0, 0, 0, 0};
        se = hipGetLastError();
        if (se == hipSuccess) se = hipMemcpy(w4, d_word.p, sizeof(w4), hipMemcpyDeviceToHost);
        word[1] = w4[1];
        if (se == hipSuccess && word[1] == 0) {
            if (w4[2] > capacity - total) word[1] |= 2u;
            total += w4[2];
            levels = std::max(levels, w4[3]);
        }
    }
    if (se == hipSuccess && word[1] == 0) {
        hipLaunchKernelGGL(k_sah_mark_inner, grid_for(total), dim3(kBlock), 0, nullptr, d_nodes.p, total, d_inner.p);
        se = rocprim::exclusive_scan(d_temp.p, scan2_bytes, d_inner.p, d_index.p, 0u, (size_t)total, rocprim::plus<uint32_t>(), nullptr);
        if (se == hipSuccess) {
            hipLaunchKernelGGL(k_sah_emit, grid_for(total), dim3(kBlock), 0, nullptr, d_nodes.p, total, d_inner.p, d_index.p, n, d_out.p, d_word.p + 1);
            se = hipGetLastError();
        }
        uint32_t last[2] = {0, 0};
        if (se == hipSuccess) se = hipMemcpy(&last[0], d_index.p + (total - 1), sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (se == hipSuccess) se = hipMemcpy(&last[1], d_inner.p + (total - 1), sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (se == hipSuccess) se = hipMemcpy(word, d_word.p, sizeof(word), hipMemcpyDeviceToHost);
        n_inner = last[0] + last[1];
    }
    hipError_t re = hipEventRecord(e1, nullptr);
    hipError_t ye = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (se == hipSuccess && re == hipSuccess && ye == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    RRT_HIP(se);
    RRT_HIP(re);
    RRT_HIP(ye);
    if (device_ms) *device_ms = ms;
    lap("kernels (wall)");
    if (timing) fprintf(stderr, "[sah] kernels (HIP events)         %8.2f ms, %u levels, %u nodes\n", ms, levels, total);
    if (word[1] == 1u) return 1;  // non-finite input: the host builder's business
    // refusals — the node workspace is full (2), a level is wider or the tree deeper than the builder is laid out for (32):
    // RRT_ERR_LIMIT, which RRT_BVH_AUTO answers with the host builder; everything else is a failed self-check
    if (word[1] && !(word[1] & ~(2u | 32u))) return fail(RRT_ERR_LIMIT, "device SAH build: workspace or level limit reached");
    if (word[1] || n_inner == 0 || n_inner >= n || total > capacity) {
        char what[160];
        snprintf(what, sizeof(what), "device SAH build produced an invalid tree (internal error): check bits 0x%x, %u inner nodes of %u, %u levels", word[1],
                 n_inner, total, levels);
        return fail(RRT_ERR_DEVICE, what);
    }
    B.order = order_a;
    B.n_inner = n_inner;
    B.levels = levels;
    return RRT_OK;
}
}  // namespace

// ---- the rest of a mesh's flattening on the device (pre_render without host stages) -----------------
// After the build the tree, the triangle order and the positions are in HBM; the traversal's arrays are made from them
// there: the hull of the tree's boxes (-> the mesh's 16-bit grid, a few words read back), the four-wide collapse level by level
// (bvh_quant.h's own functions: the host's bytes), the triangle records.  Nothing of the mesh goes back to the host but the
// grid and the node count.
namespace {
// hull of every child box of the tree: order-preserving words under atomicMin / atomicMax; acc[6]: wild
__global__ __launch_bounds__(kBlock) void k_tree_hull(const BvhNode *nodes, uint32_t n_nodes, uint32_t *acc) {
    __shared__ uint32_t s_acc[7];
    if (threadIdx.x < 3) s_acc[threadIdx.x] = 0xFFFFFFFFu;
    if (threadIdx.x >= 3 && threadIdx.x < 7) s_acc[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_nodes) {
        const float lim = 1152921504606846976.0f;  // 2^60
        const BvhNode n = nodes[i];
        bool wild = false;
        for (int a = 0; a < 3; a++) {
            const float4 b = bq::box_axis(n, a);
            const float v[4] = {b.x, b.y, b.z, b.w};
            for (int c = 0; c < 2; c++) {
                const float l = v[c], h = v[2 + c];
                if (l > h) continue;  // empty child
                if (!(fabsf(l) < lim) || !(fabsf(h) < lim)) wild = true;
                if (l == l) atomicMin(&s_acc[a], ordered(l));      // (std::min(lo, l) never takes a NaN)
                if (h == h) atomicMax(&s_acc[3 + a], ordered(h));
            }
            if (b.x != b.x || b.y != b.y || b.z != b.z || b.w != b.w) wild = true;
        }
        if (wild) s_acc[6] = 1u;
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&acc[threadIdx.x], s_acc[threadIdx.x]);
    if (threadIdx.x >= 3 && threadIdx.x < 6) atomicMax(&acc[threadIdx.x], s_acc[threadIdx.x]);
    if (threadIdx.x == 6 && s_acc[6]) acc[6] = 1u;
}
// One level of the collapse.  Work item i: output node dst0 + i stands for BVH2 node src[i].
//   count: how many inner children it will have (-> exclusive scan = where its children's indices start)
//   emit : the node itself, its inner children as the next level's work items, their stack need
__global__ void k_collapse_count(const BvhNode *nodes, const uint32_t *src, uint32_t n_items, uint32_t wild, uint32_t *count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    bq::Slot sl[4];
    const int m = bq::collapse(nodes, src[i], wild != 0u, sl);
    uint32_t inner = 0;
    for (int k = 0; k < m; k++) inner += sl[k].ref >= 0 ? 1u : 0u;
    count[i] = inner;
}
__global__ void k_collapse_emit(const BvhNode *nodes, const uint32_t *src, const uint32_t *need, uint32_t n_items, bq::Grid grid, const uint32_t *offset,
                                uint32_t dst0, uint32_t next0, uint32_t capacity, BvhNodeQ *out, uint32_t *next_src, uint32_t *next_need, uint32_t *stack_need) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    bq::Slot sl[4];
    const int m = bq::collapse(nodes, src[i], grid.wild != 0u, sl);
    const uint32_t below = need[i] + (m > 0 ? (uint32_t)m - 1u : 0u);  // stack entries while below this node
    atomicMax(stack_need, below);
    const uint32_t first = next0 + offset[i];
    if (dst0 + i < capacity) out[dst0 + i] = bq::encode(grid, sl, m, first);
    uint32_t rank = 0;
    for (int k = 0; k < m; k++)
        if (sl[k].ref >= 0) {
            next_src[offset[i] + rank] = (uint32_t)sl[k].ref;
            next_need[offset[i] + rank] = below;
            rank++;
        }
}
// what Triangle::get_plane caches (triangle_object.rs:51-52, 85-89; plane_object.rs:37), in BVH order
__global__ void k_tri_records(const float *pos, const uint32_t *order, uint32_t n, TriHot *tris, uint32_t *tri_index) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t src = order[k];
    const float *p = pos + (size_t)src * 9;
    const f3 a = mk3(p[0], p[1], p[2]), b = mk3(p[3], p[4], p[5]), c = mk3(p[6], p[7], p[8]);
    const f3 ab = b - a, bc = c - b;
    const f3 nrm = norm3(cross3(ab, bc));
    const float d = -dot3(nrm, a);
    TriHot t;
    t.a_nx = make_float4(a.x, a.y, a.z, nrm.x);
    t.b_ny = make_float4(b.x, b.y, b.z, nrm.y);
    t.c_nz = make_float4(c.x, c.y, c.z, nrm.z);
    t.d_pad = make_float4(d, __uint_as_float(src), 0.0f, 0.0f);
    tris[k] = t;
    tri_index[k] = src;
}
}  // namespace

int build_mesh_device_sah(const float *positions, uint32_t n, MeshInfo &mi, DeviceMesh &M, uint32_t &stack_need, float *device_ms) {
    SahBuffers B;
    int rc = sah_build(positions, n, B, device_ms);
    if (rc) return rc;
    const bool timing = getenv("RRT_LBVH_TIMING") != nullptr;
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipDeviceSynchronize();
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[flatten, device] %-17s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count());
        T0 = t;
    };
    const uint32_t n_inner = B.n_inner;
    // triangle records: independent of everything below
    if ((rc = M.tris.ensure(n)) || (rc = M.tri_index.ensure(n))) return rc;
    hipLaunchKernelGGL(k_tri_records, grid_for(n), dim3(kBlock), 0, nullptr, B.d_pos.p, B.order, n, M.tris.p, M.tri_index.p);
    M.n_tris = n;
    // the grid
    uint32_t acc[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};  // [7]: stack need
    DBuf<uint32_t> d_acc;
    if ((rc = d_acc.ensure(8))) return rc;
    RRT_HIP(hipMemcpy(d_acc.p, acc, sizeof(acc), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_tree_hull, grid_for(n_inner), dim3(kBlock), 0, nullptr, B.d_out.p, n_inner, d_acc.p);
    RRT_HIP(hipGetLastError());
    RRT_HIP(hipMemcpy(acc, d_acc.p, sizeof(acc), hipMemcpyDeviceToHost));
    bq::Hull hull;
    for (int a = 0; a < 3; a++) {
        hull.lo[a] = acc[a] == 0xFFFFFFFFu ? INFINITY : unordered(acc[a]);
        hull.hi[a] = acc[3 + a] == 0u ? -INFINITY : unordered(acc[3 + a]);
    }
    hull.wild = acc[6] != 0u;
    bq::make_grid(hull, mi);
    const bq::Grid grid = bq::grid_of(mi);
    lap("triangles + grid");
    // the collapse, a level per round: every output node has at least two children and every BVH2 node stands for at most
    // one output node, so there are at most n_inner of them
    DBuf<uint32_t> src[2], need[2], count, offset;
    DBuf<char> temp;
    if ((rc = M.qnodes.ensure(n_inner)) || (rc = src[0].ensure(n_inner)) || (rc = src[1].ensure(n_inner)) || (rc = need[0].ensure(n_inner)) ||
        (rc = need[1].ensure(n_inner)) || (rc = count.ensure(n_inner + 1)) || (rc = offset.ensure(n_inner + 1)))
        return rc;
    size_t scan_bytes = 0;
    RRT_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, count.p, offset.p, 0u, (size_t)n_inner + 1, rocprim::plus<uint32_t>(), nullptr));
    if ((rc = temp.ensure(std::max<size_t>(scan_bytes, 1)))) return rc;
    const uint32_t zero[2] = {0u, 0u};
    RRT_HIP(hipMemcpy(src[0].p, zero, sizeof(uint32_t), hipMemcpyHostToDevice));   // the root stands for BVH2 node 0 ...
    RRT_HIP(hipMemcpy(need[0].p, zero, sizeof(uint32_t), hipMemcpyHostToDevice));  // ... and is reached with an empty stack
    uint32_t dst0 = 0, items = 1, rounds = 0;
    int cur = 0;
    while (items) {
        if (dst0 + items > n_inner || ++rounds > 4096) return fail(RRT_ERR_DEVICE, "device BVH collapse produced an invalid tree (internal error)");
        hipLaunchKernelGGL(k_collapse_count, grid_for(items), dim3(kBlock), 0, nullptr, B.d_out.p, src[cur].p, items, grid.wild, count.p);
        // (the scan runs over items + 1 entries: offset[items] = the next level's size; count[items] itself is not read)
        RRT_HIP(rocprim::exclusive_scan(temp.p, scan_bytes, count.p, offset.p, 0u, (size_t)items + 1, rocprim::plus<uint32_t>(), nullptr));
        hipLaunchKernelGGL(k_collapse_emit, grid_for(items), dim3(kBlock), 0, nullptr, B.d_out.p, src[cur].p, need[cur].p, items, grid, offset.p, dst0,
                           dst0 + items, n_inner, M.qnodes.p, src[cur ^ 1].p, need[cur ^ 1].p, d_acc.p + 7);
        RRT_HIP(hipGetLastError());
        uint32_t next_items = 0;
        RRT_HIP(hipMemcpy(&next_items, offset.p + items, sizeof(uint32_t), hipMemcpyDeviceToHost));
        dst0 += items;
        items = next_items;
        cur ^= 1;
    }
    uint32_t need_max = 0;
    RRT_HIP(hipMemcpy(&need_max, d_acc.p + 7, sizeof(uint32_t), hipMemcpyDeviceToHost));
    M.n_nodes = dst0;
    stack_need = need_max + 1;  // entries below the current node + one word of slack (quantise_bvh)
    lap("collapse");
    if (timing) fprintf(stderr, "[flatten, device] %u four-wide nodes in %u levels, stack need %u\n", dst0, rounds, stack_need);
    return RRT_OK;
}

}  // namespace rrt

// rrt_scene_create touches one kernel of every translation unit, so that the code objects are loaded before the first
// rrt_pre_render is timed (HIP loads them on first use: tens of milliseconds per unit)
namespace rrt {
void preload_bvh_sah() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_tri_records));
}
}  // namespace rrt
