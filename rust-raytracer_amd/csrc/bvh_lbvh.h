// bvh_lbvh.h — the pieces of the device-side LBVH build as functions of a thread index, shared by the HIP kernels
// (bvh_lbvh.hip) and by tools/cpu_emu/lbvh_check.cpp, which runs them serially under AddressSanitizer.
//
// moves the mesh part of pre_render onto the GPU (the reference has no acceleration structure at
// all: polymesh_object.rs:219-236 loops over every triangle).  Like the host builder (bvh_build.cpp) the tree may only
// prune: every child box is the padded union of the boxes of the triangles below it.
//
//   1. triangle boxes, centroid bounds of the mesh                                   lbvh::tri_box
//   2. 63-bit Morton key of each centroid (21 bits per axis)                          lbvh::morton63
//   3. radix sort of (key, triangle)                                                  rocPRIM
//   4. hierarchy over the sorted keys: Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and
//      k-d Trees" (HPG 2012); equal keys are told apart by their sorted position     lbvh::karras_node
//   5. a sparse table (range-union table) over the sorted triangle boxes: level k holds the union of 2^k consecutive
//      boxes, so the box of ANY sorted range is the union of two entries.  No bottom-up pass, no atomics, no fences:
//      every node's child boxes are pure functions of its range.                      lbvh::table_level, range_box
//   6. node emit: ranges of <= 4 triangles become leaves                              lbvh::emit_node
//   7. VALIDATION on the device, without a traversal (lbvh::check_node, check_slot, check_order): the sorted order is a
//      permutation; the root covers [0, n - 1]; every emitted node splits its range into two non-empty parts, an inner
//      child's own range is exactly the part it is given, and every triangle slot is claimed by exactly one leaf.  Together
//      these say what a DFS would: starting from the root the ranges nest (induction over the local rule), so every
//      reachable node is reached once and the reachable leaves cover [0, n - 1] exactly; an emitted node that is NOT
//      reachable would still end in leaves, which would claim slots a second time.  Anything else raises an error word:
//      an invalid tree is an error, never traversed.
//   8. depth of every emitted node by walking the parent links check_node wrote (lbvh::node_depth), its maximum;
//      compaction by a prefix sum over the emitted flags (Karras order is kept: a node's index is a position in the
//      Morton order, so neighbours in memory are neighbours in space), child indices renumbered (lbvh::compact_node).
//      The host's one-DFS version of 7 + 8 (lbvh::finish) stays for the CPU preflight, which checks both against each other.
#pragma once
#include <math.h>
#include <stdint.h>

#include <vector>

#include "dev_scene.h"

#if defined(__HIPCC__)
#define LBVH_HD __host__ __device__ __forceinline__
#else
#define LBVH_HD inline
#endif

namespace rrt {
namespace lbvh {

constexpr uint32_t kLeafMax = 4;

struct Box6 {
    float lo[3], hi[3];
};
struct Range {
    uint32_t first, last, split;  // node covers sorted slots [first, last]; children [first, split] and [split + 1, last]
};

LBVH_HD int clz32(uint32_t v) { return v ? __builtin_clz(v) : 32; }
LBVH_HD int clz64(uint64_t v) { return v ? __builtin_clzll(v) : 64; }

LBVH_HD Box6 box_union(const Box6 &a, const Box6 &b) {
    Box6 r;
    for (int k = 0; k < 3; k++) {
        r.lo[k] = fminf(a.lo[k], b.lo[k]);
        r.hi[k] = fmaxf(a.hi[k], b.hi[k]);
    }
    return r;
}

// box of triangle i of positions[n][3][3]
LBVH_HD Box6 tri_box(const float *positions, uint32_t i) {
    const float *p = positions + (size_t)i * 9;
    Box6 b;
    for (int k = 0; k < 3; k++) {
        b.lo[k] = fminf(fminf(p[k], p[3 + k]), p[6 + k]);
        b.hi[k] = fmaxf(fmaxf(p[k], p[3 + k]), p[6 + k]);
    }
    return b;
}

// order-preserving map float -> uint32 (for atomicMin / atomicMax on floats)
LBVH_HD uint32_t ordered(float f) {
    union {
        float f;
        uint32_t u;
    } v;
    v.f = f;
    return (v.u & 0x80000000u) ? ~v.u : (v.u | 0x80000000u);
}
LBVH_HD float unordered(uint32_t u) {
    union {
        float f;
        uint32_t u;
    } v;
    v.u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return v.f;
}

LBVH_HD uint64_t expand21(uint32_t v) {  // 21 bits -> every third bit
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x001F00000000FFFFull;
    x = (x | x << 16) & 0x001F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
// centroid bounds as 6 ordered words: lo xyz, hi xyz
LBVH_HD uint64_t morton63(const Box6 &b, const uint32_t *bounds) {
    uint64_t key = 0;
    for (int k = 0; k < 3; k++) {
        float lo = unordered(bounds[k]), hi = unordered(bounds[3 + k]);
        float c = 0.5f * (b.lo[k] + b.hi[k]);
        float ext = hi - lo;
        float q = ext > 0.0f ? (c - lo) / ext * 2097152.0f : 0.0f;
        q = fminf(fmaxf(q, 0.0f), 2097151.0f);  // NaN -> 0
        key |= expand21((uint32_t)q) << (2 - k);
    }
    return key;
}

// length of the common prefix of the keys at sorted positions i and j (-1 outside the array)
LBVH_HD int delta(const uint64_t *keys, uint32_t n, int64_t i, int64_t j) {
    if (j < 0 || j >= (int64_t)n) return -1;
    uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + clz32((uint32_t)i ^ (uint32_t)j);
    return clz64(a ^ b);
}

// internal node i of n - 1 (n >= 2)
LBVH_HD Range karras_node(const uint64_t *keys, uint32_t n, uint32_t node) {
    const int64_t i = node;
    const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int64_t lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int64_t l = 0;
    for (int64_t t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int64_t j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int64_t s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int64_t split = i + s * d + (d < 0 ? -1 : 0);
    Range r;
    r.first = (uint32_t)(i < j ? i : j);
    r.last = (uint32_t)(i < j ? j : i);
    r.split = (uint32_t)split;
    return r;
}

// sparse table: level k, entry i = union of the sorted boxes [i, i + 2^k); valid for i + 2^k <= n
LBVH_HD void table_level(const Box6 *prev, Box6 *cur, uint32_t n, uint32_t k, uint32_t i) {
    const uint32_t half = 1u << (k - 1);
    if ((uint64_t)i + 2ull * half <= n) cur[i] = box_union(prev[i], prev[i + half]);
}
// table = levels stored one after the other, n entries each
LBVH_HD Box6 range_box(const Box6 *table, uint32_t n, uint32_t a, uint32_t b) {
    const uint32_t len = b - a + 1;
    const uint32_t k = 31u - (uint32_t)clz32(len);
    const Box6 *lv = table + (size_t)k * n;
    return box_union(lv[a], lv[b + 1 - (1u << k)]);
}
inline uint32_t table_levels(uint32_t n) { return 32u - (uint32_t)clz32(n); }  // floor(log2 n) + 1

// same padding as bvh_build.cpp: 2^-16 of the coordinate magnitude
LBVH_HD Box6 padded(const Box6 &b) {
    Box6 r;
    for (int k = 0; k < 3; k++) {
        float mag = fmaxf(fmaxf(fabsf(b.lo[k]), fabsf(b.hi[k])), b.hi[k] - b.lo[k]);
        float pad = mag * 1.52587890625e-5f + 1e-30f;
        r.lo[k] = b.lo[k] - pad;
        r.hi[k] = b.hi[k] + pad;
    }
    return r;
}

// node `i` in Karras numbering (children of a node are `split` and `split + 1`); nodes whose range fits a leaf are
// written as unreachable placeholders
LBVH_HD BvhNode emit_node(const Range *ranges, const Box6 *table, uint32_t n, uint32_t i) {
    BvhNode nd;
    const Range r = ranges[i];
    if (i != 0 && r.last - r.first + 1 <= kLeafMax) {
        nd.lo_x_hi_x = nd.lo_y_hi_y = nd.lo_z_hi_z = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        nd.child[0] = nd.child[1] = ~0;
        nd.cnt[0] = nd.cnt[1] = 0;
        return nd;
    }
    Box6 bx[2];
    for (int c = 0; c < 2; c++) {
        const uint32_t a = c == 0 ? r.first : r.split + 1, b = c == 0 ? r.split : r.last;
        bx[c] = padded(range_box(table, n, a, b));
        const uint32_t size = b - a + 1;
        if (size <= kLeafMax) {
            nd.child[c] = ~(int32_t)a;
            nd.cnt[c] = size;
        } else {
            nd.child[c] = (int32_t)(c == 0 ? r.split : r.split + 1);
            nd.cnt[c] = 0;
        }
    }
    nd.lo_x_hi_x = make_float4(bx[0].lo[0], bx[1].lo[0], bx[0].hi[0], bx[1].hi[0]);
    nd.lo_y_hi_y = make_float4(bx[0].lo[1], bx[1].lo[1], bx[0].hi[1], bx[1].hi[1]);
    nd.lo_z_hi_z = make_float4(bx[0].lo[2], bx[1].lo[2], bx[0].hi[2], bx[1].hi[2]);
    return nd;
}

// ---- 7 + 8 on the device --------------------------------------------------------------------------------------------
enum : uint32_t {
    kErrOrder = 1u,     // the sorted order is not a permutation of the triangles
    kErrRoot = 2u,      // the root does not cover [0, n - 1]
    kErrSplit = 4u,     // a split outside its node's range
    kErrChild = 8u,     // an inner child whose own range is not the part it was given (or whose index is out of range)
    kErrCover = 16u,    // a triangle slot in two leaves, or in none
    kErrDepth = 32u,    // parent links do not lead to the root within 4096 steps
};
LBVH_HD bool is_emitted(const Range &r, uint32_t i) { return i == 0 || r.last - r.first + 1 > kLeafMax; }

#if defined(__HIP_DEVICE_COMPILE__)
#define LBVH_FETCH_ADD(p, v) atomicAdd((p), (v))
#define LBVH_FETCH_OR(p, v) atomicOr((p), (v))
#else  // the CPU preflight runs the same functions one thread after the other
LBVH_HD uint32_t lbvh_host_add(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
LBVH_HD uint32_t lbvh_host_or(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o | v; return o; }
#define LBVH_FETCH_ADD(p, v) rrt::lbvh::lbvh_host_add((p), (v))
#define LBVH_FETCH_OR(p, v) rrt::lbvh::lbvh_host_or((p), (v))
#endif

// sorted slot i: order[i] must be a triangle nobody else names (seen[] starts at zero)
LBVH_HD void check_order(const uint32_t *order, uint32_t n, uint32_t i, uint32_t *seen, uint32_t *err) {
    const uint32_t t = order[i];
    if (t >= n || LBVH_FETCH_ADD(&seen[t], 1u) != 0u) LBVH_FETCH_OR(err, (uint32_t)kErrOrder);
}
// node i (Karras numbering): the local rule of step 7; claims the slots of its leaf children in covered[] (starts at
// zero) and writes parent[] of its inner children.  Returns whether the node is emitted.
LBVH_HD bool check_node(const Range *ranges, uint32_t n, uint32_t i, uint32_t *covered, uint32_t *parent, uint32_t *err) {
    const Range r = ranges[i];
    if (!(r.first <= r.last && r.last < n)) {  // nothing below can be trusted
        LBVH_FETCH_OR(err, (uint32_t)kErrSplit);
        return false;
    }
    if (!is_emitted(r, i)) return false;
    if (i == 0 && !(r.first == 0 && r.last == n - 1)) LBVH_FETCH_OR(err, (uint32_t)kErrRoot);
    if (!(r.split >= r.first && r.split < r.last)) {
        LBVH_FETCH_OR(err, (uint32_t)kErrSplit);
        return true;
    }
    for (int c = 0; c < 2; c++) {
        const uint32_t a = c == 0 ? r.first : r.split + 1, b = c == 0 ? r.split : r.last;
        if (b - a + 1 <= kLeafMax) {
            for (uint32_t k = a; k <= b; k++)
                if (LBVH_FETCH_ADD(&covered[k], 1u) != 0u) LBVH_FETCH_OR(err, (uint32_t)kErrCover);
        } else {
            const uint32_t ci = c == 0 ? r.split : r.split + 1;  // what emit_node writes
            if (ci >= n - 1 || ranges[ci].first != a || ranges[ci].last != b)
                LBVH_FETCH_OR(err, (uint32_t)kErrChild);
            else
                parent[ci] = i;
        }
    }
    return true;
}
LBVH_HD void check_slot(const uint32_t *covered, uint32_t i, uint32_t *err) {
    if (covered[i] != 1u) LBVH_FETCH_OR(err, (uint32_t)kErrCover);
}
// depth of emitted node i the way bvh_build.cpp counts (root = 1); 0 when the links do not reach the root
LBVH_HD uint32_t node_depth(const uint32_t *parent, uint32_t n, uint32_t i, uint32_t *err) {
    uint32_t d = 1, j = i;
    while (j != 0) {
        j = parent[j];
        d++;
        if (j >= n - 1 || d > 4096) {
            LBVH_FETCH_OR(err, (uint32_t)kErrDepth);
            return 0;
        }
    }
    return d;
}
// emitted node i -> its place in the compacted array (new_index = exclusive prefix sum of the emitted flags)
LBVH_HD BvhNode compact_node(const BvhNode *raw, const uint32_t *new_index, uint32_t i) {
    BvhNode nd = raw[i];
    for (int c = 0; c < 2; c++)
        if (nd.child[c] >= 0) nd.child[c] = (int32_t)new_index[nd.child[c]];
    return nd;
}

// Host: validate the emitted nodes (Karras numbering, n - 1 of them), measure the depth the way bvh_build.cpp does
// (root = 1, a leaf counts one more than its node) and compact into DFS pre-order.  Returns nullptr or what is wrong.
inline const char *finish(const std::vector<BvhNode> &raw, uint32_t n_tris, std::vector<BvhNode> &nodes, uint32_t &max_depth) {
    const uint32_t n_raw = (uint32_t)raw.size();
    if (n_tris <= kLeafMax || n_raw != n_tris - 1) return "bad node count";
    std::vector<uint8_t> covered(n_tris, 0);
    std::vector<int32_t> new_index(n_raw, -1);
    struct Item {
        uint32_t node, depth;
    };
    std::vector<Item> stack;
    std::vector<uint32_t> pre;  // raw ids in DFS pre-order
    stack.push_back({0u, 1u});
    uint64_t covered_total = 0;
    max_depth = 1;
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        if (it.node >= n_raw) return "child index out of range";
        if (new_index[it.node] >= 0) return "node reached twice";
        new_index[it.node] = (int32_t)pre.size();
        pre.push_back(it.node);
        if (it.depth > 4096) return "tree too deep";
        const BvhNode &nd = raw[it.node];
        for (int c = 1; c >= 0; c--) {  // right first, so the left subtree follows its parent in memory
            if (nd.child[c] >= 0) {
                stack.push_back({(uint32_t)nd.child[c], it.depth + 1});
            } else {
                uint32_t first = (uint32_t)~nd.child[c], cnt = nd.cnt[c];
                if (cnt == 0 || cnt > kLeafMax || (uint64_t)first + cnt > n_tris) return "bad leaf";
                for (uint32_t k = 0; k < cnt; k++) {
                    if (covered[first + k]) return "triangle in two leaves";
                    covered[first + k] = 1;
                }
                covered_total += cnt;
                if (it.depth + 1 > max_depth) max_depth = it.depth + 1;
            }
        }
    }
    if (covered_total != n_tris) return "triangles missing from the tree";
    nodes.resize(pre.size());
    for (size_t k = 0; k < pre.size(); k++) {
        BvhNode nd = raw[pre[k]];
        for (int c = 0; c < 2; c++)
            if (nd.child[c] >= 0) nd.child[c] = new_index[nd.child[c]];
        nodes[k] = nd;
    }
    return nullptr;
}

}  // namespace lbvh
}  // namespace rrt
