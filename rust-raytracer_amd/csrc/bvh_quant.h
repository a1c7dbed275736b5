// bvh_quant.h — from a builder's binary tree (BvhNode) to the traversal's four-wide quantised nodes (BvhNodeQ): the pieces
// the host version (quantise_bvh, bvh_build.cpp) and the device version (bvh_sah.hip, k_collapse_*) share, so that both
// produce the same bytes: the mesh's 16-bit grid, the outward rounding onto it, the choice of a node's four children.
//
// The grid.  Containment is checked in exact arithmetic: qbase + q * qscale is exact in double (24-bit x 16-bit product,
// 24-bit addend), so "decoded lower face <= box lower face" holds for the real numbers the device's fma approximates; the
// traversal's own rounding is covered by its pad (dev_trace.h, make_slabs).
//
// The collapse.  A node starts with the two children of its BVH2 node; as long as it has room, the inner child with the
// largest surface takes its own two children's place (the box it would have been tested with is then never tested: the
// grandchildren's boxes lie inside it).  The tree keeps exactly the BVH2's leaves and their boxes, so it prunes nothing the
// BVH2 would not prune.  Nodes are numbered breadth first: a level's nodes in the order of their parents, a node's children
// in slot order — a numbering every level of which can be computed at once (a prefix sum over the parents' inner children).
#pragma once
#include <math.h>

#include "dev_scene.h"

namespace rrt {
namespace bq {

// what the grid is made from: the hull of every child box of the tree, and whether any coordinate is unusable
struct Hull {
    float lo[3], hi[3];
    bool wild;
};
RRT_HD const float4 &box_axis(const BvhNode &n, int a) { return a == 0 ? n.lo_x_hi_x : (a == 1 ? n.lo_y_hi_y : n.lo_z_hi_z); }

// MeshInfo's grid fields from the hull (host side: a few dozen operations per mesh)
inline void make_grid(Hull h, MeshInfo &mi) {
    bool wild = h.wild;
    for (int a = 0; a < 3; a++)
        if (!(h.lo[a] <= h.hi[a])) wild = true;
    mi.wild = wild ? 1u : 0u;
    mi.qmag = 0.0f;
    for (int a = 0; a < 3; a++) {
        mi.qbase[a] = wild ? 0.0f : h.lo[a];
        // a little more than extent / 65535, so that the top of the range decodes at or above hi; never 0
        float sc = wild ? 0.0f : (float)(((double)h.hi[a] - (double)h.lo[a]) / 65535.0 * (1.0 + 1e-6));
        if (!wild && !(sc > 0.0f)) sc = fmaxf(fabsf(h.lo[a]), 1.0f) * 1e-7f;  // flat mesh on this axis
        mi.qscale[a] = sc;
        if (!wild) mi.qmag = fmaxf(mi.qmag, fmaxf(fabsf(h.lo[a]), fabsf(h.hi[a])) + (h.hi[a] - h.lo[a]));
    }
}

struct Grid {
    float base[3], scale[3];
    uint32_t wild;
};
inline Grid grid_of(const MeshInfo &mi) {
    Grid g;
    for (int a = 0; a < 3; a++) g.base[a] = mi.qbase[a], g.scale[a] = mi.qscale[a];
    g.wild = mi.wild;
    return g;
}
// the largest grid line at or below v / the smallest at or above v
RRT_HD uint32_t q_down(const Grid &g, float v, int a) {
    if (g.wild) return 0u;
    const double base = g.base[a], sc = g.scale[a];
    double q = ::floor(((double)v - base) / sc);
    q = ::fmin(::fmax(q, 0.0), 65535.0);
    while (q > 0.0 && base + q * sc > (double)v) q -= 1.0;
    return (uint32_t)q;
}
RRT_HD uint32_t q_up(const Grid &g, float v, int a) {
    if (g.wild) return 65535u;
    const double base = g.base[a], sc = g.scale[a];
    double q = ::ceil(((double)v - base) / sc);
    q = ::fmin(::fmax(q, 0.0), 65535.0);
    while (q < 65535.0 && base + q * sc < (double)v) q += 1.0;
    return (uint32_t)q;
}

struct Slot {
    float lo[3], hi[3];
    int32_t ref;   // >= 0: BVH2 node index; < 0: leaf, ~ref = first slot
    uint32_t cnt;  // leaf: triangles
};
RRT_HD bool slot_of(const BvhNode &n, int c, bool wild, Slot &s) {
    for (int a = 0; a < 3; a++) {
        const float4 &b = box_axis(n, a);
        s.lo[a] = c == 0 ? b.x : b.y;
        s.hi[a] = c == 0 ? b.z : b.w;
    }
    s.ref = n.child[c];
    s.cnt = n.cnt[c];
    // an empty child (lower face above upper face on some axis; bvh_build.cpp writes them for meshes of <= 4 triangles)
    // takes no slot, unless the mesh is wild and boxes mean nothing
    if (!wild && (s.lo[0] > s.hi[0] || s.lo[1] > s.hi[1] || s.lo[2] > s.hi[2])) return false;
    return true;
}
RRT_HD double area(const Slot &s) {
    const double x = (double)s.hi[0] - s.lo[0], y = (double)s.hi[1] - s.lo[1], z = (double)s.hi[2] - s.lo[2];
    const double A = x * y + y * z + z * x;
    return A == A ? A : (double)INFINITY;  // NaN boxes first: whatever is below them gets its own test
}
// the (up to four) children of the output node that stands for BVH2 node `src`; returns how many
RRT_HD int collapse(const BvhNode *nodes, uint32_t src, bool wild, Slot sl[4]) {
    int m = 0;
    for (int c = 0; c < 2; c++)
        if (slot_of(nodes[src], c, wild, sl[m])) m++;
    for (;;) {
        if (m >= 4) break;
        int pick = -1;
        double best = -1.0;
        for (int k = 0; k < m; k++)
            if (sl[k].ref >= 0 && area(sl[k]) > best) best = area(sl[k]), pick = k;
        if (pick < 0) break;
        const BvhNode &n = nodes[sl[pick].ref];
        Slot a, b;
        const bool ha = slot_of(n, 0, wild, a), hb = slot_of(n, 1, wild, b);
        // the picked slot is replaced in place by the first grandchild, the second goes to the end
        if (ha && hb) {
            sl[pick] = a;
            sl[m++] = b;
        } else if (ha || hb) {
            sl[pick] = ha ? a : b;
        } else {  // both empty: the slot disappears
            sl[pick] = sl[--m];
        }
    }
    return m;
}
// the output node of slots sl[0..m): boxes on the grid, leaf codes; an inner child k gets index first_inner + (its rank
// among the node's inner children)
RRT_HD BvhNodeQ encode(const Grid &g, const Slot sl[4], int m, uint32_t first_inner) {
    BvhNodeQ o;
    for (int k = 0; k < 4; k++) {
        for (int a = 0; a < 3; a++) o.q[4 * a + k] = 65535u | (0u << 16);  // empty: lower face above upper face
        o.child[k] = RRT_BVH_EMPTY;
    }
    uint32_t rank = 0;
    for (int k = 0; k < m; k++) {
        for (int a = 0; a < 3; a++) o.q[4 * a + k] = q_down(g, sl[k].lo[a], a) | (q_up(g, sl[k].hi[a], a) << 16);
        if (sl[k].ref >= 0)
            o.child[k] = (int32_t)(first_inner + rank++);
        else
            o.child[k] = -(int32_t)((((uint32_t)(~sl[k].ref)) << 4) | (sl[k].cnt & 15u)) - 1;
    }
    return o;
}

}  // namespace bq
}  // namespace rrt
