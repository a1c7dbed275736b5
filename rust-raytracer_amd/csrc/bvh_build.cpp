// bvh_build.cpp — host-side binned-SAH BVH2 over a mesh's triangles, emitted in the 64-B
// two-boxes-per-node layout the traversal kernel reads (dev_scene.h: BvhNode).
//
// The BVH replaces the reference's O(N) loop over all triangles (polymesh_object.rs:219-236); it
// must only ever prune, so leaf boxes are padded by 2^-16 of their coordinate magnitude (the
// reference's same-side test is evaluated at a rounded hit point that may sit a few ulps outside
// the exact triangle).
#include <math.h>
#include <cmath>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "bvh_quant.h"
#include "dev_scene.h"

namespace rrt {

namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int a = 0; a < 3; a++) {
            lo[a] = INFINITY;
            hi[a] = -INFINITY;
        }
    }
    void grow(const Box &b) {
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], b.lo[a]);
            hi[a] = fmaxf(hi[a], b.hi[a]);
        }
    }
    void grow_pt(const float *p) {
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], p[a]);
            hi[a] = fmaxf(hi[a], p[a]);
        }
    }
    float half_area() const {
        float e[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (e[0] < 0 || e[1] < 0 || e[2] < 0) return 0.0f;
        return e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
    }
};

Box padded(const Box &b) {
    Box r = b;
    for (int a = 0; a < 3; a++) {
        float mag = fmaxf(fmaxf(fabsf(b.lo[a]), fabsf(b.hi[a])), b.hi[a] - b.lo[a]);
        float pad = mag * 1.52587890625e-5f + 1e-30f;  // 2^-16
        r.lo[a] = b.lo[a] - pad;
        r.hi[a] = b.hi[a] + pad;
    }
    return r;
}

struct TmpNode {
    Box box;
    int left = -1, right = -1;  // children (TmpNode index) or -1 for leaf
    uint32_t first = 0, count = 0;
};

constexpr uint32_t kLeafMax = 4;
constexpr int kBins = 16;

}  // namespace

// positions: [n][3][3].  Outputs: order[n] (BVH slot -> original triangle), nodes, max depth.
void build_bvh(const float *positions, uint32_t n, std::vector<uint32_t> &order, std::vector<BvhNode> &nodes, uint32_t &max_depth) {
    order.resize(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::vector<Box> tb(n);
    std::vector<float> cen((size_t)n * 3);
    for (uint32_t i = 0; i < n; i++) {
        tb[i].reset();
        for (int c = 0; c < 3; c++) tb[i].grow_pt(positions + (size_t)i * 9 + c * 3);
        for (int a = 0; a < 3; a++) cen[(size_t)i * 3 + a] = 0.5f * (tb[i].lo[a] + tb[i].hi[a]);
    }
    struct Work {
        int node;
        uint32_t lo, hi, depth;
    };
    auto range_box = [&](uint32_t lo, uint32_t hi) {
        Box b;
        b.reset();
        for (uint32_t i = lo; i < hi; i++) b.grow(tb[order[i]]);
        return b;
    };
    // One split: either makes pool[w.node] a leaf or partitions order[w.lo, w.hi) (nobody else's part of `order`), appends
    // the two children to `pool` and returns their work items.
    auto split = [&](std::vector<TmpNode> &pool, const Work &w, Work out[2]) -> bool {
        uint32_t cnt = w.hi - w.lo;
        if (cnt <= kLeafMax) {
            pool[w.node].first = w.lo;
            pool[w.node].count = cnt;
            return false;
        }
        // centroid bounds
        Box cb;
        cb.reset();
        for (uint32_t i = w.lo; i < w.hi; i++) cb.grow_pt(&cen[(size_t)order[i] * 3]);
        int best_axis = -1, best_split = -1;
        float best_cost = INFINITY;
        for (int a = 0; a < 3; a++) {
            float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.0f)) continue;
            Box bins[kBins];
            uint32_t counts[kBins] = {0};
            for (int b = 0; b < kBins; b++) bins[b].reset();
            float scale = (float)kBins / ext;
            for (uint32_t i = w.lo; i < w.hi; i++) {
                uint32_t t = order[i];
                int b = (int)((cen[(size_t)t * 3 + a] - cb.lo[a]) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                bins[b].grow(tb[t]);
                counts[b]++;
            }
            float right_area[kBins];
            uint32_t right_cnt[kBins];
            Box acc;
            acc.reset();
            uint32_t c = 0;
            for (int b = kBins - 1; b > 0; b--) {
                acc.grow(bins[b]);
                c += counts[b];
                right_area[b] = acc.half_area();
                right_cnt[b] = c;
            }
            acc.reset();
            c = 0;
            for (int b = 0; b < kBins - 1; b++) {
                acc.grow(bins[b]);
                c += counts[b];
                if (c == 0 || right_cnt[b + 1] == 0) continue;
                float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = a;
                    best_split = b;
                }
            }
        }
        uint32_t mid;
        if (best_axis < 0) {
            mid = w.lo + cnt / 2;  // all centroids coincide: split by index
        } else {
            float ext = cb.hi[best_axis] - cb.lo[best_axis];
            float scale = (float)kBins / ext;
            float lo_c = cb.lo[best_axis];
            auto it = std::partition(order.begin() + w.lo, order.begin() + w.hi, [&](uint32_t t) {
                int b = (int)((cen[(size_t)t * 3 + best_axis] - lo_c) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                return b <= best_split;
            });
            mid = (uint32_t)(it - order.begin());
            if (mid == w.lo || mid == w.hi) mid = w.lo + cnt / 2;
        }
        int l = (int)pool.size();
        pool.push_back(TmpNode());
        pool.push_back(TmpNode());
        pool[l].box = range_box(w.lo, mid);
        pool[l + 1].box = range_box(mid, w.hi);
        pool[w.node].left = l;
        pool[w.node].right = l + 1;
        out[0] = {l, w.lo, mid, w.depth + 1};
        out[1] = {l + 1, mid, w.hi, w.depth + 1};
        return true;
    };
    std::vector<TmpNode> tmp;
    tmp.reserve(n ? (size_t)n + 16 : 16);
    tmp.push_back(TmpNode());
    tmp[0].box = range_box(0, n);
    max_depth = 1;
    // The top of the tree serially, down to sub-ranges of at most n / 16 triangles; those are independent jobs (disjoint
    // parts of `order`, node pools of their own) and run on host threads.  The tree is the one a serial build makes.
    const uint32_t job_size = std::max<uint32_t>(n / 16, 4096);
    std::vector<Work> stack, jobs;
    stack.push_back({0, 0, n, 1});
    while (!stack.empty()) {
        Work w = stack.back();
        stack.pop_back();
        max_depth = std::max(max_depth, w.depth);
        if (w.hi - w.lo <= job_size && w.hi - w.lo > kLeafMax && n > 2 * job_size) {
            jobs.push_back(w);
            continue;
        }
        Work out[2];
        if (split(tmp, w, out)) {
            stack.push_back(out[0]);
            stack.push_back(out[1]);
        }
    }
    if (!jobs.empty()) {
        std::vector<std::vector<TmpNode>> pools(jobs.size());
        std::vector<uint32_t> depths(jobs.size(), 1);
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= jobs.size()) return;
                std::vector<TmpNode> &pool = pools[k];
                pool.reserve((size_t)(jobs[k].hi - jobs[k].lo) + 16);
                pool.push_back(tmp[jobs[k].node]);  // local index 0 = the job's root
                std::vector<Work> st;
                st.push_back({0, jobs[k].lo, jobs[k].hi, jobs[k].depth});
                while (!st.empty()) {
                    Work w = st.back();
                    st.pop_back();
                    depths[k] = std::max(depths[k], w.depth);
                    Work out[2];
                    if (split(pool, w, out)) {
                        st.push_back(out[0]);
                        st.push_back(out[1]);
                    }
                }
            }
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t n_threads = std::min<size_t>(std::min<size_t>(hw ? hw : 4, 16), jobs.size());
        std::vector<std::thread> threads;
        for (size_t t = 1; t < n_threads; t++) threads.emplace_back(worker);
        worker();
        for (auto &t : threads) t.join();
        for (size_t k = 0; k < jobs.size(); k++) {  // splice the pools in: local index i > 0 becomes offset + i - 1
            const std::vector<TmpNode> &pool = pools[k];
            const int offset = (int)tmp.size();
            auto fix = [&](TmpNode t) {
                if (t.left >= 0) t.left += offset - 1, t.right += offset - 1;
                return t;
            };
            tmp[jobs[k].node] = fix(pool[0]);
            for (size_t i = 1; i < pool.size(); i++) tmp.push_back(fix(pool[i]));
            max_depth = std::max(max_depth, depths[k]);
        }
    }
    // leaves keep ascending original order inside a leaf (cheap tie handling, better locality)
    for (const TmpNode &t : tmp)
        if (t.left < 0 && t.count > 1) std::sort(order.begin() + t.first, order.begin() + t.first + t.count);

    // emit: inner TmpNodes become BvhNodes; root is always an inner node
    nodes.clear();
    std::vector<int> emit_index(tmp.size(), -1);
    // assign indices in DFS order so that near subtrees are close in memory
    std::vector<int> dfs;
    dfs.push_back(0);
    std::vector<int> inner_order;
    if (tmp[0].left < 0) {
        // a mesh with <= kLeafMax triangles: synthesise a root whose right child is empty
        BvhNode nd;
        Box b = padded(tmp[0].box);
        Box e;
        e.reset();
        nd.lo_x_hi_x = make_float4(b.lo[0], e.lo[0], b.hi[0], e.hi[0]);
        nd.lo_y_hi_y = make_float4(b.lo[1], e.lo[1], b.hi[1], e.hi[1]);
        nd.lo_z_hi_z = make_float4(b.lo[2], e.lo[2], b.hi[2], e.hi[2]);
        nd.child[0] = ~(int32_t)tmp[0].first;
        nd.child[1] = ~0;
        nd.cnt[0] = tmp[0].count;
        nd.cnt[1] = 0;
        nodes.push_back(nd);
        max_depth = 2;
        return;
    }
    while (!dfs.empty()) {
        int t = dfs.back();
        dfs.pop_back();
        if (tmp[t].left < 0) continue;
        emit_index[t] = (int)inner_order.size();
        inner_order.push_back(t);
        dfs.push_back(tmp[t].right);
        dfs.push_back(tmp[t].left);
    }
    nodes.resize(inner_order.size());
    for (size_t k = 0; k < inner_order.size(); k++) {
        const TmpNode &t = tmp[inner_order[k]];
        const TmpNode &L = tmp[t.left], &R = tmp[t.right];
        Box bl = padded(L.box), br = padded(R.box);
        BvhNode nd;
        nd.lo_x_hi_x = make_float4(bl.lo[0], br.lo[0], bl.hi[0], br.hi[0]);
        nd.lo_y_hi_y = make_float4(bl.lo[1], br.lo[1], bl.hi[1], br.hi[1]);
        nd.lo_z_hi_z = make_float4(bl.lo[2], br.lo[2], bl.hi[2], br.hi[2]);
        nd.child[0] = L.left >= 0 ? emit_index[t.left] : ~(int32_t)L.first;
        nd.child[1] = R.left >= 0 ? emit_index[t.right] : ~(int32_t)R.first;
        nd.cnt[0] = L.left >= 0 ? 0 : L.count;
        nd.cnt[1] = R.left >= 0 ? 0 : R.count;
        nodes[k] = nd;
    }
}

// The traversal's node format from a builder's binary tree: the mesh's 16-bit grid, the four-wide collapse and the outward
// rounding are bvh_quant.h's (shared with the device version in bvh_sah.hip, which makes the same bytes); this is the
// serial, breadth-first driver for trees that are on the host.
bq::Hull hull_of(const std::vector<BvhNode> &nodes) {
    const float lim = 1152921504606846976.0f;  // 2^60
    bq::Hull h;
    for (int a = 0; a < 3; a++) h.lo[a] = INFINITY, h.hi[a] = -INFINITY;
    h.wild = nodes.empty();
    for (const BvhNode &n : nodes)
        for (int a = 0; a < 3; a++) {
            const float4 &b = bq::box_axis(n, a);
            const float v[4] = {b.x, b.y, b.z, b.w};  // l.lo, r.lo, l.hi, r.hi
            for (int c = 0; c < 2; c++) {
                const float l = v[c], hh = v[2 + c];
                if (l > hh) continue;  // empty child (bvh_build.cpp: a mesh of <= 4 triangles) or NaN
                if (!(std::fabs(l) < lim) || !(std::fabs(hh) < lim)) h.wild = true;
                h.lo[a] = std::min(h.lo[a], l);
                h.hi[a] = std::max(h.hi[a], hh);
            }
            if (std::isnan(b.x) || std::isnan(b.y) || std::isnan(b.z) || std::isnan(b.w)) h.wild = true;
        }
    return h;
}

void quantise_bvh(const std::vector<BvhNode> &nodes, MeshInfo &mi, std::vector<BvhNodeQ> &out, uint32_t &stack_need) {
    bq::make_grid(hull_of(nodes), mi);
    const bq::Grid grid = bq::grid_of(mi);
    const bool wild = mi.wild != 0;
    out.clear();
    stack_need = 0;
    if (nodes.empty()) return;
    // breadth first: out[k] stands for BVH2 node src[k]; need[k] = stack entries in use when the traversal arrives at it
    std::vector<uint32_t> src, need;
    src.push_back(0);
    need.push_back(0);
    out.emplace_back();
    for (size_t k = 0; k < src.size(); k++) {
        bq::Slot sl[4];
        const int m = bq::collapse(nodes.data(), src[k], wild, sl);
        // stack entries while below this node: what was there on arrival + its other children
        const uint32_t below = need[k] + (m > 0 ? (uint32_t)m - 1u : 0u);
        stack_need = std::max(stack_need, below);
        out[k] = bq::encode(grid, sl, m, (uint32_t)src.size());
        for (int c = 0; c < m; c++)
            if (sl[c].ref >= 0) {
                src.push_back((uint32_t)sl[c].ref);
                need.push_back(below);
                out.emplace_back();
            }
    }
    stack_need += 1;  // the count above is of entries below the current node; one word of slack
}

}  // namespace rrt
