// lbvh_check.cpp — CPU pre-flight of the device LBVH build: runs the per-thread functions of bvh_lbvh.h (the bodies of
// the kernels in bvh_lbvh.hip) serially under AddressSanitizer, then checks the result exhaustively:
//   * lbvh::finish accepts the tree (every node reached once, every triangle slot in exactly one leaf, indices in range);
//   * every leaf box contains its triangles and every child box contains the boxes below it (so traversal can only
//     prune what it may prune), against boxes recomputed here from the positions;
//   * the compacted tree's depth is what finish reports;
//   * the device's own validation (check_order / check_node / check_slot), depth (node_depth) and compaction
//     (compact_node) accept the same tree, report the same depth and node count, give the same tree node for node, and
//     refuse damaged ranges and orders.
// Inputs: random soups, degenerate sets (all triangles identical, coplanar grids, NaN vertices, huge coordinates), sizes
// 5 .. 200 000, and optionally an OBJ-free raw float file of positions.   usage: lbvh_check [positions.f32]
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <random>
#include <vector>

#include "bvh_lbvh.h"

using namespace rrt;
using lbvh::Box6;
using lbvh::Range;

static bool contains(const float *lo, const float *hi, const Box6 &b) {
    for (int k = 0; k < 3; k++) {
        if (b.lo[k] == b.lo[k] && !(lo[k] <= b.lo[k])) return false;  // NaN coordinates are nobody's to contain
        if (b.hi[k] == b.hi[k] && !(hi[k] >= b.hi[k])) return false;
    }
    return true;
}

struct Sub {
    Box6 box;
    uint32_t depth;
};

static int check(const char *name, const std::vector<float> &pos) {
    const uint32_t n = (uint32_t)(pos.size() / 9);
    // --- the kernels, thread by thread -------------------------------------------------------------
    std::vector<Box6> boxes(n);
    uint32_t bounds[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    for (uint32_t i = 0; i < n; i++) {
        boxes[i] = lbvh::tri_box(pos.data(), i);
        for (int k = 0; k < 3; k++) {
            float c = 0.5f * (boxes[i].lo[k] + boxes[i].hi[k]);
            if (c == c) {
                bounds[k] = std::min(bounds[k], lbvh::ordered(c));
                bounds[3 + k] = std::max(bounds[3 + k], lbvh::ordered(c));
            }
        }
    }
    std::vector<std::pair<uint64_t, uint32_t>> kv(n);
    for (uint32_t i = 0; i < n; i++) kv[i] = {lbvh::morton63(boxes[i], bounds), i};
    std::stable_sort(kv.begin(), kv.end(), [](auto &a, auto &b) { return a.first < b.first; });  // what a radix sort does
    std::vector<uint64_t> keys(n);
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) {
        keys[i] = kv[i].first;
        order[i] = kv[i].second;
    }
    std::vector<Range> ranges(n - 1);
    for (uint32_t i = 0; i + 1 < n; i++) ranges[i] = lbvh::karras_node(keys.data(), n, i);
    const uint32_t levels = lbvh::table_levels(n);
    std::vector<Box6> table((size_t)levels * n);  // exactly the device allocation: ASan sees any access beyond it
    for (uint32_t i = 0; i < n; i++) table[i] = boxes[order[i]];
    for (uint32_t k = 1; k < levels; k++)
        for (uint32_t i = 0; i < n; i++) lbvh::table_level(table.data() + (size_t)(k - 1) * n, table.data() + (size_t)k * n, n, k, i);
    std::vector<BvhNode> raw(n - 1);
    for (uint32_t i = 0; i + 1 < n; i++) raw[i] = lbvh::emit_node(ranges.data(), table.data(), n, i);
    // --- host finish ------------------------------------------------------------------------------------
    std::vector<BvhNode> nodes;
    uint32_t depth = 0;
    if (const char *what = lbvh::finish(raw, n, nodes, depth)) {
        printf("%-28s n=%u: finish refused the tree: %s\n", name, n, what);
        return 1;
    }
    // --- containment, bottom-up over the compacted tree (children follow parents in DFS order) -----------
    std::vector<Sub> sub(nodes.size());
    uint32_t max_depth = 0;
    uint64_t leaf_tris = 0;
    for (size_t k = nodes.size(); k-- > 0;) {
        const BvhNode &nd = nodes[k];
        const float *X = &nd.lo_x_hi_x.x, *Y = &nd.lo_y_hi_y.x, *Z = &nd.lo_z_hi_z.x;
        Box6 total;
        uint32_t d = 0;
        for (int c = 0; c < 2; c++) {
            const float lo[3] = {X[c], Y[c], Z[c]}, hi[3] = {X[2 + c], Y[2 + c], Z[2 + c]};
            Box6 below;
            uint32_t dc;
            if (nd.child[c] >= 0) {
                if ((size_t)nd.child[c] <= k) { printf("%s: child before parent\n", name); return 1; }
                below = sub[nd.child[c]].box;
                dc = sub[nd.child[c]].depth + 1;
            } else {
                uint32_t first = (uint32_t)~nd.child[c];
                below = boxes[order[first]];
                for (uint32_t t = 1; t < nd.cnt[c]; t++) below = lbvh::box_union(below, boxes[order[first + t]]);
                leaf_tris += nd.cnt[c];
                dc = 2;
            }
            if (!contains(lo, hi, below)) { printf("%-28s n=%u: node %zu child %d does not contain what is below it\n", name, n, k, c); return 1; }
            total = c == 0 ? below : lbvh::box_union(total, below);
            d = std::max(d, dc);
        }
        sub[k] = {total, d};
        max_depth = std::max(max_depth, d);
    }
    if (leaf_tris != n || sub[0].depth != depth) {
        printf("%-28s n=%u: leaves hold %llu triangles, depth %u vs reported %u\n", name, n, (unsigned long long)leaf_tris, sub[0].depth, depth);
        return 1;
    }
    // --- the device's own validation, depth and compaction (steps 7 and 8), thread by thread, against finish ---------
    {
        std::vector<uint32_t> seen(n, 0), covered(n, 0), parent(n, 0xFFFFFFFFu), emitted(n, 0), new_index(n, 0);
        uint32_t err = 0, dmax = 0;
        for (uint32_t i = 0; i < n; i++) lbvh::check_order(order.data(), n, i, seen.data(), &err);
        for (uint32_t i = 0; i + 1 < n; i++) emitted[i] = lbvh::check_node(ranges.data(), n, i, covered.data(), parent.data(), &err) ? 1u : 0u;
        for (uint32_t i = 0; i < n; i++) lbvh::check_slot(covered.data(), i, &err);
        if (err) { printf("%-28s n=%u: device validation refused a tree finish accepted (bits 0x%x)\n", name, n, err); return 1; }
        for (uint32_t i = 0; i + 1 < n; i++)
            if (emitted[i]) dmax = std::max(dmax, lbvh::node_depth(parent.data(), n, i, &err) + 1u);
        uint32_t total = 0;
        for (uint32_t i = 0; i < n; i++) { new_index[i] = total; total += emitted[i]; }
        if (err || dmax != depth || total != nodes.size()) {
            printf("%-28s n=%u: device finish disagrees: err 0x%x depth %u vs %u, %u vs %zu nodes\n", name, n, err, dmax, depth, total, nodes.size());
            return 1;
        }
        std::vector<BvhNode> packed(total);
        for (uint32_t i = 0; i + 1 < n; i++)
            if (emitted[i]) packed[new_index[i]] = lbvh::compact_node(raw.data(), new_index.data(), i);
        // same tree as finish's DFS-ordered one: walk both in lockstep
        std::vector<std::pair<uint32_t, uint32_t>> stack{{0u, 0u}};
        size_t visited = 0;
        while (!stack.empty()) {
            auto [a, b] = stack.back();
            stack.pop_back();
            visited++;
            const BvhNode &x = nodes[a], &y = packed[b];
            if (memcmp(&x.lo_x_hi_x, &y.lo_x_hi_x, 3 * sizeof(float4)) != 0 || x.cnt[0] != y.cnt[0] || x.cnt[1] != y.cnt[1]) { printf("%s: compacted node differs\n", name); return 1; }
            for (int c = 0; c < 2; c++) {
                if ((x.child[c] >= 0) != (y.child[c] >= 0)) { printf("%s: compacted child kind differs\n", name); return 1; }
                if (x.child[c] >= 0) stack.push_back({(uint32_t)x.child[c], (uint32_t)y.child[c]});
                else if (x.child[c] != y.child[c]) { printf("%s: compacted leaf differs\n", name); return 1; }
            }
        }
        if (visited != nodes.size()) { printf("%s: compacted tree has another size\n", name); return 1; }
        // negative controls: one damaged range / order entry must be refused
        if (n > 64) {
            auto refused = [&](std::vector<Range> rg, std::vector<uint32_t> ord) {
                std::vector<uint32_t> sn(n, 0), cv(n, 0), pr(n, 0xFFFFFFFFu);
                uint32_t e = 0;
                for (uint32_t i = 0; i < n; i++) lbvh::check_order(ord.data(), n, i, sn.data(), &e);
                for (uint32_t i = 0; i + 1 < n; i++) lbvh::check_node(rg.data(), n, i, cv.data(), pr.data(), &e);
                for (uint32_t i = 0; i < n; i++) lbvh::check_slot(cv.data(), i, &e);
                return e != 0;
            };
            std::vector<Range> r1 = ranges, r2 = ranges, r3 = ranges;
            std::vector<uint32_t> o1 = order;
            uint32_t big = 0;  // some emitted inner node other than the root
            for (uint32_t i = 1; i + 1 < n; i++)
                if (lbvh::is_emitted(ranges[i], i) && ranges[i].last - ranges[i].first > 12) { big = i; break; }
            r1[0].last = n - 2;
            r2[big].split = r2[big].last;
            r3[big].first += 1;
            o1[3] = o1[4];
            if (!refused(r1, order) || !refused(r2, order) || !refused(r3, order) || !refused(ranges, o1) || refused(ranges, order)) {
                printf("%-28s n=%u: device validation let a damaged tree through\n", name, n);
                return 1;
            }
        }
    }
    printf("%-28s n=%7u: ok, %zu nodes, depth %u\n", name, n, nodes.size(), depth);
    return 0;
}

int main(int argc, char **argv) {
    int bad = 0;
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-1.0f, 1.0f);
    auto soup = [&](uint32_t n, float spread, float size) {
        std::vector<float> p((size_t)n * 9);
        for (uint32_t i = 0; i < n; i++) {
            float c[3] = {U(rng) * spread, U(rng) * spread, U(rng) * spread};
            for (int v = 0; v < 9; v++) p[(size_t)i * 9 + v] = c[v % 3] + U(rng) * size;
        }
        return p;
    };
    for (uint32_t n : {5u, 6u, 7u, 8u, 9u, 16u, 17u, 63u, 64u, 65u, 1000u, 4097u, 200000u}) bad += check("random soup", soup(n, 10.0f, 0.5f));
    for (uint32_t n : {5u, 33u, 5000u}) {
        std::vector<float> p = soup(1, 1.0f, 1.0f);
        std::vector<float> all;
        for (uint32_t i = 0; i < n; i++) all.insert(all.end(), p.begin(), p.end());
        bad += check("identical triangles", all);  // every key equal: the tie-break by position carries the whole tree
    }
    {
        std::vector<float> p;
        for (int y = 0; y < 60; y++)
            for (int x = 0; x < 60; x++) {
                float q[9] = {(float)x, 0, (float)y, (float)x + 1, 0, (float)y, (float)x, 0, (float)y + 1};
                p.insert(p.end(), q, q + 9);
            }
        bad += check("coplanar grid (one flat axis)", p);
    }
    {
        std::vector<float> p = soup(3000, 5.0f, 0.2f);
        for (int k = 0; k < 40; k++) p[(size_t)(rng() % 3000) * 9 + rng() % 9] = NAN;
        for (int k = 0; k < 10; k++) p[(size_t)(rng() % 3000) * 9 + rng() % 9] = INFINITY;
        bad += check("NaN and inf vertices", p);
    }
    bad += check("huge coordinates", soup(2000, 1e30f, 1e28f));
    bad += check("tiny extent", soup(2000, 1e-30f, 1e-32f));
    {
        std::vector<float> p = soup(20000, 100.0f, 0.01f);  // clusters: many equal high bits
        for (size_t i = 0; i < p.size(); i++) p[i] = floorf(p[i] / 25.0f) * 25.0f + (p[i] - floorf(p[i])) * 1e-3f;
        bad += check("tight clusters", p);
    }
    if (argc > 1) {
        FILE *f = fopen(argv[1], "rb");
        if (!f) { printf("cannot open %s\n", argv[1]); return 2; }
        fseek(f, 0, SEEK_END);
        long bytes = ftell(f);
        fseek(f, 0, SEEK_SET);
        std::vector<float> p((size_t)bytes / 36 * 9);
        if (fread(p.data(), 4, p.size(), f) != p.size()) return 2;
        fclose(f);
        bad += check(argv[1], p);
    }
    printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad ? 1 : 0;
}
