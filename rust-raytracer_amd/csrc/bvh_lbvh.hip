// bvh_lbvh.hip — device-side LBVH build: steps 1-8 of bvh_lbvh.h as kernels (build, validation, depth, compaction).
// Selected with rrt_params.bvh_builder = RRT_BVH_DEVICE_LBVH; the default is the binned-SAH tree (bvh_build.cpp on the
// host, bvh_sah.hip on the device), which traverses faster.  This one is for scenes that are rebuilt often (the reference's watch loop, main.rs:86-99).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>

#include <rocprim/rocprim.hpp>

#include "bvh_lbvh.h"
#include "runtime.h"

namespace rrt {

using lbvh::Box6;
using lbvh::Range;

namespace {

constexpr int kBlock = 256;
inline dim3 grid_for(uint64_t n) { return dim3((uint32_t)((n + kBlock - 1) / kBlock)); }

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, off, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, off, 64));
    return v;
}

// 1. triangle boxes + centroid bounds (6 ordered words: lo xyz start at 0xFFFFFFFF, hi xyz at 0)
__global__ __launch_bounds__(kBlock) void k_lbvh_boxes(const float *positions, uint32_t n, Box6 *boxes, uint32_t *bounds) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    if (i < n) {
        const Box6 b = lbvh::tri_box(positions, i);
        boxes[i] = b;
        for (int k = 0; k < 3; k++) {
            const float c = 0.5f * (b.lo[k] + b.hi[k]);
            if (c == c) lo[k] = hi[k] = lbvh::ordered(c);  // a NaN centroid takes no part in the bounds
        }
    }
    // all lanes of the block get here: whole-wave butterflies are safe.  One atomic per block and word (atomics of many
    // waves on one address serialise in the L2)
    __shared__ uint32_t s_lo[kBlock / 64][3], s_hi[kBlock / 64][3];
    for (int k = 0; k < 3; k++) {
        const uint32_t l = wave_min_u32(lo[k]), h = wave_max_u32(hi[k]);
        if ((threadIdx.x & 63) == 0) {
            s_lo[threadIdx.x >> 6][k] = l;
            s_hi[threadIdx.x >> 6][k] = h;
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        uint32_t l = s_lo[0][k], h = s_hi[0][k];
        for (int w = 1; w < kBlock / 64; w++) {
            l = min(l, s_lo[w][k]);
            h = max(h, s_hi[w][k]);
        }
        atomicMin(&bounds[k], l);
        atomicMax(&bounds[3 + k], h);
    }
}
// 2. keys
__global__ __launch_bounds__(kBlock) void k_lbvh_keys(const Box6 *boxes, uint32_t n, const uint32_t *bounds, uint64_t *keys, uint32_t *idx) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    keys[i] = lbvh::morton63(boxes[i], bounds);
    idx[i] = i;
}
// 5. sparse table
__global__ __launch_bounds__(kBlock) void k_lbvh_level0(const Box6 *boxes, const uint32_t *order, uint32_t n, Box6 *table) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) table[i] = boxes[order[i]];
}
__global__ __launch_bounds__(kBlock) void k_lbvh_level(const Box6 *prev, Box6 *cur, uint32_t n, uint32_t k) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) lbvh::table_level(prev, cur, n, k, i);
}
// 4. hierarchy
__global__ __launch_bounds__(kBlock) void k_lbvh_hierarchy(const uint64_t *keys, uint32_t n, Range *ranges) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i + 1 < n) ranges[i] = lbvh::karras_node(keys, n, i);
}
// 6. nodes
__global__ __launch_bounds__(kBlock) void k_lbvh_emit(const Range *ranges, const Box6 *table, uint32_t n, BvhNode *nodes) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i + 1 < n) nodes[i] = lbvh::emit_node(ranges, table, n, i);
}

// 7. validation (see bvh_lbvh.h)
__global__ __launch_bounds__(kBlock) void k_lbvh_check_order(const uint32_t *order, uint32_t n, uint32_t *seen, uint32_t *err) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) lbvh::check_order(order, n, i, seen, err);
}
__global__ __launch_bounds__(kBlock) void k_lbvh_check_nodes(const Range *ranges, uint32_t n, uint32_t *covered, uint32_t *parent, uint32_t *emitted,
                                                             uint32_t *err) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i + 1 < n) emitted[i] = lbvh::check_node(ranges, n, i, covered, parent, err) ? 1u : 0u;
}
__global__ __launch_bounds__(kBlock) void k_lbvh_check_slots(const uint32_t *covered, uint32_t n, uint32_t *err) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) lbvh::check_slot(covered, i, err);
}
// 8. depth (runs only on a tree that passed 7: its parent links form a tree) and compaction
__global__ __launch_bounds__(kBlock) void k_lbvh_depth(const uint32_t *parent, const uint32_t *emitted, uint32_t n, uint32_t *max_depth, uint32_t *err) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t d = 0;
    if (i + 1 < n && emitted[i]) d = lbvh::node_depth(parent, n, i, err) + 1u;  // the deepest node has a leaf below it: one more
    __shared__ uint32_t s_max[kBlock / 64];
    const uint32_t w = wave_max_u32(d);
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = s_max[0];
        for (int k = 1; k < kBlock / 64; k++) m = max(m, s_max[k]);
        if (m) atomicMax(max_depth, m);
    }
}
__global__ __launch_bounds__(kBlock) void k_lbvh_compact(const BvhNode *raw, const uint32_t *emitted, const uint32_t *new_index, uint32_t n, BvhNode *out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i + 1 < n && emitted[i]) out[new_index[i]] = lbvh::compact_node(raw, new_index, i);
}

}  // namespace

// positions: host [n][3][3].  Outputs as build_bvh (bvh_build.cpp).  n > lbvh::kLeafMax.
int build_bvh_device(const float *positions, uint32_t n, std::vector<uint32_t> &order, std::vector<BvhNode> &nodes, uint32_t &max_depth,
                     float *device_ms) {
    if (n <= lbvh::kLeafMax) return fail(RRT_ERR_INVALID, "device BVH build needs more than 4 triangles");
    if (n >= (1u << 27)) return fail(RRT_ERR_LIMIT, "more than 2^27 triangles in one mesh");  // leaf ids carry 27 bits on the traversal stack
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) return fail(RRT_ERR_DEVICE, "no HIP device: the device BVH builder has no CPU fallback");
    const uint32_t levels = lbvh::table_levels(n);
    auto T0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!getenv("RRT_LBVH_TIMING")) return;
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[lbvh] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - T0).count());
        T0 = t;
    };
    DBuf<float> d_pos;
    DBuf<Box6> d_boxes, d_table;
    DBuf<uint32_t> d_bounds, d_idx, d_idx2;
    DBuf<uint64_t> d_keys, d_keys2;
    DBuf<Range> d_ranges;
    DBuf<BvhNode> d_nodes;
    DBuf<char> d_temp, d_temp2;
    DBuf<uint32_t> d_seen, d_covered, d_parent, d_emitted, d_newidx, d_word;  // validation + compaction; d_word: [0] error bits, [1] depth
    DBuf<BvhNode> d_out;
    int rc;
    if ((rc = d_pos.ensure((size_t)n * 9)) || (rc = d_boxes.ensure(n)) || (rc = d_table.ensure((size_t)levels * n)) || (rc = d_bounds.ensure(6)) ||
        (rc = d_idx.ensure(n)) || (rc = d_idx2.ensure(n)) || (rc = d_keys.ensure(n)) || (rc = d_keys2.ensure(n)) || (rc = d_ranges.ensure(n - 1)) ||
        (rc = d_nodes.ensure(n - 1)) || (rc = d_seen.ensure(n)) || (rc = d_covered.ensure(n)) || (rc = d_parent.ensure(n)) ||
        (rc = d_emitted.ensure(n)) || (rc = d_newidx.ensure(n)) || (rc = d_word.ensure(2)) || (rc = d_out.ensure(n - 1)))
        return rc;
    size_t temp_bytes = 0, scan_bytes = 0;
    RRT_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys.p, d_keys2.p, d_idx.p, d_idx2.p, n, 0, 63, nullptr));
    if ((rc = d_temp.ensure(temp_bytes ? temp_bytes : 1))) return rc;
    RRT_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, d_emitted.p, d_newidx.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), nullptr));
    if ((rc = d_temp2.ensure(scan_bytes ? scan_bytes : 1))) return rc;
    lap("allocations");
    RRT_HIP(hipMemcpy(d_pos.p, positions, (size_t)n * 9 * sizeof(float), hipMemcpyHostToDevice));
    const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
    RRT_HIP(hipMemcpy(d_bounds.p, init, sizeof(init), hipMemcpyHostToDevice));
    lap("upload");
    hipEvent_t e0, e1;
    RRT_HIP(hipEventCreate(&e0));
    RRT_HIP(hipEventCreate(&e1));
    RRT_HIP(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(k_lbvh_boxes, grid_for(n), dim3(kBlock), 0, nullptr, d_pos.p, n, d_boxes.p, d_bounds.p);
    hipLaunchKernelGGL(k_lbvh_keys, grid_for(n), dim3(kBlock), 0, nullptr, d_boxes.p, n, d_bounds.p, d_keys.p, d_idx.p);
    hipError_t se = rocprim::radix_sort_pairs(d_temp.p, temp_bytes, d_keys.p, d_keys2.p, d_idx.p, d_idx2.p, n, 0, 63, nullptr);
    if (se == hipSuccess) {
        hipLaunchKernelGGL(k_lbvh_hierarchy, grid_for(n - 1), dim3(kBlock), 0, nullptr, d_keys2.p, n, d_ranges.p);
        hipLaunchKernelGGL(k_lbvh_level0, grid_for(n), dim3(kBlock), 0, nullptr, d_boxes.p, d_idx2.p, n, d_table.p);
        for (uint32_t k = 1; k < levels; k++)
            hipLaunchKernelGGL(k_lbvh_level, grid_for(n), dim3(kBlock), 0, nullptr, d_table.p + (size_t)(k - 1) * n, d_table.p + (size_t)k * n, n, k);
        hipLaunchKernelGGL(k_lbvh_emit, grid_for(n - 1), dim3(kBlock), 0, nullptr, d_ranges.p, d_table.p, n, d_nodes.p);
        // 7. validation; 8. depth + compaction.  (emitted[n - 1] stays 0: the scan runs over n entries so that
        // new_index[n - 1] is the number of emitted nodes)
        (void)hipMemsetAsync(d_seen.p, 0, (size_t)n * sizeof(uint32_t), nullptr);
        (void)hipMemsetAsync(d_covered.p, 0, (size_t)n * sizeof(uint32_t), nullptr);
        (void)hipMemsetAsync(d_parent.p, 0xFF, (size_t)n * sizeof(uint32_t), nullptr);
        (void)hipMemsetAsync(d_emitted.p, 0, (size_t)n * sizeof(uint32_t), nullptr);
        (void)hipMemsetAsync(d_word.p, 0, 2 * sizeof(uint32_t), nullptr);
        hipLaunchKernelGGL(k_lbvh_check_order, grid_for(n), dim3(kBlock), 0, nullptr, d_idx2.p, n, d_seen.p, d_word.p);
        hipLaunchKernelGGL(k_lbvh_check_nodes, grid_for(n - 1), dim3(kBlock), 0, nullptr, d_ranges.p, n, d_covered.p, d_parent.p, d_emitted.p, d_word.p);
        hipLaunchKernelGGL(k_lbvh_check_slots, grid_for(n), dim3(kBlock), 0, nullptr, d_covered.p, n, d_word.p);
        se = hipGetLastError();
    }
    uint32_t word[2] = {0, 0};
    if (se == hipSuccess) se = hipMemcpy(word, d_word.p, sizeof(word), hipMemcpyDeviceToHost);  // the verdict, before anything walks the tree
    uint32_t n_emitted = 0;
    if (se == hipSuccess && word[0] == 0) {
        hipLaunchKernelGGL(k_lbvh_depth, grid_for(n - 1), dim3(kBlock), 0, nullptr, d_parent.p, d_emitted.p, n, d_word.p + 1, d_word.p);
        se = rocprim::exclusive_scan(d_temp2.p, scan_bytes, d_emitted.p, d_newidx.p, 0u, (size_t)n, rocprim::plus<uint32_t>(), nullptr);
        if (se == hipSuccess) {
            hipLaunchKernelGGL(k_lbvh_compact, grid_for(n - 1), dim3(kBlock), 0, nullptr, d_nodes.p, d_emitted.p, d_newidx.p, n, d_out.p);
            se = hipGetLastError();
        }
        if (se == hipSuccess) se = hipMemcpy(word, d_word.p, sizeof(word), hipMemcpyDeviceToHost);
        if (se == hipSuccess) se = hipMemcpy(&n_emitted, d_newidx.p + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost);
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
    if (getenv("RRT_LBVH_TIMING")) fprintf(stderr, "[lbvh] kernels (HIP events)         %8.2f ms\n", ms);
    if (word[0] || n_emitted == 0 || n_emitted > n - 1 || word[1] < 2) {
        char what[160];
        snprintf(what, sizeof(what), "device BVH build produced an invalid tree (internal error): check bits 0x%x, %u nodes, depth %u", word[0], n_emitted,
                 word[1]);
        return fail(RRT_ERR_DEVICE, what);
    }
    nodes.resize(n_emitted);
    order.resize(n);
    RRT_HIP(hipMemcpy(nodes.data(), d_out.p, (size_t)n_emitted * sizeof(BvhNode), hipMemcpyDeviceToHost));
    RRT_HIP(hipMemcpy(order.data(), d_idx2.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    max_depth = word[1];
    lap("download (validated, compacted)");
    return RRT_OK;
}

}  // namespace rrt

// rrt_scene_create touches one kernel of every translation unit, so that the code objects are loaded before the first
// rrt_pre_render is timed (HIP loads them on first use: tens of milliseconds per unit)
namespace rrt {
void preload_bvh_lbvh() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_lbvh_boxes));
}
}  // namespace rrt
