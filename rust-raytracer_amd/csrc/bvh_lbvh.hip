// bvh_lbvh.hip — device-side LBVH build (steps 1-6 of bvh_lbvh.h as kernels) + the host finish.
// Selected with rrt_params.bvh_builder = RRT_BVH_DEVICE_LBVH; the default stays the host's binned-SAH builder, whose
// trees traverse faster.  This one is for scenes that are rebuilt often (the reference's watch loop, main.rs:86-99).
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
    DBuf<char> d_temp;
    int rc;
    if ((rc = d_pos.ensure((size_t)n * 9)) || (rc = d_boxes.ensure(n)) || (rc = d_table.ensure((size_t)levels * n)) || (rc = d_bounds.ensure(6)) ||
        (rc = d_idx.ensure(n)) || (rc = d_idx2.ensure(n)) || (rc = d_keys.ensure(n)) || (rc = d_keys2.ensure(n)) || (rc = d_ranges.ensure(n - 1)) ||
        (rc = d_nodes.ensure(n - 1)))
        return rc;
    size_t temp_bytes = 0;
    RRT_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys.p, d_keys2.p, d_idx.p, d_idx2.p, n, 0, 63, nullptr));
    if ((rc = d_temp.ensure(temp_bytes ? temp_bytes : 1))) return rc;
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
        se = hipGetLastError();
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
    std::vector<BvhNode> raw(n - 1);
    order.resize(n);
    RRT_HIP(hipMemcpy(raw.data(), d_nodes.p, (size_t)(n - 1) * sizeof(BvhNode), hipMemcpyDeviceToHost));
    RRT_HIP(hipMemcpy(order.data(), d_idx2.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    lap("download");
    {
        std::vector<uint8_t> seen(n, 0);  // the order must be a permutation of the triangles
        for (uint32_t i = 0; i < n; i++) {
            if (order[i] >= n || seen[order[i]]) return fail(RRT_ERR_DEVICE, "device BVH build: triangle order is not a permutation (internal error)");
            seen[order[i]] = 1;
        }
    }
    lap("permutation check");
    if (const char *what = lbvh::finish(raw, n, nodes, max_depth))
        return fail(RRT_ERR_DEVICE, std::string("device BVH build produced an invalid tree (internal error): ") + what);
    lap("validate + compact (host)");
    return RRT_OK;
}

}  // namespace rrt
