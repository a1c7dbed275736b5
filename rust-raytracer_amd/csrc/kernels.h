// kernels.h — launch interface between the host runtime (runtime.cpp) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "dev_scene.h"

namespace rrt {

struct DevCamera {
    f3 position, right, up, lookat;
    float fov;
    uint32_t width, height;
    // interleaved row groups (rrt_render_interleaved_device): the rows being rendered are the image rows y with
    // (y / row_group) % row_stride == row_first, packed; row_stride <= 1 = a plain band starting at y0
    uint32_t row_group, row_stride, row_first;
};

// photon index in HBM (dev_photon.h)
#define RRT_DENSE_GRIDS 1  // (several grids per map were measured in round 2: no gain for the frames, slower builds; the loops over grids remain)
struct DenseGrid {
    float lo_x, lo_y, lo_z;   // the grid covers [lo, lo + n * cell)
    uint32_t nx, ny, nz;
    uint32_t cell_base;       // index of its first cell in dense_start
    uint32_t pad;
};
struct DevPhotonMap {
    const float4 *pos_cell;  // [n] position.xyz, w unused (cell-sorted: dense cells first, then the outliers' hash cells)
    const float4 *incident;  // [n] normalised-again incident.xyz (photon_scene.rs:405), w unused
    const float4 *intensity; // [n] intensity.rgb
    const uint64_t *keys;    // [table_size] open-addressing hash table of the outliers: cell key or ~0
    const uint2 *ranges;     // [table_size] (first, count), absolute record indices
    uint32_t table_mask;
    uint32_t count;
    float inv_cell;          // hash grid: 1 / (2 * half_width)
    // Dense grids, cell edge = half_width, tried in order: a record lies in the first grid whose box holds it.  Cell
    // (x, y, z) of grid g holds records [dense_start[i], dense_start[i + 1]), i = cell_base + (z * ny + y) * nx + x: one start
    // table over all grids, records sorted by that index.
    const uint32_t *dense_start;  // [cells of all grids + 1]
    DenseGrid grid[RRT_DENSE_GRIDS];
    uint32_t n_grids;
    uint32_t n_dense;        // records [0, n_dense) lie in dense cells, [n_dense, count) are outliers
    float inv_dcell;         // div / half_width
    const uint64_t *meta;    // [n] record keys (cell order): ties between equal squared distances (k-nearest estimator, queries)
    // fine grids (div > 1; dev_photon.h): dense cells of edge half_width / div, and per dense cell the sums of its records
    const float4 *agg_dir;   // [cells] sum of the normalised incidents, w = records of the cell as uint bits
    const float4 *agg_rgb;   // [cells] sum of the intensities
    uint32_t div;
};

// What the index build decides on the device (photon.hip): the build's kernels read it from HBM, the host reads it back once,
// when both maps are done.
struct IndexHeader {
    DenseGrid grid[RRT_DENSE_GRIDS];
    uint32_t n_grids, dense_cells;  // k_choose_dense
    uint32_t n_dense;               // k_dense_starts: records in dense cells (the outliers follow them)
    uint32_t out_cells, table_size; // k_table_size: occupied outlier cells, slots of their hash table (a power of two)
    uint32_t error;                 // 1: more dense records than records (internal error)
    uint32_t trim_steps;            // k_choose_dense: faces moved while trimming the boxes (RRT_INDEX_INFO prints it)
    uint32_t pad;
    uint32_t choose_ticks[4];       // k_choose_dense: 100 MHz ticks spent loading the samples / sorting / trimming / in all (RRT_INDEX_INFO)
    unsigned long long crowding_sum;  // k_finish_header: sum over the dense cells of count^2
};

// error bits of TraceArgs::err (1, 2, 4: dev_shade.h)
#define RRT_ERR_BIT_QUEUE 8u  // a level queue outgrew the size predicted from the previous frame (internal: retried)
#define RRT_ERR_BIT_INTERNAL 16u  // a device-side index guard tripped: the kernel refused an out-of-range index

enum StatSlot { ST_NODES = 0, ST_TRIS, ST_SHADOW, ST_GATHERED, ST_QUERIES, ST_CANDIDATES, ST_TRUNCATED, ST_COUNT };  // ST_TRUNCATED: see rrt_stats
// The device-side counters are kept in RRT_STAT_COPIES copies, each on its own 64-byte line, picked by block index, and
// summed on the host: thousands of waves adding to ONE address serialise in the L2 (measured 4.4 ns per 64-bit atomic,
// 0.14 ms for the 32 k waves of a 1080p level).
#define RRT_STAT_COPIES 64
#define RRT_STAT_STRIDE 8  // slots per copy (ST_COUNT rounded up to a 64-byte line)
#define RRT_STAT_WORDS (RRT_STAT_COPIES * RRT_STAT_STRIDE)

// A photon-mode hit whose two radiance-estimate queries would read more than `heavy_threshold` candidate records is
// not gathered by its lane: the lane files this record and k_gather_heavy gives the query a whole wave.
struct HeavyRec {
    float4 pos_u;    // hit position, u
    float4 nrm_v;    // hit normal, v
    float4 dir_mat;  // ray direction, material index as uint bits
    uint32_t node, has_tex, pad0, pad1;
};
#define RRT_HEAVY_THRESHOLD 2048u       // default for rrt_params.gather_heavy_threshold
#define RRT_HEAVY_COUNTER 32            // lane counters [32 + level] count the records filed by that level

// The queues of the recursion levels >= 1 are STRIPED: RRT_STRIPES dense queues side by side, each with a counter on a
// cache line of its own.  Block b appends its child rays to stripe b % RRT_STRIPES; the next level's launch deals every
// stripe's rays to all XCDs (node_slot in kernels.hip).  One counter for the whole level made every appending wave queue up at ONE address (the
// L2 serialises same-address atomics at about 4.4 ns each: 0.57 ms for the 129 600 one-wave blocks of a 4K level whose
// every pixel reflects, measured as 1.50 ms against 0.75 ms for the same rays without appends); RRT_STRIPES counters take
// that many times fewer each, in parallel.
// Node indices: stripe s of a level holds the nodes [s * stripe_cap, s * stripe_cap + min(counter[s], stripe_cap)).
#ifndef RRT_STRIPES
#define RRT_STRIPES 8u
#endif
#define RRT_STRIPE_STRIDE 16u  // 32-bit words from one stripe's counter to the next (64 bytes)

struct TraceArgs {
    DevScene S;
    LevelBuffers cur, next;
    DevCamera cam;
    DevPhotonMap maps[2];
    uint32_t count;       // level 0: rays in this level.  Striped levels: rays per stripe this launch covers
    const uint32_t *count_ptr;  // striped levels: the level's stripe counters (the previous level's appends wrote them); level 0: null
    uint32_t striped;     // 0: level 0 (nodes [0, count)); 1: a striped level
    uint32_t lane_shift;  // striped trace launches: a block works on 64 >> lane_shift rays (2: sixteen lanes of every wave busy).
                          // Small levels of incoherent rays last as long as their slowest wave, and a wave as long as the longest
                          // of its lanes' walks, phase by phase: fewer rays per wave, shorter waves (set_level_args)
    uint32_t stripe_cap;  // striped levels: nodes per stripe the level's arrays hold
    uint32_t next_stripe_cap;  // nodes per stripe of the next level: appends beyond it raise RRT_ERR_BIT_QUEUE (0: nothing may be appended)
    uint32_t level;       // recursion depth of these rays
    uint32_t max_depth;   // Scene: global_material.rs:72 (5); PhotonScene: max_vueon_depth
    uint32_t mode;        // rrt_mode
    uint32_t y0;          // first image row of the band being rendered
    uint32_t pixel_base;  // first node (band-relative) of this chunk
    uint32_t band_rows;   // rows of the band being rendered (camera renders)
    uint32_t lds_slots;   // hit-list slots per lane (0 = scene has no CSG)
    uint32_t fuse;        // 1: raytracing mode and level 0 is the whole recursion: k_trace_shade writes the frame itself
    uint32_t tile_order;  // 1: level 0 takes the frame's 8x8 tiles in the XCD-interleaved supertile order (node_pixel), which
                          // assumes one-wave blocks
    const uint32_t *tile_table;  // camera frames: place of full tile t of the band, (x / 8) | (y / 8) << 16 (k_tile_table), or null
    float gather_half_width;
    uint32_t *next_count;  // the next level's stripe counters
    uint32_t *err;
    unsigned long long *stats;
    HeavyRec *heavy;           // null: every lane gathers for itself
    uint32_t heavy_capacity;   // records the queue holds (the chunk size: every query of a level may be heavy)
    uint32_t *heavy_count;     // this level's counter
    uint32_t heavy_threshold;  // candidate records (both maps) above which a query is filed
    // PhotonTree::get_n_within_radius as the estimator (rrt_params.gather_k > 0): a heap of knn_k (distance, record) pairs
    // per ray of the launch, entry e of ray i at knn_ws[e * knn_stride + i]
    uint32_t knn_k;
    uint32_t knn_stride;
    uint2 *knn_ws;
    // level-0 outputs of the combine pass (any may be null)
    float *out_rgb;
    float *out_depth;
    uint32_t *out_id;
};

void launch_trace_shade(const TraceArgs &a, size_t lds_bytes, hipStream_t stream);
void launch_combine(const TraceArgs &a, hipStream_t stream);
void launch_tile_table(uint32_t *table, uint32_t n_tiles, uint32_t width, uint32_t rows, uint32_t xcd_order, hipStream_t stream);
void launch_gather_heavy(const TraceArgs &a, hipStream_t stream);  // after launch_trace_shade of the same level (photon mode)

struct BatchTraceArgs {
    DevScene S;
    const float *origins, *directions;  // [n][3]
    const float *limits;                // shadowtrace only
    uint32_t n;
    uint32_t lds_slots;
    uint32_t *object_id, *prim_id;
    float *distance, *position, *normal, *uv;
    uint8_t *occluded;
    uint32_t *err;
    unsigned long long *stats;
};
void launch_batch_trace(const BatchTraceArgs &a, size_t lds_bytes, hipStream_t stream);
void launch_batch_shadow(const BatchTraceArgs &a, size_t lds_bytes, hipStream_t stream);
// rays given explicitly (rrt_raytrace): fills level-0 ray buffers from host-style [n][3] arrays
// multi-GPU frames: a share quantised and packed for the wire, the shares put together on rank 0 (kernels.hip)
void launch_quantise_pack(const float *rgb, const float *depth, uint32_t n, uint8_t *out, hipStream_t stream);
void launch_assemble_frame(const uint8_t *parts, uint32_t world, uint32_t rows_max, uint32_t width, uint32_t height, uint32_t row_group, uint8_t *rgb,
                           uint8_t *depth, hipStream_t stream);
void launch_load_rays(const float *origins, const float *directions, uint32_t n, LevelBuffers lv, hipStream_t stream);

}  // namespace rrt
