// runtime.h — host runtime of librrt: the scene handle, device storage and queue workspace.
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <string>
#include <mutex>
#include <vector>

#include "host_scene.h"
#include "kernels.h"

namespace rrt {

const char *last_error_cstr();

#define RRT_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return ::rrt::fail(RRT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

// grow-only device array
template <class T>
struct DBuf {
    T *p = nullptr;
    size_t cap = 0;
    ~DBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    int ensure(size_t n) {
        if (n <= cap) return RRT_OK;
        release();
        if (n == 0) return RRT_OK;
        RRT_HIP(hipMalloc((void **)&p, n * sizeof(T)));
        cap = n;
        return RRT_OK;
    }
    int upload(const std::vector<T> &v) {
        int rc = ensure(v.size() ? v.size() : 1);
        if (rc) return rc;
        if (!v.empty()) RRT_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
        return RRT_OK;
    }
};

struct LevelStorage {
    DBuf<float4> ray_o, ray_d, once, glob, lights, result;
    DBuf<int2> child;
    DBuf<uint32_t> object_id;
    uint32_t capacity = 0;
    int ensure(uint32_t n, uint32_t n_lights, bool need_rays);
    LevelBuffers view() const;
};

struct DeviceScene {
    DBuf<ProgRec> prog;
    DBuf<PlaneCold> plane_cold;
    DBuf<MeshInfo> meshes;
    DBuf<TriHot> tris;
    DBuf<uint32_t> tri_index;
    DBuf<float> tri_normals;
    DBuf<BvhNodeQ> nodes;
    DBuf<uint32_t> mesh_inf;
    DBuf<DevMaterial> materials;
    DBuf<DevTexture> textures;
    DBuf<uchar4> texels8;
    DBuf<ushort4> texels16;
    DBuf<DevLight> lights;
    DevScene view{};
    uint32_t lds_slots = 0;
    uint32_t bvh_stack = 0;
    size_t lds_bytes = 0;
    bool reflective = false;
    uint64_t flops_per_trace = 0;  // analytic primitives only
};

// photon buffers in HBM (photon.hip)
struct PhotonBuffers {
    DBuf<float4> pos, dir, rgb;  // pos.w = photon type as uint bits
    DBuf<uint64_t> meta;         // (light*2+pass) << 56 | key
    uint64_t count = 0;
};
struct PhotonIndex {
    DBuf<float4> pos, dir, rgb;
    DBuf<uint64_t> meta;  // record keys in cell order (the query entry points and the k-nearest estimator break ties by them)
    DBuf<uint64_t> keys;     // outliers' hash table
    DBuf<uint2> ranges;
    DBuf<uint32_t> dense_start;  // dense grid: first record of every cell (+ one past the last)
    DBuf<float4> agg_dir, agg_rgb;  // fine grids (div > 1): per dense cell the sums of its records
    uint32_t div = 1;            // dense cell edge = gather half width / div
    uint32_t table_size = 0;
    uint64_t count = 0;
    float inv_cell = 0.0f;   // hash grid, cell edge 2r
    float inv_dcell = 0.0f;  // dense grid, cell edge r
    DenseGrid grids[RRT_DENSE_GRIDS] = {};  // dense grids in the order records are tried against them
    uint32_t n_grids = 0, dense_cells = 0;
    uint32_t n_dense = 0;
    double crowding = 0.0;  // sum over cells of count^2 / records: the occupancy of the cell a random photon lies in
    DevPhotonMap view() const;
};
// Work arrays of the sorts and the index build, kept between calls (grow-only): hipMalloc / hipFree cost more than the
// sorts themselves at a few million records.
struct PhotonScratch {
    DBuf<uint32_t> idx, idx2, ncell;
    DBuf<uint64_t> keys, keys2;
    DBuf<float4> t;
    DBuf<uint8_t> sort_temp;
    PhotonBuffers stage[2];    // segmented staging of one rrt_photon_shoot call (photon.hip)
    DBuf<uint32_t> seg_counts;
};
// Work arrays of one map's index build.  The two maps are built side by side on streams of their own (the reference builds its
// two kd-trees in two threads, photon_scene.rs:70-82), so each has its own set.
struct IndexScratch {
    DBuf<uint32_t> idx, idx2, ncell;
    DBuf<uint64_t> keys, keys2;
    DBuf<unsigned long long> acc;
    DBuf<uint8_t> sort_temp;
    DBuf<IndexHeader> header;
    IndexHeader *host_header = nullptr;  // pinned
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    ~IndexScratch() {
        if (host_header) (void)hipHostFree(host_header);
        if (stream) (void)hipStreamDestroy(stream);
        if (done) (void)hipEventDestroy(done);
    }
};
struct CausticSeeds {
    DBuf<float4> pos, dir;
    DBuf<uint64_t> key;
    uint64_t count = 0;
};

}  // namespace rrt

namespace rrt {
// One chunk of a frame in flight: its own level queues, counters, readback staging and stream.  Lane 0 runs on the
// caller's stream; the others on streams of their own, forked from and joined to the caller's stream with events.
struct RenderLane {
    std::vector<LevelStorage> levels;
    DBuf<uint32_t> counters;  // lanes > 0 own their counters; lane 0 borrows the tail of the statistics buffer
    uint32_t *ctr = nullptr;  // [32 + k] = heavy radiance estimates filed by level k, [63] = error bits
    // the stripe counters of the level queues (kernels.h: RRT_STRIPES): counter s of level k at [(k * RRT_STRIPES + s) * RRT_STRIPE_STRIDE]
    // (lane 0's lie behind its other counters in the statistics buffers, device and pinned host copies alike: one memset,
    // one copy per frame; the other lanes own theirs)
    DBuf<uint32_t> stripe_own;
    uint32_t *stripe = nullptr;
    uint32_t *stripe_pinned = nullptr;
    bool owns_stripe_pinned = false;
    uint32_t *host_pinned = nullptr;
    bool owns_pinned = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev_trace0 = nullptr, ev_trace1 = nullptr, ev_done = nullptr;
    DBuf<HeavyRec> heavy;  // photon mode: the queries of the current level filed for k_gather_heavy
    DBuf<uint2> knn;       // photon mode with rrt_params.gather_k: gather_k heap entries per ray of the widest level
    DBuf<uint32_t> tile_table;  // camera frames: where the band's full 8x8 tiles lie (kernels.hip: k_tile_table), for ...
    uint32_t tt_width = 0, tt_rows = 0, tt_order = 0;  // ... this frame width, band height and tile order
    // per round of chunks of a frame (a frame larger than the chunk size takes several rounds, each over another part of
    // the image): [0] = chunk size it was measured on, [1 + k] = size of level k
    std::vector<std::vector<uint32_t>> prediction;
};
}  // namespace rrt

struct rrt_scene {
    int mode = RRT_MODE_SCENE;
    rrt_params params{};
    std::vector<rrt::HostTexture> textures;
    std::vector<rrt_material_desc> materials;
    std::vector<std::unique_ptr<rrt::HostObject>> detached;
    std::vector<std::unique_ptr<rrt::HostObject>> objects;
    std::vector<rrt_light_desc> lights;
    rrt_builder builder{};

    bool flattened = false;
    bool maps_built = false;
    int device = -1;  // the HIP device rrt_pre_render ran on: every later call on the handle makes it current for its thread
    uint64_t seed = 0;
    rrt::DeviceScene dev;
    std::vector<std::unique_ptr<rrt::RenderLane>> lanes;  // see render_rays (DBuf is not movable: lanes stay put)
    rrt::DBuf<uint32_t> counters;             // error word of the batch entry points
    rrt::DBuf<unsigned long long> stats_dev;  // StatSlot counters
    uint32_t *host_pinned = nullptr;          // readback staging
    rrt_stats stats{};         // what rrt_get_stats reports; written under render_mutex by render calls
    rrt_stats render_stats{};  // the render fields of the call that currently holds this handle / worker
    std::vector<hipEvent_t> events;

    // Concurrent render calls on one handle (the reference's Environment is Sync: raytrace(&self) runs on every render
    // thread, scene.rs:83, camera.rs:22-40).  The handle's own queues / counters serve the first caller; callers that
    // arrive while it is busy lease a WORKER: a light clone that shares the flattened scene and the photon maps by
    // pointer and owns only queues, counters, events and statistics.  Anything that rebuilds device data drops them.
    std::mutex render_mutex;
    bool busy = false;                                 // the handle's own render state is in use
    std::vector<std::unique_ptr<rrt_scene>> workers;   // their `busy` flags are guarded by render_mutex too
    bool is_worker = false;
    uint64_t epoch = 0;                                // bumped by whatever changes `stats` outside a render call
    uint32_t heavy_auto = RRT_HEAVY_THRESHOLD;         // gather_heavy_threshold = 0 resolves to this (set by rrt_photon_build)
    rrt::DevPhotonMap map_view[2]{};                   // what the render kernels read (index[].view(), shared with workers)

    // photon state
    rrt::PhotonBuffers local[2];  // this rank's records: regular, caustic
    std::vector<std::unique_ptr<rrt::CausticSeeds>> seeds;  // per light
    rrt::PhotonIndex index[2];
    rrt::PhotonScratch photon_scratch;
    rrt::IndexScratch index_scratch[2];
    // bits of the record keys in use (rrt_photon_shoot knows its index range; imported / loaded records are looked at on the
    // device): the canonical sort of the index build only passes over those
    uint64_t meta_or = 0;
    rrt::DBuf<unsigned long long> meta_or_dev;  // OR of the keys of imported records (read at the next build)
    bool meta_or_pending = false;

    ~rrt_scene();
};

namespace rrt {
int flatten_scene(rrt_scene *s);
int ensure_workspace(rrt_scene *s);
// sums the RRT_STAT_COPIES device-side copies of the counters (synchronous)
int read_stats(rrt_scene *s, unsigned long long out[ST_COUNT]);
int render_rays(rrt_scene *s, const DevCamera *cam, uint32_t y0, uint32_t n_pixels, const float *origins_dev,
                const float *directions_dev, float *rgb_dev, float *depth_dev, uint32_t *id_dev, hipStream_t stream);
void build_bvh(const float *positions, uint32_t n, std::vector<uint32_t> &order, std::vector<BvhNode> &nodes, uint32_t &max_depth);
// bvh_build.cpp: the traversal's 32-byte nodes from either builder's output; fills the mesh's grid (qbase, qscale, qmag, wild)
void quantise_bvh(const std::vector<BvhNode> &nodes, MeshInfo &mi, std::vector<BvhNodeQ> &out, uint32_t &stack_need);
// bvh_lbvh.hip: the same outputs from the device-side LBVH build (n > 4); device_ms = HIP-event time of its kernels
int build_bvh_device(const float *positions, uint32_t n, std::vector<uint32_t> &order, std::vector<BvhNode> &nodes, uint32_t &max_depth,
                     float *device_ms);
// bvh_sah.hip: the host builder's tree, built on the device.  Returns 1 (not an rrt error) when a coordinate is not finite.
int build_bvh_device_sah(const float *positions, uint32_t n, std::vector<uint32_t> &order, std::vector<BvhNode> &nodes, uint32_t &max_depth,
                     float *device_ms);
// bvh_sah.hip: the same build with everything that follows it done on the device as well — the mesh's grid (mi.qbase, qscale,
// qmag, wild), the four-wide quantised nodes (quantise_bvh's bytes), the triangle records and their original indices stay in HBM
struct DeviceMesh {
    DBuf<BvhNodeQ> qnodes;
    DBuf<TriHot> tris;
    DBuf<uint32_t> tri_index;
    uint32_t n_nodes = 0, n_tris = 0;
};
int build_mesh_device_sah(const float *positions, uint32_t n, MeshInfo &mi, DeviceMesh &M, uint32_t &stack_need, float *device_ms);
void drop_render_workers(rrt_scene *s);
// HIP's current device is per thread (new threads start on device 0): entry points that touch the scene's memory call this
int use_scene_device(const rrt_scene *s);  // before device data of the scene or its photon maps is rebuilt
int read_ppm(const std::string &path, HostTexture &tex);
int load_obj(const std::string &path, bool smooth, std::vector<float> &tri_pos, std::vector<float> &tri_nrm);
}  // namespace rrt
