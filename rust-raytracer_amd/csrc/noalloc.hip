// noalloc.hip — rrt_trace_noalloc: the batched Environment::trace (rrt_trace) for callers that cannot allocate device
// memory (tight serving loops, harnesses that forbid hipMalloc; not capturable into a HIP graph: it synchronises the null stream).
//
// rrt_trace stages its host arrays through device buffers it allocates per call, and keeps its error word and the
// statistics counters in buffers the scene handle allocated.  Here all of that lives in ONE file-scope device array of a
// fixed size, `scratch` below: every GPU the library runs on holds a copy of it from module load until the process ends.
// The entry point makes no hipMalloc / hipFree call and uses no allocating library routine; the scene itself was put
// into HBM earlier, by rrt_pre_render.
//
// `scratch` as written in host code is only a host-side stand-in for the symbol, not a device address.  The address of
// each device's copy comes from hipGetSymbolAddress, once per device with that device current, and from then on host
// code uses that pointer and never the array's name (kernel arguments, memsets, copies).
//
// Every call on a GPU shares that GPU's array, which still holds the previous call's contents: the error word and the
// counters are cleared at the start of a call, the staging areas are overwritten for the first n rays only, and only
// those n are read back.  Calls that overlap on one GPU (threads, or scene handles, of one process) race.
#include <string.h>

#include <mutex>

#include "runtime.h"

using namespace rrt;

namespace {

constexpr size_t kMaxRays = RRT_NOALLOC_MAX_RAYS;
constexpr size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
// layout of the array, every part 256-byte aligned
constexpr size_t kOffErr = 0;                                                   // 64 x u32, [63] = error bits
constexpr size_t kOffStats = kOffErr + align256(64 * sizeof(uint32_t));          // RRT_STAT_WORDS x u64
constexpr size_t kClearBytes = kOffStats + align256(RRT_STAT_WORDS * sizeof(unsigned long long));  // what a call zeroes
constexpr size_t kOffOrigins = kClearBytes;                                     // in:  [n][3] f32
constexpr size_t kOffDirections = kOffOrigins + align256(kMaxRays * 12);        // in:  [n][3] f32
constexpr size_t kOffObject = kOffDirections + align256(kMaxRays * 12);         // out: [n] u32
constexpr size_t kOffPrim = kOffObject + align256(kMaxRays * 4);                // out: [n] u32
constexpr size_t kOffDistance = kOffPrim + align256(kMaxRays * 4);              // out: [n] f32
constexpr size_t kOffPosition = kOffDistance + align256(kMaxRays * 4);          // out: [n][3] f32
constexpr size_t kOffNormal = kOffPosition + align256(kMaxRays * 12);           // out: [n][3] f32
constexpr size_t kOffUv = kOffNormal + align256(kMaxRays * 12);                 // out: [n][2] f32
constexpr size_t kScratchBytes = kOffUv + align256(kMaxRays * 8);               // 68 B per ray + 4.25 KiB = 4.25 MiB

static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char *g_scratch_ptr[kMaxDevices] = {};  // per device: each GPU holds its own copy at its own address

// the address of the current device's copy of `scratch` (the caller has made `device` current)
int scratch_of_device(int device, unsigned char **out) {
    if (device < 0 || device >= kMaxDevices) return fail(RRT_ERR_LIMIT, "rrt_trace_noalloc: device index above 63");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        RRT_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(RRT_ERR_DEVICE, "hipGetSymbolAddress returned a null address for the scratch array");
        g_scratch_ptr[device] = static_cast<unsigned char *>(p);
    }
    *out = g_scratch_ptr[device];
    return RRT_OK;
}

}  // namespace

extern "C" int rrt_trace_noalloc(rrt_scene *s, const float *origins, const float *directions, size_t n, uint32_t *object_id,
                                 uint32_t *prim_id, float *distance, float *position, float *normal, float *uv) {
    if (!s || !origins || !directions) return fail(RRT_ERR_INVALID, "null argument");
    if (n > kMaxRays) return fail(RRT_ERR_LIMIT, "rrt_trace_noalloc: more than RRT_NOALLOC_MAX_RAYS (65536) rays in one call");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (!s->flattened) return fail(RRT_ERR_STATE, "rrt_pre_render has not been called");
    if (n == 0) return RRT_OK;
    int device = -1;
    RRT_HIP(hipGetDevice(&device));
    unsigned char *ptr = nullptr;
    if (int rc = scratch_of_device(device, &ptr)) return rc;

    hipStream_t stream = nullptr;  // the null stream, like rrt_trace
    RRT_HIP(hipMemsetAsync(ptr + kOffErr, 0, kClearBytes, stream));
    RRT_HIP(hipMemcpyAsync(ptr + kOffOrigins, origins, n * 12, hipMemcpyHostToDevice, stream));
    RRT_HIP(hipMemcpyAsync(ptr + kOffDirections, directions, n * 12, hipMemcpyHostToDevice, stream));
    BatchTraceArgs A;
    memset(&A, 0, sizeof(A));
    A.S = s->dev.view;
    A.origins = reinterpret_cast<const float *>(ptr + kOffOrigins);
    A.directions = reinterpret_cast<const float *>(ptr + kOffDirections);
    A.n = (uint32_t)n;
    A.lds_slots = s->dev.lds_slots;
    A.err = reinterpret_cast<uint32_t *>(ptr + kOffErr) + 63;
    A.stats = reinterpret_cast<unsigned long long *>(ptr + kOffStats);
    A.object_id = reinterpret_cast<uint32_t *>(ptr + kOffObject);
    A.prim_id = reinterpret_cast<uint32_t *>(ptr + kOffPrim);
    A.distance = reinterpret_cast<float *>(ptr + kOffDistance);
    A.position = reinterpret_cast<float *>(ptr + kOffPosition);
    A.normal = reinterpret_cast<float *>(ptr + kOffNormal);
    A.uv = reinterpret_cast<float *>(ptr + kOffUv);
    launch_batch_trace(A, s->dev.lds_bytes, stream);
    RRT_HIP(hipGetLastError());
    uint32_t err = 0;
    unsigned long long st_all[RRT_STAT_WORDS];
    RRT_HIP(hipMemcpyAsync(&err, A.err, sizeof(err), hipMemcpyDeviceToHost, stream));
    RRT_HIP(hipMemcpyAsync(st_all, A.stats, sizeof(st_all), hipMemcpyDeviceToHost, stream));
    RRT_HIP(hipStreamSynchronize(stream));
    unsigned long long nodes = 0, tris = 0;
    for (int c = 0; c < RRT_STAT_COPIES; c++) {
        nodes += st_all[c * RRT_STAT_STRIDE + ST_NODES];
        tris += st_all[c * RRT_STAT_STRIDE + ST_TRIS];
    }
    s->stats.bvh_nodes_visited = nodes;
    s->stats.bvh_tris_tested = tris;
    if (err & RRT_ERR_BIT_INTERNAL) return fail(RRT_ERR_DEVICE, "device-side index guard tripped (internal error)");
    if (err & 1u) return fail(RRT_ERR_OVERFLOW, "HitVec overflow");
    if (object_id) RRT_HIP(hipMemcpyAsync(object_id, A.object_id, n * 4, hipMemcpyDeviceToHost, stream));
    if (prim_id) RRT_HIP(hipMemcpyAsync(prim_id, A.prim_id, n * 4, hipMemcpyDeviceToHost, stream));
    if (distance) RRT_HIP(hipMemcpyAsync(distance, A.distance, n * 4, hipMemcpyDeviceToHost, stream));
    if (position) RRT_HIP(hipMemcpyAsync(position, A.position, n * 12, hipMemcpyDeviceToHost, stream));
    if (normal) RRT_HIP(hipMemcpyAsync(normal, A.normal, n * 12, hipMemcpyDeviceToHost, stream));
    if (uv) RRT_HIP(hipMemcpyAsync(uv, A.uv, n * 8, hipMemcpyDeviceToHost, stream));
    RRT_HIP(hipStreamSynchronize(stream));
    return RRT_OK;
}
