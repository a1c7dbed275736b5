// kernels.hip — gfx950 kernels of the render pass.
//
//  k_trace_shade  one lane per ray of one recursion level: nearest-entering-hit search, hit
//                 attributes, local shading terms (ambient part, per-light terms with inline shadow
//                 rays, or the photon-map estimate in photon mode) and emission of the reflection /
//                 refraction child rays into the next level's queue.  Children are appended with one
//                 atomic per wave: a 64-bit ballot + lane-prefix popcount gives each lane its slot
//                 (the compaction that keeps the next level's waves full).
//  k_combine      bottom-up evaluation of the recursion (global_material.rs:100-107,
//                 compound_material.rs:80-87, scene.rs:116-121 / photon_scene.rs:354-355) in the
//                 reference's own operation order; level 0 writes the frame.
//
// Block = 256 threads = 4 waves; LDS per block = max(CSG hit lists, BVH stacks).
#include "dev_photon.h"
#include "dev_shade.h"
#include "kernels.h"

namespace rrt {

#ifndef RRT_BLOCK
#define RRT_BLOCK 256
#endif

extern __shared__ float4 g_smem[];

__device__ __forceinline__ Lds make_lds(const DevScene &S, uint32_t lds_slots, uint32_t nthreads = RRT_BLOCK) {
    Lds l;
    l.nthreads = nthreads;
    l.stack_base = (S.has_csg & 2u) ? lds_slots * RRT_LIST_CAP : 0u;
    l.tid = threadIdx.x;
    l.le = reinterpret_cast<float2 *>(g_smem);
    l.stack = reinterpret_cast<int *>(g_smem);
    return l;
}

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// Sum of a 32-bit value over the 64 lanes of a wave in which EVERY lane is active, as a wave-uniform 64-bit number: the
// halves of the value are summed apart (no carry can be lost) with four DPP adds each -- quad permutes, then the mirrors of
// the half row and the row -- and the four rows' sums are read out with v_readlane.  No LDS traffic (a 64-bit __shfl_xor
// butterfly costs two ds_bpermute per step: the seven statistics of a wave came to some 300 instructions).
__device__ __forceinline__ uint32_t row_sum_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false);  // row_mirror
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) + (uint32_t)__builtin_amdgcn_readlane((int)v, 32) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ unsigned long long wave_sum_u32(uint32_t v) {
    return (unsigned long long)row_sum_dpp(v & 0xFFFFu) + ((unsigned long long)row_sum_dpp(v >> 16) << 16);
}
__device__ __forceinline__ void flush_counters(const Counters &cnt, unsigned long long gathered, unsigned long long queries,
                                               unsigned long long candidates, unsigned long long *stats, uint32_t *err) {
    unsigned long long n = wave_sum(cnt.nodes), t = wave_sum(cnt.tris), s = wave_sum(cnt.shadow);
    unsigned long long g = wave_sum(gathered), q = wave_sum(queries), c = wave_sum(candidates);
    stats += (blockIdx.x % RRT_STAT_COPIES) * RRT_STAT_STRIDE;
    if ((threadIdx.x & 63) == 0) {
        if (n) atomicAdd(&stats[ST_NODES], n);
        if (t) atomicAdd(&stats[ST_TRIS], t);
        if (s) atomicAdd(&stats[ST_SHADOW], s);
        if (g) atomicAdd(&stats[ST_GATHERED], g);
        if (q) atomicAdd(&stats[ST_QUERIES], q);
        if (c) atomicAdd(&stats[ST_CANDIDATES], c);
    }
    if (cnt.overflow) atomicOr(err, cnt.overflow);
}

// FullCamera::get_ray_pixel (full_camera.rs:61-81): d = M^T (fx-0.5, fy-0.5, fov), rows of M =
// right, up, lookat; no aspect correction
__device__ __forceinline__ void camera_ray(const DevCamera &C, uint32_t x, uint32_t y, f3 &o, f3 &d) {
    float fx = ((float)x + 0.5f) / (float)C.width;
    float fy = ((float)y + 0.5f) / (float)C.height;
    f3 v = mk3(fx - 0.5f, fy - 0.5f, C.fov);
    d = norm3(mk3(C.right.x * v.x + C.up.x * v.y + C.lookat.x * v.z, C.right.y * v.x + C.up.y * v.y + C.lookat.y * v.z,
                  C.right.z * v.x + C.up.z * v.y + C.lookat.z * v.z));
    o = C.position;
}

// Camera renders walk the band in 8x8 pixel tiles (one wave = one tile): neighbouring rays hit and miss the same
// objects, which is what the wave-uniform bound skip and the shading branches want.  Pixels right of the last full
// tile column, then rows below the last full tile row, follow in row-major order.  Returns the band-relative pixel.
__device__ __forceinline__ uint32_t node_pixel(uint32_t g, uint32_t width, uint32_t rows, bool xcd_order) {
    uint32_t tiles_x = width >> 3, full_rows = rows & ~7u;
    uint32_t tiled = tiles_x * 8u * full_rows;
    if (g < tiled) {
        uint32_t tile = g >> 6, in = g & 63u;
        // One-wave blocks: wave b runs on XCD b % 8 (workgroups are dealt round-robin over the 8 XCDs, each with an L2 of
        // its own).  Tiles are taken in supertile-major order (4 x 4 tiles, row-major inside) and the j-th wave of an XCD
        // works in chunk (j / 64) * 8 + xcd of 64 consecutive tiles of that order: each XCD's L2 sees compact patches of
        // the frame (128 x 32 pixels), and the patches of the eight XCDs alternate finely enough to keep their loads equal.
        // Measured on the 4K mesh frame: 0.993 -> 0.956 ms; whole bands per XCD took TWICE as long (1.89 ms: the XCD that
        // holds the teapots works alone), 2 x 2 / 8 x 8 supertiles 0.962 / 0.983, chunks of 16 / 32 / 128 / 256 tiles
        // 0.982 / 0.983 / 0.955 / 1.020.
        if (xcd_order) {
            const uint32_t ST = 4u, n_tiles = tiles_x * (full_rows >> 3);
            uint32_t u = tile;
            // (chunks of 128 tiles for frames of 65 536 tiles and more: at four waves per SIMD the 4K mesh frame takes 0.759 ms
            //  with chunks of 64, 0.746 with 128, 0.815 with 256; smaller frames lose balance with three of them in flight)
            const uint32_t CL = n_tiles >= 65536u ? 7u : 6u;
            const uint32_t x = tile & 7u, j = tile >> 3, c = j >> CL;
            if (((c * 8u + 8u) << CL) <= n_tiles) u = ((c * 8u + x) << CL) + (j & ((1u << CL) - 1u));
            // u -> (tx, ty): rows of supertiles, the last one lower; in a row full supertiles, then a narrower one
            const uint32_t tiles_y = full_rows >> 3;
            const uint32_t per_row = tiles_x * ST;
            const uint32_t sy = u / per_row;
            const uint32_t hs = min(ST, tiles_y - sy * ST);
            const uint32_t r = u - sy * per_row;           // index inside this row of supertiles (rows above are all ST high)
            const uint32_t full_w = (tiles_x / ST) * ST;   // tiles covered by full-width supertiles
            uint32_t tx, ty;
            if (r < full_w * hs) {
                const uint32_t sx = r / (ST * hs), w = r - sx * ST * hs;
                tx = sx * ST + w % ST;
                ty = sy * ST + w / ST;
            } else {
                const uint32_t wlast = tiles_x - full_w, w = r - full_w * hs;
                tx = full_w + w % wlast;
                ty = sy * ST + w / wlast;
            }
            return (ty * 8u + (in >> 3)) * width + tx * 8u + (in & 7u);
        }
        uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
        return (ty * 8u + (in >> 3)) * width + tx * 8u + (in & 7u);
    }
    uint32_t r = g - tiled, wrem = width - tiles_x * 8u;
    if (r < wrem * full_rows) {
        uint32_t y = r / wrem;
        return y * width + tiles_x * 8u + (r - y * wrem);
    }
    return full_rows * width + (r - wrem * full_rows);
}

// node_pixel with its column and band row.  The tiles' places come from a table (k_tile_table: node_pixel itself, once per
// tile and frame size), because node_pixel's chain of integer divisions cost every wave of a frame some 150 vector
// instructions, twice (ray and pixel write): 8 % of all the 4K mesh frame issued.  The pixels beside the full tiles keep
// the arithmetic.
__device__ __forceinline__ uint32_t frame_pixel(const TraceArgs &A, uint32_t g, uint32_t &x, uint32_t &y) {
    const uint32_t width = A.cam.width;
    if (A.tile_table && g < (width >> 3) * 8u * (A.band_rows & ~7u)) {
        const uint32_t w = A.tile_table[g >> 6], in = g & 63u;
        x = (w & 0xFFFFu) * 8u + (in & 7u);
        y = (w >> 16) * 8u + (in >> 3);
        return y * width + x;
    }
    const uint32_t pixel = node_pixel(g, width, A.band_rows, A.tile_order != 0u);
    y = pixel / width;
    x = pixel - y * width;
    return pixel;
}
__global__ void k_tile_table(uint32_t *table, uint32_t n_tiles, uint32_t width, uint32_t rows, uint32_t xcd_order) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint32_t pixel = node_pixel(t * 64u, width, rows, xcd_order != 0u);  // the tile's first pixel is its upper left one
    const uint32_t y = pixel / width, x = pixel - y * width;
    table[t] = (x >> 3) | ((y >> 3) << 16);
}

// A node's record is written by k_trace_shade and read once, a whole frame later, by k_combine: non-temporal stores, so
// that it does not push BVH nodes and scene records out of the L2 on its way to memory.
// (Mirror / glass teapots at 4K: kernels 3.33 / 2.83 -> 3.25 / 2.75 ms.)
typedef float rrt_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void node_store(float4 *p, float4 v) {
    rrt_v4f w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<rrt_v4f *>(p));
}
// ... and read that way by k_combine (frames 1 % shorter)
__device__ __forceinline__ float4 node_load(const float4 *p) {
    rrt_v4f w = __builtin_nontemporal_load(reinterpret_cast<const rrt_v4f *>(p));
    return make_float4(w.x, w.y, w.z, w.w);
}

// (The ray queues are another matter: the next level's launch finds much of them in the L2.  Non-temporal ray stores and
//  loads were measured -- mirror / glass teapots 1 % slower, the C2 frame 4 % -- and so were non-temporal triangle loads,
//  which cost the 4K mesh frame 11 %: neighbouring rays test the same triangles.)
// the ray of node i of the level being processed: generated from the camera (level 0 of a camera render) or read from
// the level's queue
__device__ __forceinline__ void level_ray(const TraceArgs &A, uint32_t i, f3 &o, f3 &d) {
    if (A.level == 0 && A.cam.width != 0) {
        uint32_t x, row;  // row of the band (or of the packed rows of an interleaved share)
        frame_pixel(A, A.pixel_base + i, x, row);
        if (A.cam.row_stride > 1) row = ((row / A.cam.row_group) * A.cam.row_stride + A.cam.row_first) * A.cam.row_group + row % A.cam.row_group;
        camera_ray(A.cam, x, A.y0 + row, o, d);
    } else {
        float4 ro = A.cur.ray_o[i], rd = A.cur.ray_d[i];
        o = mk3(ro.x, ro.y, ro.z);
        d = mk3(rd.x, rd.y, rd.z);
    }
}

// level-0 outputs: the frame (any pointer may be null)
__device__ __forceinline__ void write_pixel(const TraceArgs &A, uint32_t pixel, f3 colour, float depth, uint32_t object) {
    // (non-temporal stores: the frame is written once and never read by the kernels, so it need not push BVH nodes out of
    //  the L2: 0.753 -> 0.748 ms on the 4K mesh frame)
    if (A.out_rgb) {
        __builtin_nontemporal_store(colour.x, &A.out_rgb[(size_t)pixel * 3 + 0]);
        __builtin_nontemporal_store(colour.y, &A.out_rgb[(size_t)pixel * 3 + 1]);
        __builtin_nontemporal_store(colour.z, &A.out_rgb[(size_t)pixel * 3 + 2]);
    }
    if (A.out_depth) __builtin_nontemporal_store(depth, &A.out_depth[pixel]);
    if (A.out_id) __builtin_nontemporal_store(object, &A.out_id[pixel]);
}

// Material part of Scene::raytrace's colour for one node, in the reference's operation order: the fold of the parts'
// compute_once (`once`), the Global part's Fresnel mix of the children (global_material.rs:100-107) and the compound
// mean (compound_material.rs:86).  Shared by k_combine and by the fused single-level path of k_trace_shade.
__device__ __forceinline__ f3 scene_node_total(f3 once, uint32_t flags, float wr, float wt, float div, float kr, bool have_a, f3 ca,
                                               bool have_b, f3 cb) {
    f3 total = once;
    if (flags & NF_HAS_GLOBAL) {
        f3 gl = mk3(0.0f, 0.0f, 0.0f);
        if (flags & NF_GLOBAL_ACTIVE) {
            f3 ra = ca * wr, rb = cb * wt;
            if (have_a && have_b)
                gl = (ra * kr) + (rb * (1.0f - kr));
            else if (have_a)
                gl = ra;
            else if (have_b)
                gl = rb;
        }
        total = (div != 0.0f) ? once + gl : gl;
    }
    if (div != 0.0f) total = total / div;
    return total;
}

#ifndef RRT_MIN_WAVES
#define RRT_MIN_WAVES 3
#endif

// What a lane carries out of shade_node into the block-wide tail of the kernel (child rays, a deferred radiance estimate).
struct NodeOut {
    bool want_a = false, want_b = false;  // a: reflection child, b: refraction child
    f3 ao, ad, bo, bd;
    bool defer = false;  // photon mode: this lane's radiance estimate goes to k_gather_heavy
    HitAttr h;
};
struct GatherCounts {
    unsigned long long gathered = 0, queries = 0, candidates = 0;
};

// Everything Scene::raytrace / PhotonScene::vueontrace do with the hit of ray (o, d), node i of the level: hit attributes,
// local shading terms, the per-light loop, the child rays a Global part asks for.  `occluded(l, so, sd, limit)` answers
// Scene::shadowtrace for light l (k_trace_shade traces the shadow ray there and then; the pooled-walk experiment,
// tools/experiments/pool_kernel.patch, had traced it before and looked the answer up).
// MODE: RRT_MODE_SCENE, RRT_MODE_PHOTON, or MODE_PHOTON_FINE = photon mode over fine photon grids (dev_photon.h), an
// instance of its own so that the coarse gather keeps its registers
#define MODE_PHOTON_FINE 2
template <int MODE, int FEAT, bool FUSE, class Occluded>
__device__ __forceinline__ void shade_node(const TraceArgs &A, uint32_t i, f3 o, f3 d, TraceResult tr, Counters &cnt, GatherCounts &gc, NodeOut &N,
                                           Occluded occluded) {
    const DevScene &S = A.S;
    HitAttr &h = N.h;
    if (tr.object != 0xFFFFFFFFu) {
        eval_hit<FEAT>(S, o, d, tr.instr, tr.sub, tr.t, h);
        if ((uint32_t)h.material >= S.n_materials) {  // index guard of eval_hit tripped
            cnt.overflow |= RRT_ERR_BIT_INTERNAL;
            tr.object = 0xFFFFFFFFu;
        }
    }
    // Fused single-level frames (A.fuse: raytracing mode, no material spawns rays, so level 0 is the whole
    // recursion): the pixel is finished here with k_combine's arithmetic and nothing goes through the node arrays.
    constexpr bool fuse = MODE == RRT_MODE_SCENE && FUSE;
    uint32_t out_x, out_y;
    const uint32_t out_pixel = fuse ? (A.cam.width ? frame_pixel(A, A.pixel_base + i, out_x, out_y) : A.pixel_base + i) : 0u;
    if (tr.object == 0xFFFFFFFFu) {  // scene.rs:85-90 / photon_scene.rs:311-313
        if (fuse) {
            write_pixel(A, out_pixel, mk3(0.0f, 0.0f, 0.0f), 0.0f, 0xFFFFFFFFu);
        } else {
            node_store(&A.cur.glob[i], make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u)));  // (depth 0, no flags)
            A.cur.object_id[i] = 0xFFFFFFFFu;
        }
        return;
    }
    const DevMaterial &M = S.materials[h.material];
    uint32_t flags = NF_HIT;
    float kr = 0.0f;
    if (!fuse) A.cur.object_id[i] = tr.object;
    if (MODE == RRT_MODE_SCENE) {
        f3 once = material_once_pre(S, M, h, cnt);
        const float div = M.is_compound ? (float)M.n_parts : 0.0f;
        f3 colour = mk3(0.0f, 0.0f, 0.0f);
        // (a fused frame has no material with a Global part: that is what makes it single-level)
        if (fuse) colour = scene_node_total(once, flags, 0.0f, 0.0f, div, kr, false, colour, false, colour);
        // scene.rs:96-122
        for (uint32_t l = 0; l < S.n_lights; l++) {
            const DevLight L = S.lights[l];
            f3 viewer = -norm3(h.position);  // eye assumed at the origin
            f3 ldir;
            bool lit = light_direction(L, h.position, ldir);
            if (lit && dot3(ldir, h.normal) > 0.0f) lit = false;
            if (lit) {
                f3 sd = -ldir;
                f3 so = h.position + sd * 0.0001f;
                if (occluded(l, so, sd, len3(ldir))) lit = false;
            }
            f3 term = mk3(0.0f, 0.0f, 0.0f);
            if (lit) term = material_per_light(S, M, h, viewer, ldir, cnt) * light_intensity(L, h.position);
            if (fuse) {
                if (lit) colour = colour + term;  // scene.rs:116-121, as k_combine adds them
            } else {
                node_store(&A.cur.lights[(size_t)l * A.cur.capacity + i], make_float4(term.x, term.y, term.z, lit ? 1.0f : 0.0f));
            }
        }
        if (fuse) {
            write_pixel(A, out_pixel, colour, tr.t, tr.object);
        } else {
            // the child rays come after the light loop: nothing of them is live across the shadow traversals
            float wr = 0.0f, wt = 0.0f;
            if (M.global_part >= 0) {  // GlobalMaterial::compute_once, global_material.rs:71-98
                const MatPart &G = M.parts[M.global_part];
                flags |= NF_HAS_GLOBAL;
                wr = G.wr;
                wt = G.wt;
                if (A.level < A.max_depth) {
                    flags |= NF_GLOBAL_ACTIVE;
                    if (wr > 0.0f) {
                        N.ad = norm3(reflect3(h.normal, d));
                        N.ao = h.position + N.ad * 0.0001f;
                        N.want_a = true;
                    }
                    if (wt > 0.0f) N.want_b = global_refraction(G, h, true, d, N.bo, N.bd, kr);
                }
            }
            node_store(&A.cur.once[i], make_float4(once.x, once.y, once.z, kr));
            // depth = hit.distance (scene.rs:124-127); the compound divisor rides in the flag word (NF_PARTS_SHIFT)
            node_store(&A.cur.glob[i], make_float4(wr, wt, tr.t, __uint_as_float(flags | (M.is_compound ? (uint32_t)M.n_parts << NF_PARTS_SHIFT : 0u))));
        }
    } else {  // PhotonScene::vueontrace, photon_scene.rs:309-361
        float sw = M.surface_weight, rw = M.w[3];
        f3 surface = mk3(0.0f, 0.0f, 0.0f);
        if (sw > 0.0f) {
            f3 inc, inten;
            gc.queries += 2;
            const int got = average_photon_at<MODE == MODE_PHOTON_FINE>(A.maps, h.position, A.gather_half_width, A.heavy ? A.heavy_threshold : 0xFFFFFFFFu, inc, inten,
                                              gc.gathered, gc.candidates, A.knn_k, KnnColumn{A.knn_ws + i, A.knn_stride});
            if (got == 1) surface = material_render_vueon(S, M, h, inc, inten, -d, cnt);
            N.defer = got == 2;  // `once` is written by k_gather_heavy (or below, if the queue is full)
        }
        // photon_scene.rs:329-352 recurses without limit; vueons deeper than max_vueon_depth are not traced, and counted
        bool deeper = A.level + 1 <= A.max_depth;
        if (rw > 0.0f) {
            if (deeper) {
                N.ao = h.position + h.normal * 0.0001f;
                N.ad = norm3(reflect3(h.normal, d));
                N.want_a = true;
            } else {
                cnt.truncated++;
            }
        }
        if (M.global_part >= 0) {
            f3 to, td;
            float k;
            if (global_refraction(M.parts[M.global_part], h, true, d, to, td, k)) {
                if (deeper) {
                    N.bo = to + td * 0.0001f;
                    N.bd = td;
                    N.want_b = true;
                } else {
                    cnt.truncated++;
                }
            }
        }
        node_store(&A.cur.once[i], make_float4(surface.x, surface.y, surface.z, 0.0f));
        node_store(&A.cur.glob[i], make_float4(rw, sw, tr.t, __uint_as_float(flags)));
    }
}

// Which node of the level a thread works on (TraceArgs: level 0 is one dense range, the levels behind it are striped).
struct NodeSlot {
    uint32_t i;       // node index into the level's arrays
    bool active;      // the thread has a node
    bool block_idle;  // no thread of the block has one (launched for the predicted size of the level)
};
// (LEVEL0: the caller is an instance that only ever runs level 0, the fused single-level kernels)
template <int BLOCK, bool LEVEL0 = false>
__device__ __forceinline__ NodeSlot node_slot(const TraceArgs &A) {
    NodeSlot n;
    if (LEVEL0 || !A.striped) {
        n.i = blockIdx.x * BLOCK + threadIdx.x;
        n.active = n.i < A.count;
        n.block_idle = blockIdx.x * BLOCK >= A.count;
    } else {
        // Eight consecutive blocks (one per XCD) take eight consecutive blocks' worth of ONE stripe, the next eight of the
        // next stripe: every stripe is worked on by all XCDs.  (Stripe s on XCD s % 8 alone was measured: a level whose costly
        // rays come from one patch of the frame then runs on an eighth of the GPU -- level 1 of the photon-mapped scene2
        // frame, 114 k rays, 0.56 -> 0.67 ms.)
        const uint32_t per_block = BLOCK >> A.lane_shift;  // rays of the block (k_combine: lane_shift 0)
        const uint32_t s = (blockIdx.x >> 3) % RRT_STRIPES, first = ((blockIdx.x & 7u) + ((blockIdx.x / (8u * RRT_STRIPES)) << 3)) * per_block;
        const uint32_t have = min(A.count_ptr[s * RRT_STRIPE_STRIDE], A.stripe_cap);
        n.i = s * A.stripe_cap + first + threadIdx.x;
        n.active = threadIdx.x < per_block && first + threadIdx.x < have;
        n.block_idle = first >= have;
    }
    return n;
}

// The block-wide tail of a level's kernel: append of the child rays to the next level's queue and (STATS) the statistics,
// ONE atomic per block each.  Per wave a 64-bit ballot + lane-prefix popcount gives each lane its slot inside the wave's
// share; the waves' shares are laid end to end through LDS and thread 0 reserves the block's total with a single atomicAdd
// (one atomic per wave on the one queue counter serialises in the L2: 32 k waves at 1080p).  Every thread of the block
// must call it (barriers inside).
template <int BLOCK, bool STATS, bool PHOTON>
__device__ __forceinline__ void finish_block(const TraceArgs &A, uint32_t i, bool write_child, NodeOut &N, const Counters &cnt, const GatherCounts &gc) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long ma = __ballot(N.want_a), mb = __ballot(N.want_b);
    uint32_t total = __popcll(ma) + __popcll(mb);
    // the wave's statistics: the ray counters are 32-bit per lane (DPP sums), the gather's 64-bit (photon mode only)
    unsigned long long v[ST_COUNT] = {0, 0, 0, 0, 0, 0, 0};
    if (STATS) {
        v[ST_NODES] = wave_sum_u32(cnt.nodes), v[ST_TRIS] = wave_sum_u32(cnt.tris), v[ST_SHADOW] = wave_sum_u32(cnt.shadow);
        v[ST_TRUNCATED] = wave_sum_u32(cnt.truncated);
#ifdef RRT_WALK_STATS
        if (A.mode == RRT_MODE_SCENE) v[ST_GATHERED] = wave_sum(cnt.ws[0]), v[ST_QUERIES] = wave_sum(cnt.ws[1]), v[ST_CANDIDATES] = wave_sum(cnt.ws[2]);
#endif
        if (PHOTON) v[ST_GATHERED] = wave_sum(gc.gathered), v[ST_QUERIES] = wave_sum(gc.queries), v[ST_CANDIDATES] = wave_sum(gc.candidates);
    }
    uint32_t block_base = 0, all_a = 0, base_a = 0, base_b = 0, block_total = total;
    if (BLOCK == 64) {
        // One-wave blocks: no LDS, no barriers.  Lane k adds statistic k; lane 0 reserves the wave's share of the queue.
        if (STATS) {
            const unsigned long long err = __ballot(cnt.overflow != 0);
            unsigned long long mine = 0;
#pragma unroll
            for (int k = 0; k < ST_COUNT; k++) mine = lane == (uint32_t)k ? v[k] : mine;
            if (lane < ST_COUNT && mine) atomicAdd(&A.stats[(blockIdx.x % RRT_STAT_COPIES) * RRT_STAT_STRIDE + lane], mine);
            if (err && cnt.overflow) atomicOr(A.err, cnt.overflow);
        }
        if (total) {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(A.next_count + ((blockIdx.x >> 3) % RRT_STRIPES) * RRT_STRIPE_STRIDE, total);
            block_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        }
        all_a = __popcll(ma);
        base_a = base_b = block_base;
    } else {
        __shared__ uint32_t s_wave_a[BLOCK / 64], s_wave_b[BLOCK / 64];
        __shared__ uint32_t s_block_base;
        __shared__ unsigned long long s_stat[BLOCK / 64][ST_COUNT];
        __shared__ uint32_t s_err;
        if (threadIdx.x == 0) s_err = 0;
        if (lane == 0) {
            s_wave_a[wave] = __popcll(ma);
            s_wave_b[wave] = __popcll(mb);
            if (STATS) {
#pragma unroll
                for (int k = 0; k < ST_COUNT; k++) s_stat[wave][k] = v[k];
            }
        }
        __syncthreads();
        if (STATS) {
            if (cnt.overflow) atomicOr(&s_err, cnt.overflow);
            if (threadIdx.x < ST_COUNT) {  // thread k owns statistic k
                unsigned long long sum = 0;
#pragma unroll
                for (int w = 0; w < BLOCK / 64; w++) sum += s_stat[w][threadIdx.x];
                if (sum) atomicAdd(&A.stats[(blockIdx.x % RRT_STAT_COPIES) * RRT_STAT_STRIDE + threadIdx.x], sum);
            }
        }
        if (threadIdx.x == 0) {
            uint32_t bt = 0;
#pragma unroll
            for (int w = 0; w < BLOCK / 64; w++) bt += s_wave_a[w] + s_wave_b[w];
            s_block_base = bt ? atomicAdd(A.next_count + ((blockIdx.x >> 3) % RRT_STRIPES) * RRT_STRIPE_STRIDE, bt) : 0u;
        }
        __syncthreads();
        if (STATS && threadIdx.x == 0 && s_err) atomicOr(A.err, s_err);
        block_base = s_block_base;
        block_total = 0;
        base_a = base_b = block_base;
        for (uint32_t w = 0; w < BLOCK / 64; w++) {
            all_a += s_wave_a[w];
            block_total += s_wave_a[w] + s_wave_b[w];
            if (w < wave) {
                base_a += s_wave_a[w];
                base_b += s_wave_b[w];
            }
        }
    }
    int2 child = make_int2(-1, -1);
    if (total) {
        // the block's reflection rays first, then its refraction rays: a wave of the next level then holds one kind of
        // ray from one 8x8-tile neighbourhood, which keeps its rays together at the bound boxes
        base_b += all_a;
        unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
        uint32_t off_a = base_a + __popcll(ma & below), off_b = base_b + __popcll(mb & below);
        // (positions inside the block's stripe so far; its nodes begin at stripe * next_stripe_cap)
        const uint32_t stripe_first = ((blockIdx.x >> 3) % RRT_STRIPES) * A.next_stripe_cap;
        off_a += stripe_first, off_b += stripe_first;
        if (block_base + block_total > A.next_stripe_cap) {  // prediction too small: the host repeats the frame level by level
            if (lane == 0) atomicOr(A.err, RRT_ERR_BIT_QUEUE);
            N.want_a = N.want_b = false;
        }
        if (N.want_a) {
            child.x = (int)off_a;
            A.next.ray_o[off_a] = make_float4(N.ao.x, N.ao.y, N.ao.z, __uint_as_float(i));
            A.next.ray_d[off_a] = make_float4(N.ad.x, N.ad.y, N.ad.z, __uint_as_float(0u));
        }
        if (N.want_b) {
            child.y = (int)off_b;
            A.next.ray_o[off_b] = make_float4(N.bo.x, N.bo.y, N.bo.z, __uint_as_float(i));
            A.next.ray_d[off_b] = make_float4(N.bd.x, N.bd.y, N.bd.z, __uint_as_float(1u));
        }
    }
    if (write_child) A.cur.child[i] = child;
}

// MODE is a template parameter so that the raytracing-mode instance does not carry the registers of the photon
// gather and vice versa (occupancy is VGPR-limited here).
// BLOCK: threads per block, 64 or 256 (launch_trace_shade says which and why).
// FUSE: single-level frames (TraceArgs.fuse; raytracing mode only): a compile-time constant so that this instance carries
// neither the child-ray state nor the node arrays' addresses across the shadow traversal.
// (Raytraced scenes of planes and meshes only: four waves per SIMD.  The fused instance needs 128 registers as it is; the
// one with child rays takes 152 and spills 15 of them under the bound, outside its loops: mirror / glass teapots 4.14 / 3.06
// -> 3.83 / 2.83 ms.)
template <int MODE, int FEAT, int BLOCK, bool FUSE>
__global__ __launch_bounds__(BLOCK, (MODE == RRT_MODE_SCENE && (FEAT & FEAT_NO_SOLIDS) && RRT_MIN_WAVES < 4) ? 4 : RRT_MIN_WAVES) void k_trace_shade(TraceArgs A) {
    const DevScene &S = A.S;
    Lds lds = make_lds(A.S, A.lds_slots, BLOCK);
    const NodeSlot slot = node_slot<BLOCK, MODE == RRT_MODE_SCENE && FUSE>(A);
    const uint32_t i = slot.i;
    const bool active = slot.active;
    if (slot.block_idle) return;  // launched for the predicted size of the level
    // every instance but the fused single-level one of raytracing mode takes its triangle records in one round trip (tri_test)
    constexpr int TF = FEAT | ((MODE == RRT_MODE_SCENE && FUSE) ? 0 : FEAT_TRI_EAGER);
    Counters cnt = {0, 0, 0, 0};
    GatherCounts gc;
    NodeOut N;
    N.ao = N.ad = N.bo = N.bd = mk3(0, 0, 0);
    N.h.material = 0;
    f3 o, d;
    RRT_WS_STAMP(ws_k0);
    if (active) {
        level_ray(A, i, o, d);
        TraceResult tr;
        RRT_WS_STAMP(ws_k1);
        trace<TF>(S, o, d, lds, cnt, tr);
        RRT_WS_STAMP(ws_k2);
#if RRT_WALK_STATS == 3
        RRT_WS_ADD(0, ws_k2 - ws_k1);
#endif
        shade_node<MODE, FEAT, FUSE>(A, i, o, d, tr, cnt, gc, N, [&](uint32_t, f3 so, f3 sd, float limit) {
            RRT_WS_STAMP(ws_s0);
            const bool occ = shadowtrace<TF>(S, so, sd, limit, lds, cnt);
            RRT_WS_STAMP(ws_s1);
#if RRT_WALK_STATS == 3
            RRT_WS_ADD(1, ws_s1 - ws_s0);
#endif
            return occ;
        });
    }
#if RRT_WALK_STATS == 3
    {
        RRT_WS_STAMP(ws_k9);
        RRT_WS_ADD(2, ws_k9 - ws_k0);  // the whole wave
    }
#endif

    // ---- photon mode: file the heavy radiance estimates (one atomic per wave; few waves have any) ----
    if (MODE != RRT_MODE_SCENE) {
        const HitAttr &h = N.h;
        const unsigned long long md = __ballot(N.defer);
        if (md) {
            const uint32_t ln = threadIdx.x & 63u;
            uint32_t base = 0;
            if (ln == (uint32_t)__ffsll((long long)md) - 1u) base = atomicAdd(A.heavy_count, (uint32_t)__popcll(md));
            base = __shfl(base, __ffsll((long long)md) - 1, 64);
            const uint32_t slot = base + (uint32_t)__popcll(md & (ln == 0 ? 0ULL : (~0ULL >> (64 - ln))));
            if (N.defer) {
                if (slot < A.heavy_capacity) {
                    HeavyRec R;
                    R.pos_u = make_float4(h.position.x, h.position.y, h.position.z, h.u);
                    R.nrm_v = make_float4(h.normal.x, h.normal.y, h.normal.z, h.v);
                    R.dir_mat = make_float4(d.x, d.y, d.z, __uint_as_float((uint32_t)h.material));
                    R.node = i;
                    R.has_tex = h.has_tex ? 1u : 0u;
                    R.pad0 = R.pad1 = 0u;
                    A.heavy[slot] = R;
                } else {  // queue full: this lane gathers for itself after all
                    f3 inc, inten;
                    f3 surface = mk3(0.0f, 0.0f, 0.0f);
                    if (average_photon_at<MODE == MODE_PHOTON_FINE>(A.maps, h.position, A.gather_half_width, 0xFFFFFFFFu, inc, inten, gc.gathered, gc.candidates) == 1)
                        surface = material_render_vueon(S, S.materials[h.material], h, inc, inten, -d, cnt);
                    node_store(&A.cur.once[i], make_float4(surface.x, surface.y, surface.z, 0.0f));
                }
            }
        }
    }
    finish_block<BLOCK, true, MODE != RRT_MODE_SCENE>(A, i, active && !(MODE == RRT_MODE_SCENE && FUSE), N, cnt, gc);
}

// Photon mode: the radiance estimates the trace kernel filed (HeavyRec), one query per wave.  Grid-stride over the
// records: every wave leaves the loop once the records run out, nothing waits on anything.
template <bool FINE>
__global__ __launch_bounds__(64) void k_gather_heavy(TraceArgs A) {
    const DevScene &S = A.S;
    const uint32_t n = min(*A.heavy_count, A.heavy_capacity);
    Counters cnt = {0, 0, 0, 0};
    unsigned long long gathered = 0, candidates = 0;
    for (uint32_t q = blockIdx.x; q < n; q += gridDim.x) {
        const HeavyRec R = A.heavy[q];
        HitAttr h;
        h.position = mk3(R.pos_u.x, R.pos_u.y, R.pos_u.z);
        h.normal = mk3(R.nrm_v.x, R.nrm_v.y, R.nrm_v.z);
        h.u = R.pos_u.w;
        h.v = R.nrm_v.w;
        h.has_tex = R.has_tex != 0u;
        h.material = (int32_t)__float_as_uint(R.dir_mat.w);
        const f3 d = mk3(R.dir_mat.x, R.dir_mat.y, R.dir_mat.z);
        if ((uint32_t)h.material >= S.n_materials || R.node >= A.cur.capacity) {  // index guards
            cnt.overflow |= RRT_ERR_BIT_INTERNAL;
            continue;
        }
        GatherSums reg, cau;
        if (FINE) {
            wave_gather_fine(A.maps[0], h.position, A.gather_half_width, reg);
            wave_gather_fine(A.maps[1], h.position, A.gather_half_width, cau);
        } else {
            wave_gather_box(A.maps[0], h.position, A.gather_half_width, reg);
            wave_gather_box(A.maps[1], h.position, A.gather_half_width, cau);
        }
        gathered += reg.k + cau.k;
        candidates += reg.candidates + cau.candidates;
        f3 inc, inten;
        f3 surface = mk3(0.0f, 0.0f, 0.0f);
        if (blend_estimates(reg, cau, inc, inten)) surface = material_render_vueon(S, S.materials[h.material], h, inc, inten, -d, cnt);
        if ((threadIdx.x & 63u) == 0) A.cur.once[R.node] = make_float4(surface.x, surface.y, surface.z, 0.0f);
    }
    if ((threadIdx.x & 63u) == 0) {  // every lane holds the same totals
        unsigned long long *stats = A.stats + (blockIdx.x % RRT_STAT_COPIES) * RRT_STAT_STRIDE;
        if (gathered) atomicAdd(&stats[ST_GATHERED], gathered);
        if (candidates) atomicAdd(&stats[ST_CANDIDATES], candidates);
        if (cnt.overflow) atomicOr(A.err, cnt.overflow);
    }
}

__global__ __launch_bounds__(RRT_BLOCK) void k_combine(TraceArgs A) {
    const NodeSlot slot = node_slot<RRT_BLOCK>(A);
    const uint32_t i = slot.i;
    if (!slot.active) return;
    float4 g = node_load(&A.cur.glob[i]);
    uint32_t flags = __float_as_uint(g.w);
    const float depth = g.z;
    f3 colour = mk3(0.0f, 0.0f, 0.0f);
    if (flags & NF_HIT) {
        float4 on = node_load(&A.cur.once[i]);
        f3 once = mk3(on.x, on.y, on.z);
        int2 ch = A.cur.child[i];
        f3 ca = mk3(0, 0, 0), cb = mk3(0, 0, 0);
        if (ch.x >= 0) {
            float4 r = A.next.result[ch.x];
            ca = mk3(r.x, r.y, r.z);
        }
        if (ch.y >= 0) {
            float4 r = A.next.result[ch.y];
            cb = mk3(r.x, r.y, r.z);
        }
        if (A.mode == RRT_MODE_SCENE) {
            float wr = g.x, wt = g.y, div = (float)(flags >> NF_PARTS_SHIFT), kr = on.w;
            f3 total = scene_node_total(once, flags, wr, wt, div, kr, ch.x >= 0, ca, ch.y >= 0, cb);
            colour = total;
            for (uint32_t l = 0; l < A.S.n_lights; l++) {  // scene.rs:116-121
                float4 t = node_load(&A.cur.lights[(size_t)l * A.cur.capacity + i]);
                if (t.w != 0.0f) colour = colour + mk3(t.x, t.y, t.z);
            }
        } else {  // photon_scene.rs:336-355
            float rw = g.x, sw = g.y;
            f3 refl = mk3(0, 0, 0), refr = mk3(0, 0, 0);
            if (ch.x >= 0) refl = ca * rw;
            if (ch.y >= 0) refr = cb * rw;
            colour = once + refl + refr;
            colour = colour / (sw + rw + rw);
        }
    }
    if (A.level == 0) {  // (nobody reads level 0's result: the colour goes to the frame)
        uint32_t px, py;
        uint32_t pixel = A.cam.width ? frame_pixel(A, A.pixel_base + i, px, py) : A.pixel_base + i;
        write_pixel(A, pixel, colour, depth, A.cur.object_id[i]);
    } else {
        A.cur.result[i] = make_float4(colour.x, colour.y, colour.z, depth);
    }
}

// ---- batched Environment::trace / Scene::shadowtrace for the C ABI's test entry points ----
#ifndef RRT_BATCH_WAVES
#define RRT_BATCH_WAVES 1
#endif
__global__ __launch_bounds__(RRT_BLOCK, RRT_BATCH_WAVES) void k_batch_trace(BatchTraceArgs A) {
    Lds lds = make_lds(A.S, A.lds_slots);
    uint32_t i = blockIdx.x * RRT_BLOCK + threadIdx.x;
    Counters cnt = {0, 0, 0, 0};
    if (i < A.n) {
        f3 o = mk3(A.origins[i * 3], A.origins[i * 3 + 1], A.origins[i * 3 + 2]);
        f3 d = mk3(A.directions[i * 3], A.directions[i * 3 + 1], A.directions[i * 3 + 2]);
        TraceResult tr;
        trace(A.S, o, d, lds, cnt, tr);
        bool ok = tr.object != 0xFFFFFFFFu;
        HitAttr h;
        h.position = h.normal = mk3(0, 0, 0);
        h.u = h.v = 0.0f;
        h.has_tex = false;
        uint32_t prim = 0;
        if (ok) {
            eval_hit(A.S, o, d, tr.instr, tr.sub, tr.t, h);
            if ((uint32_t)h.material >= A.S.n_materials) {  // index guard of eval_hit tripped
                cnt.overflow |= RRT_ERR_BIT_INTERNAL;
                ok = false;
            }
        }
        if (ok) {
            const ProgRec &in = A.S.prog[tr.instr];
            uint32_t op = in.op & 0xFFu;
            if (in.op & PF_CSG) {  // leaf number in post-order (a mesh leaf too: the leaf, not its triangle)
                uint32_t leaf = 0;
                for (uint32_t k = tr.instr; k > 0 && !(A.S.prog[k].op & PF_FIRST);) {
                    k--;
                    if ((A.S.prog[k].op & 0xFFu) < OP_CSG_UNION) leaf++;
                }
                prim = leaf;
            } else if (op == OP_MESH) {
                const MeshInfo &M = A.S.meshes[in.aux];
                prim = A.S.tri_index[M.tri_first + tr.sub];
            } else if (op == OP_CUBOID) {
                prim = tr.sub;
            }
        }
        if (A.object_id) A.object_id[i] = tr.object;
        if (A.prim_id) A.prim_id[i] = prim;
        if (A.distance) A.distance[i] = ok ? tr.t : 0.0f;
        if (A.position) {
            A.position[i * 3] = h.position.x; A.position[i * 3 + 1] = h.position.y; A.position[i * 3 + 2] = h.position.z;
        }
        if (A.normal) {
            A.normal[i * 3] = h.normal.x; A.normal[i * 3 + 1] = h.normal.y; A.normal[i * 3 + 2] = h.normal.z;
        }
        if (A.uv) {
            float nanv = __uint_as_float(0x7FC00000u);
            A.uv[i * 2] = (ok && h.has_tex) ? h.u : nanv;
            A.uv[i * 2 + 1] = (ok && h.has_tex) ? h.v : nanv;
        }
    }
    flush_counters(cnt, 0, 0, 0, A.stats, A.err);
}

__global__ __launch_bounds__(RRT_BLOCK, RRT_BATCH_WAVES) void k_batch_shadow(BatchTraceArgs A) {
    Lds lds = make_lds(A.S, A.lds_slots);
    uint32_t i = blockIdx.x * RRT_BLOCK + threadIdx.x;
    Counters cnt = {0, 0, 0, 0};
    if (i < A.n) {
        f3 o = mk3(A.origins[i * 3], A.origins[i * 3 + 1], A.origins[i * 3 + 2]);
        f3 d = mk3(A.directions[i * 3], A.directions[i * 3 + 1], A.directions[i * 3 + 2]);
        A.occluded[i] = shadowtrace(A.S, o, d, A.limits[i], lds, cnt) ? 1 : 0;
    }
    flush_counters(cnt, 0, 0, 0, A.stats, A.err);
}

__global__ void k_load_rays(const float *origins, const float *directions, uint32_t n, LevelBuffers lv) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    lv.ray_o[i] = make_float4(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2], __uint_as_float(i));
    lv.ray_d[i] = make_float4(directions[i * 3], directions[i * 3 + 1], directions[i * 3 + 2], 0.0f);
}

// ---- what a rank sends to rank 0, and what rank 0 makes of it (multi-GPU frames) -------------------------
// framebuffer.rs:105-107,124: `(c * 255.0) as u8`, Rust's saturating cast (NaN -> 0)
__device__ __forceinline__ uint8_t quantise_u8(float v) {
    const float f = v * 255.0f;
    return !(f > 0.0f) ? (uint8_t)0 : (f >= 255.0f ? (uint8_t)255 : (uint8_t)f);
}
// a share's pixels as the reference would finally write them (P6 + P5 payloads), 4 bytes r g b d per pixel instead of 20
__global__ void k_quantise_pack(const float *rgb, const float *depth, uint32_t n, uchar4 *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = make_uchar4(quantise_u8(rgb[3 * (size_t)i]), quantise_u8(rgb[3 * (size_t)i + 1]), quantise_u8(rgb[3 * (size_t)i + 2]), quantise_u8(depth[i]));
}
// the ranks' packed shares [world][rows_max][width] -> the frame's P6 and P5 payloads in image order.  Which rank rendered
// image row y and as which of its rows follows from the split alone (rrt_render_interleaved_device: groups of row_group rows
// dealt round-robin; row_group = 0: the contiguous bands of camera.rs:14-31, height / world rows each, the rest on the last).
__global__ void k_assemble_frame(const uchar4 *parts, uint32_t world, uint32_t rows_max, uint32_t width, uint32_t height, uint32_t row_group,
                                 uint8_t *rgb, uint8_t *depth) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= width || y >= height) return;
    uint32_t rank, local;
    if (row_group) {
        const uint32_t g = y / row_group;
        rank = g % world;
        local = (g / world) * row_group + y % row_group;
    } else {
        const uint32_t per = height / world;
        rank = per ? min(y / per, world - 1u) : world - 1u;
        local = y - rank * per;
    }
    const uchar4 v = parts[((size_t)rank * rows_max + local) * width + x];
    const size_t at = (size_t)y * width + x;
    rgb[3 * at] = v.x;
    rgb[3 * at + 1] = v.y;
    rgb[3 * at + 2] = v.z;
    depth[at] = v.w;
}
void launch_quantise_pack(const float *rgb, const float *depth, uint32_t n, uint8_t *out, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_quantise_pack, dim3((n + 255u) / 256u), dim3(256), 0, stream, rgb, depth, n, reinterpret_cast<uchar4 *>(out));
}
void launch_assemble_frame(const uint8_t *parts, uint32_t world, uint32_t rows_max, uint32_t width, uint32_t height, uint32_t row_group, uint8_t *rgb,
                           uint8_t *depth, hipStream_t stream) {
    if (width == 0 || height == 0) return;
    hipLaunchKernelGGL(k_assemble_frame, dim3((width + 255u) / 256u, height), dim3(256), 0, stream, reinterpret_cast<const uchar4 *>(parts), world, rows_max,
                       width, height, row_group, rgb, depth);
}

static inline uint32_t blocks_for(uint32_t n) { return (n + RRT_BLOCK - 1) / RRT_BLOCK; }
// grid of a level's launch: `blocks` blocks (level 0), or that many per stripe in groups of eight (node_slot)
static inline uint32_t stripe_blocks(uint32_t blocks, uint32_t striped) { return striped ? (blocks + 7u) / 8u * 8u * RRT_STRIPES : blocks; }

void launch_trace_shade(const TraceArgs &a, size_t lds_bytes, hipStream_t stream) {
    if (a.count == 0) return;
    const uint32_t per_block = a.striped ? 64u >> a.lane_shift : 64u;  // rays per block (node_slot)
    // one instance per (mode, scene feature set): see SceneFeatures in dev_trace.h
    int feat = (a.S.has_csg ? FEAT_CSG : 0) | (a.S.n_meshes ? FEAT_MESH : 0);
    if (feat == FEAT_MESH && a.S.no_solids) feat |= FEAT_NO_SOLIDS;  // planes and meshes only
    // The hit lists / BVH stacks are laid out [element][thread]: the bytes scale with the block.
    // Block size: 64 threads.  A block's registers and LDS return to the pool when its LAST wave is done, and rays differ
    // widely in cost (BVH depth, photon gathers), so one-wave blocks keep the SIMDs fuller: 4K mesh frame 1.35 -> 1.21 ms,
    // with mirror / glass teapots +4 / +9 %, photon-mapped scene2 +2 % (+10 % with three frames in flight).  Every wave then
    // pays an atomic of its own on the queue counter, which used to make the raytraced CSG scene the exception (most of its
    // pixels append two children: 9.6 -> 7.6 Grays/s with one-wave blocks and ONE counter per level); with the queues
    // striped over RRT_STRIPES counters it gains as well (kernels 0.450 -> 0.437 ms per frame) and the 256-thread
    // instances are gone.
#define RRT_NARROW(M, F, U) \
    hipLaunchKernelGGL((k_trace_shade<M, F, 64, U>), dim3(stripe_blocks((a.count + per_block - 1u) / per_block, a.striped)), dim3(64), lds_bytes / 256 * 64, stream, a)
#define RRT_EITHER(WHICH, M, F)    \
    do {                           \
        if (a.fuse)                \
            WHICH(M, F, true);     \
        else                       \
            WHICH(M, F, false);    \
    } while (0)
    if (a.mode == RRT_MODE_SCENE) {
        switch (feat) {
            case 0: RRT_EITHER(RRT_NARROW, RRT_MODE_SCENE, 0); break;
            case FEAT_CSG: RRT_EITHER(RRT_NARROW, RRT_MODE_SCENE, FEAT_CSG); break;
            case FEAT_MESH: RRT_EITHER(RRT_NARROW, RRT_MODE_SCENE, FEAT_MESH); break;
            case FEAT_MESH | FEAT_NO_SOLIDS: RRT_EITHER(RRT_NARROW, RRT_MODE_SCENE, FEAT_MESH | FEAT_NO_SOLIDS); break;
            default: RRT_EITHER(RRT_NARROW, RRT_MODE_SCENE, FEAT_ALL); break;
        }
    } else if (a.maps[0].div > 1u) {  // fine photon grids (dev_photon.h): the gather adds whole cells
        switch (feat) {
            case 0: RRT_NARROW(MODE_PHOTON_FINE, 0, false); break;
            case FEAT_CSG: RRT_NARROW(MODE_PHOTON_FINE, FEAT_CSG, false); break;
            case FEAT_MESH: RRT_NARROW(MODE_PHOTON_FINE, FEAT_MESH, false); break;
            case FEAT_MESH | FEAT_NO_SOLIDS: RRT_NARROW(MODE_PHOTON_FINE, FEAT_MESH | FEAT_NO_SOLIDS, false); break;
            default: RRT_NARROW(MODE_PHOTON_FINE, FEAT_ALL, false); break;
        }
    } else {
        switch (feat) {
            case 0: RRT_NARROW(RRT_MODE_PHOTON, 0, false); break;
            case FEAT_CSG: RRT_NARROW(RRT_MODE_PHOTON, FEAT_CSG, false); break;
            case FEAT_MESH: RRT_NARROW(RRT_MODE_PHOTON, FEAT_MESH, false); break;
            case FEAT_MESH | FEAT_NO_SOLIDS: RRT_NARROW(RRT_MODE_PHOTON, FEAT_MESH | FEAT_NO_SOLIDS, false); break;
            default: RRT_NARROW(RRT_MODE_PHOTON, FEAT_ALL, false); break;
        }
    }
#undef RRT_EITHER
#undef RRT_NARROW
}
void launch_tile_table(uint32_t *table, uint32_t n_tiles, uint32_t width, uint32_t rows, uint32_t xcd_order, hipStream_t stream) {
    if (n_tiles == 0) return;
    hipLaunchKernelGGL(k_tile_table, dim3((n_tiles + 255u) / 256u), dim3(256), 0, stream, table, n_tiles, width, rows, xcd_order);
}
void launch_gather_heavy(const TraceArgs &a, hipStream_t stream) {
    if (!a.heavy || a.mode != RRT_MODE_PHOTON) return;
    if (a.maps[0].div > 1u)
        hipLaunchKernelGGL(k_gather_heavy<true>, dim3(4096), dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL(k_gather_heavy<false>, dim3(4096), dim3(64), 0, stream, a);
}

void launch_combine(const TraceArgs &a0, hipStream_t stream) {
    if (a0.count == 0) return;
    TraceArgs a = a0;
    a.lane_shift = 0;  // (full blocks: the pass is bound by its bytes)
    hipLaunchKernelGGL(k_combine, dim3(stripe_blocks(blocks_for(a.count), a.striped)), dim3(RRT_BLOCK), 0, stream, a);
}
void launch_batch_trace(const BatchTraceArgs &a, size_t lds_bytes, hipStream_t stream) {
    if (a.n == 0) return;
    hipLaunchKernelGGL(k_batch_trace, dim3(blocks_for(a.n)), dim3(RRT_BLOCK), lds_bytes / 256 * RRT_BLOCK, stream, a);
}
void launch_batch_shadow(const BatchTraceArgs &a, size_t lds_bytes, hipStream_t stream) {
    if (a.n == 0) return;
    hipLaunchKernelGGL(k_batch_shadow, dim3(blocks_for(a.n)), dim3(RRT_BLOCK), lds_bytes / 256 * RRT_BLOCK, stream, a);
}
void launch_load_rays(const float *origins, const float *directions, uint32_t n, LevelBuffers lv, hipStream_t stream) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_load_rays, dim3(blocks_for(n)), dim3(RRT_BLOCK), 0, stream, origins, directions, n, lv);
}

}  // namespace rrt

// rrt_scene_create touches one kernel of every translation unit, so that the code objects are loaded before the first
// rrt_pre_render is timed (HIP loads them on first use: tens of milliseconds per unit)
namespace rrt {
void preload_kernels() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_combine));
}
}  // namespace rrt
