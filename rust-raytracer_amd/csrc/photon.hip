// photon.hip — the photon pass on gfx950: emission + bounce tracing into a global photon buffer
// (environments/photon_scene.rs:42-307, lights/{point_light,directional_point_light}.rs), and the
// build of the hash-grid index the render pass gathers from (replaces core/photon_tree.rs).
//
//  k_photon_shoot   one lane per photon in flight; the Russian-roulette path is a loop (every branch
//                   of PhotonScene::photontrace is a tail call), and a lane whose path has ended takes
//                   the next photon of its wave's run (the grid is what the GPU holds at once).  Each landing stores the absorbed
//                   photon plus one shadow photon per entering surface behind it; records are
//                   appended with one atomic per wave per destination (ballot + prefix popcount).
//                   RNG = Philox4x32-10 keyed by (seed, light, pass, photon index): the record SET
//                   is independent of wave scheduling, atomics order and GPU count; every record
//                   carries a unique key so canonical order can be restored by sorting.
//  index build      cell key per record -> rocPRIM radix sort (by record key, then stable by cell)
//                   -> gather into cell order -> open-addressing table cell -> (first, count).
#include <string.h>

#include <array>

#include <rocprim/device/device_radix_sort.hpp>

#include "dev_photon.h"
#include "dev_shade.h"
#include "runtime.h"

using namespace rrt;

namespace rrt {

#define RRT_BLOCK 256
extern __shared__ float4 g_smem_ph[];

// Records are appended with one atomic per wave and store call.  A single counter per buffer serialises in the L2
// (same-address atomics: ~4 ns each, tens of them per wave and bounce), so a call writes into PH_SEGMENTS staging
// segments, each with its own counter on its own cache line and picked by block index; k_compact_photons then copies
// the segments end to end behind the records the buffer already holds.  The order records land in is irrelevant:
// the index build sorts them by key.
#define PH_SEGMENTS 64
#define PH_COUNT_STRIDE 16  // counters 64 bytes apart
struct PhotonOut {
    float4 *pos, *dir, *rgb;
    uint64_t *meta;
    uint32_t *count;        // [PH_SEGMENTS * PH_COUNT_STRIDE]
    uint32_t seg_capacity;  // records per segment
};

struct ShootArgs {
    DevScene S;
    PhotonOut out[2];  // regular, caustic
    const float4 *seed_pos, *seed_dir;
    uint32_t n_seeds;
    uint64_t seed;
    uint64_t first;
    uint32_t count;
    uint32_t light, pass;
    uint32_t max_bounces;
    uint32_t lds_slots;
    uint32_t *err;  // bit 0 HitVec overflow, bit 2 shadow cap, bit 3 buffer overflow
    unsigned long long *stats;
};

#define PH_ERR_CAPACITY 8u

__device__ __forceinline__ uint32_t wave_append(bool want, uint32_t *counter) {
    unsigned long long mask = __ballot(want);
    if (!want) return 0xFFFFFFFFu;
    uint32_t lane = threadIdx.x & 63;
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader, 64);
    unsigned long long below = (lane == 0) ? 0ULL : (~0ULL >> (64 - lane));
    return base + (uint32_t)__popcll(mask & below);
}

__device__ __forceinline__ void store_photon(const PhotonOut &o, bool want, f3 p, f3 d, f3 c, uint32_t type, uint64_t meta, uint32_t &err) {
    const uint32_t seg = blockIdx.x % PH_SEGMENTS;
    uint32_t slot = wave_append(want, o.count + seg * PH_COUNT_STRIDE);
    if (!want) return;
    if (slot >= o.seg_capacity) {
        err |= PH_ERR_CAPACITY;
        return;
    }
    slot += seg * o.seg_capacity;
    o.pos[slot] = make_float4(p.x, p.y, p.z, __uint_as_float(type));
    o.dir[slot] = make_float4(d.x, d.y, d.z, 0.0f);
    o.rgb[slot] = make_float4(c.x, c.y, c.z, 0.0f);
    o.meta[slot] = meta;
}

// Three waves per SIMD (168 VGPRs, 400 bytes of spills per lane): the kernel is one long chain of traces per photon and
// waits on memory half of the time; left to itself the compiler takes all 256 registers and one wave per SIMD runs.
// Measured on scene2, 1 M photons: 1 / 2 / 3 / 4 waves 2.82 / 1.93 / 1.76 / 2.13 ms (the hit lists' LDS allows three);
// with lanes refilled from the wave's run of photons (below) 1.5 ms.
#ifndef RRT_SHOOT_WAVES
#define RRT_SHOOT_WAVES 3
#endif
__global__ __launch_bounds__(RRT_BLOCK, RRT_SHOOT_WAVES) void k_photon_shoot(ShootArgs A) {
    const DevScene &S = A.S;
    Lds lds;
    lds.nthreads = RRT_BLOCK;
    lds.tid = threadIdx.x;
    lds.le = reinterpret_cast<float2 *>(g_smem_ph);
    lds.stack = reinterpret_cast<int *>(g_smem_ph);
    lds.stack_base = (S.has_csg & 2u) ? A.lds_slots * RRT_LIST_CAP : 0u;  // see make_lds (kernels.hip)
    // Each wave owns a run of photons [next, end) and deals them to its lanes as they fall free: a path ends after 1.7
    // landings on average but the longest of 64 takes many more, and a lane whose photon is absorbed would idle until then
    // (34 % of the lanes were busy when a VALU instruction issued).  A photon's records depend on its index alone (RNG
    // stream, keys), so which lane traces it, and when, changes nothing.
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_waves = gridDim.x * (RRT_BLOCK / 64u), wave_id = blockIdx.x * (RRT_BLOCK / 64u) + (threadIdx.x >> 6);
    const uint32_t per_wave = (A.count + n_waves - 1u) / n_waves;
    uint32_t next = (uint32_t)min((uint64_t)wave_id * per_wave, (uint64_t)A.count);
    uint32_t end = (uint32_t)min((uint64_t)next + per_wave, (uint64_t)A.count);
    if (A.max_bounces == 0) end = next;  // no landing allowed: nothing is traced, nothing stored (as the bounce loop of photontrace would)
    bool alive = false;
    uint32_t bounce = 0;
    Counters cnt = {0, 0, 0, 0};
    uint32_t err = 0;
    uint64_t index = 0;
    const uint64_t tag = (uint64_t)(A.light * 2 + A.pass) << 56;
    const DevLight L = S.lights[A.light];
    Philox rng;
    rng_init(rng, A.seed, A.light, A.pass, 0);
    f3 origin = L.position, dir = mk3(0, 0, 1), intensity = L.intensity;
    const int dest_absorb_default = A.pass == 1 ? 1 : 0;
    for (;;) {
      // ---- free lanes take the wave's next photons (light.rs:28-72: emission) ----
      bool take = false;
      {
        const unsigned long long idle = __ballot(!alive);
        const uint32_t avail = end - next;
        if (idle && avail) {
            const uint32_t rank = (uint32_t)__popcll(idle & (lane == 0 ? 0ULL : (~0ULL >> (64 - lane))));
            take = !alive && rank < avail;
            if (take) index = A.first + next + rank;
            next += min((uint32_t)__popcll(idle), avail);
        }
      }
      if (take) {
        alive = true;
        bounce = 0;
        rng_init(rng, A.seed, A.light, A.pass, index);
        origin = L.position, intensity = L.intensity;
        f3 direction;
        if (A.pass == 0) {
            if (L.kind == RRT_LIGHT_POINT) {  // point_light.rs:61-70
                for (;;) {
                    float x = rng_sym(rng), y = rng_sym(rng), z = rng_sym(rng);
                    direction = mk3(x, y, z);
                    if (lensq3(direction) <= 1.0f) break;
                }
            } else {  // directional_point_light.rs:68 -> Vector::random_on_surface (vector.rs:44-52)
                direction = rng_unit_vector(rng);
                if (dot3(direction, L.direction) < 0.0f) direction = -direction;
            }
        } else {  // point_light.rs:115-121 / directional_point_light.rs:108-115
            uint32_t pick = (uint32_t)(((uint64_t)rng_u32(rng) * (uint64_t)A.n_seeds) >> 32);
            f3 offset = rng_unit_vector(rng) * 0.1f;
            if (L.kind == RRT_LIGHT_POINT) {
                float4 sd = A.seed_dir[pick];
                direction = mk3(sd.x, sd.y, sd.z) + offset;
            } else {
                float4 sp = A.seed_pos[pick];
                direction = (mk3(sp.x, sp.y, sp.z) - L.position) + offset;
            }
        }
        dir = norm3(direction);
      }
      if (!__any(alive)) break;  // and none left to take: `take` would have set a lane alive
      {
        // PhotonScene::photontrace, photon_scene.rs:150-186
        TraceResult tr;
        tr.object = 0xFFFFFFFFu;
        if (alive) trace(S, origin, dir, lds, cnt, tr);
        if (tr.object == 0xFFFFFFFFu) alive = false;
        HitAttr h;
        h.position = h.normal = mk3(0, 0, 0);
        h.material = 0;
        h.u = h.v = 0.0f;
        h.has_tex = false;
        int choice = 0;
        if (alive) {
            eval_hit(S, origin, dir, tr.instr, tr.sub, tr.t, h);
            if ((uint32_t)h.material >= S.n_materials) {  // index guard of eval_hit tripped
                err |= RRT_ERR_BIT_INTERNAL;
                alive = false;
                h.material = 0;
            }
            const DevMaterial &M = S.materials[h.material];
            // choose_weighted over [Absorb, Diffuse, Specular, ReflectOrRefract], photon_scene.rs:160-167
            float c1 = M.w[0], c2 = c1 + M.w[1], c3 = c2 + M.w[2], total = c3 + M.w[3];
            float x = rng_u01(rng) * total;
            choice = x < c1 ? 0 : (x < c2 ? 1 : (x < c3 ? 2 : 3));
        }
        // 56 bits: index (< 2^40) | landing (< 64) | sub-index (0 = the absorbed photon, 1..1023 = the shadow photons behind it)
        uint64_t key_base = tag | (index * 65536ull + (uint64_t)bounce * 1024ull);
        // absorb_photon (photon_scene.rs:188-200); retagged Caustic on ReflectOrRefract (:176)
        bool caustic = choice == 3;
        store_photon(A.out[0], alive && !(caustic || dest_absorb_default), h.position, dir, intensity, RRT_PHOTON_COLOUR, key_base, err);
        store_photon(A.out[1], alive && (caustic || dest_absorb_default), h.position, dir, intensity,
                     caustic ? RRT_PHOTON_CAUSTIC : RRT_PHOTON_COLOUR, key_base, err);
        // shadowphotontrace, photon_scene.rs:202-228: every entering hit with d >= 0 of every object
        {
            f3 so = h.position + dir * 0.0001f, sd = dir;
            uint32_t sub = 1;
            const PhotonOut &dst = A.out[dest_absorb_default];
            CsgState csg;
            RayBox rb;
            if (S.has_bounds) rb = make_raybox(so, sd);
            for (uint32_t k = 0; k < S.n_prog; k++) {
                const ProgRec &rec = S.prog[k];
                if ((rec.op & 0xFFu) == OP_BOUND) {
                    if (rec.op & PF_FIRST) csg.reset();
                    if (!__any(alive && bound_hit(rec, rb, so))) {
                        k += rec.aux;
                        if (rec.op & PF_CSG) csg.push_empty();
                    }
                    continue;
                }
                if (!(rec.op & PF_CSG)) {
                    float t = 0.0f;
                    uint32_t sb;
                    bool have = alive && single_entering_hit(S, rec, so, sd, RRT_INF, lds, cnt, t, sb);
                    if (have && sub >= 1024) {
                        err |= RRT_ERR_BIT_SHADOW_CAP;
                        have = false;
                    }
                    f3 p = isinf(t) ? mk3(0, 0, 0) : so + sd * t;
                    store_photon(dst, have, p, sd, mk3(0, 0, 0), RRT_PHOTON_SHADOW, key_base + sub, err);
                    if (have) sub++;
                } else {
                    if (rec.op & PF_FIRST) csg.reset();
                    if (alive) csg.step(S, rec, k, so, sd, lds, cnt);
                    if (!(rec.op & PF_LAST)) continue;
                    uint32_t slot = csg.out_slot(), n = alive ? csg.out_len() : 0;
                    for (uint32_t e = 0; e < RRT_LIST_CAP; e++) {
                        bool have = false;
                        float t = 0.0f;
                        if (e < n) {
                            uint32_t m;
                            lds.get(slot, e, t, m);
                            have = (m & META_ENTER) && !(t < 0.0f);
                        }
                        if (!__any(have)) continue;
                        if (have && sub >= 1024) {
                            err |= RRT_ERR_BIT_SHADOW_CAP;
                            have = false;
                        }
                        f3 p = isinf(t) ? mk3(0, 0, 0) : so + sd * t;
                        store_photon(dst, have, p, sd, mk3(0, 0, 0), RRT_PHOTON_SHADOW, key_base + sub, err);
                        if (have) sub++;
                    }
                }
            }
        }
        if (alive) {
            const DevMaterial &M = S.materials[h.material];
            if (choice == 0) {
                alive = false;
            } else if (choice == 1) {  // diffuse_photon, photon_scene.rs:230-251
                f3 d = rng_unit_vector(rng);
                if (dot3(d, h.normal) < 0.0f) d = -d;
                f3 bounced;
                material_bounced(S, M, h, dir, intensity, cnt, bounced);
                origin = h.position;
                dir = norm3(d);
                intensity = bounced;
            } else if (choice == 2) {  // specular_photon, photon_scene.rs:253-269
                f3 refl = norm3(reflect3(h.normal, dir));
                f3 bounced;
                material_bounced(S, M, h, dir, intensity, cnt, bounced);
                origin = h.position;
                dir = refl;
                intensity = bounced;
            } else {  // reflect_or_refract_photon, photon_scene.rs:271-307
                f3 refl = norm3(reflect3(h.normal, dir));
                bool refract = false;
                f3 ro, rd;
                if (M.global_part >= 0) {
                    const MatPart &G = M.parts[M.global_part];
                    float kr;
                    if (global_refraction(G, h, true, dir, ro, rd, kr)) {
                        // refract_chance: global_material.rs:129-133, compound_material.rs:148-152
                        float reflect = G.wr * kr;
                        float refr = G.wt * (1.0f - kr);
                        float chance = refr / (reflect + refr);
                        if (M.is_compound) chance = (0.0f + chance) / (float)M.n_parts;
                        refract = rng_u01(rng) < chance;  // gen_bool
                    }
                }
                if (refract) {
                    origin = ro;
                    dir = rd;
                } else {
                    origin = h.position + refl * 0.0001f;
                    dir = refl;
                }
            }
        }
        // a photon lands at most max_bounces times (the cap of the reference's unbounded recursion, rrt.h)
        if (alive && ++bounce >= A.max_bounces) {
            alive = false;
            cnt.truncated++;  // the reference would have traced on (photon_scene.rs:171-179): rrt_stats.paths_truncated
        }
      }
    }
    unsigned long long n = cnt.nodes, t = cnt.tris, cut = cnt.truncated;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_xor(n, off, 64);
        t += __shfl_xor(t, off, 64);
        cut += __shfl_xor(cut, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *stats = A.stats + (blockIdx.x % RRT_STAT_COPIES) * RRT_STAT_STRIDE;
        if (n) atomicAdd(&stats[ST_NODES], n);
        if (t) atomicAdd(&stats[ST_TRIS], t);
        if (cut) atomicAdd(&stats[ST_TRUNCATED], cut);
    }
    err |= cnt.overflow;
    if (err) atomicOr(A.err, err);
}

// ---- index build kernels ------------------------------------------------------------------------------
// Everything the build decides (dense region, record counts, table size) stays on the device in an IndexHeader; kernels that
// depend on such a number are launched for its upper bound and leave early.
#define PH_OUTLIER_BIT (1ULL << 63)

// canonical sort key of a record: its key with the (light, pass) tag moved down next to the bits in use, so that the radix
// sort passes over `low_bits + tag bits` instead of 64
__global__ void k_meta_keys(const uint64_t *meta, uint32_t n, uint32_t low_bits, uint64_t *keys, uint32_t *idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t m = meta[i];
    keys[i] = low_bits >= 56 ? m : ((m & ((1ULL << low_bits) - 1ULL)) | ((m >> 56) << low_bits));
    idx[i] = i;
}
__global__ void k_meta_or(const uint64_t *meta, uint64_t n, unsigned long long *acc) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = i < n ? meta[i] : 0ULL;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicOr(acc, v);
}

// The dense grids.  Photon maps have heavy tails: the reference's infinite planes catch photons (and, behind every
// landing, shadow photons) thousands of units away, while the queries of a frame fall where the scene is; and they are
// mostly surfaces, so one box over everything that matters would be nearly all empty cells.  From a strided sample of the
// positions (canonical order): a grid's box starts as the box of (all but the outermost 0.2 % per side of) the samples no
// earlier grid holds and is trimmed face by face — always the face whose outer 20 % slab holds the fewest samples — until it
// fits the cell budget (4 cells per record of the map, between 2^20 and 2^24 cells per grid): sparse tails and empty heights
// go first, the dense part stays.  Grids are added until fewer than 3 % of the samples are left over (or max_grids).
// Deterministic in the record set, and only performance depends on it: records outside every box go to the hash grid.
// One block; every thread carries the same box in registers, the samples live in LDS.
#define PH_SAMPLES 2048
#define PH_CHOOSE_THREADS 256
#define PH_PER_THREAD (PH_SAMPLES / PH_CHOOSE_THREADS)
// sums of six per-thread counts over the block: shuffles inside the waves, one LDS word per wave and count, ONE barrier (the
// buffers alternate, so the next reduction cannot overwrite what a slow thread is still reading)
__device__ __forceinline__ void block_sum6(uint32_t v[6], uint32_t (*part)[PH_CHOOSE_THREADS / 64][6], uint32_t &parity) {
#pragma unroll
    for (int f = 0; f < 6; f++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[f] += (uint32_t)__shfl_xor((int)v[f], off, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int f = 0; f < 6; f++) part[parity][threadIdx.x >> 6][f] = v[f];
    }
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 6; f++) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < PH_CHOOSE_THREADS / 64; w++) t += part[parity][w][f];
        v[f] = t;
    }
    parity ^= 1u;
}
__global__ __launch_bounds__(PH_CHOOSE_THREADS) void k_choose_dense(const float4 *pos, const uint32_t *order, uint32_t n, float cell, float inv_dcell,
                                                                    uint32_t div, uint32_t max_grids, double budget, IndexHeader *H) {
    __shared__ float s_v[3][PH_SAMPLES];
    __shared__ uint32_t s_part[2][PH_CHOOSE_THREADS / 64][6];
    const uint32_t tid = threadIdx.x;
    const uint32_t n_samples = min(n, (uint32_t)PH_SAMPLES), stride = max(1u, n / n_samples);
    uint32_t parity = 0;
    const uint64_t t_start = wall_clock64();
    uint64_t t_loaded = t_start, t_sort = 0, t_trim = 0, t_mark;
    // sample tid + j * threads lives in this thread's registers
    float px[PH_PER_THREAD], py[PH_PER_THREAD], pz[PH_PER_THREAD];
    uint32_t alive = 0;  // bit j
    uint32_t cnt[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < PH_PER_THREAD; j++) {
        const uint32_t k = tid + (uint32_t)j * PH_CHOOSE_THREADS;
        float4 p = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (k < n_samples) {
            p = pos[order[min((uint64_t)k * stride, (uint64_t)n - 1)]];
            if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) alive |= 1u << j;
        }
        px[j] = p.x, py[j] = p.y, pz[j] = p.z;
    }
    cnt[0] = (uint32_t)__popc(alive);
    block_sum6(cnt, s_part, parity);
    t_loaded = wall_clock64();
    if (cnt[0] == 0) {  // no finite position at all: one sample at the origin
        if (tid == 0) {
            px[0] = py[0] = pz[0] = 0.0f;
            alive = 1u;
        }
        cnt[0] = 1;
    }
    const uint32_t all = cnt[0];
    uint32_t left = all, n_grids = 0, dense_cells = 0, trim_steps = 0;
    const float inv = inv_dcell;  // (what the records are filed with)
    while (n_grids < max_grids && left > 0 && (n_grids == 0 || (uint64_t)left * 100u >= (uint64_t)all * 3u)) {
        // quantiles of the samples still without a grid: the three axes through one bitonic sort, the other samples at +inf
        t_mark = wall_clock64();
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PH_PER_THREAD; j++) {
            const uint32_t k = tid + (uint32_t)j * PH_CHOOSE_THREADS;
            const bool on = (alive >> j) & 1u;
            s_v[0][k] = on ? px[j] : INFINITY;
            s_v[1][k] = on ? py[j] : INFINITY;
            s_v[2][k] = on ? pz[j] : INFINITY;
        }
        for (uint32_t size = 2; size <= PH_SAMPLES; size <<= 1)
            for (uint32_t step = size >> 1; step > 0; step >>= 1) {
                __syncthreads();
                // PH_SAMPLES / 2 pairs over the threads; all loads first, then all stores (no branch in between: the twelve
                // LDS round trips of a step overlap instead of queueing up behind each other)
                float x[PH_PER_THREAD / 2][3], y[PH_PER_THREAD / 2][3];
#pragma unroll
                for (int q = 0; q < PH_PER_THREAD / 2; q++) {
                    const uint32_t t = tid + (uint32_t)q * PH_CHOOSE_THREADS;
                    const uint32_t i = 2u * t - (t & (step - 1u));
#pragma unroll
                    for (int a = 0; a < 3; a++) x[q][a] = s_v[a][i], y[q][a] = s_v[a][i + step];
                }
#pragma unroll
                for (int q = 0; q < PH_PER_THREAD / 2; q++) {
                    const uint32_t t = tid + (uint32_t)q * PH_CHOOSE_THREADS;
                    const uint32_t i = 2u * t - (t & (step - 1u));
                    const bool up = (i & size) == 0;
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        const float mn = fminf(x[q][a], y[q][a]), mx = fmaxf(x[q][a], y[q][a]);  // (no NaNs here: dead samples are +inf)
                        s_v[a][i] = up ? mn : mx;
                        s_v[a][i + step] = up ? mx : mn;
                    }
                }
            }
        __syncthreads();
        double lo[3], hi[3];
        {
            const uint32_t m = left - 1u;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                lo[a] = (double)s_v[a][(uint32_t)(0.002 * (double)m)];
                hi[a] = (double)s_v[a][(uint32_t)(0.998 * (double)m)];
            }
        }
        t_sort += wall_clock64() - t_mark;
        t_mark = wall_clock64();
        auto cells_of = [&](int a) { return fmax(1.0, ceil((hi[a] - lo[a]) / (double)cell) + 4.0); };
        double nc[3] = {cells_of(0), cells_of(1), cells_of(2)};  // cells along each axis of the box as it stands
        for (int it = 0; it < 400 && nc[0] * nc[1] * nc[2] > budget; it++) {
            double cut[6];  // new position of face 2a (low) / 2a + 1 (high)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                cut[2 * a] = lo[a] + 0.2 * (hi[a] - lo[a]);
                cut[2 * a + 1] = hi[a] - 0.2 * (hi[a] - lo[a]);
            }
            // samples of the box that each new face would leave outside: six counts of at most 8 per thread and 512 per wave,
            // ten bits each in two words through the wave's shuffles
            uint32_t pk0 = 0, pk1 = 0;
#pragma unroll
            for (int j = 0; j < PH_PER_THREAD; j++) {
                const double x = (double)px[j], y = (double)py[j], z = (double)pz[j];
                const bool in = ((alive >> j) & 1u) && x >= lo[0] && x <= hi[0] && y >= lo[1] && y <= hi[1] && z >= lo[2] && z <= hi[2];
                pk0 += ((in && x < cut[0]) ? 1u : 0u) + ((in && x > cut[1]) ? 1u << 10 : 0u) + ((in && y < cut[2]) ? 1u << 20 : 0u);
                pk1 += ((in && y > cut[3]) ? 1u : 0u) + ((in && z < cut[4]) ? 1u << 10 : 0u) + ((in && z > cut[5]) ? 1u << 20 : 0u);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                pk0 += (uint32_t)__shfl_xor((int)pk0, off, 64);
                pk1 += (uint32_t)__shfl_xor((int)pk1, off, 64);
            }
            if ((tid & 63u) == 0) {
                s_part[parity][tid >> 6][0] = pk0;
                s_part[parity][tid >> 6][1] = pk1;
            }
            __syncthreads();
            uint32_t loss[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int w = 0; w < PH_CHOOSE_THREADS / 64; w++) {
                const uint32_t a0 = s_part[parity][w][0], a1 = s_part[parity][w][1];
                loss[0] += a0 & 1023u, loss[1] += (a0 >> 10) & 1023u, loss[2] += a0 >> 20;
                loss[3] += a1 & 1023u, loss[4] += (a1 >> 10) & 1023u, loss[5] += a1 >> 20;
            }
            parity ^= 1u;
            // the face that loses the fewest samples (the first of equals), among the axes that are not thin already; written
            // with constant indices only: an array indexed by `best` would live in scratch memory, a few microseconds per step
            int best = -1;
            uint32_t best_loss = 0;
#pragma unroll
            for (int f = 0; f < 6; f++) {
                if (nc[f / 2] <= 8.0) continue;  // an axis this thin is not what makes the grid large
                if (best < 0 || loss[f] < best_loss) {
                    best = f;
                    best_loss = loss[f];
                }
            }
            if (best < 0) break;
#pragma unroll
            for (int f = 0; f < 6; f++)
                if (f == best) {
                    if (f & 1)
                        hi[f / 2] = cut[f];
                    else
                        lo[f / 2] = cut[f];
                    nc[f / 2] = cells_of(f / 2);
                }
            trim_steps++;
        }
        t_trim += wall_clock64() - t_mark;
        double dims[3];
        for (int a = 0; a < 3; a++) {
            lo[a] -= 2.0 * (double)cell;
            dims[a] = cells_of(a);
        }
        while (dims[0] * dims[1] * dims[2] > budget) {  // (only if the loop above ran out of iterations)
            int a = dims[0] >= dims[1] && dims[0] >= dims[2] ? 0 : (dims[1] >= dims[2] ? 1 : 2);
            dims[a] = fmax(1.0, floor(dims[a] / 2.0));
        }
        DenseGrid D;
        D.lo_x = (float)lo[0], D.lo_y = (float)lo[1], D.lo_z = (float)lo[2];
        // (the box is chosen in cells of edge r whatever the grid's own cell size, so that it does not depend on `div`)
        D.nx = (uint32_t)dims[0] * div, D.ny = (uint32_t)dims[1] * div, D.nz = (uint32_t)dims[2] * div;
        D.cell_base = dense_cells;
        D.pad = 0;
        dense_cells += D.nx * D.ny * D.nz;
        if (tid == 0) H->grid[n_grids] = D;
        n_grids++;
        // the samples this grid does not hold, filed like the records will be
        uint32_t rest[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < PH_PER_THREAD; j++) {
            if (!((alive >> j) & 1u)) continue;
            const float fx = dense_coord(px[j], D.lo_x, inv), fy = dense_coord(py[j], D.lo_y, inv), fz = dense_coord(pz[j], D.lo_z, inv);
            if (fx >= 0.0f && fx < (float)D.nx && fy >= 0.0f && fy < (float)D.ny && fz >= 0.0f && fz < (float)D.nz)
                alive &= ~(1u << j);
            else
                rest[0]++;
        }
        block_sum6(rest, s_part, parity);
        if (rest[0] == left) break;  // a box that holds none of its samples: stop rather than repeat it
        left = rest[0];
    }
    if (tid == 0) {
        for (uint32_t g = n_grids; g < RRT_DENSE_GRIDS; g++) H->grid[g] = DenseGrid{0.0f, 0.0f, 0.0f, 0u, 0u, 0u, 0u, 0u};
        H->n_grids = n_grids;
        H->dense_cells = dense_cells;
        H->n_dense = 0;
        H->out_cells = 0;
        H->table_size = 0;
        H->error = 0;
        H->trim_steps = trim_steps;
        H->pad = 0;
        H->choose_ticks[0] = (uint32_t)(t_loaded - t_start), H->choose_ticks[1] = (uint32_t)t_sort, H->choose_ticks[2] = (uint32_t)t_trim;
        H->choose_ticks[3] = (uint32_t)(wall_clock64() - t_start);
        H->crowding_sum = 0;
    }
}

// the grid a position is filed in (the first whose box holds it) and its cell there; false: an outlier
__device__ __forceinline__ bool dense_cell_in(const IndexHeader &H, float inv_dcell, float x, float y, float z, uint32_t &cell) {
    for (uint32_t g = 0; g < H.n_grids; g++) {
        const DenseGrid &D = H.grid[g];
        const float fx = dense_coord(x, D.lo_x, inv_dcell), fy = dense_coord(y, D.lo_y, inv_dcell), fz = dense_coord(z, D.lo_z, inv_dcell);
        if (fx >= 0.0f && fx < (float)D.nx && fy >= 0.0f && fy < (float)D.ny && fz >= 0.0f && fz < (float)D.nz) {
            cell = D.cell_base + ((uint32_t)fz * D.ny + (uint32_t)fy) * D.nx + (uint32_t)fx;
            return true;
        }
    }
    return false;
}
// sort key of the i-th record in canonical order: its dense cell index (x fastest), or bit 63 + its hash cell for records
// outside the dense region
__global__ void k_cell_keys(const float4 *pos, const uint32_t *order, uint32_t n, const IndexHeader *Hp, float inv_cell, float inv_dcell, uint64_t *keys) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pos[order[i]];
    uint32_t cell;
    if (dense_cell_in(*Hp, inv_dcell, p.x, p.y, p.z, cell))
        keys[i] = cell;
    else
        keys[i] = PH_OUTLIER_BIT | cell_key(cell_coord(p.x, inv_cell), cell_coord(p.y, inv_cell), cell_coord(p.z, inv_cell));
}
// start[c] = first record (of the sorted keys) whose dense cell is >= c, for c = 0 .. cells; n_dense = first outlier.
// A block's 256 consecutive cells share the part of the keys their answers lie in: two threads find it with a search over all
// records, the others search inside it (a handful of steps instead of log2 n).
__device__ __forceinline__ uint32_t first_at_least(const uint64_t *keys, uint32_t lo, uint32_t hi, uint64_t want) {
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (keys[mid] < want)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}
__global__ __launch_bounds__(256) void k_dense_starts(const uint64_t *keys, uint32_t n, IndexHeader *H, uint32_t *start) {
    __shared__ uint32_t s_range[2];
    const uint32_t cells = H->dense_cells;
    const uint32_t c0 = blockIdx.x * 256u;
    if (c0 > cells) return;  // (block-uniform)
    auto key_of = [&](uint32_t c) { return c >= cells ? PH_OUTLIER_BIT : (uint64_t)c; };  // every dense key is below `cells`
    if (threadIdx.x < 2) s_range[threadIdx.x] = first_at_least(keys, 0, n, key_of(threadIdx.x == 0 ? c0 : min(c0 + 255u, cells)));
    __syncthreads();
    const uint32_t c = c0 + threadIdx.x;
    if (c > cells) return;
    const uint32_t lo = first_at_least(keys, s_range[0], s_range[1], key_of(c));
    start[c] = lo;
    if (c == cells) {
        H->n_dense = lo;
        if (lo > n) H->error |= 1u;
    }
}
// sum of count^2 over the dense cells, into 64 line-separated partial sums
__global__ void k_dense_crowding(const uint32_t *start, const IndexHeader *H, unsigned long long *acc) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    if (c < H->dense_cells) {
        unsigned long long k = start[c + 1] - start[c];
        v = k * k;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&acc[(blockIdx.x & 63u) * 8u], v);
}
// occupied outlier cells = run starts in the outliers' part of the sorted keys.  One atomic per BLOCK into one of 64
// line-separated counters: an atomic per wave on a single address serialises in the L2 (640 us for 9 M records; this is 4 us)
#define PH_CELL_COUNTERS 64
#define PH_CELL_COUNTER_STRIDE 16
__global__ __launch_bounds__(256) void k_count_cells(const uint64_t *keys, uint32_t n, const IndexHeader *H, uint32_t *count) {
    __shared__ uint32_t s_part[4];
    const uint32_t base = min(H->n_dense, n);
    if ((uint64_t)blockIdx.x * blockDim.x >= n - base) return;  // (block-uniform)
    uint32_t i = base + blockIdx.x * blockDim.x + threadIdx.x;
    bool start = i < n && (i == base || keys[i] != keys[i - 1]);
    unsigned long long m = __ballot(start);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (total) atomicAdd(&count[(blockIdx.x % PH_CELL_COUNTERS) * PH_CELL_COUNTER_STRIDE], total);
    }
}
// the table of the outliers: a power of two of at least twice the occupied cells (the capacity allows for every record being one)
__global__ void k_table_size(IndexHeader *H, const uint32_t *count, uint32_t capacity) {
    if (blockIdx.x || threadIdx.x) return;
    uint32_t cells = 0;
    for (int k = 0; k < PH_CELL_COUNTERS; k++) cells += count[k * PH_CELL_COUNTER_STRIDE];
    uint32_t size = 64;
    while (size < capacity && size < 2u * cells) size <<= 1;
    H->out_cells = cells;
    H->table_size = size;
}
__global__ void k_fill_table(uint64_t *p, const IndexHeader *H) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < H->table_size) p[i] = ~0ULL;
}
// the outliers' runs -> table; the table holds absolute record indices
__global__ void k_insert_cells(const uint64_t *keys, uint32_t n, const IndexHeader *H, uint64_t *table_keys, uint2 *ranges) {
    const uint32_t base = min(H->n_dense, n);
    uint32_t i = base + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t key = keys[i];
    if (i != base && keys[i - 1] == key) return;
    // run length by binary search for the first index whose key differs
    uint32_t lo = i + 1, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (keys[mid] == key)
            lo = mid + 1;
        else
            hi = mid;
    }
    const uint64_t cell = key & ~PH_OUTLIER_BIT;  // what lookup_cell is asked for
    const uint32_t mask = H->table_size - 1u;
    uint32_t slot = hash_key(cell) & mask;
    for (;;) {
        unsigned long long prev = atomicCAS((unsigned long long *)&table_keys[slot], ~0ULL, (unsigned long long)cell);
        if (prev == ~0ULL) {
            ranges[slot] = make_uint2(i, lo - i);
            return;
        }
        slot = (slot + 1) & mask;
    }
}
// Fine grids: per dense cell the sums the gather adds as a whole (dev_photon.h) — normalised incidents and intensities of the
// cell's records in cell (= canonical) order; a cell of more than 256 records is summed by its whole wave (strided partial sums,
// fixed butterfly).  agg_dir.w = records of the cell; agg_rgb of an empty cell is not written (nor read).
__global__ __launch_bounds__(256) void k_cell_sums(const uint32_t *start, const IndexHeader *H, const float4 *dir, const float4 *rgb, float4 *agg_dir,
                                                   float4 *agg_rgb) {
    const uint32_t cells = H->dense_cells;
    if (blockIdx.x * 256u >= cells) return;  // (block-uniform)
    const uint32_t c = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u;
    uint32_t first = 0, count = 0;
    if (c < cells) {
        first = start[c];
        count = start[c + 1] - first;
    }
    f3 sd = mk3(0.0f, 0.0f, 0.0f), si = mk3(0.0f, 0.0f, 0.0f);
    const bool big = count > 256u;
    if (!big)
        for (uint32_t k = 0; k < count; k++) {
            const float4 d = dir[first + k], i = rgb[first + k];
            sd = sd + mk3(d.x, d.y, d.z);
            si = si + mk3(i.x, i.y, i.z);
        }
    unsigned long long todo = __ballot(big);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1ULL;
        const uint32_t f = (uint32_t)__shfl((int)first, src, 64), n = (uint32_t)__shfl((int)count, src, 64);
        f3 pd = mk3(0.0f, 0.0f, 0.0f), pi = mk3(0.0f, 0.0f, 0.0f);
        for (uint32_t k = lane; k < n; k += 64u) {
            const float4 d = dir[f + k], i = rgb[f + k];
            pd = pd + mk3(d.x, d.y, d.z);
            pi = pi + mk3(i.x, i.y, i.z);
        }
        pd = mk3(wave_sum_f32(pd.x), wave_sum_f32(pd.y), wave_sum_f32(pd.z));
        pi = mk3(wave_sum_f32(pi.x), wave_sum_f32(pi.y), wave_sum_f32(pi.z));
        if ((int)lane == src) {
            sd = pd;
            si = pi;
        }
    }
    if (c < cells) {
        agg_dir[c] = make_float4(sd.x, sd.y, sd.z, __uint_as_float(count));
        if (count) agg_rgb[c] = make_float4(si.x, si.y, si.z, 0.0f);
    }
}
__global__ void k_finish_header(IndexHeader *H, const unsigned long long *acc) {
    if (blockIdx.x || threadIdx.x) return;
    unsigned long long sum = 0;
    for (int k = 0; k < 64; k++) sum += acc[k * 8];
    H->crowding_sum = sum;
}
// the records in cell order, straight from the buffers as they were filled (`order` = canonical order, then stably by cell);
// `photon.incident.normalised()` of photon_scene.rs:405 is evaluated once here instead of per query
__global__ void k_permute_records(const float4 *pos, const float4 *dir, const float4 *rgb, const uint64_t *meta, const uint32_t *order, uint32_t n,
                                  float4 *pos_out, float4 *dir_out, float4 *rgb_out, uint64_t *meta_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = order[i];
    const float4 d = dir[j];
    const f3 v = norm3(mk3(d.x, d.y, d.z));
    pos_out[i] = pos[j];
    dir_out[i] = make_float4(v.x, v.y, v.z, 0.0f);
    rgb_out[i] = rgb[j];
    meta_out[i] = meta[j];
}
__global__ void k_iota(uint32_t *idx, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = i;
}
template <class T>
__global__ void k_gather(const T *src, const uint32_t *idx, uint32_t n, T *dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}
// `photon.incident.normalised()` of photon_scene.rs:405 is evaluated once here instead of per query
__global__ void k_gather_norm(const float4 *src, const uint32_t *idx, uint32_t n, float4 *dst) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 d = src[idx[i]];
    f3 v = norm3(mk3(d.x, d.y, d.z));
    dst[i] = make_float4(v.x, v.y, v.z, 0.0f);
}
__global__ void k_strip_key(const uint64_t *meta, uint32_t n, uint64_t *key) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) key[i] = meta[i] & 0x00FFFFFFFFFFFFFFull;
}

// PhotonTree::get_within_distance / get_n_within_radius (photon_tree.rs:15-50), batched: one lane per query point.
// Photons inside the closed cube, ascending by (squared distance, key); the first k_max are written out.
__global__ void k_photon_query(DevPhotonMap G, const uint64_t *meta, const float *positions, uint32_t n, float r, uint32_t k_max, uint32_t *count,
                               uint64_t *key_out, float *dist2_out) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    f3 p = mk3(positions[q * 3], positions[q * 3 + 1], positions[q * 3 + 2]);
    uint32_t found = 0;
    if (G.count) {
        f3 lo = p - mk3(r, r, r), hi = p + mk3(r, r, r);  // photon_tree.rs:16-18
        uint64_t *keys = key_out + (size_t)q * k_max;
        float *d2s = dist2_out + (size_t)q * k_max;
        visit_runs(G, lo, hi, [&](uint2 range) {
            for (uint32_t i = range.x; i < range.x + range.y && i < G.count; i++) {
                float4 c = G.pos_cell[i];
                if (!(c.x >= lo.x && c.x <= hi.x && c.y >= lo.y && c.y <= hi.y && c.z >= lo.z && c.z <= hi.z)) continue;
                float d2 = lensq3(mk3(c.x, c.y, c.z) - p);  // (photon.position.vector() - position.vector()).len_sqrd()
                uint64_t key = meta[i];
                uint32_t have = min(found, k_max);
                found++;
                // insertion into the sorted prefix kept so far (rows are private to the lane)
                uint32_t at = have;
                while (at > 0 && (d2s[at - 1] > d2 || (d2s[at - 1] == d2 && keys[at - 1] > key))) at--;
                if (at >= k_max) continue;
                for (uint32_t m = min(have, k_max - 1); m > at; m--) {
                    d2s[m] = d2s[m - 1];
                    keys[m] = keys[m - 1];
                }
                d2s[at] = d2;
                keys[at] = key;
            }
        });
    }
    count[q] = found;
}

struct SegmentPrefix {
    uint32_t count[PH_SEGMENTS], first[PH_SEGMENTS];
};
// staging segments -> contiguous records behind dst_base
__global__ void k_compact_photons(PhotonOut src, SegmentPrefix P, float4 *pos, float4 *dir, float4 *rgb, uint64_t *meta, uint64_t dst_base) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t seg = i / src.seg_capacity, l = i - seg * src.seg_capacity;
    if (seg >= PH_SEGMENTS || l >= P.count[seg]) return;
    uint64_t to = dst_base + P.first[seg] + l;
    pos[to] = src.pos[i];
    dir[to] = src.dir[i];
    rgb[to] = src.rgb[i];
    meta[to] = src.meta[i];
}

static inline dim3 grid_for(uint64_t n, uint32_t block = 256) { return dim3((uint32_t)((n + block - 1) / block)); }

// stable radix sort of (key, index) pairs with rocPRIM
static int sort_pairs(PhotonScratch &W, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *idx_in, uint32_t *idx_out, uint32_t n,
                      hipStream_t st) {
    size_t temp_bytes = 0;
    RRT_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, idx_in, idx_out, n, 0, 64, st));
    int rc = W.sort_temp.ensure(temp_bytes ? temp_bytes : 1);
    if (rc) return rc;
    RRT_HIP(rocprim::radix_sort_pairs(W.sort_temp.p, temp_bytes, keys_in, keys_out, idx_in, idx_out, n, 0, 64, st));
    return RRT_OK;
}

template <class T>
static int grow_preserve(DBuf<T> &b, size_t need, size_t used) {
    if (need <= b.cap) return RRT_OK;
    T *np = nullptr;
    RRT_HIP(hipMalloc((void **)&np, need * sizeof(T)));
    if (used && b.p) RRT_HIP(hipMemcpy(np, b.p, used * sizeof(T), hipMemcpyDeviceToDevice));
    if (b.p) (void)hipFree(b.p);
    b.p = np;
    b.cap = need;
    return RRT_OK;
}
static int grow_photons(PhotonBuffers &b, uint64_t need) {
    int rc;
    if (need > b.pos.cap && b.pos.cap) need = std::max<uint64_t>(need, b.pos.cap + b.pos.cap / 2);  // repeated appends: amortised copies
    if ((rc = grow_preserve(b.pos, need, b.count)) || (rc = grow_preserve(b.dir, need, b.count)) || (rc = grow_preserve(b.rgb, need, b.count)) ||
        (rc = grow_preserve(b.meta, need, b.count)))
        return rc;
    return RRT_OK;
}

// canonical order: sort the records of a buffer by meta (light, pass, key)
static int sort_by_meta(PhotonScratch &W, PhotonBuffers &b) {
    uint32_t n = (uint32_t)b.count;
    if (n < 2) return RRT_OK;
    int rc;
    // the sorted arrays replace the unsorted ones by pointer swap, so the work arrays are as large as the buffers
    size_t cap = std::max<size_t>(n, b.pos.cap);
    if ((rc = W.idx.ensure(n)) || (rc = W.idx2.ensure(n)) || (rc = W.keys2.ensure(std::max<size_t>(n, b.meta.cap))) || (rc = W.t.ensure(cap))) return rc;
    hipLaunchKernelGGL(k_iota, grid_for(n), dim3(256), 0, nullptr, W.idx.p, n);
    if ((rc = sort_pairs(W, b.meta.p, W.keys2.p, W.idx.p, W.idx2.p, n, nullptr))) return rc;
    std::swap(b.meta.p, W.keys2.p);
    std::swap(b.meta.cap, W.keys2.cap);
    DBuf<float4> *arrs[3] = {&b.pos, &b.dir, &b.rgb};
    for (auto *a : arrs) {
        if ((rc = W.t.ensure(std::max<size_t>(n, a->cap)))) return rc;
        hipLaunchKernelGGL(k_gather<float4>, grid_for(n), dim3(256), 0, nullptr, a->p, W.idx2.p, n, W.t.p);
        std::swap(a->p, W.t.p);
        std::swap(a->cap, W.t.cap);
    }
    return RRT_OK;
}

// stable radix sort of (key, index) pairs over the key bits [0, end_bit) on a stream, with the map's own temporary storage
static int sort_pairs_bits(IndexScratch &W, const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *idx_in, uint32_t *idx_out, uint32_t n,
                           uint32_t end_bit, hipStream_t st) {
    size_t temp_bytes = 0;
    RRT_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys_out, idx_in, idx_out, n, 0, end_bit, st));
    int rc = W.sort_temp.ensure(temp_bytes ? temp_bytes : 1);
    if (rc) return rc;
    RRT_HIP(rocprim::radix_sort_pairs(W.sort_temp.p, temp_bytes, keys_in, keys_out, idx_in, idx_out, n, 0, end_bit, st));
    return RRT_OK;
}
static uint32_t bit_length(uint64_t v) {
    uint32_t b = 0;
    while (v) {
        b++;
        v >>= 1;
    }
    return b;
}
// Cells of edge r per dense grid: 4 per record of the map, between 2^20 and 2^24.  A fine grid (cells of edge r / div) covers
// the same box with div^3 times as many cells, at most 2^27 of them (0.5 GB of start table, 4 GB of per-cell sums).
static double dense_cell_budget(uint32_t n, uint32_t div) {
    const uint64_t coarse = std::min<uint64_t>(std::max<uint64_t>(4ull * n, 1ull << 20), 1ull << 24);
    return (double)std::min<uint64_t>(coarse, (1ull << 27) / ((uint64_t)div * div * div));
}
// Dense cell edge = gather half width / div.  Fine cells pay once a query cube holds hundreds of photons (the gather then adds
// the cells inside the cube as a whole instead of streaming their records): maps of RRT_FINE_GRID_RECORDS records and more.
// Both maps of a scene get the same div (decided by the regular map).  RRT_DENSE_DIV overrides (1 = the per-record gather
// everywhere: the parity reference of the fine path).
#define RRT_FINE_GRID_RECORDS (24u << 20)
static uint32_t dense_div_for(uint64_t regular_records) {
    const char *env = getenv("RRT_DENSE_DIV");  // (read at every build: tools switch it between builds)
    if (env) return (uint32_t)std::min(4, std::max(1, atoi(env)));
    return regular_records >= RRT_FINE_GRID_RECORDS ? 3u : 1u;  // (100 M photons, 4K frame: 0.31 s with div 1, 0.16 / 0.13 / 0.14 s with 2 / 3 / 4)
}
static uint32_t dense_grids_max() {
    static const char *max_env = getenv("RRT_DENSE_GRIDS_MAX");  // experiments: more than one grid per map
    return max_env ? std::min<uint32_t>(std::max(1, atoi(max_env)), RRT_DENSE_GRIDS) : 1u;
}

// One map's index build, enqueued on the map's own stream; nothing here waits for the device.  The numbers the device
// decides on the way (the dense region, how many records it holds, the outliers' table size) stay in the IndexHeader; the
// kernels that need them are launched for their upper bounds.  finish_index reads the header once the stream has drained.
static int enqueue_index(rrt_scene *s, int which) {
    PhotonBuffers &b = s->local[which];
    PhotonIndex &X = s->index[which];
    IndexScratch &W = s->index_scratch[which];
    uint32_t n = (uint32_t)b.count;
    const float cell = s->params.gather_half_width;
    const uint32_t div = dense_div_for(s->local[0].count);
    X.count = n;
    X.div = div;
    X.inv_cell = 1.0f / (2.0f * cell);
    X.inv_dcell = (float)div / cell;
    X.table_size = 0;
    X.crowding = 0.0;
    X.n_dense = 0;
    X.n_grids = 0;
    X.dense_cells = 0;
    s->map_view[which] = X.view();
    if (n == 0) return RRT_OK;
    if (!(cell > 0.0f) || !std::isfinite(X.inv_dcell)) return fail(RRT_ERR_INVALID, "gather_half_width must be positive");
    int rc;
    hipStream_t st = W.stream;
    const bool timing = getenv("RRT_INDEX_INFO") != nullptr;  // debugging aid: host wall clock of the phases (each ends in a sync)
    auto t_phase = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        auto t = std::chrono::steady_clock::now();
        fprintf(stderr, "[index %d] %-22s %7.3f ms\n", which, what, std::chrono::duration<double, std::milli>(t - t_phase).count());
        t_phase = t;
    };
    const uint32_t max_grids = dense_grids_max();
    const double budget = dense_cell_budget(n, div);  // in cells of edge r
    const size_t max_cells = (size_t)budget * div * div * div * max_grids;
    size_t table_cap = 64;  // every record an outlier in a cell of its own, at most half full
    while (table_cap < 2 * (size_t)n) table_cap <<= 1;
    // (a sixteenth of slack where a buffer has to grow: the next pass of the same scene stores a few records more or less, and
    // a reallocation costs more than the build)
    auto roomy = [](size_t need, size_t have) { return need <= have ? need : need + need / 16 + 1024; };
    const size_t rn = roomy(n, X.pos.cap), wn = roomy(n, W.keys.cap);
    if ((rc = W.keys.ensure(wn)) || (rc = W.keys2.ensure(wn)) || (rc = W.idx.ensure(wn)) || (rc = W.idx2.ensure(wn)) ||
        (rc = W.ncell.ensure(PH_CELL_COUNTERS * PH_CELL_COUNTER_STRIDE)) || (rc = W.acc.ensure(512)) || (rc = W.header.ensure(1)) ||
        (rc = X.pos.ensure(rn)) || (rc = X.dir.ensure(rn)) || (rc = X.rgb.ensure(rn)) || (rc = X.meta.ensure(rn)) ||
        (rc = X.dense_start.ensure(roomy(max_cells + 1, X.dense_start.cap))) || (rc = X.keys.ensure(table_cap)) || (rc = X.ranges.ensure(table_cap)))
        return rc;
    if (div > 1 && ((rc = X.agg_dir.ensure(roomy(max_cells, X.agg_dir.cap))) || (rc = X.agg_rgb.ensure(roomy(max_cells, X.agg_rgb.cap))))) return rc;
    IndexHeader *H = W.header.p;
    // canonical order: the records by key (light, pass, photon index, landing, sub-index), over the key bits in use only
    const uint32_t low_bits = std::min<uint32_t>(56u, bit_length(s->meta_or & 0x00FFFFFFFFFFFFFFull)), tag_bits = bit_length(s->meta_or >> 56);
    hipLaunchKernelGGL(k_meta_keys, grid_for(n), dim3(256), 0, st, b.meta.p, n, low_bits, W.keys.p, W.idx.p);
    if ((rc = sort_pairs_bits(W, W.keys.p, W.keys2.p, W.idx.p, W.idx2.p, n, std::max(1u, low_bits + tag_bits), st))) return rc;
    lap("sort by key");
    hipLaunchKernelGGL(k_choose_dense, dim3(1), dim3(PH_CHOOSE_THREADS), 0, st, b.pos.p, W.idx2.p, n, cell, X.inv_dcell, div, max_grids, budget, H);
    lap("choose dense grids");
    // then stably by cell
    hipLaunchKernelGGL(k_cell_keys, grid_for(n), dim3(256), 0, st, b.pos.p, W.idx2.p, n, H, X.inv_cell, X.inv_dcell, W.keys.p);
    if ((rc = sort_pairs_bits(W, W.keys.p, W.keys2.p, W.idx2.p, W.idx.p, n, 64, st))) return rc;
    hipLaunchKernelGGL(k_permute_records, grid_for(n), dim3(256), 0, st, b.pos.p, b.dir.p, b.rgb.p, b.meta.p, W.idx.p, n, X.pos.p, X.dir.p, X.rgb.p,
                       X.meta.p);
    lap("cell sort + permute");
    // dense part: the start table, and where the outliers begin; outliers: occupied hash cells -> table
    hipLaunchKernelGGL(k_dense_starts, grid_for((uint64_t)max_cells + 1), dim3(256), 0, st, W.keys2.p, n, H, X.dense_start.p);
    RRT_HIP(hipMemsetAsync(W.ncell.p, 0, PH_CELL_COUNTERS * PH_CELL_COUNTER_STRIDE * sizeof(uint32_t), st));
    RRT_HIP(hipMemsetAsync(W.acc.p, 0, 512 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_count_cells, grid_for(n), dim3(256), 0, st, W.keys2.p, n, H, W.ncell.p);
    hipLaunchKernelGGL(k_table_size, dim3(1), dim3(64), 0, st, H, W.ncell.p, (uint32_t)std::min<size_t>(table_cap, 1u << 31));
    hipLaunchKernelGGL(k_fill_table, grid_for(table_cap), dim3(256), 0, st, X.keys.p, H);
    hipLaunchKernelGGL(k_insert_cells, grid_for(n), dim3(256), 0, st, W.keys2.p, n, H, X.keys.p, X.ranges.p);
    // how crowded the cells are where the photons are (sizes the default heavy-query threshold, rrt_photon_build)
    hipLaunchKernelGGL(k_dense_crowding, grid_for(max_cells), dim3(256), 0, st, X.dense_start.p, H, W.acc.p);
    hipLaunchKernelGGL(k_finish_header, dim3(1), dim3(64), 0, st, H, W.acc.p);
    if (div > 1)
        hipLaunchKernelGGL(k_cell_sums, grid_for(max_cells), dim3(256), 0, st, X.dense_start.p, H, X.dir.p, X.rgb.p, X.agg_dir.p, X.agg_rgb.p);
    RRT_HIP(hipGetLastError());
    lap("tables + crowding");
    RRT_HIP(hipMemcpyAsync(W.host_header, H, sizeof(IndexHeader), hipMemcpyDeviceToHost, st));
    return RRT_OK;
}
// after the map's stream has drained: what the device decided becomes the view the render kernels get
static int finish_index(rrt_scene *s, int which) {
    PhotonIndex &X = s->index[which];
    if (X.count == 0) return RRT_OK;
    const IndexHeader &H = *s->index_scratch[which].host_header;
    const uint32_t n = (uint32_t)X.count;
    if (H.error || H.n_dense > n || H.n_grids > RRT_DENSE_GRIDS || (H.table_size & (H.table_size - 1u)) || H.table_size < 64u || H.table_size > X.keys.cap)
        return fail(RRT_ERR_DEVICE, "photon index: inconsistent header (internal error)");
    for (uint32_t g = 0; g < RRT_DENSE_GRIDS; g++) X.grids[g] = H.grid[g];
    X.n_grids = H.n_grids;
    X.dense_cells = H.dense_cells;
    X.n_dense = H.n_dense;
    X.table_size = H.table_size;
    X.crowding = (double)H.crowding_sum / (double)n;
    if (getenv("RRT_INDEX_INFO")) {
        fprintf(stderr, "[index %d] %u records: %u in %u dense grids (cell %.3g, %u cells), %u outliers in %u hash cells (table %u); crowding %.1f; %u trimming steps (choice: load %.1f, sort %.1f, trim %.1f, all %.1f us)\n",
                which, n, X.n_dense, X.n_grids, s->params.gather_half_width, X.dense_cells, n - X.n_dense, H.out_cells, H.table_size, X.crowding, H.trim_steps,
                H.choose_ticks[0] * 0.01, H.choose_ticks[1] * 0.01, H.choose_ticks[2] * 0.01, H.choose_ticks[3] * 0.01);
        for (uint32_t g = 0; g < X.n_grids; g++)
            fprintf(stderr, "[index %d]   grid %u: %u x %u x %u from %.9g %.9g %.9g\n", which, g, X.grids[g].nx, X.grids[g].ny, X.grids[g].nz, X.grids[g].lo_x,
                    X.grids[g].lo_y, X.grids[g].lo_z);
    }
    s->map_view[which] = X.view();
    return RRT_OK;
}
static int ensure_index_streams(rrt_scene *s) {
    for (int w = 0; w < 2; w++) {
        IndexScratch &W = s->index_scratch[w];
        if (!W.stream) RRT_HIP(hipStreamCreateWithFlags(&W.stream, hipStreamNonBlocking));
        if (!W.done) RRT_HIP(hipEventCreateWithFlags(&W.done, hipEventDisableTiming));
        if (!W.host_header) RRT_HIP(hipHostMalloc((void **)&W.host_header, sizeof(IndexHeader), hipHostMallocDefault));
    }
    return RRT_OK;
}
// the key bits of records that did not come from rrt_photon_shoot (imports, appended segments, files)
static int note_foreign_keys(rrt_scene *s, const uint64_t *meta_dev, uint64_t count) {
    if (count == 0) return RRT_OK;
    int rc;
    if (!s->meta_or_dev.p) {
        if ((rc = s->meta_or_dev.ensure(1))) return rc;
        RRT_HIP(hipMemsetAsync(s->meta_or_dev.p, 0, sizeof(unsigned long long), nullptr));
    }
    hipLaunchKernelGGL(k_meta_or, grid_for(count), dim3(256), 0, nullptr, meta_dev, count, s->meta_or_dev.p);
    RRT_HIP(hipGetLastError());
    s->meta_or_pending = true;
    return RRT_OK;
}

static int rebuild_seeds(rrt_scene *s, int light, const float4 *pos, const float4 *dir, const uint64_t *key, uint64_t count) {
    CausticSeeds &cs = *s->seeds[light];
    uint32_t n = (uint32_t)count;
    cs.count = n;
    if (n == 0) return RRT_OK;
    int rc;
    PhotonScratch &W = s->photon_scratch;
    DBuf<uint32_t> &idx = W.idx, &idx2 = W.idx2;
    DBuf<uint64_t> &k2 = W.keys2;
    if ((rc = idx.ensure(n)) || (rc = idx2.ensure(n)) || (rc = k2.ensure(n))) return rc;
    if ((rc = cs.pos.ensure(n)) || (rc = cs.dir.ensure(n)) || (rc = cs.key.ensure(n))) return rc;
    hipLaunchKernelGGL(k_iota, grid_for(n), dim3(256), 0, nullptr, idx.p, n);
    if ((rc = sort_pairs(W, key, k2.p, idx.p, idx2.p, n, nullptr))) return rc;
    hipLaunchKernelGGL(k_gather<float4>, grid_for(n), dim3(256), 0, nullptr, pos, idx2.p, n, cs.pos.p);
    hipLaunchKernelGGL(k_gather<float4>, grid_for(n), dim3(256), 0, nullptr, dir, idx2.p, n, cs.dir.p);
    RRT_HIP(hipMemcpy(cs.key.p, k2.p, (size_t)n * 8, hipMemcpyDeviceToDevice));
    RRT_HIP(hipDeviceSynchronize());
    return RRT_OK;
}

}  // namespace rrt

extern "C" {

int rrt_photon_begin(rrt_scene *s, uint64_t seed) {
    if (!s) return fail(RRT_ERR_INVALID, "null scene");
    if (s->mode != RRT_MODE_PHOTON) return fail(RRT_ERR_STATE, "scene is not a PhotonScene");
    for (const rrt_light_desc &l : s->lights)
        if (l.kind == RRT_LIGHT_DIRECTIONAL) return fail(RRT_ERR_INVALID, "Light does not support photon mapping");  // light.rs:16-18
    int rc = flatten_scene(s);  // drops the render workers
    if (rc) return rc;
    s->seed = seed;
    s->maps_built = false;
    s->local[0].count = s->local[1].count = 0;
    s->meta_or = 0;
    s->meta_or_pending = false;
    if (s->meta_or_dev.p) RRT_HIP(hipMemset(s->meta_or_dev.p, 0, sizeof(unsigned long long)));
    s->seeds.clear();
    for (size_t i = 0; i < s->lights.size(); i++) s->seeds.push_back(std::make_unique<CausticSeeds>());
    s->stats.photons_emitted = 0;
    s->stats.paths_truncated = 0;
    s->stats.ms_photon_shoot = 0;
    s->stats.ms_photon_build = 0;
    return RRT_OK;
}

int rrt_photon_shoot(rrt_scene *s, int light, int pass, uint64_t first, uint64_t count) {
    if (!s) return fail(RRT_ERR_INVALID, "null scene");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (!s->flattened || s->seeds.size() != s->lights.size()) return fail(RRT_ERR_STATE, "rrt_photon_begin has not been called");
    if (light < 0 || light >= (int)s->lights.size() || pass < 0 || pass > 1) return fail(RRT_ERR_INVALID, "bad light/pass");
    if (count == 0) return RRT_OK;
    if (count >= (1ull << 31)) return fail(RRT_ERR_LIMIT, "more than 2^31 photons in one call");
    if (first + count >= (1ull << 40)) return fail(RRT_ERR_LIMIT, "photon index above 2^40");
    // key bits this call can set (index * 65536 + landing * 1024 + sub-index, tag above bit 56): the index build sorts over those
    s->meta_or |= ((uint64_t)(light * 2 + pass) << 56) | ((1ULL << bit_length((first + count - 1) * 65536ULL + 65535ULL)) - 1ULL);
    CausticSeeds &cs = *s->seeds[light];
    if (pass == 1 && cs.count == 0) {
        // point_light.rs:106-108 returns an empty Vec; directional_point_light.rs:110 would panic on unwrap
        if (s->lights[light].kind == RRT_LIGHT_DIRPOINT) return fail(RRT_ERR_INVALID, "no caustic photons to aim at (reference: unwrap on None)");
        return RRT_OK;
    }
    int rc;
    PhotonScratch &W = s->photon_scratch;
    if ((rc = W.seg_counts.ensure(2 * PH_SEGMENTS * PH_COUNT_STRIDE))) return rc;
    double est[2] = {pass == 0 ? 4.0 : 0.0, pass == 0 ? 1.0 : 5.0};
    hipEvent_t e0 = s->events[0], e1 = s->events[1];
    uint64_t got[2] = {0, 0};
    for (int attempt = 0;; attempt++) {
        if (attempt == 8) return fail(RRT_ERR_LIMIT, "photon staging buffers kept overflowing");
        uint32_t seg_cap[2];
        for (int w = 0; w < 2; w++) {
            uint64_t per_seg = (uint64_t)(est[w] * (double)count * 1.3 / PH_SEGMENTS) + 512;
            if (per_seg * PH_SEGMENTS >= (1ull << 32) - 4096) return fail(RRT_ERR_LIMIT, "photon buffer above 2^32 records per GPU");
            seg_cap[w] = (uint32_t)per_seg;
            size_t n = (size_t)per_seg * PH_SEGMENTS;
            if ((rc = W.stage[w].pos.ensure(n)) || (rc = W.stage[w].dir.ensure(n)) || (rc = W.stage[w].rgb.ensure(n)) || (rc = W.stage[w].meta.ensure(n)))
                return rc;
        }
        RRT_HIP(hipMemset(W.seg_counts.p, 0, 2 * PH_SEGMENTS * PH_COUNT_STRIDE * sizeof(uint32_t)));
        RRT_HIP(hipMemset(s->counters.p + 63, 0, 4));
        RRT_HIP(hipMemset(s->stats_dev.p, 0, RRT_STAT_WORDS * sizeof(unsigned long long)));
        ShootArgs A;
        memset(&A, 0, sizeof(A));
        A.S = s->dev.view;
        for (int w = 0; w < 2; w++) {
            A.out[w].pos = W.stage[w].pos.p;
            A.out[w].dir = W.stage[w].dir.p;
            A.out[w].rgb = W.stage[w].rgb.p;
            A.out[w].meta = W.stage[w].meta.p;
            A.out[w].count = W.seg_counts.p + w * PH_SEGMENTS * PH_COUNT_STRIDE;
            A.out[w].seg_capacity = seg_cap[w];
        }
        A.seed_pos = cs.pos.p;
        A.seed_dir = cs.dir.p;
        A.n_seeds = (uint32_t)cs.count;
        A.seed = s->seed;
        A.first = first;
        A.count = (uint32_t)count;
        A.light = (uint32_t)light;
        A.pass = (uint32_t)pass;
        A.max_bounces = s->params.max_photon_bounces;
        A.lds_slots = s->dev.lds_slots;
        A.err = s->counters.p + 63;
        A.stats = s->stats_dev.p;
        RRT_HIP(hipEventRecord(e0, nullptr));
        // as many blocks as the GPU holds at once (or fewer, for few photons): every wave then works through a run of
        // photons of its own, refilling its lanes as paths end (k_photon_shoot)
        dim3 shoot_grid = grid_for(count, RRT_BLOCK);
        {
            int per_cu = 0, dev = 0, n_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_photon_shoot, RRT_BLOCK, s->dev.lds_bytes) == hipSuccess && per_cu > 0 &&
                hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n_cu > 0)
                shoot_grid.x = std::min<uint32_t>(shoot_grid.x, (uint32_t)per_cu * (uint32_t)n_cu);  // (2x / 4x as many, shorter runs: 2 / 10 % slower)
        }
        hipLaunchKernelGGL(k_photon_shoot, shoot_grid, dim3(RRT_BLOCK), s->dev.lds_bytes, nullptr, A);
        RRT_HIP(hipGetLastError());
        uint32_t err = 0;
        std::vector<uint32_t> raw(2 * PH_SEGMENTS * PH_COUNT_STRIDE);
        RRT_HIP(hipMemcpy(&err, s->counters.p + 63, 4, hipMemcpyDeviceToHost));
        RRT_HIP(hipMemcpy(raw.data(), W.seg_counts.p, raw.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (err & RRT_ERR_BIT_INTERNAL) return fail(RRT_ERR_DEVICE, "device-side index guard tripped (internal error)");
        if (err & 1u) return fail(RRT_ERR_OVERFLOW, "HitVec overflow");
        if (err & 2u) return fail(RRT_ERR_INVALID, "No texture coordinates");
        if (err & 4u) return fail(RRT_ERR_LIMIT, "more than 1023 shadow photons behind one landing");
        SegmentPrefix P[2];
        bool overflow = (err & PH_ERR_CAPACITY) != 0;
        for (int w = 0; w < 2; w++) {
            uint64_t total = 0, worst = 0;
            for (int g = 0; g < PH_SEGMENTS; g++) {
                uint32_t c = raw[(w * PH_SEGMENTS + g) * PH_COUNT_STRIDE];  // counts what the segment was ASKED to hold
                worst = std::max<uint64_t>(worst, c);
                if (c > seg_cap[w]) overflow = true;
                P[w].count[g] = std::min(c, seg_cap[w]);
                P[w].first[g] = (uint32_t)total;
                total += P[w].count[g];
            }
            got[w] = total;
            if (overflow) est[w] = std::max(est[w] * 1.5, 1.25 * (double)(worst * PH_SEGMENTS) / (double)count);
        }
        if (overflow) continue;  // rerun with room for what this attempt wanted to store
        for (int w = 0; w < 2; w++) {
            PhotonBuffers &b = s->local[w];
            if (b.count + got[w] >= (1ull << 32) - 4096) return fail(RRT_ERR_LIMIT, "photon buffer above 2^32 records per GPU");
            if ((rc = grow_photons(b, b.count + got[w] + 1))) return rc;
            if (got[w])
                hipLaunchKernelGGL(k_compact_photons, grid_for((uint64_t)seg_cap[w] * PH_SEGMENTS), dim3(256), 0, nullptr, A.out[w], P[w], b.pos.p, b.dir.p,
                                   b.rgb.p, b.meta.p, (uint64_t)b.count);
        }
        RRT_HIP(hipEventRecord(e1, nullptr));
        RRT_HIP(hipGetLastError());
        RRT_HIP(hipDeviceSynchronize());
        float ms = 0.0f;
        RRT_HIP(hipEventElapsedTime(&ms, e0, e1));
        got[0] += s->local[0].count;
        got[1] += s->local[1].count;
        s->epoch++;                      // statistics change outside a render call
        s->stats.ms_photon_shoot += ms;  // shooting kernel + compaction
        s->stats.photons_emitted += count;
        uint64_t caustic_before = s->local[1].count;
        s->local[0].count = got[0];
        s->local[1].count = got[1];
        unsigned long long st[ST_COUNT];
        if ((rc = read_stats(s, st))) return rc;
        s->stats.bvh_nodes_visited = st[ST_NODES];
        s->stats.bvh_tris_tested = st[ST_TRIS];
        s->stats.paths_truncated += st[ST_TRUNCATED];
        if (pass == 0) {
            // photon_scene.rs:48-57: the Caustic-tagged records of pass 1 are what pass 2 aims at.  Everything
            // this call appended to the caustic buffer is Caustic-tagged; merge with earlier ranges.
            uint64_t add = got[1] - caustic_before;
            uint64_t total = cs.count + add;
            DBuf<float4> p, d;
            DBuf<uint64_t> k;
            if ((rc = p.ensure(total ? total : 1)) || (rc = d.ensure(total ? total : 1)) || (rc = k.ensure(total ? total : 1))) return rc;
            if (cs.count) {
                RRT_HIP(hipMemcpy(p.p, cs.pos.p, cs.count * 16, hipMemcpyDeviceToDevice));
                RRT_HIP(hipMemcpy(d.p, cs.dir.p, cs.count * 16, hipMemcpyDeviceToDevice));
                RRT_HIP(hipMemcpy(k.p, cs.key.p, cs.count * 8, hipMemcpyDeviceToDevice));
            }
            if (add) {
                RRT_HIP(hipMemcpy(p.p + cs.count, s->local[1].pos.p + caustic_before, add * 16, hipMemcpyDeviceToDevice));
                RRT_HIP(hipMemcpy(d.p + cs.count, s->local[1].dir.p + caustic_before, add * 16, hipMemcpyDeviceToDevice));
                hipLaunchKernelGGL(k_strip_key, grid_for(add), dim3(256), 0, nullptr, s->local[1].meta.p + caustic_before, (uint32_t)add, k.p + cs.count);
            }
            if ((rc = rebuild_seeds(s, light, p.p, d.p, k.p, total))) return rc;
        }
        return RRT_OK;
    }
    return fail(RRT_ERR_LIMIT, "photon buffer kept overflowing");
}

int rrt_photon_build(rrt_scene *s) {
    if (!s) return fail(RRT_ERR_INVALID, "null scene");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (!s->flattened) return fail(RRT_ERR_STATE, "rrt_photon_begin has not been called");
    drop_render_workers(s);  // the index arrays they read are rebuilt
    int rc;
    if ((rc = ensure_index_streams(s))) return rc;
    if (s->meta_or_pending) {  // records that were imported, appended or loaded: which key bits they use
        unsigned long long v = 0;
        RRT_HIP(hipMemcpy(&v, s->meta_or_dev.p, sizeof(v), hipMemcpyDeviceToHost));
        s->meta_or |= v;
        s->meta_or_pending = false;
    }
    // The two maps side by side on a stream each (photon_scene.rs:70-82 builds its two trees in two threads), both behind
    // whatever filled the buffers on the null stream; the host waits once, for both, and reads the two headers.
    hipEvent_t e0 = s->events[0], e1 = s->events[1];
    IndexScratch &W0 = s->index_scratch[0], &W1 = s->index_scratch[1];
    RRT_HIP(hipEventRecord(e0, nullptr));
    RRT_HIP(hipStreamWaitEvent(W0.stream, e0, 0));
    RRT_HIP(hipStreamWaitEvent(W1.stream, e0, 0));
    for (int w = 0; w < 2; w++)
        if ((rc = enqueue_index(s, w))) return rc;
    RRT_HIP(hipEventRecord(W1.done, W1.stream));
    RRT_HIP(hipStreamWaitEvent(W0.stream, W1.done, 0));
    RRT_HIP(hipEventRecord(e1, W0.stream));
    RRT_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    RRT_HIP(hipEventElapsedTime(&ms, e0, e1));
    for (int w = 0; w < 2; w++)
        if ((rc = finish_index(s, w))) return rc;
    s->stats.ms_photon_build = ms;
    s->stats.photons_stored_regular = s->local[0].count;
    s->stats.photons_stored_caustic = s->local[1].count;
    // Default heavy-query threshold: a query pair on a lit surface reads about 13 x crowding(regular map) candidate records
    // (crowding = occupancy of the dense cell, edge r, a random photon lies in; measured on scene2: 148 candidates per pair
    // at crowding 11); queries 4 x heavier than that are the tail of a launch and get a wave of their own.  A property of
    // the maps, not of the camera: the same frame always takes the same path.  Fine grids: crowding is that of the smaller
    // cells and a lane reads the cube's shell only; measured at 100 M photons (4K frame, cells of r / 3): factors 26 / 52 / 104 /
    // 208 / 416 give 0.175 / 0.145 / 0.133 / 0.144 / 0.164 s.
    {
        const char *f_env = getenv("RRT_HEAVY_FACTOR");  // experiments
        const double factor = f_env ? atof(f_env) : (s->index[0].div > 1u ? 104.0 : 52.0);
        double t = factor * s->index[0].crowding;
        s->heavy_auto = t > 4.0e9 ? 0xFFFFFFFEu : std::max<uint32_t>(RRT_HEAVY_THRESHOLD, (uint32_t)t);
    }
    s->maps_built = true;
    return RRT_OK;
}

int rrt_photon_count(rrt_scene *s, int which, uint64_t *count) {
    if (!s || !count || which < 0 || which > 1) return fail(RRT_ERR_INVALID, "bad argument");
    *count = s->local[which].count;
    return RRT_OK;
}

int rrt_photon_read(rrt_scene *s, int which, float *position, float *incident, float *intensity, uint8_t *type, uint64_t *key) {
    if (!s || which < 0 || which > 1) return fail(RRT_ERR_INVALID, "bad argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    PhotonBuffers &b = s->local[which];
    size_t n = b.count;
    if (n == 0) return RRT_OK;
    int rc = sort_by_meta(s->photon_scratch, b);
    if (rc) return rc;
    std::vector<float4> tmp(n);
    auto pull = [&](const float4 *src, float *dst, uint8_t *ty) -> int {
        RRT_HIP(hipMemcpy(tmp.data(), src, n * 16, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) {
            if (dst) {
                dst[i * 3] = tmp[i].x; dst[i * 3 + 1] = tmp[i].y; dst[i * 3 + 2] = tmp[i].z;
            }
            if (ty) {
                uint32_t t;
                memcpy(&t, &tmp[i].w, 4);
                ty[i] = (uint8_t)t;
            }
        }
        return RRT_OK;
    };
    if ((position || type) && (rc = pull(b.pos.p, position, type))) return rc;
    if (incident && (rc = pull(b.dir.p, incident, nullptr))) return rc;
    if (intensity && (rc = pull(b.rgb.p, intensity, nullptr))) return rc;
    if (key) RRT_HIP(hipMemcpy(key, b.meta.p, n * 8, hipMemcpyDeviceToHost));
    return RRT_OK;
}

// ---- multi-GPU exchange points: plain device pointers in, plain device pointers out ----
int rrt_photon_caustic_count(rrt_scene *s, int light, uint64_t *count) {
    if (!s || !count || light < 0 || light >= (int)s->seeds.size()) return fail(RRT_ERR_INVALID, "bad argument");
    *count = s->seeds[light]->count;
    return RRT_OK;
}
int rrt_photon_caustic_export_device(rrt_scene *s, int light, float *pos4_dev, float *incident4_dev, uint64_t *key_dev) {
    if (!s || light < 0 || light >= (int)s->seeds.size() || !pos4_dev || !incident4_dev || !key_dev) return fail(RRT_ERR_INVALID, "bad argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    CausticSeeds &cs = *s->seeds[light];
    if (cs.count == 0) return RRT_OK;
    RRT_HIP(hipMemcpy(pos4_dev, cs.pos.p, cs.count * 16, hipMemcpyDeviceToDevice));
    RRT_HIP(hipMemcpy(incident4_dev, cs.dir.p, cs.count * 16, hipMemcpyDeviceToDevice));
    RRT_HIP(hipMemcpy(key_dev, cs.key.p, cs.count * 8, hipMemcpyDeviceToDevice));
    return RRT_OK;
}
int rrt_photon_caustic_import_device(rrt_scene *s, int light, const float *pos4_dev, const float *incident4_dev, const uint64_t *key_dev,
                                     uint64_t count) {
    if (!s || light < 0 || light >= (int)s->seeds.size()) return fail(RRT_ERR_INVALID, "bad argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (count && (!pos4_dev || !incident4_dev || !key_dev)) return fail(RRT_ERR_INVALID, "null argument");
    if (count >= (1ull << 32)) return fail(RRT_ERR_LIMIT, "too many caustic seeds");
    return rebuild_seeds(s, light, (const float4 *)pos4_dev, (const float4 *)incident4_dev, key_dev, count);
}
int rrt_photon_local_count(rrt_scene *s, int which, uint64_t *count) { return rrt_photon_count(s, which, count); }
int rrt_photon_export_device(rrt_scene *s, int which, float *pos4_dev, float *incident4_dev, float *intensity4_dev, uint64_t *meta_dev) {
    if (!s || which < 0 || which > 1 || !pos4_dev || !incident4_dev || !intensity4_dev || !meta_dev) return fail(RRT_ERR_INVALID, "bad argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    PhotonBuffers &b = s->local[which];
    if (b.count == 0) return RRT_OK;
    RRT_HIP(hipMemcpy(pos4_dev, b.pos.p, b.count * 16, hipMemcpyDeviceToDevice));
    RRT_HIP(hipMemcpy(incident4_dev, b.dir.p, b.count * 16, hipMemcpyDeviceToDevice));
    RRT_HIP(hipMemcpy(intensity4_dev, b.rgb.p, b.count * 16, hipMemcpyDeviceToDevice));
    RRT_HIP(hipMemcpy(meta_dev, b.meta.p, b.count * 8, hipMemcpyDeviceToDevice));
    return RRT_OK;
}
int rrt_photon_import_device(rrt_scene *s, int which, const float *pos4_dev, const float *incident4_dev, const float *intensity4_dev,
                             const uint64_t *meta_dev, uint64_t count) {
    if (!s || which < 0 || which > 1) return fail(RRT_ERR_INVALID, "bad argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (count && (!pos4_dev || !incident4_dev || !intensity4_dev || !meta_dev)) return fail(RRT_ERR_INVALID, "null argument");
    if (count >= (1ull << 32)) return fail(RRT_ERR_LIMIT, "photon buffer above 2^32 records per GPU");
    PhotonBuffers &b = s->local[which];
    b.count = 0;
    int rc = grow_photons(b, count ? count : 1);
    if (rc) return rc;
    if (count) {
        RRT_HIP(hipMemcpy(b.pos.p, pos4_dev, count * 16, hipMemcpyDeviceToDevice));
        RRT_HIP(hipMemcpy(b.dir.p, incident4_dev, count * 16, hipMemcpyDeviceToDevice));
        RRT_HIP(hipMemcpy(b.rgb.p, intensity4_dev, count * 16, hipMemcpyDeviceToDevice));
        RRT_HIP(hipMemcpy(b.meta.p, meta_dev, count * 8, hipMemcpyDeviceToDevice));
        if ((rc = note_foreign_keys(s, b.meta.p, count))) return rc;
    }
    b.count = count;
    s->maps_built = false;
    return RRT_OK;
}
int rrt_photon_reserve(rrt_scene *s, int which, uint64_t total) {
    if (!s || which < 0 || which > 1) return fail(RRT_ERR_INVALID, "bad argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (total >= (1ull << 32)) return fail(RRT_ERR_LIMIT, "photon buffer above 2^32 records per GPU");
    PhotonBuffers &b = s->local[which];
    int rc;
    const uint64_t need = std::max<uint64_t>(total, 1);
    if ((rc = grow_preserve(b.pos, need, b.count)) || (rc = grow_preserve(b.dir, need, b.count)) || (rc = grow_preserve(b.rgb, need, b.count)) ||
        (rc = grow_preserve(b.meta, need, b.count)))
        return rc;
    return RRT_OK;
}
int rrt_photon_append_device(rrt_scene *s, int which, const float *pos4_dev, const float *incident4_dev, const float *intensity4_dev,
                             const uint64_t *meta_dev, uint64_t count) {
    if (!s || which < 0 || which > 1) return fail(RRT_ERR_INVALID, "bad argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (count == 0) return RRT_OK;
    if (!pos4_dev || !incident4_dev || !intensity4_dev || !meta_dev) return fail(RRT_ERR_INVALID, "null argument");
    PhotonBuffers &b = s->local[which];
    if (b.count + count >= (1ull << 32)) return fail(RRT_ERR_LIMIT, "photon buffer above 2^32 records per GPU");
    int rc = grow_photons(b, b.count + count);
    if (rc) return rc;
    RRT_HIP(hipMemcpy(b.pos.p + b.count, pos4_dev, count * 16, hipMemcpyDeviceToDevice));
    RRT_HIP(hipMemcpy(b.dir.p + b.count, incident4_dev, count * 16, hipMemcpyDeviceToDevice));
    RRT_HIP(hipMemcpy(b.rgb.p + b.count, intensity4_dev, count * 16, hipMemcpyDeviceToDevice));
    RRT_HIP(hipMemcpy(b.meta.p + b.count, meta_dev, count * 8, hipMemcpyDeviceToDevice));
    if ((rc = note_foreign_keys(s, b.meta.p + b.count, count))) return rc;
    b.count += count;
    s->maps_built = false;
    return RRT_OK;
}

int rrt_photon_query(rrt_scene *s, int which, const float *positions, size_t n, float half_width, uint32_t k_max, uint32_t *count, uint64_t *key,
                     float *dist2) {
    if (!s || which < 0 || which > 1 || !positions || !count) return fail(RRT_ERR_INVALID, "bad argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (k_max && (!key || !dist2)) return fail(RRT_ERR_INVALID, "null output array");
    if (!s->maps_built) return fail(RRT_ERR_STATE, "Photon map not built");
    if (!(half_width >= 0.0f)) return fail(RRT_ERR_INVALID, "bad half width");
    if (n >= (1ull << 31) || (uint64_t)n * k_max >= (1ull << 32)) return fail(RRT_ERR_LIMIT, "query batch too large");
    if (n == 0) return RRT_OK;
    PhotonIndex &X = s->index[which];
    // the cube may overlap at most a few hundred cells per axis: a larger one would scan the table for minutes
    if (half_width * X.inv_cell > 64.0f) return fail(RRT_ERR_LIMIT, "query cube wider than 128 index cells");
    DBuf<float> d_pos, d_d2;
    DBuf<uint32_t> d_count;
    DBuf<uint64_t> d_key;
    int rc;
    size_t rows = (size_t)n * k_max;
    if ((rc = d_pos.ensure(n * 3)) || (rc = d_count.ensure(n)) || (rc = d_key.ensure(rows ? rows : 1)) || (rc = d_d2.ensure(rows ? rows : 1))) return rc;
    RRT_HIP(hipMemcpy(d_pos.p, positions, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_photon_query, grid_for(n, 64), dim3(64), 0, nullptr, X.view(), X.meta.p, d_pos.p, (uint32_t)n, half_width, k_max, d_count.p,
                       d_key.p, d_d2.p);
    RRT_HIP(hipGetLastError());
    RRT_HIP(hipDeviceSynchronize());
    RRT_HIP(hipMemcpy(count, d_count.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (rows) {
        RRT_HIP(hipMemcpy(key, d_key.p, rows * sizeof(uint64_t), hipMemcpyDeviceToHost));
        RRT_HIP(hipMemcpy(dist2, d_d2.p, rows * sizeof(float), hipMemcpyDeviceToHost));
    }
    return RRT_OK;
}

// PhotonTree::find_nearest (photon_tree.rs:37-39) on top of the cube query: a cube of half width r that holds k photons
// whose k-th smallest distance is <= r cannot miss a closer photon (anything outside the cube is farther than r along
// one axis already), so the cube is doubled for the points that are not there yet.
int rrt_photon_nearest(rrt_scene *s, int which, const float *positions, size_t n, uint32_t k, uint32_t *count, uint64_t *key, float *dist2) {
    if (!s || which < 0 || which > 1 || !positions || !count || !key || !dist2) return fail(RRT_ERR_INVALID, "bad argument");
    if (k == 0) return fail(RRT_ERR_INVALID, "k = 0");
    if (!s->maps_built) return fail(RRT_ERR_STATE, "Photon map not built");
    if (n == 0) return RRT_OK;
    const PhotonIndex &X = s->index[which];
    const uint64_t in_map = X.count;
    std::vector<size_t> pending(n);
    for (size_t i = 0; i < n; i++) pending[i] = i;
    float r = X.inv_cell > 0.0f ? 0.5f / X.inv_cell : 0.1f;           // half a cell: the reference's own gather cube
    const float r_max = X.inv_cell > 0.0f ? 64.0f / X.inv_cell : 6.4f;  // rrt_photon_query's limit
    std::vector<float> pos, d2;
    std::vector<uint32_t> cnt;
    std::vector<uint64_t> keys;
    for (;;) {
        const size_t m = pending.size();
        pos.resize(m * 3);
        cnt.resize(m);
        keys.assign(m * k, 0);
        d2.assign(m * k, 0.0f);
        for (size_t j = 0; j < m; j++) memcpy(&pos[j * 3], positions + pending[j] * 3, 3 * sizeof(float));
        int rc = rrt_photon_query(s, which, pos.data(), m, r, k, cnt.data(), keys.data(), d2.data());
        if (rc) return rc;
        std::vector<size_t> next;
        const float inside = r * r * (1.0f - 9.5367431640625e-7f);  // the cube test is on coordinates, the distances are rounded sums
        for (size_t j = 0; j < m; j++) {
            const uint32_t have = std::min<uint64_t>(cnt[j], k);
            const bool all = cnt[j] >= in_map;                        // the whole map is inside the cube
            const bool enough = cnt[j] >= k && d2[j * k + k - 1] <= inside;
            if (all || enough) {
                const size_t i = pending[j];
                count[i] = have;
                memcpy(key + i * k, &keys[j * k], have * sizeof(uint64_t));
                memcpy(dist2 + i * k, &d2[j * k], have * sizeof(float));
            } else {
                next.push_back(pending[j]);
            }
        }
        if (next.empty()) return RRT_OK;
        if (r >= r_max) {
            // Points whose k nearest photons are more than 64 index cells away (empty space, or k above what is nearby:
            // shadow photons on the reference's infinite planes land millions of units out).  The grid has nothing to offer
            // there: these few points are searched exhaustively on the host, same arithmetic, same order.
            if (in_map > (1ull << 26)) return fail(RRT_ERR_LIMIT, "find_nearest: query point more than 64 index cells from its k nearest photons in a map above 2^26 records");
            std::vector<float4> all_pos(in_map);
            std::vector<uint64_t> all_key(in_map);
            RRT_HIP(hipMemcpy(all_pos.data(), X.pos.p, in_map * sizeof(float4), hipMemcpyDeviceToHost));
            RRT_HIP(hipMemcpy(all_key.data(), X.meta.p, in_map * sizeof(uint64_t), hipMemcpyDeviceToHost));
            std::vector<std::pair<float, uint64_t>> cand(in_map);
            for (size_t i : next) {
                const f3 p = mk3(positions[i * 3], positions[i * 3 + 1], positions[i * 3 + 2]);
                for (uint64_t j = 0; j < in_map; j++)
                    cand[j] = {lensq3(mk3(all_pos[j].x, all_pos[j].y, all_pos[j].z) - p), all_key[j]};
                const size_t have = std::min<uint64_t>(k, in_map);
                std::partial_sort(cand.begin(), cand.begin() + have, cand.end());  // (distance, key) ascending
                count[i] = (uint32_t)have;
                for (size_t j = 0; j < have; j++) {
                    dist2[i * k + j] = cand[j].first;
                    key[i * k + j] = cand[j].second;
                }
            }
            return RRT_OK;
        }
        r = std::min(r * 2.0f, r_max);
        pending.swap(next);
    }
}

// ---- photon-map files (keep the maps between runs, re-render without re-shooting) ----
// Layout, little endian: "RRTPMAP1", u32 version (2: record keys are index * 65536 + landing * 1024 + sub-index; version 1 files
// of round 1 keyed index * 4096 + landing * 64 + sub-index and are refused), u32 bytes per record (56), u64 seed, u64 count[2], then per map
// pos[count] float4 (w = type), incident[count] float4, intensity[count] float4, meta[count] u64, in canonical order.
namespace {
struct PmapHeader {
    char magic[8];
    uint32_t version, record_bytes;
    uint64_t seed, count[2];
};
const size_t kPmapChunk = 1u << 22;  // records per host staging chunk (64 MiB of float4)
}  // namespace

int rrt_photon_save(rrt_scene *s, const char *path) {
    if (!s || !path) return fail(RRT_ERR_INVALID, "null argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (s->mode != RRT_MODE_PHOTON) return fail(RRT_ERR_STATE, "scene is not a PhotonScene");
    int rc;
    for (int w = 0; w < 2; w++)
        if ((rc = sort_by_meta(s->photon_scratch, s->local[w]))) return rc;
    FILE *f = fopen(path, "wb");
    if (!f) return fail(RRT_ERR_IO, std::string("cannot write ") + path);
    PmapHeader h;
    memcpy(h.magic, "RRTPMAP1", 8);
    h.version = 2;
    h.record_bytes = 56;
    h.seed = s->seed;
    h.count[0] = s->local[0].count;
    h.count[1] = s->local[1].count;
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    std::vector<uint8_t> stage(kPmapChunk * 16);
    for (int w = 0; w < 2 && ok; w++) {
        PhotonBuffers &b = s->local[w];
        const void *arrs[4] = {b.pos.p, b.dir.p, b.rgb.p, b.meta.p};
        const size_t elem[4] = {16, 16, 16, 8};
        for (int a = 0; a < 4 && ok; a++)
            for (size_t i = 0; i < b.count && ok; i += kPmapChunk) {
                size_t n = std::min<size_t>(kPmapChunk, b.count - i);
                if (hipMemcpy(stage.data(), (const uint8_t *)arrs[a] + i * elem[a], n * elem[a], hipMemcpyDeviceToHost) != hipSuccess) {
                    fclose(f);
                    return fail(RRT_ERR_DEVICE, "hipMemcpy failed while saving the photon maps");
                }
                ok = fwrite(stage.data(), elem[a], n, f) == n;
            }
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail(RRT_ERR_IO, std::string("short write to ") + path);
    return RRT_OK;
}

int rrt_photon_load(rrt_scene *s, const char *path) {
    if (!s || !path) return fail(RRT_ERR_INVALID, "null argument");
    if (int rc_dev = use_scene_device(s)) return rc_dev;
    if (s->mode != RRT_MODE_PHOTON) return fail(RRT_ERR_STATE, "scene is not a PhotonScene");
    if (!s->flattened || s->seeds.size() != s->lights.size()) return fail(RRT_ERR_STATE, "rrt_photon_begin has not been called");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(RRT_ERR_IO, std::string("cannot open ") + path);
    PmapHeader h;
    if (fread(&h, sizeof(h), 1, f) != 1 || memcmp(h.magic, "RRTPMAP1", 8) != 0 || h.version != 2 || h.record_bytes != 56) {
        fclose(f);
        return fail(RRT_ERR_PARSE, std::string(path) + " is not a photon-map file of this build (RRTPMAP1, version 2; version 1 files carry another key layout)");
    }
    if (h.count[0] >= (1ull << 32) || h.count[1] >= (1ull << 32)) {
        fclose(f);
        return fail(RRT_ERR_LIMIT, "photon buffer above 2^32 records per GPU");
    }
    std::vector<uint8_t> stage(kPmapChunk * 16);
    for (int w = 0; w < 2; w++) {
        PhotonBuffers &b = s->local[w];
        b.count = 0;
        int rc = grow_photons(b, h.count[w] ? h.count[w] : 1);
        if (rc) {
            fclose(f);
            return rc;
        }
        void *arrs[4] = {b.pos.p, b.dir.p, b.rgb.p, b.meta.p};
        const size_t elem[4] = {16, 16, 16, 8};
        for (int a = 0; a < 4; a++)
            for (size_t i = 0; i < h.count[w]; i += kPmapChunk) {
                size_t n = std::min<size_t>(kPmapChunk, h.count[w] - i);
                if (fread(stage.data(), elem[a], n, f) != n) {
                    fclose(f);
                    return fail(RRT_ERR_IO, std::string("short read from ") + path);
                }
                if (hipMemcpy((uint8_t *)arrs[a] + i * elem[a], stage.data(), n * elem[a], hipMemcpyHostToDevice) != hipSuccess) {
                    fclose(f);
                    return fail(RRT_ERR_DEVICE, "hipMemcpy failed while loading the photon maps");
                }
            }
        b.count = h.count[w];
        if (int rc = note_foreign_keys(s, b.meta.p, b.count)) {
            fclose(f);
            return rc;
        }
    }
    fclose(f);
    s->seed = h.seed;
    s->maps_built = false;
    return RRT_OK;
}

}  // extern "C"

// rrt_scene_create touches one kernel of every translation unit, so that the code objects are loaded before the first
// rrt_pre_render is timed (HIP loads them on first use: tens of milliseconds per unit)
namespace rrt {
void preload_photon() {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&k_iota));
}
}  // namespace rrt
