// dev_photon.h — counter-based RNG streams, the photon index and the radiance-estimate gather.
//
// Replaces core/photon_tree.rs (third-party kd-tree 0.5.1, `within` = closed axis-aligned box) with
// dense uniform grids (cell edge = half_width) over the regions that hold the photons, plus a hash
// grid for the few that lie outside them.  Records are sorted by cell so a query streams contiguous runs.
#pragma once
#include "kernels.h"

namespace rrt {

// ---- Philox4x32-10 (Salmon et al., SC'11).  Stream = (seed, light, pass, photon index); the
// reference's rand 0.8.5 thread_rng is unseeded, only the distributions are contractual. ----------
struct Philox {
    uint32_t key0, key1;
    uint32_t c0, c1, c2, c3;
    uint32_t buf[4];
    uint32_t have;
};
RRT_HD void philox_block(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
RRT_HD void rng_init(Philox &r, uint64_t seed, uint32_t light, uint32_t pass, uint64_t index) {
    r.key0 = (uint32_t)seed;
    r.key1 = (uint32_t)(seed >> 32);
    r.c0 = 0;
    r.c1 = (uint32_t)index;
    r.c2 = (uint32_t)(index >> 32);
    r.c3 = (light << 1) | pass;
    r.have = 0;
}
RRT_HD uint32_t rng_u32(Philox &r) {
    if (r.have == 0) {
        philox_block(r.c0, r.c1, r.c2, r.c3, r.key0, r.key1, r.buf);
        r.c0++;
        r.have = 4;
    }
    uint32_t i = 4 - r.have;
    r.have--;
    // select without dynamic register indexing
    return i == 0 ? r.buf[0] : (i == 1 ? r.buf[1] : (i == 2 ? r.buf[2] : r.buf[3]));
}
RRT_HD float rng_u01(Philox &r) { return (float)(rng_u32(r) >> 8) * 5.9604644775390625e-8f; }
RRT_HD float rng_sym(Philox &r) { return rng_u01(r) * 2.0f - 1.0f; }
// Vector::random (core/vector.rs:23-42): rejection-sample the unit ball, normalise
RRT_HD f3 rng_unit_vector(Philox &r) {
    for (;;) {
        float x = rng_sym(r), y = rng_sym(r), z = rng_sym(r);
        f3 v = mk3(x, y, z);
        if (lensq3(v) <= 1.0f) return norm3(v);
    }
}

// ---- hash grid --------------------------------------------------------------------------------------
#define RRT_CELL_BIAS 1048576  // 2^20
RRT_HD uint32_t cell_coord(float x, float inv_cell) {
    float c = floorf(x * inv_cell);
    if (!(c > -(float)RRT_CELL_BIAS)) c = -(float)RRT_CELL_BIAS;  // also catches NaN
    if (c > (float)(RRT_CELL_BIAS - 1)) c = (float)(RRT_CELL_BIAS - 1);
    return (uint32_t)((int32_t)c + RRT_CELL_BIAS);
}
RRT_HD uint64_t cell_key(uint32_t cx, uint32_t cy, uint32_t cz) { return ((uint64_t)cx << 42) | ((uint64_t)cy << 21) | (uint64_t)cz; }
RRT_HD uint32_t hash_key(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return (uint32_t)k;
}

struct GatherSums {
    f3 ldir;       // sum of normalised incidents
    f3 intensity;  // sum of intensities
    uint32_t k;    // photons summed: those inside the closed box, or the nearest gather_k of them
    uint32_t candidates;
    uint32_t inside;  // gather_knn only: photons inside the closed box before the truncation to k
};

// ---- the index: dense grids where the photons are, a hash grid for the stragglers ---------------------------------
// Dense part: cells of edge r (the query half width).  Records are sorted by cell with x running fastest, so the cells a
// query box [p - r, p + r] overlaps along x (3, rarely 4) are ONE contiguous run of records per (y, z) row: 9 runs per
// query, each found with two loads from the start table at computed addresses — no probing — and the records read cover
// 3 x 3 x 3 cells of edge r instead of 2 x 2 x 2 cells of edge 2r (2.25 x instead of 4 x the query's own area on a surface).
// Photon maps are mostly surfaces in a large volume, and a 3-D grid pays for the volume: ONE box over everything would
// need billions of cells.  So there are up to RRT_DENSE_GRIDS boxes, each sized to a cell budget: the first around the
// densest part of the map, every further one around the densest part of what the earlier ones left out (typically a thin
// slab along the floor, then the walls).  A record belongs to the FIRST grid whose box holds its position.  A query walks
// the grids in order and stops at the first one whose box holds the whole query box: every record inside the query box
// lies in that grid's box too, hence in it or in an earlier one.  Stragglers (the reference's infinite planes catch photons
// millions of units out): the hash grid of round 1 (cell edge 2r, open addressing), asked only by queries that no grid
// box holds completely.

// cell key -> (first record, count) through the open-addressing table of the outliers
__device__ __forceinline__ uint2 lookup_cell(const DevPhotonMap &G, uint64_t key) {
    uint32_t slot = hash_key(key) & G.table_mask;
    for (;;) {
        uint64_t k = G.keys[slot];
        if (k == key) return G.ranges[slot];
        if (k == ~0ULL) return make_uint2(0, 0);
        slot = (slot + 1) & G.table_mask;
    }
}

// dense cell coordinate along one axis as a float (NaN for NaN input): the same function files records and places queries
RRT_HD float dense_coord(float x, float lo, float inv_cell) { return floorf((x - lo) * inv_cell); }

// The cells of one grid a box overlaps: [x0, x1] x [y0, y1] x [z0, z1] clamped to the grid (empty when x1 < x0 ...), and
// whether the box reaches outside the grid (then later grids and the outliers have to be asked as well).
struct DenseBox {
    int x0, x1, y0, y1, z0, z1;
    bool outside, empty;
};
__device__ __forceinline__ DenseBox dense_box(const DenseGrid &D, float inv_dcell, f3 lo, f3 hi) {
    DenseBox B;
    const float fx0 = dense_coord(lo.x, D.lo_x, inv_dcell), fx1 = dense_coord(hi.x, D.lo_x, inv_dcell);
    const float fy0 = dense_coord(lo.y, D.lo_y, inv_dcell), fy1 = dense_coord(hi.y, D.lo_y, inv_dcell);
    const float fz0 = dense_coord(lo.z, D.lo_z, inv_dcell), fz1 = dense_coord(hi.z, D.lo_z, inv_dcell);
    const float nx = (float)D.nx, ny = (float)D.ny, nz = (float)D.nz;
    // written so that a NaN coordinate counts as outside
    B.outside = !(fx0 >= 0.0f && fx1 < nx && fy0 >= 0.0f && fy1 < ny && fz0 >= 0.0f && fz1 < nz);
    B.empty = !(fx1 >= 0.0f && fx0 < nx && fy1 >= 0.0f && fy0 < ny && fz1 >= 0.0f && fz0 < nz);
    B.x0 = (int)fminf(fmaxf(fx0, 0.0f), nx - 1.0f), B.x1 = (int)fminf(fmaxf(fx1, 0.0f), nx - 1.0f);
    B.y0 = (int)fminf(fmaxf(fy0, 0.0f), ny - 1.0f), B.y1 = (int)fminf(fmaxf(fy1, 0.0f), ny - 1.0f);
    B.z0 = (int)fminf(fmaxf(fz0, 0.0f), nz - 1.0f), B.z1 = (int)fminf(fmaxf(fz1, 0.0f), nz - 1.0f);
    return B;
}
// the grid a position is filed in (the first whose box holds it) and its cell there; false: an outlier
RRT_HD bool dense_cell_of(const DevPhotonMap &G, float x, float y, float z, uint32_t &cell) {
#pragma unroll
    for (uint32_t g = 0; g < RRT_DENSE_GRIDS; g++) {  // (constant indices into the argument block: see fine_box)
        if (g >= G.n_grids) break;
        const DenseGrid &D = G.grid[g];
        const float fx = dense_coord(x, D.lo_x, G.inv_dcell), fy = dense_coord(y, D.lo_y, G.inv_dcell), fz = dense_coord(z, D.lo_z, G.inv_dcell);
        if (fx >= 0.0f && fx < (float)D.nx && fy >= 0.0f && fy < (float)D.ny && fz >= 0.0f && fz < (float)D.nz) {
            cell = D.cell_base + ((uint32_t)fz * D.ny + (uint32_t)fy) * D.nx + (uint32_t)fx;
            return true;
        }
    }
    return false;
}
// the records of row (y, z) of a grid, cells x0..x1: one contiguous run
__device__ __forceinline__ uint2 dense_run(const DevPhotonMap &G, const DenseGrid &D, int x0, int x1, int y, int z) {
    const uint32_t row = D.cell_base + ((uint32_t)z * D.ny + (uint32_t)y) * D.nx;
    const uint32_t first = G.dense_start[row + (uint32_t)x0], end = G.dense_start[row + (uint32_t)x1 + 1u];
    return make_uint2(first, end - first);
}

// closed-box test of one candidate record; incident / intensity are read for the records inside only
__device__ __forceinline__ void take_record(const DevPhotonMap &G, uint32_t i, float4 q, f3 lo, f3 hi, GatherSums &s) {
    if (q.x >= lo.x && q.x <= hi.x && q.y >= lo.y && q.y <= hi.y && q.z >= lo.z && q.z <= hi.z) {
        float4 d = G.incident[i];
        float4 c = G.intensity[i];
        s.ldir = s.ldir + mk3(d.x, d.y, d.z);
        s.intensity = s.intensity + mk3(c.x, c.y, c.z);
        s.k++;
    }
}
__device__ __forceinline__ void stream_cell(const DevPhotonMap &G, uint2 range, f3 lo, f3 hi, GatherSums &s) {
    s.candidates += range.y;
    uint32_t i = min(range.x, G.count), end = i + min(range.y, G.count - i);  // guard: stay inside the map
    for (; i + 4 <= end; i += 4) {  // four independent position loads in flight
        float4 q0 = G.pos_cell[i], q1 = G.pos_cell[i + 1], q2 = G.pos_cell[i + 2], q3 = G.pos_cell[i + 3];
        take_record(G, i, q0, lo, hi, s);
        take_record(G, i + 1, q1, lo, hi, s);
        take_record(G, i + 2, q2, lo, hi, s);
        take_record(G, i + 3, q3, lo, hi, s);
    }
    for (; i < end; i++) take_record(G, i, G.pos_cell[i], lo, hi, s);
}

// the outliers a box may hold: the hash cells (edge 2r) it overlaps
template <class F>
__device__ __forceinline__ void visit_outlier_cells(const DevPhotonMap &G, f3 lo, f3 hi, F &&f) {
    if (G.n_dense == G.count) return;
    uint32_t x0 = cell_coord(lo.x, G.inv_cell), x1 = cell_coord(hi.x, G.inv_cell);
    uint32_t y0 = cell_coord(lo.y, G.inv_cell), y1 = cell_coord(hi.y, G.inv_cell);
    uint32_t z0 = cell_coord(lo.z, G.inv_cell), z1 = cell_coord(hi.z, G.inv_cell);
    for (uint32_t cx = x0; cx <= x1; cx++)
        for (uint32_t cy = y0; cy <= y1; cy++)
            for (uint32_t cz = z0; cz <= z1; cz++) f(lookup_cell(G, cell_key(cx, cy, cz)));
}
// every run of records that may hold photons of the closed box [lo, hi]: per grid the rows in (z, y) order, then outlier cells
template <class F>
__device__ __forceinline__ void visit_runs(const DevPhotonMap &G, f3 lo, f3 hi, F &&f) {
    if (G.count == 0) return;
#pragma unroll
    for (uint32_t g = 0; g < RRT_DENSE_GRIDS; g++) {  // (constant indices into the argument block: see fine_box)
        if (g >= G.n_grids) break;
        const DenseGrid &D = G.grid[g];
        const DenseBox B = dense_box(D, G.inv_dcell, lo, hi);
        if (!B.empty)
            for (int z = B.z0; z <= B.z1; z++)
                for (int y = B.y0; y <= B.y1; y++) f(dense_run(G, D, B.x0, B.x1, y, z));
        if (!B.outside) return;  // the whole box lies in this grid's box: nothing of it can be filed further down
    }
    visit_outlier_cells(G, lo, hi, f);
}

// PhotonTree::get_within_distance (photon_tree.rs:15-35) + the accumulation of
// average_photon_of_type_at (photon_scene.rs:401-407).  The reference adds in order of distance,
// this adds in index order: same set, f32 sums differ by rounding only.
__device__ __forceinline__ void gather_box(const DevPhotonMap &G, f3 p, float r, GatherSums &s) {
    s.ldir = mk3(0.0f, 0.0f, 0.0f);
    s.intensity = mk3(0.0f, 0.0f, 0.0f);
    s.k = 0;
    s.candidates = 0;
    f3 lo = p - mk3(r, r, r), hi = p + mk3(r, r, r);  // photon_tree.rs:16-18
    visit_runs(G, lo, hi, [&](uint2 run) { stream_cell(G, run, lo, hi, s); });
}

// The common query: the box lies inside the dense region and overlaps at most 3 x 3 rows.  Its runs are looked up first
// (independent loads, and the candidate count is known before a record is read: the heavy-query decision), then streamed.
struct CellSet {
    uint2 rg[9];
    uint32_t candidates;
};
__device__ __forceinline__ bool lookup_cells(const DevPhotonMap &G, f3 lo, f3 hi, CellSet &C) {
    C.candidates = 0;
#pragma unroll
    for (uint32_t c = 0; c < 9; c++) C.rg[c] = make_uint2(0, 0);
    if (G.count == 0) return true;
#pragma unroll
    for (uint32_t g = 0; g < RRT_DENSE_GRIDS; g++) {  // (constant indices into the argument block: see fine_box)
        if (g >= G.n_grids) break;
        const DenseGrid &D = G.grid[g];
        const DenseBox B = dense_box(D, G.inv_dcell, lo, hi);
        if (B.empty) continue;           // nothing of the box in this grid: the next one
        if (B.outside) return false;     // partly in this grid: the general walk
        if (B.y1 - B.y0 > 2 || B.z1 - B.z0 > 2) return false;  // rounding made it 4 cells along an axis
#pragma unroll
        for (int c = 0; c < 9; c++) {
            const int y = B.y0 + c % 3, z = B.z0 + c / 3;
            if (y <= B.y1 && z <= B.z1) C.rg[c] = dense_run(G, D, B.x0, B.x1, y, z);
            C.candidates += C.rg[c].y;
        }
        return true;
    }
    return G.n_dense == G.count;  // outside every grid: only outliers can be there, if there are any
}
__device__ __forceinline__ void stream_cells(const DevPhotonMap &G, const CellSet &C, f3 lo, f3 hi, GatherSums &s) {
    s.ldir = mk3(0.0f, 0.0f, 0.0f);
    s.intensity = mk3(0.0f, 0.0f, 0.0f);
    s.k = 0;
    s.candidates = 0;
#pragma unroll
    for (uint32_t c = 0; c < 9; c++) stream_cell(G, C.rg[c], lo, hi, s);
}

// ---- fine grids (DevPhotonMap::div > 1): cells of edge r / div and per-cell sums ------------------------------------
// In a crowded map (tens of millions of records) a query cube holds thousands of photons and neighbouring pixels stream
// nearly the same ones.  With cells of edge r / div the cube spans 2 div + 1 (or + 2) cells along each axis, and every cell
// STRICTLY between the cube's first and last cell along all three axes lies wholly inside the closed cube: dense_coord is
// monotone in x, so a record whose cell coordinate is above that of the cube's lower bound is itself above the bound, and
// the same at the upper bound — exactly, in float arithmetic.  Such cells are not streamed: the index keeps per cell the
// sums the estimate needs (normalised incidents, intensities, in canonical order; the count is the start table's), and the
// query adds those.  Only the cube's shell is read record by record.  Same photon SET as the reference's box query
// (photon_tree.rs:15-35), counts exact; the f32 sums are grouped by cell, i.e. differ by rounding only.
// Used when the cube lies wholly inside a grid's box (as the coarse fast path); anything else takes the general walk.
struct FineBox {
    DenseBox B;
    uint32_t cell_base, nx, ny;  // of the grid that holds the cube
    bool any;  // false: the cube meets no dense grid (and the map has no outliers): nothing to gather
};
__device__ __forceinline__ bool fine_box(const DevPhotonMap &G, f3 lo, f3 hi, FineBox &F) {
    F.any = false;
    F.cell_base = F.nx = F.ny = 0;
    if (G.count == 0) return true;
    // (unrolled: an index into the kernel's argument block that is not a constant makes the compiler copy the block to scratch)
#pragma unroll
    for (uint32_t g = 0; g < RRT_DENSE_GRIDS; g++) {
        if (g >= G.n_grids) break;
        const DenseGrid &D = G.grid[g];
        const DenseBox B = dense_box(D, G.inv_dcell, lo, hi);
        if (B.empty) continue;        // nothing of the cube in this grid: the next one
        if (B.outside) return false;  // partly in this grid: the general walk
        F.B = B;
        F.cell_base = D.cell_base, F.nx = D.nx, F.ny = D.ny;
        F.any = true;
        return true;
    }
    return G.n_dense == G.count;  // outside every grid: only outliers can be there, if there are any
}
// rows (y, z) of the box: `stream(first, count)` for runs of records to be tested one by one, `whole(first_cell, last_cell,
// count)` for runs of cells that lie inside the cube as a whole
template <class Stream, class Whole>
__device__ __forceinline__ void visit_fine(const DevPhotonMap &G, const FineBox &F, Stream &&stream, Whole &&whole) {
    if (!F.any) return;
    const DenseBox &B = F.B;
    const bool x_inner = B.x1 - B.x0 >= 2;
    for (int z = B.z0; z <= B.z1; z++)
        for (int y = B.y0; y <= B.y1; y++) {
            const uint32_t row = F.cell_base + ((uint32_t)z * F.ny + (uint32_t)y) * F.nx;
            // the row's records are one run [a, d); on an inner row its middle [b, c) is whole cells
            const uint32_t a = G.dense_start[row + (uint32_t)B.x0], d = G.dense_start[row + (uint32_t)B.x1 + 1u];
            uint32_t b = d, c = d;
            if (x_inner && z > B.z0 && z < B.z1 && y > B.y0 && y < B.y1) {
                b = G.dense_start[row + (uint32_t)B.x0 + 1u];
                c = G.dense_start[row + (uint32_t)B.x1];
                if (c != b) whole(row + (uint32_t)B.x0 + 1u, row + (uint32_t)B.x1 - 1u, c - b);
            }
            // (one call site, so that the streaming loop exists once in the code)
#pragma clang loop unroll(disable)
            for (int part = 0; part < 2; part++) stream(part ? c : a, part ? d - c : b - a);
        }
}
// records the shell of the cube holds (what a lane would read): the heavy-query decision, before any record is read
__device__ __forceinline__ uint32_t fine_candidates(const DevPhotonMap &G, const FineBox &F) {
    uint32_t n = 0;
    visit_fine(G, F, [&](uint32_t, uint32_t count) { n += count; }, [&](uint32_t, uint32_t, uint32_t) {});
    return n;
}
__device__ __forceinline__ void add_cells(const DevPhotonMap &G, uint32_t first_cell, uint32_t last_cell, GatherSums &s) {
    for (uint32_t c = first_cell; c <= last_cell; c++) {
        const float4 d = G.agg_dir[c];
        if (__float_as_uint(d.w) == 0u) continue;  // (w = records of the cell)
        const float4 i = G.agg_rgb[c];
        s.ldir = s.ldir + mk3(d.x, d.y, d.z);
        s.intensity = s.intensity + mk3(i.x, i.y, i.z);
    }
}
__device__ __forceinline__ void gather_fine(const DevPhotonMap &G, const FineBox &F, f3 lo, f3 hi, GatherSums &s) {
    s.ldir = mk3(0.0f, 0.0f, 0.0f);
    s.intensity = mk3(0.0f, 0.0f, 0.0f);
    s.k = 0;
    s.candidates = 0;
    visit_fine(
        G, F, [&](uint32_t first, uint32_t count) { stream_cell(G, make_uint2(first, count), lo, hi, s); },
        [&](uint32_t first_cell, uint32_t last_cell, uint32_t count) {
            add_cells(G, first_cell, last_cell, s);
            s.k += count;
        });
}

// ---- PhotonTree::get_n_within_radius (photon_tree.rs:41-50) as the estimator ------------------------------------------
// get_within_distance sorted by squared distance, truncated to the first k: the k photons of the cube closest to the hit.
// The lane keeps the k best (squared distance, record index) pairs seen so far in a max-heap in its column of the
// workspace (entry e at ws[e * stride]: neighbouring lanes touch neighbouring words), replaces the root whenever a closer
// photon turns up, heap-sorts the column in place at the end and adds the photons in ascending order — the order the
// reference's loop meets them in (photon_scene.rs:404-407), so the f32 sums are the reference's, bit for bit.
struct KnnColumn {
    uint2 *ws;
    uint32_t stride;
    __device__ __forceinline__ uint2 get(uint32_t e) const { return ws[(size_t)e * stride]; }
    __device__ __forceinline__ void put(uint32_t e, uint2 v) const { ws[(size_t)e * stride] = v; }
};
// does a come before b?  Squared distance first; equal distances by record key (the reference's sort_unstable leaves
// that order open, the oracle fixes it the same way)
__device__ __forceinline__ bool knn_before(const DevPhotonMap &G, uint2 a, uint2 b) {
    const float da = __uint_as_float(a.x), db = __uint_as_float(b.x);
    if (da != db) return da < db;
    return G.meta[a.y] < G.meta[b.y];
}
// put `item` where the root was and let it sink (heap of n entries, largest at the root)
__device__ __forceinline__ void knn_sift_down(const DevPhotonMap &G, const KnnColumn &H, uint32_t n, uint2 item) {
    uint32_t e = 0;
    for (;;) {
        uint32_t c = 2u * e + 1u;
        if (c >= n) break;
        uint2 child = H.get(c);
        if (c + 1u < n) {
            const uint2 other = H.get(c + 1u);
            if (knn_before(G, child, other)) {
                child = other;
                c++;
            }
        }
        if (!knn_before(G, item, child)) break;
        H.put(e, child);
        e = c;
    }
    H.put(e, item);
}
__device__ __forceinline__ void gather_knn(const DevPhotonMap &G, f3 p, float r, uint32_t k, const KnnColumn &H, GatherSums &s) {
    s.ldir = mk3(0.0f, 0.0f, 0.0f);
    s.intensity = mk3(0.0f, 0.0f, 0.0f);
    s.k = 0;
    s.candidates = 0;
    const f3 lo = p - mk3(r, r, r), hi = p + mk3(r, r, r);  // photon_tree.rs:16-18
    uint32_t n = 0, inside = 0, candidates = 0;
    visit_runs(G, lo, hi, [&](uint2 run) {
        candidates += run.y;
        uint32_t i = min(run.x, G.count);
        const uint32_t end = i + min(run.y, G.count - i);  // guard: stay inside the map
        for (; i < end; i++) {
            const float4 q = G.pos_cell[i];
            if (!(q.x >= lo.x && q.x <= hi.x && q.y >= lo.y && q.y <= hi.y && q.z >= lo.z && q.z <= hi.z)) continue;
            inside++;
            const uint2 item = make_uint2(__float_as_uint(lensq3(mk3(q.x, q.y, q.z) - p)), i);  // photon_tree.rs:25
            if (n < k) {  // room left: append and let it rise
                uint32_t e = n++;
                while (e > 0) {
                    const uint32_t parent = (e - 1u) >> 1;
                    const uint2 up = H.get(parent);
                    if (!knn_before(G, up, item)) break;
                    H.put(e, up);
                    e = parent;
                }
                H.put(e, item);
            } else if (knn_before(G, item, H.get(0))) {  // closer than the farthest kept: replace it
                knn_sift_down(G, H, n, item);
            }
        }
    });
    // heap sort in place: the farthest goes to the end, the rest is a heap again
    for (uint32_t m = n; m > 1; m--) {
        const uint2 last = H.get(m - 1u);
        H.put(m - 1u, H.get(0));
        knn_sift_down(G, H, m - 1u, last);
    }
    for (uint32_t e = 0; e < n; e++) {  // photon_scene.rs:404-407, nearest first
        const uint32_t i = H.get(e).y;
        const float4 d = G.incident[i], c = G.intensity[i];
        s.ldir = s.ldir + mk3(d.x, d.y, d.z);
        s.intensity = s.intensity + mk3(c.x, c.y, c.z);
    }
    s.k = n;               // neighbour_photons.len() after truncate(n)
    s.candidates = candidates;
    s.inside = inside;
}

// the blend of average_photon_at (photon_scene.rs:363-420) from the two maps' sums
__device__ __forceinline__ bool blend_estimates(const GatherSums &reg, const GatherSums &cau, f3 &incident, f3 &intensity) {
    float nr = (float)reg.k, nc = (float)cau.k;
    f3 r_dir = norm3(reg.ldir), r_int = reg.intensity / nr;
    f3 c_dir = norm3(cau.ldir), c_int = cau.intensity / nc;
    if (cau.k == 0) {
        if (reg.k == 0) return false;
        incident = r_dir;
        intensity = r_int;
        return true;
    }
    if (reg.k == 0) {
        incident = c_dir;
        intensity = c_int;
        return true;
    }
    incident = r_dir;
    intensity = r_int + c_int * (nc / (nc + nr));
    return true;
}

// average_photon_at (photon_scene.rs:363-420): regular and caustic estimates and their blend.
// Returns 0: no photons, 1: estimate in (incident, intensity), 2: DEFERRED -- the two boxes hold more than `defer_above`
// candidate records, nothing was read, the caller files the query for k_gather_heavy (pass 0xFFFFFFFF to never defer).
// FINE: the maps are fine grids (DevPhotonMap::div > 1; both maps of a scene are built alike): an instance of its own, so
// that the coarse path keeps its registers.
template <bool FINE>
__device__ __forceinline__ int average_photon_at(const DevPhotonMap maps[2], f3 p, float r, uint32_t defer_above, f3 &incident, f3 &intensity,
                                                 unsigned long long &gathered, unsigned long long &candidates, uint32_t knn_k = 0,
                                                 KnnColumn knn = KnnColumn{nullptr, 0}) {
    GatherSums reg, cau;
    if (knn_k) {  // get_n_within_radius instead of get_within_distance; the column is reused by the second map
        gather_knn(maps[0], p, r, knn_k, knn, reg);
        gather_knn(maps[1], p, r, knn_k, knn, cau);
        gathered += reg.inside + cau.inside;
        candidates += reg.candidates + cau.candidates;
        return blend_estimates(reg, cau, incident, intensity) ? 1 : 0;
    }
    f3 lo = p - mk3(r, r, r), hi = p + mk3(r, r, r);  // photon_tree.rs:16-18
    if (FINE) {
        FineBox f;
        uint32_t shell = 0;  // records a lane would read: the heavy-query decision, before any of them is read
        if (fine_box(maps[0], lo, hi, f)) shell += fine_candidates(maps[0], f);
        if (fine_box(maps[1], lo, hi, f)) shell += fine_candidates(maps[1], f);
        if (shell > defer_above) return 2;
        if (fine_box(maps[0], lo, hi, f))
            gather_fine(maps[0], f, lo, hi, reg);
        else
            gather_box(maps[0], p, r, reg);
        if (fine_box(maps[1], lo, hi, f))
            gather_fine(maps[1], f, lo, hi, cau);
        else
            gather_box(maps[1], p, r, cau);
        gathered += reg.k + cau.k;
        candidates += reg.candidates + cau.candidates;
        return blend_estimates(reg, cau, incident, intensity) ? 1 : 0;
    }
    // Each map on its own: the common case is a box inside one dense grid (9 runs, known before a record is read); a box
    // that straddles grid boxes or may hold outliers takes the general walk FOR THAT MAP only (the caustic map is small
    // and most queries fall outside its grids: that must not push the regular map's lookup onto the slow path as well).
    CellSet c0, c1;
    const bool fast0 = lookup_cells(maps[0], lo, hi, c0), fast1 = lookup_cells(maps[1], lo, hi, c1);
    // the heavy-query decision from what is known without reading records (a function of the query and the maps alone)
    if ((fast0 ? c0.candidates : 0u) + (fast1 ? c1.candidates : 0u) > defer_above) return 2;
    if (fast0)
        stream_cells(maps[0], c0, lo, hi, reg);
    else
        gather_box(maps[0], p, r, reg);
    if (fast1)
        stream_cells(maps[1], c1, lo, hi, cau);
    else
        gather_box(maps[1], p, r, cau);
    gathered += reg.k + cau.k;
    candidates += reg.candidates + cau.candidates;
    return blend_estimates(reg, cau, incident, intensity) ? 1 : 0;
}

// One query, a whole wave (all 64 lanes active, wave-uniform control flow): the cells' records are strided over the
// lanes and the partial sums are combined by a fixed butterfly, so the result depends only on the query and the map
// (not on which lane or wave ran it).  Same photon set as gather_box; the f32 sums differ by rounding only.
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ void wave_gather_box(const DevPhotonMap &G, f3 p, float r, GatherSums &s) {
    f3 lo = p - mk3(r, r, r), hi = p + mk3(r, r, r);
    CellSet C;
    if (!lookup_cells(G, lo, hi, C)) {  // outliers, or 4 cells along an axis: every lane does the general walk
        gather_box(G, p, r, s);
        return;
    }
    GatherSums part;
    part.ldir = mk3(0.0f, 0.0f, 0.0f);
    part.intensity = mk3(0.0f, 0.0f, 0.0f);
    part.k = 0;
    part.candidates = 0;
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t c = 0; c < 9; c++) {
        const uint32_t first = min(C.rg[c].x, G.count), n = min(C.rg[c].y, G.count - first);  // guard: stay inside the map
        for (uint32_t k = lane; k < n; k += 64u) take_record(G, first + k, G.pos_cell[first + k], lo, hi, part);
    }
    s.ldir = mk3(wave_sum_f32(part.ldir.x), wave_sum_f32(part.ldir.y), wave_sum_f32(part.ldir.z));
    s.intensity = mk3(wave_sum_f32(part.intensity.x), wave_sum_f32(part.intensity.y), wave_sum_f32(part.intensity.z));
    uint32_t k = part.k;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) k += (uint32_t)__shfl_xor((int)k, off, 64);
    s.k = k;
    s.candidates = C.candidates;
}
// the same over a fine grid: the shell's records strided over the lanes, the whole cells one per lane
__device__ __forceinline__ void wave_gather_fine(const DevPhotonMap &G, f3 p, float r, GatherSums &s) {
    f3 lo = p - mk3(r, r, r), hi = p + mk3(r, r, r);
    FineBox F;
    if (!fine_box(G, lo, hi, F)) {
        gather_box(G, p, r, s);
        return;
    }
    GatherSums part;
    part.ldir = mk3(0.0f, 0.0f, 0.0f);
    part.intensity = mk3(0.0f, 0.0f, 0.0f);
    part.k = 0;
    part.candidates = 0;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t shell = 0;
    visit_fine(
        G, F,
        [&](uint32_t first, uint32_t count) {
            shell += count;
            first = min(first, G.count);
            const uint32_t n = min(count, G.count - first);  // guard: stay inside the map
            for (uint32_t k = lane; k < n; k += 64u) take_record(G, first + k, G.pos_cell[first + k], lo, hi, part);
        },
        [&](uint32_t first_cell, uint32_t last_cell, uint32_t count) {
            const uint32_t c = first_cell + lane;
            if (c <= last_cell) add_cells(G, c, c, part);
            if (lane == 0) part.k += count;
        });
    s.ldir = mk3(wave_sum_f32(part.ldir.x), wave_sum_f32(part.ldir.y), wave_sum_f32(part.ldir.z));
    s.intensity = mk3(wave_sum_f32(part.intensity.x), wave_sum_f32(part.intensity.y), wave_sum_f32(part.intensity.z));
    uint32_t k = part.k;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) k += (uint32_t)__shfl_xor((int)k, off, 64);
    s.k = k;
    s.candidates = shell;
}

}  // namespace rrt
