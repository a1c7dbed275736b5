// dev_photon.h — counter-based RNG streams, the photon hash grid and the radiance-estimate gather.
//
// Replaces core/photon_tree.rs (third-party kd-tree 0.5.1, `within` = closed axis-aligned box) with
// a uniform hash grid whose cell edge equals the box edge (2 * half_width): a query box overlaps at
// most 2 (rarely 3) cells per axis.  Records are sorted by cell so a query streams contiguous runs.
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
    uint32_t k;    // photons inside the closed box
    uint32_t candidates;
};

// cell key -> (first record, count) through the open-addressing table
__device__ __forceinline__ uint2 lookup_cell(const DevPhotonMap &G, uint64_t key) {
    uint32_t slot = hash_key(key) & G.table_mask;
    for (;;) {
        uint64_t k = G.keys[slot];
        if (k == key) return G.ranges[slot];
        if (k == ~0ULL) return make_uint2(0, 0);
        slot = (slot + 1) & G.table_mask;
    }
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
    uint32_t i = range.x, end = range.x + range.y;
    for (; i + 4 <= end; i += 4) {  // four independent position loads in flight
        float4 q0 = G.pos_cell[i], q1 = G.pos_cell[i + 1], q2 = G.pos_cell[i + 2], q3 = G.pos_cell[i + 3];
        take_record(G, i, q0, lo, hi, s);
        take_record(G, i + 1, q1, lo, hi, s);
        take_record(G, i + 2, q2, lo, hi, s);
        take_record(G, i + 3, q3, lo, hi, s);
    }
    for (; i < end; i++) take_record(G, i, G.pos_cell[i], lo, hi, s);
}

// PhotonTree::get_within_distance (photon_tree.rs:15-35) + the accumulation of
// average_photon_of_type_at (photon_scene.rs:401-407).  The reference adds in order of distance,
// this adds in cell order: same set, f32 sums differ by rounding only.
__device__ __forceinline__ void gather_box(const DevPhotonMap &G, f3 p, float r, GatherSums &s) {
    s.ldir = mk3(0.0f, 0.0f, 0.0f);
    s.intensity = mk3(0.0f, 0.0f, 0.0f);
    s.k = 0;
    s.candidates = 0;
    if (G.count == 0) return;
    f3 lo = p - mk3(r, r, r), hi = p + mk3(r, r, r);  // photon_tree.rs:16-18
    uint32_t x0 = cell_coord(lo.x, G.inv_cell), x1 = cell_coord(hi.x, G.inv_cell);
    uint32_t y0 = cell_coord(lo.y, G.inv_cell), y1 = cell_coord(hi.y, G.inv_cell);
    uint32_t z0 = cell_coord(lo.z, G.inv_cell), z1 = cell_coord(hi.z, G.inv_cell);
    // Two phases, so that the memory system sees independent requests instead of one dependent chain per lane:
    // first every overlapped cell's (first, count) is looked up, then the cells are streamed four records at a time.
    // The order of accumulation (cells x-major, records ascending) is the same either way.
    const bool small = x1 - x0 <= 1u && y1 - y0 <= 1u && z1 - z0 <= 1u;  // cell edge = 2r: always, barring rounding
    if (small) {
        uint2 rg[8];
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) {
            uint32_t cx = x0 + (c >> 2), cy = y0 + ((c >> 1) & 1u), cz = z0 + (c & 1u);
            rg[c] = (cx <= x1 && cy <= y1 && cz <= z1) ? lookup_cell(G, cell_key(cx, cy, cz)) : make_uint2(0, 0);
        }
#pragma unroll
        for (uint32_t c = 0; c < 8; c++) stream_cell(G, rg[c], lo, hi, s);
    } else {
        for (uint32_t cx = x0; cx <= x1; cx++)
            for (uint32_t cy = y0; cy <= y1; cy++)
                for (uint32_t cz = z0; cz <= z1; cz++) stream_cell(G, lookup_cell(G, cell_key(cx, cy, cz)), lo, hi, s);
    }
}

// average_photon_at (photon_scene.rs:363-420): regular and caustic estimates and their blend
__device__ __forceinline__ bool average_photon_at(const DevPhotonMap maps[2], f3 p, float r, f3 &incident, f3 &intensity,
                                                  unsigned long long &gathered, unsigned long long &candidates) {
    GatherSums reg, cau;
    gather_box(maps[0], p, r, reg);
    gather_box(maps[1], p, r, cau);
    gathered += reg.k + cau.k;
    candidates += reg.candidates + cau.candidates;
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

}  // namespace rrt
