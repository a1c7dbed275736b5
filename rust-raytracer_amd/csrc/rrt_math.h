// rrt_math.h — f32 vector helpers shared by host flattening code and gfx950 kernels.
//
// Every operation keeps the reference's evaluation order (src/core/vector.rs, vertex.rs) because
// the parity contract is bit-exact object ids / depth: the library is built with
// -ffp-contract=off (no FMA fusion; Rust never fuses) and relies on hipcc's default correctly
// rounded f32 divide and sqrt.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define RRT_HD __host__ __device__ __forceinline__

namespace rrt {

struct f3 {
    float x, y, z;
};

RRT_HD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
RRT_HD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RRT_HD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RRT_HD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
RRT_HD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RRT_HD f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
RRT_HD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
// vector.rs:78-80: (x*x + y*y) + z*z
RRT_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RRT_HD float lensq3(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RRT_HD float len3(f3 a) { return sqrtf(lensq3(a)); }
// vector.rs:61-68: three divisions by the length, not a multiply by the reciprocal
RRT_HD f3 norm3(f3 a) {
    float l = len3(a);
    return f3{a.x / l, a.y / l, a.z / l};
}
// vector.rs:106-112
RRT_HD f3 cross3(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// vector.rs:82-90: n.reflection(i) = i - (n.i * 2) n
RRT_HD f3 reflect3(f3 n, f3 i) {
    float d = dot3(n, i) * 2.0f;
    return f3{i.x - d * n.x, i.y - d * n.y, i.z - d * n.z};
}
// vector.rs:121-131 with the columns of the tangent frame given explicitly
// (to_tangent_space, vector.rs:133-146: rows are [t.x b.x n.x] etc.)
RRT_HD f3 tbn_apply(f3 t, f3 b, f3 n, f3 v) {
    return f3{t.x * v.x + b.x * v.y + n.x * v.z, t.y * v.x + b.y * v.y + n.y * v.z, t.z * v.x + b.z * v.y + n.z * v.z};
}
RRT_HD f3 to_tangent_space3(f3 self, f3 tangent, f3 normal) {
    f3 t = norm3(tangent);
    f3 n = norm3(normal);
    f3 b = cross3(n, t);
    return norm3(tbn_apply(t, b, n, self));
}

// Rust f32::rem_euclid
RRT_HD float rem_euclid_f(float a, float b) {
    float r = fmodf(a, b);
    return (r < 0.0f) ? r + fabsf(b) : r;
}
// Rust `as u32` (saturating, NaN -> 0)
RRT_HD uint32_t sat_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

}  // namespace rrt
