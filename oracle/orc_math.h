// orc_math.h — value types of the CPU oracle (TEST INFRASTRUCTURE ONLY, see oracle.h).
// Restates src/core/{vector,vertex,transform,colour}.rs with the reference's evaluation order.
// Build with -ffp-contract=off: Rust never contracts a*b+c into an FMA.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

static constexpr float PI_F = 3.14159265358979323846f;  // std::f32::consts::PI

// core/vector.rs:4-8 (Vector) and core/vertex.rs:4-8 (Vertex) share one struct here: the
// reference converts freely between them (vertex.rs:23-25, vector.rs:231-239).
struct V3 {
    float x, y, z;
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
// vector.rs:173-183 / vertex.rs:51-61
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
// vector.rs:161-171 / vertex.rs:71-81
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
// vector.rs:185-195 (and :197-207: an f64 literal is cast to f32 first, same value)
inline V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
// vector.rs:149-159
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
// vector.rs:209-219
inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
// vector.rs:70-72
inline float len_sqrd(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// vector.rs:74-76
inline float length(V3 a) { return sqrtf(len_sqrd(a)); }
// vector.rs:78-80
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// vector.rs:61-68
inline V3 normalised(V3 a) {
    float l = length(a);
    return V3{a.x / l, a.y / l, a.z / l};
}
// vector.rs:106-112
inline V3 cross(V3 a, V3 b) {
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// vector.rs:82-90: normal.reflection(initial)
inline V3 reflection(V3 normal, V3 initial) {
    float d = dot(normal, initial) * 2.0f;
    return V3{initial.x - d * normal.x, initial.y - d * normal.y, initial.z - d * normal.z};
}
// vector.rs:114-119
inline float angle(V3 a, V3 b) {
    float d = dot(a, b);
    float l = length(a) * length(b);
    return acosf(d) / l;
}

// core/transform.rs:4-6
struct Mat4 {
    float m[4][4];
};
inline Mat4 mat4_identity() {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = (i == j) ? 1.0f : 0.0f;
    return r;
}
// transform.rs:179-189
inline Mat4 transposed(const Mat4 &a) {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i];
    return r;
}
// transform.rs:192-209: four products summed left to right
inline Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] +
                        a.m[i][3] * b.m[3][j];
    return r;
}

// transform.rs:46-177: adjugate by cofactors.  Each entry of the adjugate is a sum of six signed
// triple products m[r0][c0]*m[r1][c1]*m[r2][c2], multiplied left to right and accumulated left to
// right.  The term tables below are the row/column index triples of that expansion in the
// reference's order; the sign of the leading factor is folded into the first product exactly as
// in the reference (`-a*b*c` is `(-a)*b*c`).
struct CofTerm {
    int8_t sign;  // +1 / -1
    uint8_t r0, c0, r1, c1, r2, c2;
};
// out[R][C] = sum terms.  Row index of output first.
bool mat4_inverse(const Mat4 &a, Mat4 &out);

// vector.rs:121-131: 3x3 part only
inline V3 apply_to_vector(const Mat4 &t, V3 v) {
    const float(*m)[4] = t.m;
    float x = m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z;
    float y = m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z;
    float z = m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z;
    return V3{x, y, z};
}
// vertex.rs:35-48: affine
inline V3 apply_to_vertex(const Mat4 &t, V3 v) {
    const float(*m)[4] = t.m;
    float x = m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z + m[0][3];
    float y = m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z + m[1][3];
    float z = m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z + m[2][3];
    return V3{x, y, z};
}
// vector.rs:133-146
inline V3 to_tangent_space(V3 self, V3 tangent, V3 normal) {
    tangent = normalised(tangent);
    normal = normalised(normal);
    V3 bitangent = cross(normal, tangent);
    Mat4 r = mat4_identity();
    r.m[0][0] = tangent.x; r.m[0][1] = bitangent.x; r.m[0][2] = normal.x;
    r.m[1][0] = tangent.y; r.m[1][1] = bitangent.y; r.m[1][2] = normal.y;
    r.m[2][0] = tangent.z; r.m[2][1] = bitangent.z; r.m[2][2] = normal.z;
    return normalised(apply_to_vector(r, self));
}

// core/colour.rs:2-6
struct Colour {
    float r, g, b;
};
inline Colour black() { return Colour{0.0f, 0.0f, 0.0f}; }
inline Colour white() { return Colour{1.0f, 1.0f, 1.0f}; }
inline Colour operator+(Colour a, Colour b) { return Colour{a.r + b.r, a.g + b.g, a.b + b.b}; }
inline Colour operator*(Colour a, Colour b) { return Colour{a.r * b.r, a.g * b.g, a.b * b.b}; }
inline Colour operator*(Colour a, float s) { return Colour{a.r * s, a.g * s, a.b * s}; }
inline Colour operator/(Colour a, float s) { return Colour{a.r / s, a.g / s, a.b / s}; }

// Rust `f32::rem_euclid` (core::f32): r = self % rhs; if r < 0 { r + |rhs| } else { r }
inline float rem_euclid(float a, float b) {
    float r = fmodf(a, b);
    return (r < 0.0f) ? r + fabsf(b) : r;
}
// Rust `as u32` on f32: saturating, NaN -> 0
inline uint32_t f32_as_u32(float f) {
    if (!(f > 0.0f)) return 0u;  // NaN, negatives, zero
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
// Rust `as u8` on f32: saturating, NaN -> 0 (framebuffer.rs:105-107,124)
inline uint8_t f32_as_u8(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 255.0f) return 255;
    return (uint8_t)f;
}

struct Ray {
    V3 position;
    V3 direction;
};

}  // namespace orc
