// dev_shade.h — materials and lights on the device (src/materials/*.rs, src/lights/*.rs).
// Parameters come from the host-built tables in dev_scene.h; evaluation happens per lane.
#pragma once
#include "dev_pow.h"
#include "dev_trace.h"

namespace rrt {

#define RRT_ERR_BIT_OVERFLOW 1u  // HitVec overflow (core/hit.rs:70-73)
#define RRT_ERR_BIT_NO_TEX 2u    // "No texture coordinates" (texture.rs:157-161)
#define RRT_ERR_BIT_SHADOW_CAP 4u

// Phong::colour_at_hit (phong_material.rs:79-81, texture.rs:156-163)
__device__ __forceinline__ f3 part_colour(const DevScene &S, const MatPart &P, const HitAttr &h, Counters &cnt) {
    if (P.kind == RRT_MAT_MONOCHROME) return mk3(P.colour[0], P.colour[1], P.colour[2]);
    if (!h.has_tex) {
        cnt.overflow |= RRT_ERR_BIT_NO_TEX;
        return mk3(0.0f, 0.0f, 0.0f);
    }
    float k = 1.0f / P.scale;
    return tex_fetch(S, P.diffuse_tex, h.u * k, h.v * k);
}
// falsecolour_material.rs:18-24
__device__ __forceinline__ f3 falsecolour(const HitAttr &h) {
    return mk3((h.normal.x + 1.0f) * 0.5f, (h.normal.y + 1.0f) * 0.5f, (h.normal.z + 1.0f) * 0.5f);
}
// phong_material.rs:26-29: unclamped
__device__ __forceinline__ f3 phong_diffuse(f3 colour, const HitAttr &h, f3 ldir) {
    float s = -dot3(h.normal, ldir);
    return colour * s;
}
// phong_material.rs:30-34
__device__ __forceinline__ f3 phong_specular(const HitAttr &h, f3 ldir, f3 viewer, float shininess) {
    f3 refl = norm3(reflect3(h.normal, ldir));
    // through f64, rounded to f32 once: a correctly rounded powf on both sides of the parity check (see eval_hit);
    // dev_pow.h says why this is not ocml's pow
    float s = pow_f32(fmaxf(dot3(viewer, refl), 0.0f), shininess);
    return mk3(1.0f * s, 1.0f * s, 1.0f * s);
}

// compute_once of every part that is not a Global, folded like compound_material.rs:80-87
// (a Global part is last and is added by the combine pass).
__device__ __forceinline__ f3 material_once_pre(const DevScene &S, const DevMaterial &M, const HitAttr &h, Counters &cnt) {
    f3 acc = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < RRT_MAX_COMPOUND_PARTS; i++) {
        if (i >= M.n_parts) break;
        const MatPart &P = M.parts[i];
        if (P.kind == RRT_MAT_GLOBAL) continue;
        f3 v = (P.kind == RRT_MAT_FALSECOLOUR) ? falsecolour(h) : part_colour(S, P, h, cnt) * P.ambient;
        acc = M.is_compound ? acc + v : v;
    }
    return acc;
}
// compute_per_light (phong_material.rs:43-51, compound_material.rs:89-102)
__device__ __forceinline__ f3 material_per_light(const DevScene &S, const DevMaterial &M, const HitAttr &h, f3 viewer, f3 ldir,
                                                 Counters &cnt) {
    f3 acc = mk3(0.0f, 0.0f, 0.0f);
    // not unrolled: the f64 pow of the specular term is some 300 instructions and three dozen live doubles; four copies of
    // it pushed the mesh kernels 37 VGPRs over their budget (1.8 GB of scratch writes per 4K frame, profiles/r02)
#pragma unroll 1
    for (int i = 0; i < M.n_parts; i++) {
        const MatPart &P = M.parts[i];
        f3 v = mk3(0.0f, 0.0f, 0.0f);
        if (P.kind == RRT_MAT_MONOCHROME || P.kind == RRT_MAT_TEXTURE)
            v = phong_diffuse(part_colour(S, P, h, cnt), h, ldir) + phong_specular(h, ldir, viewer, P.shininess);
        acc = M.is_compound ? acc + v : v;
    }
    if (M.is_compound) acc = acc / (float)M.n_parts;
    return acc;
}
// PhotonMaterial::render_vueon (phong_material.rs:101-105, falsecolour_material.rs:52-59,
// compound_material.rs:141-146: a plain sum, not a mean)
__device__ __forceinline__ f3 material_render_vueon(const DevScene &S, const DevMaterial &M, const HitAttr &h, f3 incident,
                                                    f3 intensity, f3 viewer, Counters &cnt) {
    f3 acc = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < RRT_MAX_COMPOUND_PARTS; i++) {
        if (i >= M.n_parts) break;
        const MatPart &P = M.parts[i];
        f3 v = mk3(0.0f, 0.0f, 0.0f);
        if (P.kind == RRT_MAT_MONOCHROME || P.kind == RRT_MAT_TEXTURE)
            v = (phong_diffuse(part_colour(S, P, h, cnt), h, incident) + phong_specular(h, incident, viewer, P.shininess)) * intensity;
        else if (P.kind == RRT_MAT_FALSECOLOUR)
            v = falsecolour(h) * intensity;
        acc = M.is_compound ? acc + v : v;
    }
    return acc;
}
// PhotonMaterial::bounced_photon (phong_material.rs:97-99, falsecolour_material.rs:48-50,
// compound_material.rs:130-139: sum of the parts that return Some)
__device__ __forceinline__ bool material_bounced(const DevScene &S, const DevMaterial &M, const HitAttr &h, f3 incident, f3 intensity,
                                                 Counters &cnt, f3 &out) {
    bool have = false;
    f3 acc = mk3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < RRT_MAX_COMPOUND_PARTS; i++) {
        if (i >= M.n_parts) break;
        const MatPart &P = M.parts[i];
        if (P.kind == RRT_MAT_GLOBAL) continue;
        f3 v = (P.kind == RRT_MAT_FALSECOLOUR) ? falsecolour(h) * intensity
                                               : phong_diffuse(part_colour(S, P, h, cnt), h, incident) * intensity;
        acc = have ? acc + v : v;
        have = true;
    }
    out = acc;
    return have;
}

// GlobalMaterial::refraction (global_material.rs:28-67).  `entering` is always true for hits that
// come out of trace (environment.rs:42-44), the parameter keeps the formula whole.
__device__ __forceinline__ bool global_refraction(const MatPart &G, const HitAttr &h, bool entering, f3 incoming, f3 &origin,
                                                  f3 &direction, float &kr) {
    if (G.wt == 0.0f) return false;
    f3 N = h.normal;
    f3 I = -incoming;
    float cos_i = dot3(N, I);
    float n1, n2;
    if (entering) {
        N = -N;
        n1 = 1.0f;
        n2 = G.ior;
    } else {
        cos_i = -cos_i;
        n1 = G.ior;
        n2 = 1.0f;
    }
    float eta = n1 / n2;
    float cos_t = sqrtf(1.0f - (eta * eta) * (1.0f - cos_i * cos_i));
    if (isnan(cos_t)) return false;  // total internal reflection
    f3 T = I * eta - N * (cos_t - eta * cos_i);
    T = -norm3(T);
    float r_par = (n2 * cos_i - n1 * cos_t) / (n2 * cos_i + n1 * cos_t);
    float r_per = (n1 * cos_i - n2 * cos_t) / (n1 * cos_i + n2 * cos_t);
    kr = (r_par * r_par + r_per * r_per) / 2.0f;
    origin = h.position + T * 0.0001f;
    direction = T;
    return true;
}

// Light::get_direction (point_light.rs:32-35, directional_light.rs:20-22, directional_point_light.rs:34-43)
__device__ __forceinline__ bool light_direction(const DevLight &L, f3 surface, f3 &ldir) {
    if (L.kind == RRT_LIGHT_DIRECTIONAL) {
        ldir = L.direction;
        return true;
    }
    f3 d = surface - L.position;
    if (L.kind == RRT_LIGHT_DIRPOINT && dot3(d, L.direction) < 0.0f) return false;
    ldir = norm3(d);
    return true;
}
// Light::get_intensity (point_light.rs:37-39, directional_point_light.rs:45-50)
__device__ __forceinline__ f3 light_intensity(const DevLight &L, f3 surface) {
    if (L.kind == RRT_LIGHT_DIRPOINT) {
        f3 d = norm3(surface - L.position);
        return L.intensity * dot3(d, L.direction);
    }
    return L.intensity;
}

}  // namespace rrt
