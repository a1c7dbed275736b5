// orc_scene.h — objects, materials, lights of the CPU oracle (TEST INFRASTRUCTURE ONLY).
// Restates src/objects/*.rs, src/materials/*.rs, src/lights/*.rs.  Class layout follows the
// reference's traits (Object, Material + PhotonMaterial, Light) so that each method can be read
// next to the Rust it follows.
#pragma once
#include <memory>
#include <stdexcept>
#include <vector>

#include "orc_math.h"

namespace orc {

struct Object;
struct Material;
struct Env;

struct HitVecOverflow : std::runtime_error {
    HitVecOverflow() : std::runtime_error("HitVec overflow") {}
};

// core/hit.rs:5-13.  `what`/`prim` identify the producer for the object-id contract
// (SURVEY.md §8a quirk 11); the reference only holds a borrowed pointer.
struct Hit {
    float distance;
    bool entering;
    const Object *what;
    const Material *material;
    V3 position;
    V3 normal;
    bool has_tex;
    float u, v;
    uint32_t prim;  // triangle index (mesh) / leaf index (csg) / 0
};

// core/hit.rs:56-94: inline list of at most 6 hits, push panics on overflow
struct HitVec {
    Hit hits[6];
    int len = 0;
    void push(const Hit &h) {
        if (len >= 6) throw HitVecOverflow();
        hits[len++] = h;
    }
};

// core/hit.rs:16-34
inline Hit hit_new(const Object *what, bool entering, float distance, V3 position, V3 normal,
                   const Material *material, bool has_tex, float u, float v) {
    return Hit{distance, entering, what, material, position, normal, has_tex, u, v, 0u};
}
// core/hit.rs:37-52
inline Hit hit_infinity(const Object *what, bool entering, float distance, const Material *material) {
    return Hit{distance, entering, what, material, V3{0, 0, 0}, V3{0, 0, 0}, false, 0.0f, 0.0f, 0u};
}

// objects/object.rs:3-6
struct Object {
    virtual ~Object() {}
    virtual HitVec intersect(const Ray &ray) const = 0;
    virtual void apply_transform(const Mat4 &t) = 0;
    // leaf numbering inside CSG trees (post-order), see Env::finalise
    virtual void number_leaves(uint32_t &next) { leaf_index = next++; }
    uint32_t leaf_index = 0;
};

// core/photon.rs:8-20 (+ key, see include/rrt.h)
enum PhotonType : uint8_t { PT_COLOUR = 0, PT_SHADOW = 1, PT_CAUSTIC = 2, PT_VUEON = 3 };
struct Photon {
    V3 position;
    V3 incident;
    Colour intensity;
    uint8_t type;
    uint64_t key;
};

enum Behaviour { B_ABSORB = 0, B_DIFFUSE = 1, B_SPECULAR = 2, B_REFLECT_OR_REFRACT = 3 };

// materials/material.rs:10-36 (Material) and :51-77 (PhotonMaterial) in one class
struct Material {
    virtual ~Material() {}
    virtual Colour compute_once(const Env &scene, const Ray &viewer, const Hit &hit, int depth) const = 0;
    virtual Colour compute_per_light(V3 viewer, const Hit &hit, V3 ldir) const { return black(); }
    virtual bool normal(float u, float v, V3 &out) const { return false; }
    // PhotonMaterial defaults, material.rs:54-76
    virtual float behaviour_weight(int b) const {
        switch (b) {
            case B_ABSORB: return 0.6f;
            case B_DIFFUSE: return 0.3f;
            case B_SPECULAR: return 0.1f;
            default: return 0.0f;
        }
    }
    virtual bool bounced_photon(const Photon &photon, const Hit &hit, Colour &out) const = 0;
    virtual Colour render_vueon(const Hit &hit, const Photon &photon, V3 viewer) const { return black(); }
    virtual float refract_chance(float kr) const { return 0.0f; }
    virtual bool refracted_direction(const Hit &hit, V3 incoming, Ray &ray, float &kr) const { return false; }
};

// materials/texture.rs:15-125
struct Image {
    uint32_t width = 0, height = 0;
    std::vector<Colour> pixels;
    Colour get_xy(uint32_t x, uint32_t y) const {  // texture.rs:105-109
        uint32_t idx = (y % height) * width + (x % width);
        return pixels[idx];
    }
    Colour get_uv(float u, float v) const {  // texture.rs:111-119
        uint32_t x = f32_as_u32(floorf(rem_euclid(-u, 1.0f) * (float)(width - 1)));
        float max_y = (float)width / (float)height;
        uint32_t y = f32_as_u32(floorf(rem_euclid(-v, 1.0f) * max_y * (float)(height - 1)));
        return get_xy(x, y);
    }
};

// lights/light.rs:6-19
struct Light {
    int kind;
    V3 position, direction;
    Colour intensity;
    bool get_direction(V3 surface, V3 &out) const;
    Colour get_intensity(V3 surface) const;
};

}  // namespace orc
