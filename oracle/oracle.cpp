// oracle.cpp — CPU restatement of the reference's ray/scene-intersection and photon-mapping path.
//
// TEST INFRASTRUCTURE ONLY (see oracle.h): loaded by tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py, never by the product.  Every function cites the reference
// file:line it follows (paths under /root/reference/src).  Same brute-force algorithms as the
// reference: a flat loop over objects, a flat loop over mesh triangles, recursive shading.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <string>
#include <thread>

#include "orc_scene.h"

namespace orc {

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

// ------------------------------------------------------------------------------------------
// core/transform.rs:46-177
bool mat4_inverse(const Mat4 &a, Mat4 &out) {
    const float(*m)[4] = a.m;
    float inv[4][4];
    for (int R = 0; R < 4; R++) {
        for (int C = 0; C < 4; C++) {
            int rows[3], cols[3], nr = 0, nc = 0;
            for (int i = 0; i < 4; i++) {
                if (i != C) rows[nr++] = i;
                if (i != R) cols[nc++] = i;
            }
            // six terms in the reference's order: leading row rows[k] (column cols[0]), the other
            // two rows ascending, first with columns (cols[1], cols[2]) then (cols[2], cols[1]);
            // signs (+,-,-,+,+,-) times (-1)^(R+C)
            static const int base_sign[6] = {+1, -1, -1, +1, +1, -1};
            int s = ((R + C) & 1) ? -1 : +1;
            float acc = 0.0f;
            for (int k = 0; k < 3; k++) {
                int ra = rows[k];
                int rb = rows[k == 0 ? 1 : 0];
                int rc = rows[k == 2 ? 1 : 2];
                for (int j = 0; j < 2; j++) {
                    int cb = cols[j == 0 ? 1 : 2], cc = cols[j == 0 ? 2 : 1];
                    float term = m[ra][cols[0]] * m[rb][cb] * m[rc][cc];
                    int sign = base_sign[k * 2 + j] * s;
                    if (k == 0 && j == 0)
                        acc = (sign > 0) ? term : -term;
                    else
                        acc = (sign > 0) ? acc + term : acc - term;
                }
            }
            inv[R][C] = acc;
        }
    }
    float det = m[0][0] * inv[0][0] + m[0][1] * inv[1][0] + m[0][2] * inv[2][0] + m[0][3] * inv[3][0];
    if (det == 0.0f) return false;  // transform.rs:165-167 panics
    det = 1.0f / det;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out.m[i][j] = inv[i][j] * det;
    return true;
}

// ------------------------------------------------------------------------------------------
// lights: point_light.rs:31-39, directional_light.rs:19-27, directional_point_light.rs:33-50
bool Light::get_direction(V3 surface, V3 &out) const {
    switch (kind) {
        case RRT_LIGHT_DIRECTIONAL:
            out = direction;  // not normalised (directional_light.rs:20-22)
            return true;
        case RRT_LIGHT_POINT:
            out = normalised(surface - position);
            return true;
        default: {
            V3 d = surface - position;
            if (dot(d, direction) < 0.0f) return false;
            out = normalised(d);
            return true;
        }
    }
}
Colour Light::get_intensity(V3 surface) const {
    if (kind == RRT_LIGHT_DIRPOINT) {
        V3 d = normalised(surface - position);
        float k = dot(d, direction);
        return intensity * k;
    }
    return intensity;
}

// ------------------------------------------------------------------------------------------
// objects/plane_object.rs
struct Plane : Object {
    V3 centre, up, normal;
    float d;
    const Material *material;
    Plane() {}
    Plane(V3 point, V3 up_, V3 normal_, const Material *m)  // new_raw, plane_object.rs:27-40
        : centre(point), up(up_), normal(normal_), d(-dot(normal_, point)), material(m) {}

    HitVec intersect(const Ray &ray) const override {  // plane_object.rs:54-113
        HitVec out;
        float U = dot(normal, ray.position) + d;
        float V = dot(normal, ray.direction);
        if (V == 0.0f) {
            if (U >= 0.0f) {
                out.push(hit_infinity(this, true, -INFINITY, material));
                out.push(hit_infinity(this, false, INFINITY, material));
            }
            return out;
        }
        float t = U / -V;
        if (V > 0.0f) {
            out.push(hit_infinity(this, true, -INFINITY, material));
            out.push(hit_new(this, false, t, ray.position + ray.direction * t, -normal, material, false, 0, 0));
        } else {
            V3 n = normal;
            V3 position = ray.position + ray.direction * t;
            V3 from_center = position - centre;
            V3 v_unit = normalised(up);
            V3 u_unit = normalised(cross(v_unit, normal));
            float u = dot(from_center, u_unit);
            float v = dot(from_center, v_unit);
            V3 nm;
            if (material->normal(u, v, nm)) {
                V3 right = normalised(cross(normal, up));
                n = to_tangent_space(nm, right, n);
            }
            out.push(hit_new(this, true, t, position, n, material, true, u, v));
            out.push(hit_infinity(this, false, INFINITY, material));
        }
        return out;
    }
    void apply_transform(const Mat4 &t) override {  // plane_object.rs:115-124
        centre = apply_to_vertex(t, centre);
        up = apply_to_vector(t, up);
        normal = apply_to_vector(t, normal);
        normal = normalised(normal);
        up = normalised(up);
        d = -dot(normal, centre);
    }
};

// objects/sphere_object.rs
struct Sphere : Object {
    V3 centre;
    float radius;
    const Material *material;
    HitVec intersect(const Ray &ray) const override {  // sphere_object.rs:35-91
        HitVec out;
        V3 ro = ray.position - centre;
        float a = dot(ray.direction, ray.direction);
        float b = 2.0f * dot(ray.direction, ro);
        float c = dot(ro, ro) - radius * radius;
        float disc = b * b - 4.0f * a * c;
        if (disc < 0.0f) return out;
        float ds = sqrtf(disc);
        float t0 = (-b - ds) / 2.0f;
        float t1 = (-b + ds) / 2.0f;
        out.push(create_hit(ray, t0, true));
        out.push(create_hit(ray, t1, false));
        return out;
    }
    Hit create_hit(const Ray &ray, float distance, bool entering) const {  // sphere_object.rs:55-88
        V3 position = ray.position + ray.direction * distance;
        V3 n = normalised(position - centre);
        if (dot(n, ray.direction) > 0.0f) n = -n;
        // Rust's f32::atan2 / acos call the platform libm, whose f32 results differ between implementations by an ulp.
        // The oracle pins them as the correctly rounded value: evaluate in f64, round to f32 once.
        float theta = (float)std::atan2((double)n.x, (double)n.z) + PI_F;
        float phi = (float)std::acos((double)-n.y);
        float u = theta / (2.0f * PI_F);
        float v = (PI_F - phi) / PI_F;
        V3 nm;
        if (material->normal(u, v, nm)) {
            V3 tangent = normalised(cross(V3{1.0f, 0.0f, 0.0f}, position - centre));
            nm = to_tangent_space(nm, tangent, n);
            n = normalised(nm);
        }
        return hit_new(this, entering, distance, position, n, material, true, u, v);
    }
    void apply_transform(const Mat4 &t) override {  // sphere_object.rs:93-100
        centre = apply_to_vertex(t, centre);
        if (t.m[0][0] == t.m[1][1] && t.m[1][1] == t.m[2][2]) radius *= t.m[0][0];
    }
};

// objects/cuboid_object.rs
struct Cuboid : Object {
    V3 corner, size;
    const Material *material;
    Plane planes[6];  // left, right, up, down, front, back (iteration order, cuboid_object.rs:27-37)
    void build_planes() {  // cuboid_object.rs:58-90
        V3 fdl = corner;
        V3 ful = corner + V3{0.0f, size.y, 0.0f};
        V3 bdl = corner + V3{0.0f, 0.0f, size.z};
        V3 bdr = corner + V3{size.x, 0.0f, size.z};
        V3 up{0, 1, 0}, down{0, -1, 0}, left{-1, 0, 0}, right{1, 0, 0}, forwards{0, 0, 1}, backwards{0, 0, -1};
        planes[0] = Plane(fdl, up, left, material);
        planes[1] = Plane(bdr, up, right, material);
        planes[2] = Plane(ful, forwards, up, material);
        planes[3] = Plane(fdl, forwards, down, material);
        planes[4] = Plane(fdl, up, backwards, material);
        planes[5] = Plane(bdl, down, forwards, material);
    }
    HitVec intersect(const Ray &ray) const override {  // cuboid_object.rs:94-148
        bool have_first = false, have_back = false;
        Hit first_hit{}, back_hit{};
        for (int p = 0; p < 6; p++) {
            HitVec hits = planes[p].intersect(ray);
            for (int i = 0; i < hits.len; i++) {
                Hit hit = hits.hits[i];
                V3 hp = hit.position;
                bool inside = hp.x >= corner.x - 0.0001f && hp.x <= corner.x + size.x + 0.0001f &&
                              hp.y >= corner.y - 0.0001f && hp.y <= corner.y + size.y + 0.0001f &&
                              hp.z >= corner.z - 0.0001f && hp.z <= corner.z + size.z + 0.0001f;
                if (!inside) continue;
                if (hit.distance < 0.0f) continue;
                if (std::isinf(hit.distance)) continue;
                hit.prim = (uint32_t)(p * 2 + i);
                if (hit.entering) {
                    if (!(have_first && first_hit.distance > hit.distance)) {
                        first_hit = hit;
                        have_first = true;
                    }
                } else if (!(have_back && back_hit.distance < hit.distance)) {
                    back_hit = hit;
                    have_back = true;
                }
            }
        }
        HitVec out;
        if (have_first) out.push(first_hit);
        if (have_back) out.push(back_hit);
        return out;
    }
    void apply_transform(const Mat4 &t) override {  // cuboid_object.rs:150-160
        corner = apply_to_vertex(t, corner);
        size = V3{size.x * t.m[0][0], size.y * t.m[1][1], size.z * t.m[2][2]};
        build_planes();
    }
};

// objects/quadratic_object.rs
struct Quadratic : Object {
    float q[10];  // a b c d e f g h i j
    const Material *material;
    HitVec intersect(const Ray &ray) const override {  // quadratic_object.rs:43-124
        HitVec out;
        V3 P = ray.position, D = ray.direction;
        float a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6], h = q[7], i = q[8], j = q[9];
        float Aq = a * (D.x * D.x) + 2.0f * b * D.x * D.y + 2.0f * c * D.x * D.z + e * (D.y * D.y) +
                   2.0f * f * D.y * D.z + h * (D.z * D.z);
        float Bq = 2.0f * (a * P.x * D.x + b * (P.x * D.y + P.y * D.x) + c * (P.x * D.z + D.x * P.z) + d * D.x +
                           e * P.y * D.y + f * (P.y * D.z + D.y * P.z) + g * D.y + h * P.z * D.z + i * D.z);
        float Cq = a * (P.x * P.x) + 2.0f * b * P.x * P.y + 2.0f * c * P.x * P.z + 2.0f * d * P.x +
                   e * (P.y * P.y) + 2.0f * f * P.y * P.z + 2.0f * g * P.y + h * (P.z * P.z) + 2.0f * i * P.z + j;
        if (fabsf(Aq) == 0.0f) return out;
        float disc = Bq * Bq - 4.0f * Aq * Cq;
        if (disc < 0.0f) return out;
        disc = sqrtf(disc);
        float t0 = (-Bq - disc) / (2.0f * Aq);
        float t1 = (-Bq + disc) / (2.0f * Aq);
        if (t0 > t1) std::swap(t0, t1);
        out.push(create_hit(ray, t0, true));
        out.push(create_hit(ray, t1, false));
        return out;
    }
    Hit create_hit(const Ray &ray, float t, bool entering) const {  // quadratic_object.rs:91-118
        float a = q[0], b = q[1], c = q[2], d = q[3], e = q[4], f = q[5], g = q[6], h = q[7], i = q[8];
        V3 p = ray.position + ray.direction * t;
        V3 n = normalised(V3{a * p.x + b * p.y + c * p.z + d, b * p.x + e * p.y + f * p.z + g,
                             c * p.x + f * p.y + h * p.z + i});
        if (dot(n, ray.direction) > 0.0f) n = -n;
        return hit_new(this, entering, t, p, n, material, false, 0, 0);
    }
    void apply_transform(const Mat4 &T) override {  // quadratic_object.rs:127-148
        Mat4 Q = {{{q[0], q[1], q[2], q[3]}, {q[1], q[4], q[5], q[6]}, {q[2], q[5], q[7], q[8]}, {q[3], q[6], q[8], q[9]}}};
        Mat4 Ti;
        if (!mat4_inverse(T, Ti)) throw std::runtime_error("Matrix is not invertible");
        Mat4 Tt = transposed(Ti);
        Mat4 R = mul(mul(Tt, Q), Ti);
        q[0] = R.m[0][0]; q[1] = R.m[0][1]; q[2] = R.m[0][2]; q[3] = R.m[0][3];
        q[4] = R.m[1][1]; q[5] = R.m[1][2]; q[6] = R.m[1][3];
        q[7] = R.m[2][2]; q[8] = R.m[2][3];
        q[9] = R.m[3][3];
    }
};

// objects/triangle_object.rs
struct Triangle {
    V3 a, b, c;
    V3 an, bn, cn;
    bool has_normals;
    V3 ab, bc, ca;
    Plane plane;
    void rebuild(const Material *m) {  // triangle_object.rs:51-53, 85-96
        ab = b - a;
        bc = c - b;
        ca = a - c;
        V3 plane_normal = normalised(cross(ab, bc));
        plane = Plane(a, ab, plane_normal, m);
    }
    HitVec intersect(const Ray &ray, bool smooth) const {  // triangle_object.rs:129-160
        HitVec plane_hits = plane.intersect(ray);
        HitVec out;
        for (int k = 0; k < plane_hits.len; k++) {
            Hit h = plane_hits.hits[k];
            V3 ip = h.position;
            V3 ai = ip - a, bi = ip - b, ci = ip - c;
            V3 ab_n = cross(ai, ab), bc_n = cross(bi, bc), ca_n = cross(ci, ca);
            bool inside = dot(ab_n, bc_n) > 0.0f && dot(bc_n, ca_n) > 0.0f;
            if (!inside) continue;
            if (smooth) {  // smoothen_hit + get_barycentric, triangle_object.rs:98-125
                if (!has_normals) throw std::runtime_error("Vertex normals not set in smoothen_hit()");
                float gamma = length(cross(ai, ab));
                float alpha = length(cross(bi, bc));
                float beta = length(cross(ci, ca));
                V3 n = an * alpha + bn * beta + cn * gamma;
                h.normal = normalised(n);
            }
            out.push(h);
        }
        return out;
    }
};

// objects/polymesh_object.rs
struct PolyMesh : Object {
    std::vector<Triangle> triangles;
    bool smooth;
    const Material *material;
    HitVec intersect(const Ray &ray) const override {  // polymesh_object.rs:216-242
        bool have = false;
        Hit closest{};
        for (size_t t = 0; t < triangles.size(); t++) {
            HitVec th = triangles[t].intersect(ray, smooth);
            for (int k = 0; k < th.len; k++) {
                const Hit &h = th.hits[k];
                if (h.distance < 0.0f) continue;
                if (!have || h.distance < closest.distance) {
                    if (!have || h.distance < closest.distance) {
                        closest = h;
                        closest.prim = (uint32_t)t;
                        have = true;
                    }
                }
            }
        }
        HitVec out;
        if (have) out.push(closest);
        return out;
    }
    void apply_transform(const Mat4 &t) override {  // polymesh_object.rs:244-252, triangle_object.rs:162-173
        for (auto &tri : triangles) {
            tri.a = apply_to_vertex(t, tri.a);
            tri.b = apply_to_vertex(t, tri.b);
            tri.c = apply_to_vertex(t, tri.c);
            if (tri.has_normals) {  // vertex.rs:119-126
                tri.an = normalised(apply_to_vector(t, tri.an));
                tri.bn = normalised(apply_to_vector(t, tri.bn));
                tri.cn = normalised(apply_to_vector(t, tri.cn));
            }
            tri.rebuild(material);
        }
    }
};

// objects/csg_object.rs
enum CsgAction { AEnter, AExit, ADrop, BEnter, BExit, BDrop };
struct Csg : Object {
    int mode;
    std::unique_ptr<Object> left, right;
    bool left_is_csg = false, right_is_csg = false;
    void number_leaves(uint32_t &next) override {
        left->number_leaves(next);
        right->number_leaves(next);
    }
    HitVec intersect(const Ray &ray) const override {  // csg_object.rs:39-108
        static const CsgAction tables[3][8] = {
            {ADrop, BDrop, AExit, BDrop, ADrop, BExit, AEnter, BEnter},   // Union
            {AExit, BExit, ADrop, BEnter, AEnter, BDrop, ADrop, BDrop},   // Intersection
            {ADrop, BEnter, AExit, BExit, ADrop, BDrop, AEnter, BDrop}};  // Difference
        const CsgAction *actions = tables[mode];
        HitVec L = left->intersect(ray), R = right->intersect(ray);
        if (!left_is_csg)
            for (int i = 0; i < L.len; i++) L.hits[i].prim = left->leaf_index;
        if (!right_is_csg)
            for (int i = 0; i < R.len; i++) R.hits[i].prim = right->leaf_index;
        int li = 0, ri = 0;
        HitVec out;
        while (li < L.len && ri < R.len) {
            const Hit &lh = L.hits[li];
            const Hit &rh = R.hits[ri];
            int state = 0;
            if (lh.entering) state += 4;
            if (rh.entering) state += 2;
            if (lh.distance > rh.distance) state += 1;
            CsgAction act = actions[state];
            switch (act) {
                case AEnter:
                case AExit: {
                    Hit h = L.hits[li++];
                    h.entering = (act == AEnter);
                    out.push(h);
                    break;
                }
                case ADrop: li++; break;
                case BEnter:
                case BExit: {
                    Hit h = R.hits[ri++];
                    h.entering = (act == BEnter);
                    out.push(h);
                    break;
                }
                case BDrop: ri++; break;
            }
        }
        if (mode == RRT_CSG_DIFFERENCE) {
            while (li < L.len) out.push(L.hits[li++]);
        } else if (mode == RRT_CSG_UNION) {
            while (li < L.len) out.push(L.hits[li++]);
            while (ri < R.len) out.push(R.hits[ri++]);
        }
        return out;
    }
    void apply_transform(const Mat4 &t) override {  // csg_object.rs:110-113
        left->apply_transform(t);
        right->apply_transform(t);
    }
};

// ------------------------------------------------------------------------------------------
// materials
struct Phong : Material {  // phong_material.rs:12-60, 96-106
    float ambient_strength, shininess;
    virtual Colour colour_at_hit(const Hit &hit) const = 0;
    Colour diffuse(const Hit &hit, V3 ldir) const {
        float s = -dot(hit.normal, ldir);
        return colour_at_hit(hit) * s;
    }
    Colour specular(const Hit &hit, V3 ldir, V3 viewer) const {
        V3 refl = normalised(reflection(hit.normal, ldir));
        float s = (float)std::pow((double)fmaxf(dot(viewer, refl), 0.0f), (double)shininess);  // correctly rounded powf, as above
        return white() * s;
    }
    Colour compute_once(const Env &, const Ray &, const Hit &hit, int) const override {
        return colour_at_hit(hit) * ambient_strength;
    }
    Colour compute_per_light(V3 viewer, const Hit &hit, V3 ldir) const override {
        return diffuse(hit, ldir) + specular(hit, ldir, viewer);
    }
    bool bounced_photon(const Photon &photon, const Hit &hit, Colour &out) const override {
        out = diffuse(hit, photon.incident) * photon.intensity;
        return true;
    }
    Colour render_vueon(const Hit &hit, const Photon &photon, V3 viewer) const override {
        Colour c = diffuse(hit, photon.incident) + specular(hit, photon.incident, viewer);
        return c * photon.intensity;
    }
};
struct Monochrome : Phong {  // phong_material.rs:62-94
    Colour colour;
    Colour colour_at_hit(const Hit &) const override { return colour; }
};
struct Texture : Phong {  // texture.rs:127-189
    const Image *diffuse_img = nullptr;
    const Image *normal_img = nullptr;
    float scale;
    Colour colour_at_hit(const Hit &hit) const override {
        if (!hit.has_tex) throw std::runtime_error("No texture coordinates");
        float k = 1.0f / scale;
        return diffuse_img->get_uv(hit.u * k, hit.v * k);
    }
    bool normal(float u, float v, V3 &out) const override {
        if (!normal_img) return false;
        Colour c = normal_img->get_uv(u * scale, v * scale);
        V3 n = V3{c.r, c.g, c.b} * 2.0f - V3{1.0f, 1.0f, 1.0f};
        out = normalised(n);
        return true;
    }
};
struct FalseColour : Material {  // falsecolour_material.rs
    static Colour colour_at_hit(const Hit &hit) {
        return Colour{(hit.normal.x + 1.0f) * 0.5f, (hit.normal.y + 1.0f) * 0.5f, (hit.normal.z + 1.0f) * 0.5f};
    }
    Colour compute_once(const Env &, const Ray &, const Hit &hit, int) const override { return colour_at_hit(hit); }
    bool bounced_photon(const Photon &photon, const Hit &hit, Colour &out) const override {
        out = colour_at_hit(hit) * photon.intensity;
        return true;
    }
    Colour render_vueon(const Hit &hit, const Photon &photon, V3) const override {
        return colour_at_hit(hit) * photon.intensity;
    }
};
struct GlobalMaterial : Material {  // global_material.rs
    float reflect_weight, refract_weight, ior;
    bool refraction(const Hit &hit, V3 incoming, Ray &ray, float &kr) const {  // :28-67
        if (refract_weight == 0.0f) return false;
        V3 N = hit.normal;
        V3 I = -incoming;
        float cos_i = dot(N, I);
        float n1, n2;
        if (hit.entering) {
            N = -N;
            n1 = 1.0f;
            n2 = ior;
        } else {
            cos_i = -cos_i;
            n1 = ior;
            n2 = 1.0f;
        }
        float eta = n1 / n2;
        float cos_t = sqrtf(1.0f - (eta * eta) * (1.0f - cos_i * cos_i));
        if (std::isnan(cos_t)) return false;
        V3 T = I * eta - N * (cos_t - eta * cos_i);
        T = -normalised(T);
        float r_par = (n2 * cos_i - n1 * cos_t) / (n2 * cos_i + n1 * cos_t);
        float r_per = (n1 * cos_i - n2 * cos_t) / (n1 * cos_i + n2 * cos_t);
        kr = (r_par * r_par + r_per * r_per) / 2.0f;
        ray.position = hit.position + T * 0.0001f;
        ray.direction = T;
        return true;
    }
    Colour compute_once(const Env &scene, const Ray &viewer, const Hit &hit, int depth) const override;  // below
    float behaviour_weight(int b) const override { return b == B_REFLECT_OR_REFRACT ? 1.0f : 0.0f; }
    bool bounced_photon(const Photon &, const Hit &, Colour &) const override { return false; }
    float refract_chance(float kr) const override {  // :129-133
        float reflect = reflect_weight * kr;
        float refract = refract_weight * (1.0f - kr);
        return refract / (reflect + refract);
    }
    bool refracted_direction(const Hit &hit, V3 incoming, Ray &ray, float &kr) const override {
        return refraction(hit, incoming, ray, kr);
    }
};
struct CompoundMaterial : Material {  // compound_material.rs:79-168
    std::vector<const Material *> parts;
    Colour compute_once(const Env &scene, const Ray &viewer, const Hit &hit, int depth) const override {
        Colour acc = black();
        for (auto *m : parts) acc = acc + m->compute_once(scene, viewer, hit, depth);
        return acc / (float)parts.size();
    }
    Colour compute_per_light(V3 viewer, const Hit &hit, V3 ldir) const override {
        Colour acc = black();
        for (auto *m : parts) acc = acc + m->compute_per_light(viewer, hit, ldir);
        return acc / (float)parts.size();
    }
    bool normal(float u, float v, V3 &out) const override {
        for (auto *m : parts)
            if (m->normal(u, v, out)) return true;
        return false;
    }
    float behaviour_weight(int b) const override {
        float acc = 0.0f;
        for (auto *m : parts) acc = acc + m->behaviour_weight(b);
        return acc / (float)parts.size();
    }
    bool bounced_photon(const Photon &photon, const Hit &hit, Colour &out) const override {
        bool have = false;
        Colour acc = black();
        for (auto *m : parts) {
            Colour c;
            if (m->bounced_photon(photon, hit, c)) {
                acc = have ? acc + c : c;
                have = true;
            }
        }
        out = acc;
        return have;
    }
    Colour render_vueon(const Hit &hit, const Photon &photon, V3 viewer) const override {
        Colour acc = black();
        for (auto *m : parts) acc = acc + m->render_vueon(hit, photon, viewer);
        return acc;
    }
    float refract_chance(float kr) const override {
        float acc = 0.0f;
        for (auto *m : parts) acc = acc + m->refract_chance(kr);
        return acc / (float)parts.size();
    }
    bool refracted_direction(const Hit &hit, V3 incoming, Ray &ray, float &kr) const override {
        for (auto *m : parts)
            if (m->refracted_direction(hit, incoming, ray, kr)) return true;
        return false;
    }
};

// ------------------------------------------------------------------------------------------
// RNG.  The reference uses rand 0.8.5 thread_rng (unseeded ChaCha12): parity unpinned by
// construction, only the distributions are contractual.  Oracle and
// product both use Philox4x32-10 (Salmon et al., SC'11), one stream per emitted photon.
static inline void philox_round(uint32_t c[4], const uint32_t k[2]) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    uint32_t k[2] = {key[0], key[1]};
    for (int r = 0; r < 10; r++) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u;
        k[1] += 0xBB67AE85u;
    }
    for (int i = 0; i < 4; i++) out[i] = c[i];
}
// Stream = (seed, light, pass, photon index); draws are consumed in path order.
struct Rng {
    uint32_t key[2];
    uint32_t ctr[4];
    uint32_t buf[4];
    int have = 0;
    Rng(uint64_t seed, uint32_t light, uint32_t pass, uint64_t index) {
        key[0] = (uint32_t)seed;
        key[1] = (uint32_t)(seed >> 32);
        ctr[0] = 0;
        ctr[1] = (uint32_t)index;
        ctr[2] = (uint32_t)(index >> 32);
        ctr[3] = (light << 1) | pass;
    }
    uint32_t next_u32() {
        if (have == 0) {
            philox4x32_10(ctr, key, buf);
            ctr[0]++;
            have = 4;
        }
        return buf[4 - have--];
    }
    float u01() { return (float)(next_u32() >> 8) * 5.9604644775390625e-8f; }  // [0,1), 24 bits
    float sym() { return u01() * 2.0f - 1.0f; }                                // Uniform(-1,1)
    // Vector::random, core/vector.rs:23-42
    V3 unit_vector() {
        for (;;) {
            float x = sym(), y = sym(), z = sym();
            V3 v{x, y, z};
            if (len_sqrd(v) <= 1.0f) return normalised(v);
        }
    }
};

// ------------------------------------------------------------------------------------------
// core/photon_tree.rs: static kd-tree + closed-box query (kd-tree 0.5.1 `within`; parity unpinned)
struct KdTree {
    std::vector<Photon> pts;
    static float at(const Photon &p, int axis) { return axis == 0 ? p.position.x : axis == 1 ? p.position.y : p.position.z; }
    void build_range(size_t lo, size_t hi, int axis) {
        if (hi - lo <= 1) return;
        size_t mid = (lo + hi) / 2;
        std::nth_element(pts.begin() + lo, pts.begin() + mid, pts.begin() + hi,
                         [axis](const Photon &a, const Photon &b) { return at(a, axis) < at(b, axis); });
        build_range(lo, mid, (axis + 1) % 3);
        build_range(mid + 1, hi, (axis + 1) % 3);
    }
    void build(std::vector<Photon> photons) {
        pts = std::move(photons);
        build_range(0, pts.size(), 0);
    }
    void within(size_t lo, size_t hi, int axis, const float mn[3], const float mx[3], std::vector<const Photon *> &out) const {
        if (lo >= hi) return;
        size_t mid = (lo + hi) / 2;
        const Photon &p = pts[mid];
        if (p.position.x >= mn[0] && p.position.x <= mx[0] && p.position.y >= mn[1] && p.position.y <= mx[1] &&
            p.position.z >= mn[2] && p.position.z <= mx[2])
            out.push_back(&p);
        float c = at(p, axis);
        if (mn[axis] <= c) within(lo, mid, (axis + 1) % 3, mn, mx, out);
        if (mx[axis] >= c) within(mid + 1, hi, (axis + 1) % 3, mn, mx, out);
    }
};

struct Stats {
    std::atomic<uint64_t> rays_primary{0}, rays_secondary{0}, rays_shadow{0}, photons_emitted{0},
        photons_gathered{0}, gather_queries{0}, paths_truncated{0}, vueons_truncated{0};
    double ms_total = 0, ms_photon_shoot = 0, ms_photon_build = 0;
};

// ------------------------------------------------------------------------------------------
// environments/{environment,scene,photon_scene}.rs
struct Env {
    int mode;
    rrt_params params;
    std::vector<std::unique_ptr<Image>> textures;
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<std::unique_ptr<Object>> detached;
    std::vector<bool> detached_is_csg;
    std::vector<std::unique_ptr<Object>> objects;
    std::vector<Light> lights;
    rrt_builder builder;
    // photon state
    uint64_t seed = 0;
    std::vector<Photon> local[2];                       // this rank's records (regular, caustic)
    std::vector<std::vector<Photon>> caustic_seeds;     // per light, canonical order
    KdTree tree[2];
    bool maps_built = false;
    mutable Stats stats;

    struct Traced {
        Hit hit;
        uint32_t object_id;
    };

    // environment.rs:34-53 / scene.rs:21-40
    static bool select_first_hit(const HitVec &hits, Hit &out) {
        bool have = false;
        float min_distance = 3.40282347e+38f;
        for (int i = 0; i < hits.len; i++) {
            const Hit &h = hits.hits[i];
            if (h.distance < 0.0f) continue;
            if (!h.entering) continue;
            if (h.distance < min_distance) {
                min_distance = h.distance;
                out = h;
                have = true;
            }
        }
        return have;
    }
    // environment.rs:55-74 / scene.rs:42-61
    bool trace(const Ray &ray, Traced &out) const {
        bool have = false;
        float min_distance = 3.40282347e+38f;
        for (size_t i = 0; i < objects.size(); i++) {
            HitVec hits = objects[i]->intersect(ray);
            Hit h;
            if (!select_first_hit(hits, h)) continue;
            if (h.distance < min_distance) {
                min_distance = h.distance;
                out.hit = h;
                out.object_id = (uint32_t)i;
                have = true;
            }
        }
        return have;
    }
    // scene.rs:65-79
    bool shadowtrace(const Ray &ray, float limit) const {
        stats.rays_shadow.fetch_add(1, std::memory_order_relaxed);
        for (auto &o : objects) {
            HitVec hits = o->intersect(ray);
            Hit h;
            if (!select_first_hit(hits, h)) continue;
            if (h.distance > 0.0000001f && h.distance < limit) return true;
        }
        return false;
    }
    struct Result {
        Colour colour;
        float depth;
        uint32_t object_id;
    };
    // scene.rs:83-128
    Result raytrace(const Ray &ray, int depth) const {
        (depth == 0 ? stats.rays_primary : stats.rays_secondary).fetch_add(1, std::memory_order_relaxed);
        Traced tr;
        if (!trace(ray, tr)) return Result{black(), 0.0f, 0xFFFFFFFFu};
        const Hit &hit = tr.hit;
        Colour colour = hit.material->compute_once(*this, ray, hit, depth);
        for (const Light &light : lights) {
            V3 viewer = -normalised(hit.position);
            V3 ldir;
            bool lit = light.get_direction(hit.position, ldir);
            if (lit && dot(ldir, hit.normal) > 0.0f) lit = false;
            if (lit) {
                Ray shadow{hit.position, -ldir};
                shadow.position = shadow.position + shadow.direction * 0.0001f;
                if (shadowtrace(shadow, length(ldir))) lit = false;
            }
            if (lit) {
                Colour intensity = light.get_intensity(hit.position);
                colour = colour + hit.material->compute_per_light(viewer, hit, ldir) * intensity;
            }
        }
        return Result{colour, hit.distance, tr.object_id};
    }

    // ---- photon pass ----------------------------------------------------------------------
    // photon_scene.rs:202-228
    void shadowphotontrace(const Photon &absorbed, uint64_t key_base, std::vector<Photon> &out) const {
        Ray ray{absorbed.position + absorbed.incident * 0.0001f, absorbed.incident};
        uint32_t sub = 1;
        for (auto &o : objects) {
            HitVec hits = o->intersect(ray);
            for (int i = 0; i < hits.len; i++) {
                const Hit &h = hits.hits[i];
                if (!h.entering || h.distance < 0.0f) continue;
                if (sub >= 1024) throw std::runtime_error("more than 1023 shadow photons behind one landing");
                out.push_back(Photon{h.position, ray.direction, black(), PT_SHADOW, key_base + sub});
                sub++;
            }
        }
    }
    // photon_scene.rs:150-307 as a loop (every branch of the recursion is a tail call)
    void photontrace(V3 origin, V3 direction, Colour intensity, Rng &rng, uint64_t index,
                     std::vector<Photon> &out) const {
        for (uint32_t bounce = 0; bounce < params.max_photon_bounces; bounce++) {
            Ray ray{origin, direction};
            Traced tr;
            if (!trace(ray, tr)) return;
            const Hit &hit = tr.hit;
            const Material *material = hit.material;
            // choose_weighted, photon_scene.rs:159-167
            float w[4];
            for (int b = 0; b < 4; b++) w[b] = material->behaviour_weight(b);
            float c1 = w[0], c2 = c1 + w[1], c3 = c2 + w[2], total = c3 + w[3];
            float x = rng.u01() * total;
            int choice = x < c1 ? 0 : x < c2 ? 1 : x < c3 ? 2 : 3;
            // absorb_photon, photon_scene.rs:188-200
            uint64_t key_base = index * 65536 + (uint64_t)bounce * 1024;  // index | landing | sub-index, as the product keys its records
            Photon absorbed{hit.position, direction, intensity, PT_COLOUR, key_base};
            size_t slot = out.size();
            out.push_back(absorbed);
            shadowphotontrace(absorbed, key_base, out);
            if (choice == B_ABSORB) return;
            if (choice == B_DIFFUSE) {  // photon_scene.rs:230-251
                V3 d = rng.unit_vector();
                if (dot(d, hit.normal) < 0.0f) d = -d;
                Colour bounced;
                if (!material->bounced_photon(absorbed, hit, bounced)) throw std::runtime_error("bounced_photon is None");
                origin = hit.position;
                direction = normalised(d);
                intensity = bounced;
            } else if (choice == B_SPECULAR) {  // photon_scene.rs:253-269
                V3 refl = normalised(reflection(hit.normal, absorbed.incident));
                Colour bounced;
                if (!material->bounced_photon(absorbed, hit, bounced)) throw std::runtime_error("bounced_photon is None");
                origin = hit.position;
                direction = refl;
                intensity = bounced;
            } else {  // photon_scene.rs:175-178, 271-307
                out[slot].type = PT_CAUSTIC;
                V3 refl = normalised(reflection(hit.normal, absorbed.incident));
                Ray rr;
                float kr;
                bool refract = false;
                if (material->refracted_direction(hit, ray.direction, rr, kr)) {
                    float chance = material->refract_chance(kr);
                    refract = rng.u01() < chance;  // gen_bool
                }
                if (refract) {
                    origin = rr.position;
                    direction = rr.direction;
                } else {
                    origin = hit.position + refl * 0.0001f;
                    direction = refl;
                }
            }
        }
        // the reference recurses on (photon_scene.rs:171-179); the cap of rrt_params.max_photon_bounces cut this path short
        if (params.max_photon_bounces) stats.paths_truncated.fetch_add(1, std::memory_order_relaxed);
    }
    // point_light.rs:47-148, directional_point_light.rs:58-142: one emitted photon of (light, pass)
    void shoot_one(int li, int pass, uint64_t index, std::vector<Photon> &out) const {
        const Light &light = lights[li];
        Rng rng(seed, (uint32_t)li, (uint32_t)pass, index);
        V3 direction;
        if (pass == 0) {
            if (light.kind == RRT_LIGHT_POINT) {
                for (;;) {
                    float x = rng.sym(), y = rng.sym(), z = rng.sym();
                    direction = V3{x, y, z};
                    if (len_sqrd(direction) <= 1.0f) break;
                }
            } else {  // Vector::random_on_surface, vector.rs:44-52
                direction = rng.unit_vector();
                if (dot(direction, light.direction) < 0.0f) direction = -direction;
            }
        } else {
            const std::vector<Photon> &seeds = caustic_seeds[li];
            uint64_t pick = ((uint64_t)rng.next_u32() * (uint64_t)seeds.size()) >> 32;  // SliceRandom::choose
            const Photon &cp = seeds[pick];
            V3 offset = rng.unit_vector() * 0.1f;
            if (light.kind == RRT_LIGHT_POINT)
                direction = cp.incident + offset;
            else
                direction = (cp.position - light.position) + offset;
        }
        photontrace(light.position, normalised(direction), light.intensity, rng, index, out);
    }

    // photon_scene.rs:380-420
    bool average_photon_of_type_at(const Hit &hit, int which, Photon &out, float &count) const {
        float r = params.gather_half_width;
        V3 lo = hit.position - V3{r, r, r}, hi = hit.position + V3{r, r, r};  // photon_tree.rs:16-18
        float mn[3] = {lo.x, lo.y, lo.z}, mx[3] = {hi.x, hi.y, hi.z};
        std::vector<const Photon *> found;
        tree[which].within(0, tree[which].pts.size(), 0, mn, mx, found);
        stats.gather_queries.fetch_add(1, std::memory_order_relaxed);
        if (found.empty()) return false;
        stats.photons_gathered.fetch_add(found.size(), std::memory_order_relaxed);
        V3 p = hit.position;
        std::sort(found.begin(), found.end(), [p](const Photon *a, const Photon *b) {  // photon_tree.rs:33
            float da = len_sqrd(a->position - p), db = len_sqrd(b->position - p);
            return da != db ? da < db : a->key < b->key;  // the reference's sort is unstable on ties: by key, to be deterministic
        });
        if (params.gather_k && found.size() > params.gather_k) found.resize(params.gather_k);  // get_n_within_radius: truncate(n), photon_tree.rs:41-50
        float n = (float)found.size();
        V3 ldir{0, 0, 0};
        Colour inten = black();
        for (const Photon *ph : found) {
            ldir = ldir + normalised(ph->incident);
            inten = inten + ph->intensity;
        }
        ldir = normalised(ldir);
        inten = inten / n;
        out = Photon{hit.position, ldir, inten, PT_COLOUR, 0};
        count = n;
        return true;
    }
    // photon_scene.rs:363-378
    bool average_photon_at(const Hit &hit, Photon &out) const {
        Photon reg, cau;
        float nr = 0, nc = 0;
        bool have_r = average_photon_of_type_at(hit, 0, reg, nr);
        bool have_c = average_photon_of_type_at(hit, 1, cau, nc);
        if (!have_c) {
            if (have_r) out = reg;
            return have_r;
        }
        if (!have_r) {
            out = cau;
            return true;
        }
        reg.intensity = reg.intensity + cau.intensity * (nc / (nc + nr));
        out = reg;
        return true;
    }
    // photon_scene.rs:309-361
    Result vueontrace(V3 origin, V3 direction, uint32_t depth) const {
        (depth == 0 ? stats.rays_primary : stats.rays_secondary).fetch_add(1, std::memory_order_relaxed);
        Ray ray{origin, direction};
        Traced tr;
        if (!trace(ray, tr)) return Result{black(), 0.0f, 0xFFFFFFFFu};
        const Hit &hit = tr.hit;
        const Material *material = hit.material;
        float surface_weight = material->behaviour_weight(B_ABSORB) + material->behaviour_weight(B_DIFFUSE) +
                               material->behaviour_weight(B_SPECULAR);
        Colour surface = black();
        if (surface_weight > 0.0f) {
            Photon avg;
            if (average_photon_at(hit, avg)) surface = material->render_vueon(hit, avg, -direction);
        }
        float reflect_weight = material->behaviour_weight(B_REFLECT_OR_REFRACT);
        Colour reflect = black();
        if (reflect_weight > 0.0f) {
            V3 ro = hit.position + hit.normal * 0.0001f;
            V3 rd = normalised(reflection(hit.normal, direction));
            if (depth + 1 > params.max_vueon_depth) stats.vueons_truncated.fetch_add(1, std::memory_order_relaxed);  // photon_scene.rs:329-339 has no cap
            Colour c = (depth + 1 > params.max_vueon_depth) ? black() : vueontrace(ro, rd, depth + 1).colour;
            reflect = c * reflect_weight;
        }
        float refract_weight = reflect_weight;
        Colour refract = black();
        Ray rr;
        float kr;
        if (material->refracted_direction(hit, ray.direction, rr, kr)) {
            V3 ro = rr.position + rr.direction * 0.0001f;
            if (depth + 1 > params.max_vueon_depth) stats.vueons_truncated.fetch_add(1, std::memory_order_relaxed);
            Colour c = (depth + 1 > params.max_vueon_depth) ? black() : vueontrace(ro, rr.direction, depth + 1).colour;
            refract = c * refract_weight;
        }
        Colour colour = surface + reflect + refract;
        colour = colour / (surface_weight + reflect_weight + refract_weight);
        return Result{colour, hit.distance, tr.object_id};
    }
    // Environment::raytrace: scene.rs:142-144 / photon_scene.rs:437-446
    Result raytrace_primary(const Ray &ray) const {
        if (mode == RRT_MODE_PHOTON) return vueontrace(ray.position, ray.direction, 0);
        return raytrace(ray, 0);
    }
};

// global_material.rs:71-108
Colour GlobalMaterial::compute_once(const Env &scene, const Ray &viewer, const Hit &hit, int depth) const {
    if (depth >= (int)scene.params.max_depth) return black();
    bool have_refl = false, have_refr = false;
    Colour refl = black(), refr = black();
    if (reflect_weight > 0.0f) {
        V3 rd = normalised(reflection(hit.normal, viewer.direction));
        V3 ro = hit.position + rd * 0.0001f;
        refl = scene.raytrace(Ray{ro, rd}, depth + 1).colour * reflect_weight;
        have_refl = true;
    }
    float kr = 0.0f;
    if (refract_weight > 0.0f) {
        Ray rr;
        float k;
        if (refraction(hit, viewer.direction, rr, k)) {
            kr = k;
            refr = scene.raytrace(rr, depth + 1).colour * refract_weight;
            have_refr = true;
        }
    }
    if (have_refl && have_refr) return (refl * kr) + (refr * (1.0f - kr));
    if (have_refl) return refl;
    if (have_refr) return refr;
    return black();
}

// ------------------------------------------------------------------------------------------
// builder (scene_file.rs constructors + Environment::add_*)
static int b_set_mode(void *ctx, int mode) {
    ((Env *)ctx)->mode = mode;
    return RRT_OK;
}
static int b_add_texture(void *ctx, uint32_t w, uint32_t h, uint32_t maxval, const uint8_t *samples) {
    Env *e = (Env *)ctx;
    if (!samples || w == 0 || h == 0 || maxval == 0 || maxval > 65535) return fail(RRT_ERR_INVALID, "bad texture");
    auto img = std::make_unique<Image>();
    img->width = w;
    img->height = h;
    img->pixels.resize((size_t)w * h);
    int bps = maxval <= 255 ? 1 : 2;  // texture.rs:71-96
    for (size_t i = 0; i < (size_t)w * h; i++) {
        uint16_t c[3];
        for (int k = 0; k < 3; k++) {
            const uint8_t *p = samples + (i * 3 + k) * bps;
            c[k] = bps == 1 ? p[0] : (uint16_t)((p[0] << 8) | p[1]);
        }
        img->pixels[i] = Colour{(float)c[0] / (float)maxval, (float)c[1] / (float)maxval, (float)c[2] / (float)maxval};
    }
    e->textures.push_back(std::move(img));
    return (int)e->textures.size() - 1;
}
static int b_add_material(void *ctx, const rrt_material_desc *d) {
    Env *e = (Env *)ctx;
    if (!d) return fail(RRT_ERR_INVALID, "null material");
    std::unique_ptr<Material> m;
    switch (d->kind) {
        case RRT_MAT_MONOCHROME: {
            auto p = std::make_unique<Monochrome>();
            p->colour = Colour{d->colour[0], d->colour[1], d->colour[2]};
            p->ambient_strength = d->ambient;
            p->shininess = d->shininess;
            m = std::move(p);
            break;
        }
        case RRT_MAT_TEXTURE: {
            auto p = std::make_unique<Texture>();
            if (d->diffuse_tex < 0 || d->diffuse_tex >= (int)e->textures.size()) return fail(RRT_ERR_INVALID, "bad diffuse texture id");
            if (d->normal_tex >= (int)e->textures.size()) return fail(RRT_ERR_INVALID, "bad normal texture id");
            p->diffuse_img = e->textures[d->diffuse_tex].get();
            p->normal_img = d->normal_tex >= 0 ? e->textures[d->normal_tex].get() : nullptr;
            p->scale = d->scale;
            p->ambient_strength = d->ambient;
            p->shininess = d->shininess;
            m = std::move(p);
            break;
        }
        case RRT_MAT_GLOBAL: {
            auto p = std::make_unique<GlobalMaterial>();
            p->reflect_weight = d->reflect_weight;
            p->refract_weight = d->refract_weight;
            p->ior = d->ior;
            m = std::move(p);
            break;
        }
        case RRT_MAT_FALSECOLOUR: m = std::make_unique<FalseColour>(); break;
        case RRT_MAT_COMPOUND: {
            auto p = std::make_unique<CompoundMaterial>();
            if (d->n_parts < 1 || d->n_parts > RRT_MAX_COMPOUND_PARTS) return fail(RRT_ERR_INVALID, "bad compound part count");
            for (int i = 0; i < d->n_parts; i++) {
                if (d->parts[i] < 0 || d->parts[i] >= (int)e->materials.size()) return fail(RRT_ERR_INVALID, "bad compound part id");
                p->parts.push_back(e->materials[d->parts[i]].get());
            }
            m = std::move(p);
            break;
        }
        default: return fail(RRT_ERR_INVALID, "bad material kind");
    }
    e->materials.push_back(std::move(m));
    return (int)e->materials.size() - 1;
}
static int b_object_define(void *ctx, const rrt_object_desc *d) {
    Env *e = (Env *)ctx;
    if (!d) return fail(RRT_ERR_INVALID, "null object");
    const Material *mat = nullptr;
    if (d->kind != RRT_OBJ_CSG) {
        if (d->material < 0 || d->material >= (int)e->materials.size()) return fail(RRT_ERR_INVALID, "bad material id");
        mat = e->materials[d->material].get();
    }
    std::unique_ptr<Object> o;
    bool is_csg = false;
    const float *v = d->v;
    switch (d->kind) {
        case RRT_OBJ_PLANE: o = std::make_unique<Plane>(V3{v[0], v[1], v[2]}, V3{v[3], v[4], v[5]}, V3{v[6], v[7], v[8]}, mat); break;
        case RRT_OBJ_SPHERE: {
            auto p = std::make_unique<Sphere>();
            p->centre = V3{v[0], v[1], v[2]};
            p->radius = v[3];
            p->material = mat;
            o = std::move(p);
            break;
        }
        case RRT_OBJ_CUBOID: {
            auto p = std::make_unique<Cuboid>();
            p->corner = V3{v[0], v[1], v[2]};
            p->size = V3{v[3], v[4], v[5]};
            p->material = mat;
            p->build_planes();
            o = std::move(p);
            break;
        }
        case RRT_OBJ_QUADRATIC: {
            auto p = std::make_unique<Quadratic>();
            for (int i = 0; i < 10; i++) p->q[i] = v[i];
            p->material = mat;
            o = std::move(p);
            break;
        }
        case RRT_OBJ_MESH: {
            auto p = std::make_unique<PolyMesh>();
            p->smooth = d->smooth != 0;
            p->material = mat;
            if (!d->positions) return fail(RRT_ERR_INVALID, "mesh without positions");
            p->triangles.resize(d->n_triangles);
            for (uint64_t t = 0; t < d->n_triangles; t++) {
                Triangle &tri = p->triangles[t];
                const float *q = d->positions + t * 9;
                tri.a = V3{q[0], q[1], q[2]};
                tri.b = V3{q[3], q[4], q[5]};
                tri.c = V3{q[6], q[7], q[8]};
                tri.has_normals = d->normals != nullptr;
                if (d->normals) {
                    const float *n = d->normals + t * 9;
                    tri.an = V3{n[0], n[1], n[2]};
                    tri.bn = V3{n[3], n[4], n[5]};
                    tri.cn = V3{n[6], n[7], n[8]};
                }
                tri.rebuild(mat);
            }
            o = std::move(p);
            break;
        }
        case RRT_OBJ_CSG: {
            auto p = std::make_unique<Csg>();
            if (d->csg_mode < 0 || d->csg_mode > 2) return fail(RRT_ERR_INVALID, "bad csg mode");
            int n = (int)e->detached.size();
            if (d->left < 0 || d->left >= n || d->right < 0 || d->right >= n || d->left == d->right || !e->detached[d->left] ||
                !e->detached[d->right])
                return fail(RRT_ERR_INVALID, "bad csg child handle");
            p->mode = d->csg_mode;
            p->left = std::move(e->detached[d->left]);
            p->right = std::move(e->detached[d->right]);
            p->left_is_csg = e->detached_is_csg[d->left];
            p->right_is_csg = e->detached_is_csg[d->right];
            is_csg = true;
            o = std::move(p);
            break;
        }
        default: return fail(RRT_ERR_INVALID, "bad object kind");
    }
    e->detached.push_back(std::move(o));
    e->detached_is_csg.push_back(is_csg);
    return (int)e->detached.size() - 1;
}
static int b_object_transform(void *ctx, int handle, const float *m16) {
    Env *e = (Env *)ctx;
    if (handle < 0 || handle >= (int)e->detached.size() || !e->detached[handle] || !m16) return fail(RRT_ERR_INVALID, "bad handle");
    Mat4 t;
    memcpy(t.m, m16, sizeof(t.m));
    try {
        e->detached[handle]->apply_transform(t);
    } catch (std::exception &ex) {
        return fail(RRT_ERR_INVALID, ex.what());
    }
    return RRT_OK;
}
static int b_add_object(void *ctx, int handle) {
    Env *e = (Env *)ctx;
    if (handle < 0 || handle >= (int)e->detached.size() || !e->detached[handle]) return fail(RRT_ERR_INVALID, "bad handle");
    uint32_t next = 0;
    e->detached[handle]->number_leaves(next);
    if (!e->detached_is_csg[handle]) e->detached[handle]->leaf_index = 0;
    e->objects.push_back(std::move(e->detached[handle]));
    return (int)e->objects.size() - 1;
}
static int b_add_light(void *ctx, const rrt_light_desc *d) {
    Env *e = (Env *)ctx;
    if (!d || d->kind < 0 || d->kind > 2) return fail(RRT_ERR_INVALID, "bad light");
    Light l;
    l.kind = d->kind;
    l.position = V3{d->position[0], d->position[1], d->position[2]};
    l.direction = V3{d->direction[0], d->direction[1], d->direction[2]};
    l.intensity = Colour{d->colour[0], d->colour[1], d->colour[2]};
    e->lights.push_back(l);
    return (int)e->lights.size() - 1;
}

static void default_params(rrt_params *p) {
    memset(p, 0, sizeof(*p));
    p->photons_per_light = 5000000;
    p->caustic_photons_per_light = 10000;
    p->gather_half_width = 0.1f;
    p->max_depth = 5;
    p->max_photon_bounces = 64;
    p->max_vueon_depth = 12;
}

template <class F>
static void parallel_ranges(uint64_t first, uint64_t count, int threads, F fn) {
    // lights/light.rs:28-40 and cameras/camera.rs:14-31: n / threads each, remainder on the last
    if (threads < 1) threads = 1;
    uint64_t per = count / threads, extra = count % threads;
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) {
        uint64_t lo = first + per * t, hi = lo + per + (t == threads - 1 ? extra : 0);
        pool.emplace_back([=] { fn(t, lo, hi); });
    }
    for (auto &th : pool) th.join();
}

static uint64_t photon_meta(int light, int pass, uint64_t key) { return ((uint64_t)(light * 2 + pass) << 56) | key; }

}  // namespace orc

using namespace orc;

struct orc_scene {
    Env env;
};

extern "C" {

const char *orc_last_error(void) { return g_err.c_str(); }

orc_scene *orc_scene_create(int mode, const rrt_params *params) {
    auto *s = new orc_scene();
    s->env.mode = mode;
    if (params)
        s->env.params = *params;
    else
        default_params(&s->env.params);
    s->env.builder = rrt_builder{&s->env, b_set_mode, b_add_texture, b_add_material, b_object_define,
                                 b_object_transform, b_add_object, b_add_light};
    return s;
}
void orc_scene_destroy(orc_scene *s) { delete s; }
int orc_scene_get_mode(const orc_scene *s) { return s->env.mode; }
const rrt_builder *orc_scene_builder(orc_scene *s) { return &s->env.builder; }
int orc_object_count(const orc_scene *s) { return (int)s->env.objects.size(); }

// full_camera.rs:23-58
int orc_camera_init(rrt_camera *cam, uint32_t width, uint32_t height, float fov, const float position[3],
                    const float lookat_in[3], const float up_in[3]) {
    V3 lookat{lookat_in[0], lookat_in[1], lookat_in[2]}, up{up_in[0], up_in[1], up_in[2]};
    lookat.y = -lookat.y;
    up.y = -up.y;
    lookat = normalised(lookat);
    up = normalised(up);
    float ang = angle(lookat, up);
    const float right_angle = 1.57079632679489661923f;
    if (fabsf(ang - right_angle) > 0.0001f) return fail(RRT_ERR_INVALID, "FullCamera right and up are not perpendicular");
    V3 right = normalised(cross(lookat, up));
    cam->width = width;
    cam->height = height;
    cam->fov = fov;
    for (int i = 0; i < 3; i++) cam->position[i] = position[i];
    cam->lookat[0] = lookat.x; cam->lookat[1] = lookat.y; cam->lookat[2] = lookat.z;
    cam->up[0] = up.x; cam->up[1] = up.y; cam->up[2] = up.z;
    cam->right[0] = right.x; cam->right[1] = right.y; cam->right[2] = right.z;
    return RRT_OK;
}
// full_camera.rs:61-81
static Ray camera_ray(const rrt_camera *cam, uint32_t x, uint32_t y) {
    float fx = ((float)x + 0.5f) / (float)cam->width;
    float fy = ((float)y + 0.5f) / (float)cam->height;
    V3 d{fx - 0.5f, fy - 0.5f, cam->fov};
    Mat4 rot = mat4_identity();
    for (int j = 0; j < 3; j++) {
        rot.m[0][j] = cam->right[j];
        rot.m[1][j] = cam->up[j];
        rot.m[2][j] = cam->lookat[j];
    }
    rot = transposed(rot);
    d = normalised(apply_to_vector(rot, d));
    return Ray{V3{cam->position[0], cam->position[1], cam->position[2]}, d};
}
int orc_camera_ray(const rrt_camera *cam, uint32_t x, uint32_t y, float origin[3], float direction[3]) {
    if (x >= cam->width || y >= cam->height) return fail(RRT_ERR_INVALID, "pixel out of range");
    Ray r = camera_ray(cam, x, y);
    origin[0] = r.position.x; origin[1] = r.position.y; origin[2] = r.position.z;
    direction[0] = r.direction.x; direction[1] = r.direction.y; direction[2] = r.direction.z;
    return RRT_OK;
}

int orc_intersect(orc_scene *s, int object_id, const float o[3], const float d[3], int *n_hits, float distance[6],
                  int32_t entering[6], float position[18], float normal[18], float uv[12], uint32_t prim[6]) {
    Env &e = s->env;
    if (object_id < 0 || object_id >= (int)e.objects.size()) return fail(RRT_ERR_INVALID, "bad object id");
    try {
        HitVec hv = e.objects[object_id]->intersect(Ray{V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}});
        *n_hits = hv.len;
        for (int i = 0; i < hv.len; i++) {
            const Hit &h = hv.hits[i];
            if (distance) distance[i] = h.distance;
            if (entering) entering[i] = h.entering;
            if (position) { position[i * 3] = h.position.x; position[i * 3 + 1] = h.position.y; position[i * 3 + 2] = h.position.z; }
            if (normal) { normal[i * 3] = h.normal.x; normal[i * 3 + 1] = h.normal.y; normal[i * 3 + 2] = h.normal.z; }
            if (uv) { uv[i * 2] = h.has_tex ? h.u : NAN; uv[i * 2 + 1] = h.has_tex ? h.v : NAN; }
            if (prim) prim[i] = h.prim;
        }
    } catch (HitVecOverflow &) {
        return fail(RRT_ERR_OVERFLOW, "HitVec overflow");
    } catch (std::exception &ex) {
        return fail(RRT_ERR_INVALID, ex.what());
    }
    return RRT_OK;
}

int orc_photon_begin(orc_scene *s, uint64_t seed) {
    Env &e = s->env;
    e.seed = seed;
    e.local[0].clear();
    e.local[1].clear();
    e.caustic_seeds.assign(e.lights.size(), {});
    e.maps_built = false;
    e.stats.photons_emitted = 0;
    e.stats.paths_truncated = 0;
    for (auto &l : e.lights)
        if (l.kind == RRT_LIGHT_DIRECTIONAL) return fail(RRT_ERR_INVALID, "Light does not support photon mapping");  // light.rs:16-18
    return RRT_OK;
}

// lights/light.rs:22-73 + photon_scene.rs:45-65 for one (light, pass) and one index range
int orc_photon_shoot(orc_scene *s, int light, int pass, uint64_t first, uint64_t count, int threads) {
    Env &e = s->env;
    if (light < 0 || light >= (int)e.lights.size() || pass < 0 || pass > 1) return fail(RRT_ERR_INVALID, "bad light/pass");
    if (pass == 1 && e.caustic_seeds[light].empty()) {
        // point_light.rs:106-108 returns nothing; directional_point_light.rs:110 would panic
        return RRT_OK;
    }
    if (threads < 1) threads = 1;
    std::vector<std::vector<Photon>> per_thread(threads);
    std::string err;
    std::atomic<int> status{RRT_OK};
    parallel_ranges(first, count, threads, [&](int t, uint64_t lo, uint64_t hi) {
        try {
            for (uint64_t i = lo; i < hi; i++) e.shoot_one(light, pass, i, per_thread[t]);
        } catch (HitVecOverflow &) {
            status = RRT_ERR_OVERFLOW;
        } catch (std::exception &) {
            status = RRT_ERR_LIMIT;
        }
    });
    if (status != RRT_OK) return fail(status, "photon pass failed");
    e.stats.photons_emitted += count;
    for (auto &v : per_thread)
        for (Photon &p : v) {
            bool to_caustic = pass == 1 || p.type == PT_CAUSTIC;  // photon_scene.rs:48-65
            if (pass == 0 && p.type == PT_CAUSTIC) e.caustic_seeds[light].push_back(p);  // key without the (light, pass) tag
            p.key = photon_meta(light, pass, p.key);
            e.local[to_caustic ? 1 : 0].push_back(p);
        }
    if (pass == 0)
        std::sort(e.caustic_seeds[light].begin(), e.caustic_seeds[light].end(),
                  [](const Photon &a, const Photon &b) { return a.key < b.key; });
    return RRT_OK;
}

int orc_photon_build(orc_scene *s) {
    Env &e = s->env;
    for (int w = 0; w < 2; w++) {
        std::vector<Photon> v = e.local[w];
        std::sort(v.begin(), v.end(), [](const Photon &a, const Photon &b) { return a.key < b.key; });
        e.local[w] = v;
        e.tree[w].build(std::move(v));
    }
    e.maps_built = true;
    return RRT_OK;
}

// photon_scene.rs:42-83 with one rank
int orc_pre_render(orc_scene *s, uint64_t seed, int threads) {
    Env &e = s->env;
    if (e.mode != RRT_MODE_PHOTON) return RRT_OK;  // scene.rs:140
    int rc = orc_photon_begin(s, seed);
    if (rc) return rc;
    for (int l = 0; l < (int)e.lights.size(); l++) {
        rc = orc_photon_shoot(s, l, 0, 0, e.params.photons_per_light, threads);
        if (rc) return rc;
        rc = orc_photon_shoot(s, l, 1, 0, e.params.caustic_photons_per_light, threads);
        if (rc) return rc;
    }
    return orc_photon_build(s);
}

int orc_photon_count(orc_scene *s, int which, uint64_t *count) {
    if (which < 0 || which > 1) return fail(RRT_ERR_INVALID, "bad map");
    *count = s->env.local[which].size();
    return RRT_OK;
}
int orc_photon_read(orc_scene *s, int which, float *position, float *incident, float *intensity, uint8_t *type, uint64_t *key) {
    if (which < 0 || which > 1) return fail(RRT_ERR_INVALID, "bad map");
    const auto &v = s->env.local[which];
    for (size_t i = 0; i < v.size(); i++) {
        const Photon &p = v[i];
        if (position) { position[i * 3] = p.position.x; position[i * 3 + 1] = p.position.y; position[i * 3 + 2] = p.position.z; }
        if (incident) { incident[i * 3] = p.incident.x; incident[i * 3 + 1] = p.incident.y; incident[i * 3 + 2] = p.incident.z; }
        if (intensity) { intensity[i * 3] = p.intensity.r; intensity[i * 3 + 1] = p.intensity.g; intensity[i * 3 + 2] = p.intensity.b; }
        if (type) type[i] = p.type;
        if (key) key[i] = p.key;
    }
    return RRT_OK;
}
// exchange points for multi-rank tests (host arrays; the product's versions take device pointers)
int orc_photon_caustic_count(orc_scene *s, int light, uint64_t *count) {
    if (light < 0 || light >= (int)s->env.caustic_seeds.size()) return fail(RRT_ERR_INVALID, "bad light");
    *count = s->env.caustic_seeds[light].size();
    return RRT_OK;
}
int orc_photon_caustic_export(orc_scene *s, int light, float *pos4, float *incident4, uint64_t *key) {
    if (light < 0 || light >= (int)s->env.caustic_seeds.size()) return fail(RRT_ERR_INVALID, "bad light");
    const auto &v = s->env.caustic_seeds[light];
    for (size_t i = 0; i < v.size(); i++) {
        pos4[i * 4] = v[i].position.x; pos4[i * 4 + 1] = v[i].position.y; pos4[i * 4 + 2] = v[i].position.z; pos4[i * 4 + 3] = 0.0f;
        incident4[i * 4] = v[i].incident.x; incident4[i * 4 + 1] = v[i].incident.y; incident4[i * 4 + 2] = v[i].incident.z; incident4[i * 4 + 3] = 0.0f;
        key[i] = v[i].key;
    }
    return RRT_OK;
}
int orc_photon_caustic_import(orc_scene *s, int light, const float *pos4, const float *incident4, const uint64_t *key, uint64_t count) {
    if (light < 0 || light >= (int)s->env.caustic_seeds.size()) return fail(RRT_ERR_INVALID, "bad light");
    auto &v = s->env.caustic_seeds[light];
    v.clear();
    for (uint64_t i = 0; i < count; i++)
        v.push_back(Photon{V3{pos4[i * 4], pos4[i * 4 + 1], pos4[i * 4 + 2]}, V3{incident4[i * 4], incident4[i * 4 + 1], incident4[i * 4 + 2]},
                           black(), PT_CAUSTIC, key[i]});
    std::sort(v.begin(), v.end(), [](const Photon &a, const Photon &b) { return a.key < b.key; });
    return RRT_OK;
}
int orc_photon_export(orc_scene *s, int which, float *pos4, float *incident4, float *intensity4, uint64_t *meta) {
    if (which < 0 || which > 1) return fail(RRT_ERR_INVALID, "bad map");
    const auto &v = s->env.local[which];
    for (size_t i = 0; i < v.size(); i++) {
        uint32_t ty = v[i].type;
        pos4[i * 4] = v[i].position.x; pos4[i * 4 + 1] = v[i].position.y; pos4[i * 4 + 2] = v[i].position.z;
        memcpy(&pos4[i * 4 + 3], &ty, 4);
        incident4[i * 4] = v[i].incident.x; incident4[i * 4 + 1] = v[i].incident.y; incident4[i * 4 + 2] = v[i].incident.z; incident4[i * 4 + 3] = 0.0f;
        intensity4[i * 4] = v[i].intensity.r; intensity4[i * 4 + 1] = v[i].intensity.g; intensity4[i * 4 + 2] = v[i].intensity.b; intensity4[i * 4 + 3] = 0.0f;
        meta[i] = v[i].key;
    }
    return RRT_OK;
}
int orc_photon_import(orc_scene *s, int which, const float *pos4, const float *incident4, const float *intensity4, const uint64_t *meta, uint64_t count) {
    if (which < 0 || which > 1) return fail(RRT_ERR_INVALID, "bad map");
    auto &v = s->env.local[which];
    v.clear();
    for (uint64_t i = 0; i < count; i++) {
        uint32_t ty;
        memcpy(&ty, &pos4[i * 4 + 3], 4);
        v.push_back(Photon{V3{pos4[i * 4], pos4[i * 4 + 1], pos4[i * 4 + 2]}, V3{incident4[i * 4], incident4[i * 4 + 1], incident4[i * 4 + 2]},
                           Colour{intensity4[i * 4], intensity4[i * 4 + 1], intensity4[i * 4 + 2]}, (uint8_t)ty, meta[i]});
    }
    s->env.maps_built = false;
    return RRT_OK;
}

int orc_photon_query(orc_scene *s, int which, const float p[3], float r, uint64_t *k, float dir_sum[3], float intensity_sum[3]) {
    if (which < 0 || which > 1) return fail(RRT_ERR_INVALID, "bad map");
    V3 pos{p[0], p[1], p[2]};
    V3 lo = pos - V3{r, r, r}, hi = pos + V3{r, r, r};
    uint64_t n = 0;
    double ds[3] = {0, 0, 0}, is[3] = {0, 0, 0};
    for (const Photon &ph : s->env.local[which]) {
        V3 q = ph.position;
        if (q.x >= lo.x && q.x <= hi.x && q.y >= lo.y && q.y <= hi.y && q.z >= lo.z && q.z <= hi.z) {
            n++;
            V3 d = normalised(ph.incident);
            ds[0] += d.x; ds[1] += d.y; ds[2] += d.z;
            is[0] += ph.intensity.r; is[1] += ph.intensity.g; is[2] += ph.intensity.b;
        }
    }
    *k = n;
    for (int i = 0; i < 3; i++) {
        if (dir_sum) dir_sum[i] = (float)ds[i];
        if (intensity_sum) intensity_sum[i] = (float)is[i];
    }
    return RRT_OK;
}

// core/photon_tree.rs:15-35 (get_within_distance) + :41-50 (get_n_within_radius), by brute force over the map
int orc_photon_query_list(orc_scene *s, int which, const float p[3], float r, uint32_t k_max, uint32_t *count, uint64_t *key, float *dist2) {
    if (which < 0 || which > 1) return fail(RRT_ERR_INVALID, "bad map");
    V3 pos{p[0], p[1], p[2]};
    V3 lo = pos - V3{r, r, r}, hi = pos + V3{r, r, r};
    std::vector<std::pair<float, uint64_t>> found;
    for (const Photon &ph : s->env.local[which]) {
        V3 q = ph.position;
        if (q.x >= lo.x && q.x <= hi.x && q.y >= lo.y && q.y <= hi.y && q.z >= lo.z && q.z <= hi.z)
            found.push_back({len_sqrd(q - pos), ph.key});  // photon_tree.rs:25
    }
    std::sort(found.begin(), found.end());  // by squared distance (photon_tree.rs:33); ties by key to be deterministic
    *count = (uint32_t)found.size();
    for (size_t i = 0; i < found.size() && i < k_max; i++) {  // truncate(n), photon_tree.rs:48
        dist2[i] = found[i].first;
        key[i] = found[i].second;
    }
    return RRT_OK;
}

// cameras/camera.rs:7-43 + full_camera.rs:93-146
int orc_render(orc_scene *s, const rrt_camera *cam, uint32_t y0, uint32_t y1, float *rgb, float *depth, uint32_t *object_id,
               int threads) {
    Env &e = s->env;
    if (y1 > cam->height || y0 > y1) return fail(RRT_ERR_INVALID, "bad row range");
    if (e.mode == RRT_MODE_PHOTON && !e.maps_built) return fail(RRT_ERR_STATE, "Photon map not built");
    std::atomic<int> status{RRT_OK};
    uint32_t W = cam->width;
    parallel_ranges(y0, y1 - y0, threads, [&](int, uint64_t lo, uint64_t hi) {
        try {
            for (uint32_t y = (uint32_t)lo; y < (uint32_t)hi; y++)
                for (uint32_t x = 0; x < W; x++) {
                    Env::Result r = e.raytrace_primary(camera_ray(cam, x, y));
                    size_t i = (size_t)(y - y0) * W + x;
                    if (rgb) { rgb[i * 3] = r.colour.r; rgb[i * 3 + 1] = r.colour.g; rgb[i * 3 + 2] = r.colour.b; }
                    if (depth) depth[i] = r.depth;
                    if (object_id) object_id[i] = r.object_id;
                }
        } catch (HitVecOverflow &) {
            status = RRT_ERR_OVERFLOW;
        } catch (std::exception &) {
            status = RRT_ERR_INVALID;
        }
    });
    if (status != RRT_OK) return fail(status, status == RRT_ERR_OVERFLOW ? "HitVec overflow" : "render failed");
    return RRT_OK;
}

int orc_raytrace(orc_scene *s, const float *o, const float *d, size_t n, float *rgb, float *depth, uint32_t *object_id) {
    Env &e = s->env;
    if (e.mode == RRT_MODE_PHOTON && !e.maps_built) return fail(RRT_ERR_STATE, "Photon map not built");
    try {
        for (size_t i = 0; i < n; i++) {
            Ray ray{V3{o[i * 3], o[i * 3 + 1], o[i * 3 + 2]}, V3{d[i * 3], d[i * 3 + 1], d[i * 3 + 2]}};
            Env::Result r = e.raytrace_primary(ray);
            if (rgb) { rgb[i * 3] = r.colour.r; rgb[i * 3 + 1] = r.colour.g; rgb[i * 3 + 2] = r.colour.b; }
            if (depth) depth[i] = r.depth;
            if (object_id) object_id[i] = r.object_id;
        }
    } catch (HitVecOverflow &) {
        return fail(RRT_ERR_OVERFLOW, "HitVec overflow");
    } catch (std::exception &ex) {
        return fail(RRT_ERR_INVALID, ex.what());
    }
    return RRT_OK;
}

int orc_trace(orc_scene *s, const float *o, const float *d, size_t n, uint32_t *object_id, uint32_t *prim_id, float *distance,
              float *position, float *normal, float *uv) {
    Env &e = s->env;
    try {
        for (size_t i = 0; i < n; i++) {
            Ray ray{V3{o[i * 3], o[i * 3 + 1], o[i * 3 + 2]}, V3{d[i * 3], d[i * 3 + 1], d[i * 3 + 2]}};
            Env::Traced tr;
            bool ok = e.trace(ray, tr);
            const Hit &h = tr.hit;
            if (object_id) object_id[i] = ok ? tr.object_id : 0xFFFFFFFFu;
            if (prim_id) prim_id[i] = ok ? h.prim : 0;
            if (distance) distance[i] = ok ? h.distance : 0.0f;
            if (position) for (int k = 0; k < 3; k++) position[i * 3 + k] = ok ? (&h.position.x)[k] : 0.0f;
            if (normal) for (int k = 0; k < 3; k++) normal[i * 3 + k] = ok ? (&h.normal.x)[k] : 0.0f;
            if (uv) { uv[i * 2] = (ok && h.has_tex) ? h.u : NAN; uv[i * 2 + 1] = (ok && h.has_tex) ? h.v : NAN; }
        }
    } catch (HitVecOverflow &) {
        return fail(RRT_ERR_OVERFLOW, "HitVec overflow");
    } catch (std::exception &ex) {
        return fail(RRT_ERR_INVALID, ex.what());
    }
    return RRT_OK;
}

int orc_shadowtrace(orc_scene *s, const float *o, const float *d, const float *limits, size_t n, uint8_t *occluded) {
    Env &e = s->env;
    try {
        for (size_t i = 0; i < n; i++) {
            Ray ray{V3{o[i * 3], o[i * 3 + 1], o[i * 3 + 2]}, V3{d[i * 3], d[i * 3 + 1], d[i * 3 + 2]}};
            occluded[i] = e.shadowtrace(ray, limits[i]) ? 1 : 0;
        }
    } catch (HitVecOverflow &) {
        return fail(RRT_ERR_OVERFLOW, "HitVec overflow");
    }
    return RRT_OK;
}

int orc_get_stats(const orc_scene *s, rrt_stats *out) {
    memset(out, 0, sizeof(*out));
    const Stats &st = s->env.stats;
    out->rays_primary = st.rays_primary;
    out->rays_secondary = st.rays_secondary;
    out->rays_shadow = st.rays_shadow;
    out->photons_emitted = st.photons_emitted;
    out->photons_stored_regular = s->env.local[0].size();
    out->photons_stored_caustic = s->env.local[1].size();
    out->photons_gathered = st.photons_gathered;
    out->gather_queries = st.gather_queries;
    out->paths_truncated = st.paths_truncated;
    out->vueons_truncated = st.vueons_truncated;
    return RRT_OK;
}

// framebuffer.rs:96-130
void orc_quantise(const float *values, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = f32_as_u8(values[i] * 255.0f);
}
static int write_pnm(const char *path, const char *magic, const float *v, uint32_t w, uint32_t h, int ch) {
    FILE *f = fopen(path, "wb");
    if (!f) return fail(RRT_ERR_IO, std::string("cannot create ") + path);
    fprintf(f, "%s\n%u %u\n255\n", magic, w, h);
    std::vector<uint8_t> row((size_t)w * ch);
    for (uint32_t y = 0; y < h; y++) {
        orc_quantise(v + (size_t)y * w * ch, (size_t)w * ch, row.data());
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return RRT_OK;
}
int orc_write_rgb_ppm(const char *path, const float *rgb, uint32_t w, uint32_t h) { return write_pnm(path, "P6", rgb, w, h, 3); }
int orc_write_depth_pgm(const char *path, const float *depth, uint32_t w, uint32_t h) { return write_pnm(path, "P5", depth, w, h, 1); }

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }
int orc_mat4_inverse(const float *m16, float *out16) {
    Mat4 a, r;
    memcpy(a.m, m16, sizeof(a.m));
    if (!mat4_inverse(a, r)) return fail(RRT_ERR_INVALID, "Matrix is not invertible");
    memcpy(out16, r.m, sizeof(r.m));
    return RRT_OK;
}

}  // extern "C"
