// host_scene.h — host-side scene graph of librrt: what the reference keeps as
// Vec<Box<dyn Object>> / Vec<Box<dyn Light>> / Arc<dyn Material> (src/environments/scene.rs:8-11),
// held as plain records until rrt_pre_render flattens them into SoA arrays in HBM.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/rrt.h"
#include "rrt_math.h"

namespace rrt {

struct HostTexture {
    uint32_t width, height, maxval;
    std::vector<uint8_t> samples;  // RGB, 1 or 2 (big-endian) bytes per sample
};

struct HostObject {
    int kind = 0;
    int material = -1;
    // plane: p0 = point, p1 = up, p2 = normal, d;  sphere: p0 = centre, radius;  cuboid: p0 = corner, p1 = size
    f3 p0{0, 0, 0}, p1{0, 0, 0}, p2{0, 0, 0};
    float d = 0.0f;
    float radius = 0.0f;
    float q[10] = {0};            // quadric a..j
    std::vector<float> tri_pos;   // mesh [n][3][3]
    std::vector<float> tri_nrm;   // mesh [n][3][3] or empty
    bool smooth = false;
    int csg_mode = 0;
    std::unique_ptr<HostObject> left, right;

    void apply_transform(const float m[4][4]);
    size_t leaf_count() const { return kind == RRT_OBJ_CSG ? left->leaf_count() + right->leaf_count() : 1; }
    bool contains_mesh() const {
        return kind == RRT_OBJ_MESH || (kind == RRT_OBJ_CSG && (left->contains_mesh() || right->contains_mesh()));
    }
};

// Transform::inverse (transform.rs:46-177); false when the reference would panic
bool invert4(const float m[4][4], float out[4][4]);

void set_error(const std::string &msg);
int fail(int code, const std::string &msg);

}  // namespace rrt
