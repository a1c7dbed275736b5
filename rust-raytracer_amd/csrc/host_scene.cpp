// host_scene.cpp — Object::apply_transform rules per object type, host side.
#include "host_scene.h"

#include <string.h>

namespace rrt {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
int fail(int code, const std::string &msg) {
    g_error = msg;
    return code;
}
const char *last_error_cstr() { return g_error.c_str(); }

// Transform::inverse (transform.rs:46-177).  The adjugate's cofactor expansion as data: entry
// [R][C] is six signed triple products; "s rc rc rc" per term, evaluated left to right and
// accumulated in the order listed, which is the order the reference writes them in.
static const char *const kAdjugate[4][4] = {
    {"+112233-112332-211233+211332+311223-311322", "-012233+012332+210233-210332-310223+310322",
     "+011233-011332-110233+110332+310213-310312", "-011223+011322+110223-110322-210213+210312"},
    {"-102233+102332+201233-201332-301223+301322", "+002233-002332-200233+200332+300223-300322",
     "-001233+001332+100233-100332-300213+300312", "+001223-001322-100223+100322+200213-200312"},
    {"+102133-102331-201133+201331+301123-301321", "-002133+002331+200133-200331-300123+300321",
     "+001133-001331-100133+100331+300113-300311", "-001123+001321+100123-100321-200113+200311"},
    {"-102132+102231+201132-201231-301122+301221", "+002132-002231-200132+200231+300122-300221",
     "-001132+001231+100132-100231-300112+300211", "+001122-001221-100122+100221+200112-200211"}};

bool invert4(const float m[4][4], float out[4][4]) {
    float adj[4][4];
    for (int R = 0; R < 4; R++)
        for (int C = 0; C < 4; C++) {
            const char *s = kAdjugate[R][C];
            float acc = 0.0f;
            for (int k = 0; k < 6; k++, s += 7) {
                float t = m[s[1] - '0'][s[2] - '0'] * m[s[3] - '0'][s[4] - '0'] * m[s[5] - '0'][s[6] - '0'];
                bool neg = s[0] == '-';
                if (k == 0)
                    acc = neg ? -t : t;  // `-a*b*c` == -(a*b*c) bit for bit
                else
                    acc = neg ? acc - t : acc + t;
            }
            adj[R][C] = acc;
        }
    float det = m[0][0] * adj[0][0] + m[0][1] * adj[1][0] + m[0][2] * adj[2][0] + m[0][3] * adj[3][0];
    if (det == 0.0f) return false;
    det = 1.0f / det;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[i][j] = adj[i][j] * det;
    return true;
}

static f3 xform_vertex(const float m[4][4], f3 v) {  // vertex.rs:35-48
    return f3{m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z + m[0][3], m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z + m[1][3],
              m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z + m[2][3]};
}
static f3 xform_vector(const float m[4][4], f3 v) {  // vector.rs:121-131
    return f3{m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
              m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
}
static void matmul4(const float a[4][4], const float b[4][4], float r[4][4]) {  // transform.rs:192-209
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j] + a[i][3] * b[3][j];
}

void HostObject::apply_transform(const float m[4][4]) {
    switch (kind) {
        case RRT_OBJ_PLANE:  // plane_object.rs:115-124
            p0 = xform_vertex(m, p0);
            p1 = xform_vector(m, p1);
            p2 = xform_vector(m, p2);
            p2 = norm3(p2);
            p1 = norm3(p1);
            d = -dot3(p2, p0);
            break;
        case RRT_OBJ_SPHERE:  // sphere_object.rs:93-100: radius only follows uniform scaling
            p0 = xform_vertex(m, p0);
            if (m[0][0] == m[1][1] && m[1][1] == m[2][2]) radius *= m[0][0];
            break;
        case RRT_OBJ_CUBOID:  // cuboid_object.rs:150-160: size scales by the diagonal only
            p0 = xform_vertex(m, p0);
            p1 = f3{p1.x * m[0][0], p1.y * m[1][1], p1.z * m[2][2]};
            break;
        case RRT_OBJ_QUADRATIC: {  // quadratic_object.rs:127-148: Q' = T^-T Q T^-1
            float Q[4][4] = {{q[0], q[1], q[2], q[3]}, {q[1], q[4], q[5], q[6]}, {q[2], q[5], q[7], q[8]}, {q[3], q[6], q[8], q[9]}};
            float Ti[4][4], Tt[4][4], A[4][4], B[4][4];
            if (!invert4(m, Ti)) {
                set_error("Matrix is not invertible");
                throw 1;
            }
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 4; j++) Tt[i][j] = Ti[j][i];
            matmul4(Tt, Q, A);
            matmul4(A, Ti, B);
            q[0] = B[0][0]; q[1] = B[0][1]; q[2] = B[0][2]; q[3] = B[0][3];
            q[4] = B[1][1]; q[5] = B[1][2]; q[6] = B[1][3];
            q[7] = B[2][2]; q[8] = B[2][3];
            q[9] = B[3][3];
            break;
        }
        case RRT_OBJ_MESH: {  // polymesh_object.rs:244-252, triangle_object.rs:162-173, vertex.rs:119-126
            size_t n = tri_pos.size() / 3;
            for (size_t i = 0; i < n; i++) {
                f3 v = xform_vertex(m, f3{tri_pos[i * 3], tri_pos[i * 3 + 1], tri_pos[i * 3 + 2]});
                tri_pos[i * 3] = v.x; tri_pos[i * 3 + 1] = v.y; tri_pos[i * 3 + 2] = v.z;
            }
            size_t nn = tri_nrm.size() / 3;
            for (size_t i = 0; i < nn; i++) {
                f3 v = norm3(xform_vector(m, f3{tri_nrm[i * 3], tri_nrm[i * 3 + 1], tri_nrm[i * 3 + 2]}));
                tri_nrm[i * 3] = v.x; tri_nrm[i * 3 + 1] = v.y; tri_nrm[i * 3 + 2] = v.z;
            }
            break;
        }
        case RRT_OBJ_CSG:  // csg_object.rs:110-113
            left->apply_transform(m);
            right->apply_transform(m);
            break;
    }
}

}  // namespace rrt

// ---- entry points that never touch the device (also linked into the CPU-only tools under tools/) ----
using namespace rrt;
extern "C" {

const char *rrt_last_error(void) { return rrt::last_error_cstr(); }
const char *rrt_version(void) { return "rrt 0.1 (gfx950)"; }

void rrt_params_default(rrt_params *p) {
    memset(p, 0, sizeof(*p));
    p->photons_per_light = 5000000;
    p->caustic_photons_per_light = 10000;
    p->gather_half_width = 0.1f;
    p->max_depth = 5;
    p->max_photon_bounces = 64;
    p->max_vueon_depth = 12;
    p->chunk_pixels = 0;
    p->bvh_builder = RRT_BVH_AUTO;
}

// full_camera.rs:23-58
int rrt_camera_init(rrt_camera *cam, uint32_t width, uint32_t height, float fov, const float position[3], const float lookat_in[3],
                    const float up_in[3]) {
    if (!cam || !position || !lookat_in || !up_in) return fail(RRT_ERR_INVALID, "null argument");
    if (width == 0 || height == 0) return fail(RRT_ERR_INVALID, "Invalid framebuffer size");
    f3 lookat = mk3(lookat_in[0], -lookat_in[1], lookat_in[2]);
    f3 up = mk3(up_in[0], -up_in[1], up_in[2]);
    lookat = norm3(lookat);
    up = norm3(up);
    float dotp = dot3(lookat, up);
    float len = len3(lookat) * len3(up);
    float ang = acosf(dotp) / len;  // vector.rs:114-119
    const float right_angle = 1.57079632679489661923f;
    if (fabsf(ang - right_angle) > 0.0001f) return fail(RRT_ERR_INVALID, "FullCamera right and up are not perpendicular");
    f3 right = norm3(cross3(lookat, up));
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
int rrt_camera_ray(const rrt_camera *cam, uint32_t x, uint32_t y, float origin[3], float direction[3]) {
    if (!cam || x >= cam->width || y >= cam->height) return fail(RRT_ERR_INVALID, "pixel out of range");
    float fx = ((float)x + 0.5f) / (float)cam->width;
    float fy = ((float)y + 0.5f) / (float)cam->height;
    f3 v = mk3(fx - 0.5f, fy - 0.5f, cam->fov);
    f3 d = norm3(mk3(cam->right[0] * v.x + cam->up[0] * v.y + cam->lookat[0] * v.z, cam->right[1] * v.x + cam->up[1] * v.y + cam->lookat[1] * v.z,
                     cam->right[2] * v.x + cam->up[2] * v.y + cam->lookat[2] * v.z));
    for (int i = 0; i < 3; i++) origin[i] = cam->position[i];
    direction[0] = d.x; direction[1] = d.y; direction[2] = d.z;
    return RRT_OK;
}

}  // extern "C"
