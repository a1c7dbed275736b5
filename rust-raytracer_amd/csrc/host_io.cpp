// host_io.cpp — the data formats either side of the hot path: PPM textures in, OBJ meshes in,
// PPM/PGM frames out.  Host only.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "host_scene.h"

namespace rrt {

// Rust `str::parse::<f32>()`: decimal / exponent / inf / nan, whole string, no hex, no spaces
bool parse_f32(const std::string &s, float &out) {
    if (s.empty()) return false;
    for (char c : s)
        if (c == 'x' || c == 'X' || c == ' ' || c == '\t') return false;
    char *end = nullptr;
    out = strtof(s.c_str(), &end);
    return end && *end == '\0' && end != s.c_str();
}
static bool parse_usize(const std::string &s, uint64_t &out) {
    if (s.empty()) return false;
    size_t i = (s[0] == '+') ? 1 : 0;
    if (i >= s.size()) return false;
    uint64_t v = 0;
    for (; i < s.size(); i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        v = v * 10 + (uint64_t)(s[i] - '0');
    }
    out = v;
    return true;
}
std::vector<std::string> split_ws(const std::string &line) {
    std::vector<std::string> w;
    size_t i = 0, n = line.size();
    while (i < n) {
        while (i < n && isspace((unsigned char)line[i])) i++;
        size_t j = i;
        while (j < n && !isspace((unsigned char)line[j])) j++;
        if (j > i) w.push_back(line.substr(i, j - i));
        i = j;
    }
    return w;
}

// Image::from_ppm (texture.rs:41-103): "P6", width, height, maxval separated by single whitespace
// bytes, then width*height RGB samples of 1 (maxval <= 255) or 2 big-endian bytes.
int read_ppm(const std::string &path, HostTexture &tex) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return fail(RRT_ERR_IO, "cannot open " + path);
    auto token = [&](std::string &out) -> bool {  // read_until_whitespace, texture.rs:45-56
        out.clear();
        for (;;) {
            int c = fgetc(f);
            if (c == EOF) return false;
            if (c == ' ' || c == '\n' || c == '\r' || c == '\t') return true;
            out.push_back((char)c);
        }
    };
    std::string t;
    uint64_t w, h, mv;
    bool ok = token(t) && t == "P6";
    ok = ok && token(t) && parse_usize(t, w) && token(t) && parse_usize(t, h) && token(t) && parse_usize(t, mv);
    if (!ok || w == 0 || h == 0 || mv == 0 || mv > 65535 || w > 65535 || h > 65535) {
        fclose(f);
        return fail(RRT_ERR_IO, "Invalid PPM file: " + path);
    }
    size_t bytes = (size_t)w * h * 3 * (mv <= 255 ? 1 : 2);
    tex.width = (uint32_t)w;
    tex.height = (uint32_t)h;
    tex.maxval = (uint32_t)mv;
    tex.samples.resize(bytes);
    size_t got = fread(tex.samples.data(), 1, bytes, f);
    fclose(f);
    if (got != bytes) return fail(RRT_ERR_IO, "truncated PPM file: " + path);
    return RRT_OK;
}

// PolyMesh::from_obj_file + parse_face (polymesh_object.rs:34-194): `v`, `vn`, `f a/b/c ...` with
// 1-based indices, a triangle (0,1,2) plus (0,2,3) when the face has exactly four corners.  Output is
// what each Triangle ends up holding: corner positions and corner normals.
// The reference's missing-normal smoothing path (:101-128) indexes a map by face-local corner index
// and panics on real meshes; here a smooth mesh whose corners lack `vn` is rejected instead.
int load_obj(const std::string &path, bool smooth, std::vector<float> &tri_pos, std::vector<float> &tri_nrm) {
    std::ifstream in(path);
    if (!in) return fail(RRT_ERR_IO, "Could not open OBJ file at path " + path);
    std::vector<f3> verts, normals;
    std::vector<int64_t> vert_normal;  // RichVertex.normal as last written by parse_face (-1 = None)
    bool all_normals = true;
    std::string line;
    size_t line_no = 0;
    tri_pos.clear();
    tri_nrm.clear();
    std::vector<int64_t> tri_nidx;  // per triangle corner: normal index or -1
    while (std::getline(in, line)) {
        line_no++;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        std::vector<std::string> w = split_ws(line);
        if (w.empty()) continue;  // the reference would panic on words[0]
        const std::string &tag = w[0];
        if (tag == "#") continue;
        if (tag == "v" || tag == "vn") {
            float c[3];
            if (w.size() < 4 || !parse_f32(w[1], c[0]) || !parse_f32(w[2], c[1]) || !parse_f32(w[3], c[2]))
                return fail(RRT_ERR_IO, path + ":" + std::to_string(line_no) + ": Could not parse coordinate");
            if (tag == "v") {
                verts.push_back(f3{c[0], c[1], c[2]});
                vert_normal.push_back(-1);
            } else {
                normals.push_back(f3{c[0], c[1], c[2]});
            }
        } else if (tag == "f") {
            std::vector<std::pair<uint64_t, int64_t>> idx;
            for (size_t k = 1; k < w.size(); k++) {
                std::vector<std::string> parts;
                std::stringstream ss(w[k]);
                std::string piece;
                while (std::getline(ss, piece, '/')) parts.push_back(piece);
                if (!w[k].empty() && w[k].back() == '/') parts.push_back("");
                uint64_t vi, ni;
                if (parts.empty() || !parse_usize(parts[0], vi) || vi == 0)
                    return fail(RRT_ERR_IO, path + ":" + std::to_string(line_no) + ": Could not parse vertex index");
                int64_t n = -1;
                if (parts.size() > 2) {
                    if (!parse_usize(parts[2], ni) || ni == 0)
                        return fail(RRT_ERR_IO, path + ":" + std::to_string(line_no) + ": Could not parse normal index");
                    n = (int64_t)ni - 1;
                }
                idx.push_back({vi - 1, n});
            }
            if (idx.size() < 3) return fail(RRT_ERR_IO, path + ":" + std::to_string(line_no) + ": face with fewer than 3 corners");
            auto emit = [&](size_t i, size_t j, size_t k) -> int {
                size_t c3[3] = {i, j, k};
                for (size_t c : c3) {
                    if (idx[c].first >= verts.size() || (idx[c].second >= 0 && (size_t)idx[c].second >= normals.size()))
                        return fail(RRT_ERR_IO, path + ":" + std::to_string(line_no) + ": index out of range");
                    vert_normal[idx[c].first] = idx[c].second;  // polymesh_object.rs:165-168
                }
                for (size_t c : c3) {  // clones taken after all three writes, :170-173
                    f3 v = verts[idx[c].first];
                    int64_t n = vert_normal[idx[c].first];
                    tri_pos.insert(tri_pos.end(), {v.x, v.y, v.z});
                    tri_nidx.push_back(n);
                    if (n < 0) all_normals = false;
                }
                return RRT_OK;
            };
            int rc = emit(0, 1, 2);
            if (rc) return rc;
            if (idx.size() == 4) {
                rc = emit(0, 2, 3);
                if (rc) return rc;
            }
        }
    }
    if (all_normals && !tri_nidx.empty()) {
        tri_nrm.reserve(tri_nidx.size() * 3);
        for (int64_t n : tri_nidx) tri_nrm.insert(tri_nrm.end(), {normals[n].x, normals[n].y, normals[n].z});
    } else if (smooth) {
        return fail(RRT_ERR_INVALID, path + ": smooth mesh needs a `vn` index on every face corner "
                                            "(the reference's fallback, polymesh_object.rs:101-128, panics)");
    }
    return RRT_OK;
}

}  // namespace rrt

using namespace rrt;

extern "C" {

// framebuffer.rs:105-107,124: `(c * 255.0) as u8`, Rust's saturating cast (NaN -> 0)
void rrt_quantise(const float *values, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        float f = values[i] * 255.0f;
        out[i] = !(f > 0.0f) ? 0 : (f >= 255.0f ? 255 : (uint8_t)f);
    }
}

static int write_pnm(const char *path, const char *magic, const float *v, uint32_t w, uint32_t h, int ch) {
    if (!path || !v) return fail(RRT_ERR_INVALID, "null argument");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(RRT_ERR_IO, std::string("cannot create ") + path);
    fprintf(f, "%s\n%u %u\n255\n", magic, w, h);  // framebuffer.rs:100,119
    std::vector<uint8_t> row((size_t)w * ch);
    for (uint32_t y = 0; y < h; y++) {
        rrt_quantise(v + (size_t)y * w * ch, row.size(), row.data());
        fwrite(row.data(), 1, row.size(), f);
    }
    fclose(f);
    return RRT_OK;
}
int rrt_write_rgb_ppm(const char *path, const float *rgb, uint32_t w, uint32_t h) { return write_pnm(path, "P6", rgb, w, h, 3); }
int rrt_write_depth_pgm(const char *path, const float *depth, uint32_t w, uint32_t h) { return write_pnm(path, "P5", depth, w, h, 1); }

}  // extern "C"
