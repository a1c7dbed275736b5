// scene_file.cpp — the indentation-paragraph scene grammar of the reference
// (src/scene_file.rs:59-667), parsed into calls on an rrt_builder.  Host only.
//
// Quirks kept on purpose (they decide what the reference's own scene files mean):
//  * a paragraph's attribute map also receives the lines of its nested sub-paragraphs, last key
//    wins (scene_file.rs:203-253 keeps iterating after recursing), so `scale` inside a nested
//    `material Texture` overrides an object's own `scale`;
//  * unknown keys are ignored; a column-0 `#` line ends a paragraph;
//  * `Model` builds one matrix from `translate` (row 3) and `scale` (diagonal) and transposes it,
//    `MarioPipe` likewise, `MarioCoin` applies scale and translate as two transforms.
#include <math.h>
#include <string.h>

#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "host_scene.h"

namespace rrt {

bool parse_f32(const std::string &s, float &out);
std::vector<std::string> split_ws(const std::string &line);
int read_ppm(const std::string &path, HostTexture &tex);
int load_obj(const std::string &path, bool smooth, std::vector<float> &tri_pos, std::vector<float> &tri_nrm);

namespace {

struct ParseFailure {
    int code;
    std::string message;
    uint32_t line;
};
[[noreturn]] void bail(uint32_t line, const std::string &msg) { throw ParseFailure{RRT_ERR_PARSE, msg, line}; }

struct Paragraph;
struct Attribute {
    enum Kind { WORD, FLOAT, VECTOR, SUB } kind = FLOAT;
    std::string word;
    float f = 0.0f;
    f3 v{0, 0, 0};
    std::shared_ptr<Paragraph> sub;
    uint32_t line = 0;

    static Attribute Float(float f) {
        Attribute a;
        a.kind = FLOAT;
        a.f = f;
        return a;
    }
    static Attribute Vector(f3 v) {
        Attribute a;
        a.kind = VECTOR;
        a.v = v;
        return a;
    }
    std::string as_word() const {  // scene_file.rs:617-622
        if (kind != WORD) bail(line, "Invalid attribute value for word");
        return word;
    }
    f3 as_colour() const {  // :624-635
        if (kind == VECTOR) return v;
        if (kind == FLOAT) return f3{f, f, f};
        if (kind == WORD) {
            if (word == "White") return f3{1, 1, 1};
            if (word == "Black") return f3{0, 0, 0};
            bail(line, "Unknown colour name: " + word);
        }
        bail(line, "Invalid attribute value for colour");
    }
    f3 as_vector() const {  // :637-643
        if (kind == VECTOR) return v;
        if (kind == FLOAT) return f3{f, f, f};
        bail(line, "Invalid attribute value for vector");
    }
    float as_float() const {  // :650-655
        if (kind != FLOAT) bail(line, "Invalid attribute value for float");
        return f;
    }
};

size_t indentation(const std::string &s) {
    size_t n = 0;
    while (n < s.size() && isspace((unsigned char)s[n])) n++;
    return n;
}
bool blank(const std::string &s) { return indentation(s) == s.size(); }

struct Paragraph {
    std::string kind, cls;
    std::map<std::string, Attribute> attributes;
    uint32_t start_line = 0;

    // scene_file.rs:178-261
    static std::shared_ptr<Paragraph> parse(const std::vector<std::string> &lines, size_t first, uint32_t first_line_number) {
        auto p = std::make_shared<Paragraph>();
        const std::string &first_line = lines[first];
        size_t indent = indentation(first_line);
        std::vector<std::string> words = split_ws(first_line);
        if (words.empty()) bail(first_line_number, "Empty paragraph");
        if (words.size() < 2) bail(first_line_number, "Missing paragraph class");
        if (words.size() > 2) bail(first_line_number, "Too many words in paragraph header: " + words[2]);
        p->kind = words[0];
        p->cls = words[1];
        p->start_line = first_line_number;
        for (size_t i = first + 1; i < lines.size(); i++) {
            const std::string &line = lines[i];
            if (line.empty() || indentation(line) <= indent) break;
            uint32_t line_number = first_line_number + (uint32_t)(i - first);
            std::vector<std::string> w = split_ws(line);
            const std::string key = w[0];
            if (key[0] == '#') continue;
            Attribute a;
            a.line = line_number;
            size_t nvals = w.size() - 1;
            if (nvals == 3) {
                float x, y, z;
                if (!parse_f32(w[1], x) || !parse_f32(w[2], y) || !parse_f32(w[3], z)) bail(line_number, "Invalid float");
                a.kind = Attribute::VECTOR;
                a.v = f3{x, y, z};
            } else if (nvals == 1) {
                if (i + 1 < lines.size() && indentation(lines[i + 1]) > indentation(line)) {
                    a.kind = Attribute::SUB;
                    a.sub = parse(lines, i, line_number);
                } else {
                    float f;
                    if (parse_f32(w[1], f)) {
                        a.kind = Attribute::FLOAT;
                        a.f = f;
                    } else {
                        a.kind = Attribute::WORD;
                        a.word = w[1];
                    }
                }
            } else {
                bail(line_number, "Invalid word count in attribute value: " + line);
            }
            p->attributes[key] = a;  // HashMap::insert: last wins
        }
        return p;
    }

    Attribute get_attr(const std::string &key) {  // :588-592
        auto it = attributes.find(key);
        if (it == attributes.end()) bail(start_line, "Missing required attribute: " + key);
        Attribute a = it->second;
        attributes.erase(it);
        return a;
    }
    Attribute get_attr_or(const std::string &key, Attribute dflt) {  // :594-600
        auto it = attributes.find(key);
        if (it == attributes.end()) return dflt;
        Attribute a = it->second;
        attributes.erase(it);
        return a;
    }
    bool take(const std::string &key, Attribute &out) {  // self.attributes.remove(key)
        auto it = attributes.find(key);
        if (it == attributes.end()) return false;
        out = it->second;
        attributes.erase(it);
        return true;
    }
};

struct Loader {
    const rrt_builder *b;
    std::string root;
    std::map<std::string, int> texture_cache;  // the reference re-reads the file per material; same texels

    int check(int rc, uint32_t line) {
        if (rc < 0) throw ParseFailure{rc, "", line};
        return rc;
    }

    // Image::from_image (texture.rs:22-39) without the ffmpeg shell-out: the `.ppm` sibling must exist
    int texture(const std::string &name, const char *which, bool required, uint32_t line) {
        std::string path = root + "/assets/textures/" + name + "/" + which + ".ppm";
        auto it = texture_cache.find(path);
        if (it != texture_cache.end()) return it->second;
        HostTexture t;
        int rc = read_ppm(path, t);
        if (rc) {
            if (required) throw ParseFailure{rc, "", line};
            texture_cache[path] = -1;
            return -1;
        }
        int id = check(b->add_texture(b->ctx, t.width, t.height, t.maxval, t.samples.data()), line);
        texture_cache[path] = id;
        return id;
    }

    int monochrome(f3 c, float ambient, float shininess, uint32_t line) {
        rrt_material_desc d;
        memset(&d, 0, sizeof(d));
        d.kind = RRT_MAT_MONOCHROME;
        d.colour[0] = c.x; d.colour[1] = c.y; d.colour[2] = c.z;
        d.ambient = ambient;
        d.shininess = shininess;
        d.diffuse_tex = d.normal_tex = -1;
        return check(b->add_material(b->ctx, &d), line);
    }
    int global(float reflect, float refract, float ior, uint32_t line) {
        rrt_material_desc d;
        memset(&d, 0, sizeof(d));
        d.kind = RRT_MAT_GLOBAL;
        d.reflect_weight = reflect;
        d.refract_weight = refract;
        d.ior = ior;
        d.diffuse_tex = d.normal_tex = -1;
        return check(b->add_material(b->ctx, &d), line);
    }
    int texture_material(const std::string &name, float scale, float ambient, float shininess, uint32_t line) {
        rrt_material_desc d;
        memset(&d, 0, sizeof(d));
        d.kind = RRT_MAT_TEXTURE;
        d.diffuse_tex = texture(name, "diffuse", true, line);  // texture.rs:140 unwrap
        d.normal_tex = texture(name, "normal", false, line);   // texture.rs:141 ok()
        d.scale = scale;
        d.ambient = ambient;
        d.shininess = shininess;
        return check(b->add_material(b->ctx, &d), line);
    }
    int compound2(int m0, int m1, uint32_t line) {
        rrt_material_desc d;
        memset(&d, 0, sizeof(d));
        d.kind = RRT_MAT_COMPOUND;
        d.diffuse_tex = d.normal_tex = -1;
        d.n_parts = 2;
        d.parts[0] = m0;
        d.parts[1] = m1;
        return check(b->add_material(b->ctx, &d), line);
    }

    // scene_file.rs:509-551 + compound_material.rs:39-76
    int into_material(Paragraph &p) {
        const std::string &c = p.cls;
        uint32_t ln = p.start_line;
        if (c == "Simple") {
            f3 colour = p.get_attr("colour").as_colour();
            float refl = p.get_attr("reflectiveness").as_float();
            float shin = p.get_attr_or("shininess", Attribute::Float(100.0f)).as_float();
            int phong = monochrome(colour, 0.1f, shin, ln);
            int g = global(refl, 0.0f, 1.0f, ln);
            return compound2(phong, g, ln);
        }
        if (c == "Transparent") {
            f3 colour = p.get_attr("colour").as_colour();
            float t = p.get_attr("transparency").as_float();
            float ior = p.get_attr("ior").as_float();
            float shin = p.get_attr_or("shininess", Attribute::Float(100.0f)).as_float();
            float opaqueness = 1.0f - t;
            int phong = monochrome(colour * opaqueness, 0.1f, shin, ln);
            int g = global(t, t, ior, ln);
            return compound2(phong, g, ln);
        }
        if (c == "Global") {
            float refl = p.get_attr("reflect").as_float();
            float refr = p.get_attr("refract").as_float();
            float ior = p.get_attr("ior").as_float();
            return global(refl, refr, ior, ln);
        }
        if (c == "Monochrome") {
            f3 colour = p.get_attr("colour").as_colour();
            float amb = p.get_attr_or("ambient", Attribute::Float(0.1f)).as_float();
            float shin = p.get_attr("shininess").as_float();
            return monochrome(colour, amb, shin, ln);
        }
        if (c == "Texture") {
            std::string name = p.get_attr("name").as_word();
            float scale = p.get_attr("scale").as_float();
            float amb = p.get_attr("ambient").as_float();
            float shin = p.get_attr("shininess").as_float();
            return texture_material(name, scale, amb, shin, ln);
        }
        if (c == "TransparentTexture") {
            std::string name = p.get_attr("name").as_word();
            float scale = p.get_attr("scale").as_float();
            float t = p.get_attr("transparency").as_float();
            int tex = texture_material(name, scale, 0.1f, 1000000.0f, ln);
            int g = global(t, t, 1.0f, ln);
            return compound2(tex, g, ln);
        }
        if (c == "FalseColour") {
            rrt_material_desc d;
            memset(&d, 0, sizeof(d));
            d.kind = RRT_MAT_FALSECOLOUR;
            d.diffuse_tex = d.normal_tex = -1;
            return check(b->add_material(b->ctx, &d), ln);
        }
        bail(ln, "Invalid material class: " + c);
    }
    int material_attr(Paragraph &p) {  // Attribute::into_material, scene_file.rs:657-662
        Attribute a = p.get_attr("material");
        if (a.kind != Attribute::SUB) bail(a.line, "Invalid attribute value for material");
        return into_material(*a.sub);
    }

    int define(const rrt_object_desc &d, uint32_t line) { return check(b->object_define(b->ctx, &d), line); }
    int simple(int kind, int material, std::initializer_list<float> v, uint32_t line) {
        rrt_object_desc d;
        memset(&d, 0, sizeof(d));
        d.kind = kind;
        d.material = material;
        int i = 0;
        for (float f : v) d.v[i++] = f;
        return define(d, line);
    }
    int csg(int mode, int left, int right, uint32_t line) {
        rrt_object_desc d;
        memset(&d, 0, sizeof(d));
        d.kind = RRT_OBJ_CSG;
        d.material = -1;
        d.csg_mode = mode;
        d.left = left;
        d.right = right;
        return define(d, line);
    }
    int cylinder(float diameter, int material, uint32_t line) {  // Quadratic::cylinder, quadratic_object.rs:34-38
        float radius = diameter / 2.0f;
        float a = 1.0f / (radius * radius);
        return simple(RRT_OBJ_QUADRATIC, material, {a, 0, 0, 0, 0, 0, 0, a, 0, -1.0f}, line);
    }
    void transform(int handle, const float m[4][4], uint32_t line) { check(b->object_transform(b->ctx, handle, &m[0][0]), line); }

    static void identity(float m[4][4]) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) m[i][j] = i == j ? 1.0f : 0.0f;
    }
    static void transpose(float m[4][4]) {
        for (int i = 0; i < 4; i++)
            for (int j = i + 1; j < 4; j++) std::swap(m[i][j], m[j][i]);
    }

    // scene_file.rs:310-507; returns a detached handle
    int into_object(Paragraph &p) {
        const std::string &c = p.cls;
        uint32_t ln = p.start_line;
        if (c == "Plane") {
            f3 point = p.get_attr("point").as_vector();
            f3 up = p.get_attr("up").as_vector();
            f3 normal = p.get_attr("normal").as_vector();
            int m = material_attr(p);
            return simple(RRT_OBJ_PLANE, m, {point.x, point.y, point.z, up.x, up.y, up.z, normal.x, normal.y, normal.z}, ln);
        }
        if (c == "Sphere") {
            f3 centre = p.get_attr("centre").as_vector();
            float radius = p.get_attr("radius").as_float();
            int m = material_attr(p);
            return simple(RRT_OBJ_SPHERE, m, {centre.x, centre.y, centre.z, radius}, ln);
        }
        if (c == "Cuboid") {
            f3 corner = p.get_attr("corner").as_vector();
            f3 size = p.get_attr("size").as_vector();
            int m = material_attr(p);
            return simple(RRT_OBJ_CUBOID, m, {corner.x, corner.y, corner.z, size.x, size.y, size.z}, ln);
        }
        if (c == "Quadratic") {
            float q[10];
            const char *names[10] = {"a", "b", "c", "d", "e", "f", "g", "h", "i", "j"};
            for (int i = 0; i < 10; i++) q[i] = p.get_attr_or(names[i], Attribute::Float(0.0f)).as_float();
            int m = material_attr(p);
            int h = simple(RRT_OBJ_QUADRATIC, m, {q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9]}, ln);
            Attribute tr;
            if (p.take("translate", tr)) {  // Transform::from_translation, transform.rs:35-44
                f3 t = tr.as_vector();
                float mtx[4][4];
                identity(mtx);
                mtx[0][3] = t.x;
                mtx[1][3] = t.y;
                mtx[2][3] = t.z;
                transform(h, mtx, ln);
            }
            return h;
        }
        if (c == "Model") {
            std::string obj = p.get_attr("obj").as_word();
            std::string path = root + "/assets/models/" + obj;
            int m = material_attr(p);
            bool smooth = p.get_attr_or("smooth", Attribute::Float(0.0f)).as_float() != 0.0f;
            std::vector<float> pos, nrm;
            int rc = load_obj(path, smooth, pos, nrm);
            if (rc) throw ParseFailure{rc, "", ln};
            rrt_object_desc d;
            memset(&d, 0, sizeof(d));
            d.kind = RRT_OBJ_MESH;
            d.material = m;
            d.positions = pos.data();
            d.normals = nrm.empty() ? nullptr : nrm.data();
            d.n_triangles = pos.size() / 9;
            d.smooth = smooth ? 1 : 0;
            int h = define(d, ln);
            Attribute dummy;
            if (p.take("rotate_teapot", dummy)) {
                float r[4][4];
                identity(r);
                r[1][1] = 0.0f; r[1][2] = 1.0f;
                r[2][1] = -1.0f; r[2][2] = 0.0f;
                transform(h, r, ln);
            }
            float t[4][4];
            identity(t);
            Attribute a;
            if (p.take("translate", a)) {
                f3 v = a.as_vector();
                t[3][0] = v.x; t[3][1] = v.y; t[3][2] = v.z;
            }
            if (p.take("scale", a)) {
                float s = a.as_float();
                t[0][0] = s; t[1][1] = s; t[2][2] = s;
            }
            transpose(t);
            transform(h, t, ln);
            return h;
        }
        if (c == "MarioPipe") {  // scene_file.rs:403-455
            int m = material_attr(p);
            int main_cyl = cylinder(1.0f, m, ln);
            int main_box = simple(RRT_OBJ_CUBOID, m, {-0.5f, 0.0f, -0.5f, 1.0f, 3.0f, 1.0f}, ln);
            int main_part = csg(RRT_CSG_INTERSECTION, main_cyl, main_box, ln);
            int top_cyl = cylinder(1.3f, m, ln);
            int top_box = simple(RRT_OBJ_CUBOID, m, {-0.7f, 2.7f, -0.7f, 1.3f, 0.3f, 1.3f}, ln);
            int top_part = csg(RRT_CSG_INTERSECTION, top_cyl, top_box, ln);
            int pipe = csg(RRT_CSG_UNION, main_part, top_part, ln);
            int black1 = monochrome(f3{0, 0, 0}, 0.1f, 100000000.0f, ln);
            int hole = simple(RRT_OBJ_SPHERE, black1, {0.0f, 3.0f, 0.0f, 0.5f}, ln);
            int black2 = monochrome(f3{0, 0, 0}, 0.1f, 100000000.0f, ln);
            int hole_box = simple(RRT_OBJ_CUBOID, black2, {-0.5f, 3.0f, -0.5f, 1.0f, 0.0001f, 1.0f}, ln);
            int hole_part = csg(RRT_CSG_INTERSECTION, hole, hole_box, ln);
            pipe = csg(RRT_CSG_UNION, pipe, hole_part, ln);
            float t[4][4];
            identity(t);
            Attribute a;
            if (p.take("scale", a)) {
                float s = a.as_float();
                t[0][0] = s; t[1][1] = s; t[2][2] = s;
            }
            if (p.take("translate", a)) {
                f3 v = a.as_vector();
                t[3][0] = v.x; t[3][1] = v.y; t[3][2] = v.z;
            }
            transpose(t);
            transform(pipe, t, ln);
            return pipe;
        }
        if (c == "MarioCoin") {  // scene_file.rs:456-503
            int m = material_attr(p);
            int coin_s = simple(RRT_OBJ_SPHERE, m, {0.0f, 0.0f, 0.0f, 0.5f}, ln);
            int coin_b = simple(RRT_OBJ_CUBOID, m, {-0.5f, -0.5f, -0.1f, 1.0f, 1.0f, 0.2f}, ln);
            int coin = csg(RRT_CSG_INTERSECTION, coin_s, coin_b, ln);
            int sh_s = simple(RRT_OBJ_SPHERE, m, {0.0f, 0.0f, 0.0f, 0.4f}, ln);
            int sh_b = simple(RRT_OBJ_CUBOID, m, {-0.4f, -0.4f, -0.11f, 0.8f, 0.8f, 0.03f}, ln);
            int shallow = csg(RRT_CSG_INTERSECTION, sh_s, sh_b, ln);
            coin = csg(RRT_CSG_DIFFERENCE, coin, shallow, ln);
            int line_box = simple(RRT_OBJ_CUBOID, m, {-0.08f, -0.28f, -0.1f, 0.16f, 0.56f, 0.05f}, ln);
            coin = csg(RRT_CSG_DIFFERENCE, coin, line_box, ln);
            Attribute a;
            if (p.take("scale", a)) {
                float s = a.as_float();
                float t[4][4];
                identity(t);
                t[0][0] = s; t[1][1] = s; t[2][2] = s;
                transpose(t);
                transform(coin, t, ln);
            }
            if (p.take("translate", a)) {
                f3 v = a.as_vector();
                float t[4][4];
                identity(t);
                t[3][0] = v.x; t[3][1] = v.y; t[3][2] = v.z;
                transpose(t);
                transform(coin, t, ln);
            }
            return coin;
        }
        bail(ln, "Invalid object class: " + c);
    }

    void into_light(Paragraph &p) {  // scene_file.rs:287-308
        rrt_light_desc d;
        memset(&d, 0, sizeof(d));
        const std::string &c = p.cls;
        f3 pos{0, 0, 0}, dir{0, 0, 0};
        if (c == "Directional") {
            d.kind = RRT_LIGHT_DIRECTIONAL;
            dir = p.get_attr("direction").as_vector();
        } else if (c == "Point") {
            d.kind = RRT_LIGHT_POINT;
            pos = p.get_attr("position").as_vector();
        } else if (c == "DirPoint") {
            d.kind = RRT_LIGHT_DIRPOINT;
            pos = p.get_attr("position").as_vector();
            dir = p.get_attr("direction").as_vector();
        } else {
            bail(p.start_line, "Invalid light class: " + c);
        }
        f3 col = p.get_attr_or("colour", Attribute::Float(1.0f)).as_colour();
        d.position[0] = pos.x; d.position[1] = pos.y; d.position[2] = pos.z;
        d.direction[0] = dir.x; d.direction[1] = dir.y; d.direction[2] = dir.z;
        d.colour[0] = col.x; d.colour[1] = col.y; d.colour[2] = col.z;
        check(b->add_light(b->ctx, &d), p.start_line);
    }
};

const float kRadsPerDeg = 3.14159265358979323846f / 180.0f;  // f32::to_radians

void camera_or_fail(rrt_camera *cam, uint32_t w, uint32_t h, float fov, f3 pos, f3 lookat, f3 up, uint32_t line) {
    float p[3] = {pos.x, pos.y, pos.z}, l[3] = {lookat.x, lookat.y, lookat.z}, u[3] = {up.x, up.y, up.z};
    int rc = rrt_camera_init(cam, w, h, fov, p, l, u);
    if (rc) throw ParseFailure{rc, "", line};
}

int parse_text(const std::string &contents, const std::string &root, const rrt_builder *builder, rrt_camera *camera_out,
               int mode_override, uint32_t *err_line) {
    if (err_line) *err_line = 0;
    try {
        // str::lines(): split on '\n', strip one trailing '\r'
        std::vector<std::string> all;
        {
            size_t start = 0;
            while (start <= contents.size()) {
                size_t nl = contents.find('\n', start);
                if (nl == std::string::npos) {
                    if (start < contents.size()) all.push_back(contents.substr(start));
                    break;
                }
                all.push_back(contents.substr(start, nl - start));
                start = nl + 1;
            }
            for (auto &l : all)
                if (!l.empty() && l.back() == '\r') l.pop_back();
        }
        // Paragraph::parse_whole_file, scene_file.rs:132-176
        std::vector<std::shared_ptr<Paragraph>> paragraphs;
        std::vector<std::string> current;
        uint32_t current_start = 0;
        auto flush = [&]() {
            if (current.empty()) return;
            paragraphs.push_back(Paragraph::parse(current, 0, current_start));
            current.clear();
        };
        for (size_t i = 0; i < all.size(); i++) {
            const std::string &line = all[i];
            uint32_t line_number = (uint32_t)i + 1;
            if (blank(line) || line[0] == '#') {
                flush();
                continue;
            }
            size_t ind = indentation(line);
            if (current.empty() && ind == 0) {
                current.push_back(line);
                current_start = line_number;
            } else if (!current.empty() && ind > 0) {
                current.push_back(line);
            } else {
                bail(line_number, "Started next paragraph without empty newline");
            }
        }
        flush();

        Loader L{builder, root, {}};
        // SceneFile::from_contents, scene_file.rs:64-121
        int mode = RRT_MODE_SCENE;
        bool seen_scene = false;
        for (auto &p : paragraphs) {
            if (p->kind != "scene") continue;
            if (seen_scene) bail(p->start_line, "Multiple scenes in file");
            seen_scene = true;
            if (p->cls == "Scene")
                mode = RRT_MODE_SCENE;
            else if (p->cls == "PhotonScene")
                mode = RRT_MODE_PHOTON;
            else
                bail(p->start_line, "Invalid scene class: " + p->cls);
        }
        if (mode_override >= 0) mode = mode_override;
        L.check(builder->set_mode(builder->ctx, mode), 0);

        bool have_camera = false;
        for (auto &p : paragraphs) {
            if (p->kind == "scene") continue;
            if (p->kind == "light") {
                L.into_light(*p);
            } else if (p->kind == "object") {
                int h = L.into_object(*p);
                L.check(builder->add_object(builder->ctx, h), p->start_line);
            } else if (p->kind == "material") {
                L.into_material(*p);
                bail(p->start_line, "Cannot add material to scene on its own");
            } else if (p->kind == "camera") {  // into_camera, scene_file.rs:553-586
                if (p->cls != "Camera") bail(p->start_line, "Invalid camera class: " + p->cls);
                Attribute res = p->get_attr_or("res", Attribute::Float(1024.0f));
                float width = p->get_attr_or("width", res).as_float();
                float height = p->get_attr_or("height", Attribute::Float(width)).as_float();
                float fov = p->get_attr_or("fov", Attribute::Float(40.0f)).as_float() * kRadsPerDeg;
                f3 position = p->get_attr_or("position", Attribute::Vector(f3{0, 0, 0})).as_vector();
                f3 lookat = p->get_attr_or("lookat", Attribute::Vector(f3{0, 0, 1})).as_vector();
                f3 up = p->get_attr_or("up", Attribute::Vector(f3{0.0f, lookat.z, -lookat.y})).as_vector();
                if (have_camera) bail(p->start_line, "Multiple cameras in file");
                if (camera_out) camera_or_fail(camera_out, sat_u32(width), sat_u32(height), fov, position, lookat, up, p->start_line);
                have_camera = true;
            } else {
                bail(p->start_line, "Invalid paragraph kind: " + p->kind);
            }
        }
        if (!have_camera && camera_out) {  // scene_file.rs:107-119
            f3 lookat = norm3(f3{0.0f, 0.5f, 1.0f});
            f3 up{0.0f, lookat.z, -lookat.y};
            camera_or_fail(camera_out, 1024, 1024, 40.0f * kRadsPerDeg, f3{0.0f, 3.0f, 0.0f}, lookat, up, 0);
        }
    } catch (ParseFailure &pf) {
        if (err_line) *err_line = pf.line;
        if (!pf.message.empty()) set_error("line " + std::to_string(pf.line) + ": " + pf.message);
        return pf.code;
    }
    return RRT_OK;
}

}  // namespace
}  // namespace rrt

extern "C" {

int rrt_parse_scene_text(const char *text, const char *asset_root, const rrt_builder *builder, rrt_camera *camera_out,
                         int mode_override, uint32_t *err_line) {
    if (!text || !builder) return rrt::fail(RRT_ERR_INVALID, "null argument");
    return rrt::parse_text(text, asset_root ? asset_root : ".", builder, camera_out, mode_override, err_line);
}

int rrt_parse_scene_file(const char *path, const char *asset_root, const rrt_builder *builder, rrt_camera *camera_out,
                         int mode_override, uint32_t *err_line) {
    if (!path || !builder) return rrt::fail(RRT_ERR_INVALID, "null argument");
    std::ifstream in(path, std::ios::binary);
    if (!in) return rrt::fail(RRT_ERR_IO, std::string("Failed to read scene file ") + path);
    std::stringstream ss;
    ss << in.rdbuf();
    return rrt::parse_text(ss.str(), asset_root ? asset_root : ".", builder, camera_out, mode_override, err_line);
}

}  // extern "C"
