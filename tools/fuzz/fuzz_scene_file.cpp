// Mutation fuzzer for the scene-file loader ("worth hardening + fuzzing").  Built with g++ under
// AddressSanitizer + UBSan from the host sources of librrt (no GPU, no HIP runtime); the builder is a counting sink.
// Every mutant must come back with RRT_OK or a clean error code; a ParseError must name a line of the text.
//   usage: fuzz_scene_file <asset_root> <iterations per seed> <scene files...>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "rrt.h"

namespace {
struct Sink {
    int handles = 0, materials = 0, textures = 0, objects = 0, lights = 0;
};
int b_set_mode(void *, int mode) { return mode == 0 || mode == 1 ? RRT_OK : RRT_ERR_INVALID; }
int b_add_texture(void *c, uint32_t, uint32_t, uint32_t, const uint8_t *) { return ((Sink *)c)->textures++; }
int b_add_material(void *c, const rrt_material_desc *) { return ((Sink *)c)->materials++; }
int b_object_define(void *c, const rrt_object_desc *) { return ((Sink *)c)->handles++; }
int b_object_transform(void *c, int h, const float *) { return h >= 0 && h < ((Sink *)c)->handles ? RRT_OK : RRT_ERR_INVALID; }
int b_add_object(void *c, int h) {
    if (h < 0 || h >= ((Sink *)c)->handles) return RRT_ERR_INVALID;
    return ((Sink *)c)->objects++;
}
int b_add_light(void *c, const rrt_light_desc *) { return ((Sink *)c)->lights++; }

uint64_t rng_state = 0x9E3779B97F4A7C15ull;
uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}
std::vector<std::string> split_lines(const std::string &t) {
    std::vector<std::string> out;
    std::stringstream ss(t);
    std::string l;
    while (std::getline(ss, l)) out.push_back(l);
    return out;
}
const char *kWeird[] = {"nan", "inf", "-inf", "-0.0", "1e39", "-1e39", "1e-46", "", "0", "0x10", "1.0.0", "++1", "Simple", "#", "\t", "  ", "999999999999",
                        "camera Camera", "object Sphere", "material", "White", "Black", "\xff\xfe", "1,5"};
std::string mutate(const std::string &seed) {
    std::vector<std::string> L = split_lines(seed);
    int edits = 1 + rnd() % 4;
    for (int e = 0; e < edits && !L.empty(); e++) {
        size_t i = rnd() % L.size();
        switch (rnd() % 10) {
            case 0: L.erase(L.begin() + i); break;
            case 1: L.insert(L.begin() + i, L[rnd() % L.size()]); break;
            case 2: std::swap(L[i], L[rnd() % L.size()]); break;
            case 3: {  // replace one word
                std::stringstream ss(L[i]);
                std::vector<std::string> w;
                std::string x;
                while (ss >> x) w.push_back(x);
                if (w.empty()) break;
                w[rnd() % w.size()] = kWeird[rnd() % (sizeof(kWeird) / sizeof(kWeird[0]))];
                size_t indent = L[i].find_first_not_of(" \t");
                std::string nl = L[i].substr(0, indent == std::string::npos ? 0 : indent);
                for (size_t k = 0; k < w.size(); k++) nl += (k ? " " : "") + w[k];
                L[i] = nl;
                break;
            }
            case 4: L[i] = std::string(rnd() % 9, ' ') + L[i].substr(std::min(L[i].size(), L[i].find_first_not_of(" \t") == std::string::npos ? 0 : L[i].find_first_not_of(" \t"))); break;
            case 5: L[i] += std::string(" ") + kWeird[rnd() % (sizeof(kWeird) / sizeof(kWeird[0]))]; break;
            case 6: L.resize(i); break;  // truncate
            case 7: L[i] = ""; break;
            case 8: if (!L[i].empty()) L[i][rnd() % L[i].size()] = (char)(rnd() % 256); break;
            default: L.insert(L.begin() + i, std::string(kWeird[rnd() % (sizeof(kWeird) / sizeof(kWeird[0]))])); break;
        }
    }
    std::string out;
    for (auto &l : L) out += l + (rnd() % 50 == 0 ? "\r\n" : "\n");
    if (rnd() % 8 == 0 && !out.empty()) out.pop_back();  // no trailing newline
    return out;
}
}  // namespace

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const char *root = argv[1];
    int iters = atoi(argv[2]);
    long ok = 0, parse = 0, io = 0, other = 0;
    for (int a = 3; a < argc; a++) {
        std::ifstream in(argv[a], std::ios::binary);
        std::stringstream ss;
        ss << in.rdbuf();
        std::string seed = ss.str();
        for (int it = 0; it <= iters; it++) {
            std::string text = it == 0 ? seed : mutate(seed);
            Sink sink;
            rrt_builder b = {&sink, b_set_mode, b_add_texture, b_add_material, b_object_define, b_object_transform, b_add_object, b_add_light};
            rrt_camera cam;
            uint32_t line = 0;
            int rc = rrt_parse_scene_text(text.c_str(), root, &b, &cam, -1, &line);
            size_t n_lines = split_lines(text).size();
            if (it == 0 && rc != RRT_OK) { fprintf(stderr, "seed %s does not parse: %d %s\n", argv[a], rc, rrt_last_error()); return 1; }
            if (rc == RRT_OK) ok++;
            else if (rc == RRT_ERR_PARSE) {
                parse++;
                if (line > n_lines + 1) { fprintf(stderr, "ParseError line %u beyond the text (%zu lines)\n%s\n", line, n_lines, text.c_str()); return 1; }
                if (!rrt_last_error() || !*rrt_last_error()) { fprintf(stderr, "ParseError without a message\n"); return 1; }
            } else if (rc == RRT_ERR_IO) io++;
            else if (rc == RRT_ERR_INVALID || rc == RRT_ERR_LIMIT) other++;
            else { fprintf(stderr, "unexpected status %d (%s) for\n%s\n", rc, rrt_last_error(), text.c_str()); return 1; }
        }
    }
    printf("fuzz ok: %ld parsed, %ld ParseError, %ld IO, %ld invalid/limit\n", ok, parse, io, other);
    return 0;
}
