// Host check of rrt::pow_f32 (dev_pow.h, the device's specular power) against the oracle's definition, the f64 std::pow
// rounded to f32 once: random (base, shininess) pairs over the ranges materials use plus the special cases.
//   g++ -O2 -std=c++17 -ffp-contract=off -I tools/cpu_emu -I rust-raytracer_amd/csrc tools/cpu_emu/pow_check.cpp -o pow_check && ./pow_check [pairs]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#define __host__
#define __device__
#define __forceinline__ inline
#include "dev_pow.h"

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float ref(float x, float y) { return (float)std::pow((double)x, (double)y); }

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 20000000L;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> u01(0.0, 1.0);
    long bad = 0, off_by_more = 0;
    const float shin[] = {0.5f, 1.0f, 2.0f, 10.0f, 50.0f, 100.0f, 1000.0f, 1e6f, 1e8f, 1e9f, -2.0f, 3.7f};
    for (long i = 0; i < n; i++) {
        float y = shin[i % 12];
        if (i % 5 == 0) y = (float)(u01(rng) * 2000.0);
        double r = u01(rng);
        float x;
        switch (i % 4) {  // anywhere in [0, 1]; close to 1 (where large exponents still give something); just above 1; tiny
            case 0: x = (float)r; break;
            case 1: x = (float)(1.0 - r * std::pow(10.0, -8.0 * u01(rng))); break;
            case 2: x = (float)(1.0 + r * 1e-6); break;
            default: x = (float)(r * std::pow(10.0, -30.0 * u01(rng))); break;
        }
        const float a = rrt::pow_f32(x, y), b = ref(x, y);
        if (bits(a) != bits(b) && !(a != a && b != b)) {
            bad++;
            const double d = std::fabs((double)a - (double)b), ulp = std::fabs((double)std::nextafterf(b, INFINITY) - (double)b);
            if (d > 1.01 * ulp) {
                off_by_more++;
                if (off_by_more < 5) printf("pow(%.9g, %.9g): %.9g vs %.9g\n", x, y, a, b);
            }
        }
    }
    const float inf = INFINITY, nan = NAN;
    const float sx[] = {0.0f, 1.0f, 0.5f, 1.0000001f, inf, nan, 1e-45f, 3.4e38f}, sy[] = {0.0f, -0.0f, 1.0f, -1.0f, 100.0f, 1e9f, inf, -inf, nan, 0.5f};
    long special_bad = 0;
    for (float x : sx)
        for (float y : sy) {
            const float a = rrt::pow_f32(x, y), b = ref(x, y);
            if (bits(a) != bits(b) && !(a != a && b != b)) {
                special_bad++;
                printf("special pow(%g, %g): %g vs %g\n", x, y, a, b);
            }
        }
    printf("pairs %ld  differ-in-last-bit %ld (%.2e)  off-by-more %ld  special-cases-wrong %ld\n", n, bad, (double)bad / (double)n, off_by_more, special_bad);
    return (off_by_more || special_bad || bad > n / 20000) ? 1 : 0;
}
