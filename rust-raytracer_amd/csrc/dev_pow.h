// dev_pow.h — powf for the specular term (phong_material.rs:32, `max(v . r, 0).powf(shininess)`), evaluated in f64 and
// rounded to f32 once.
//
// Rust's f32::powf calls the platform libm; libms differ in the last ulp, and the parity contract cannot hang on which
// libm ran.  Both sides of the parity check therefore return the correctly rounded f32 power: the oracle through
// std::pow on doubles, this file through its own f64 exp(y * ln x).  Its relative error is below 1e-13, so the f32 it
// rounds to is the correctly rounded one unless the true value lies within 2e-6 of an ulp of a rounding boundary
// (tools/cpu_emu/pow_check.cpp counts the cases against std::pow on 20 M inputs).
//
// Why not ocml's pow: it keeps three dozen doubles live; inside k_trace_shade that pushed the mesh kernels 37 VGPRs over
// budget and every lit pixel wrote 224 bytes of scratch (1.8 GB per 4K frame, profiles/r02).  This one needs about ten.
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef RRT_HD
#define RRT_HD __host__ __device__ __forceinline__
#endif

namespace rrt {

// The coefficients of both polynomials.  gfx950's v_fma_f64 takes no literal operand, so every one of them has to sit in
// a register pair; written as literals they were hoisted out of the light loop into VGPRs and spilled from there.  On the
// device they are read from the constant segment instead: wave-uniform scalar loads into SGPR pairs, which v_fma_f64
// takes as an operand directly, and which cost no vector register at all.
#define RRT_POW_COEFFICIENTS                                                                                                \
    { /* ln: 1/25 ... 1/3 */ 1.0 / 25.0, 1.0 / 23.0, 1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0,   \
      1.0 / 9.0, 1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0,                                                                           \
      /* ln 2 split (fdlibm), 1 / ln 2 */ 6.93147180369123816490e-01, 1.90821492927058770002e-10, 1.44269504088896338700e+00, \
      0.0,                                                                                                                  \
      /* exp: 1/13! ... 1/3! */ 1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0,    \
      1.0 / 40320.0, 1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.0, 0.0, 0.0, 0.0, 0.0 }
#ifdef __HIP_DEVICE_COMPILE__
static __constant__ double pow_coefficients[32] = RRT_POW_COEFFICIENTS;
#else
static const double pow_coefficients[32] = RRT_POW_COEFFICIENTS;
#endif
enum { POW_LN = 0, POW_LN2_HI = 12, POW_LN2_LO = 13, POW_INV_LN2 = 14, POW_EXP = 16 };

RRT_HD double pow_bits_to_double(uint64_t u) {
    double d;
    memcpy(&d, &u, sizeof(d));
    return d;
}
RRT_HD uint64_t pow_double_to_bits(double d) {
    uint64_t u;
    memcpy(&u, &d, sizeof(u));
    return u;
}

// ln x for finite x > 0: x = m * 2^e with m in [0.75, 1.5), ln m = 2 atanh((m - 1) / (m + 1))
RRT_HD double pow_ln(double x) {
    uint64_t u = pow_double_to_bits(x);
    int e = (int)((u >> 52) & 0x7FF) - 1023;
    uint64_t mant = u & 0x000FFFFFFFFFFFFFull;
    // (a float converted to double is never subnormal as a double, so the exponent field is usable as it is)
    double m = pow_bits_to_double(mant | 0x3FF0000000000000ull);  // [1, 2)
    if (m >= 1.5) {
        m *= 0.5;
        e += 1;
    }
    const double s = (m - 1.0) / (m + 1.0);  // |s| <= 0.2; m - 1 is exact
    const double z = s * s;
    // 1 + z/3 + z^2/5 + ... + z^12/25   (z <= 0.04: the next term is below 2e-19)
    const double *c = pow_coefficients;
    double p = c[POW_LN];
#pragma unroll
    for (int i = 1; i < 12; i++) p = __builtin_fma(p, z, c[POW_LN + i]);
    p = __builtin_fma(p, z, 1.0);
    const double ln2_hi = c[POW_LN2_HI], ln2_lo = c[POW_LN2_LO];
    const double de = (double)e;
    return __builtin_fma(de, ln2_hi, __builtin_fma(2.0 * s, p, de * ln2_lo));
}

// e^t for t in [-130, 100]
RRT_HD double pow_exp(double t) {
    const double *c = pow_coefficients;
    const double inv_ln2 = c[POW_INV_LN2], ln2_hi = c[POW_LN2_HI], ln2_lo = c[POW_LN2_LO];
    const double kd = __builtin_rint(t * inv_ln2);
    const double r = __builtin_fma(-kd, ln2_lo, __builtin_fma(-kd, ln2_hi, t));  // |r| <= 0.347
    // Taylor to r^13 / 13!  (next term below 4e-18)
    double p = c[POW_EXP];
#pragma unroll
    for (int i = 1; i < 11; i++) p = __builtin_fma(p, r, c[POW_EXP + i]);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const int k = (int)kd;  // -188 .. 145: 2^k is a normal double
    return p * pow_bits_to_double((uint64_t)(k + 1023) << 52);
}

// x^y as f32 for x >= 0 (the specular base is max(dot, 0)), any y; IEEE pow's special cases for that domain
RRT_HD float pow_f32(float x, float y) {
    if (y == 0.0f) return 1.0f;                    // pow(x, +-0) = 1, also for NaN x
    if (x == 1.0f) return 1.0f;                    // pow(1, y) = 1, also for NaN y
    if (x != x || y != y) return x + y;            // NaN
    const float inf = __builtin_huge_valf();
    if (x == 0.0f) return y > 0.0f ? 0.0f : inf;   // (x is +0 here: fmaxf(dot, 0))
    if (x == inf) return y > 0.0f ? inf : 0.0f;
    if (y == inf) return x < 1.0f ? 0.0f : inf;
    if (y == -inf) return x < 1.0f ? inf : 0.0f;
    double t = (double)y * pow_ln((double)x);
    // beyond these the f32 result is 0 or inf anyway (e^-104 is half the smallest f32 subnormal, e^89 overflows)
    t = t < -130.0 ? -130.0 : (t > 100.0 ? 100.0 : t);
    return (float)pow_exp(t);
}

}  // namespace rrt
