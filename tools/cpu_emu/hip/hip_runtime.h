// Host stand-in for the few HIP names dev_trace.h uses, so the device functions of librrt can be compiled with g++
// and run one lane at a time under AddressSanitizer (tools/cpu_emu/emu.cpp).  Debugging aid only: never shipped.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct uchar4 { unsigned char x, y, z, w; };
struct ushort4 { unsigned short x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
bool __any(bool x);
unsigned long long __ballot(bool x);  // likewise
void emu_wave_sync();               // rendezvous of the lanes inside the current call
#define RRT_WAVE_SYNC() emu_wave_sync()
void emu_cull_stat(int visited, int total);
#define RRT_CULL_STAT(v, t) emu_cull_stat((int)(v), (int)(t))
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline uint32_t __builtin_amdgcn_readfirstlane(uint32_t x) { return x; }
static inline uint32_t atomicMin(uint32_t *p, uint32_t v) { uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
static inline uint32_t atomicMax(uint32_t *p, uint32_t v) { uint32_t o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
static inline uint32_t atomicOr(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }  // wave collective over the lanes still inside the current call: emu.cpp
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
static inline uint32_t __builtin_amdgcn_alignbit(uint32_t hi, uint32_t lo, uint32_t s) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (s & 31u)); }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
using std::min;
using std::max;
using std::isinf;
