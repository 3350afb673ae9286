// Host stand-in used ONLY by tools/hostcheck (x86 build of mpm_device_math.hpp for quick numerical checks without a GPU).
#pragma once
#include <math.h>
#include <cstdint>
#include <cstring>
#define __device__
#define __forceinline__ inline
#define __global__
#define __restrict__
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / std::sqrt(x); }
static inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }
static inline float __builtin_amdgcn_logf(float x) { return std::log2(x); }
static inline float __builtin_amdgcn_exp2f(float x) { return std::exp2(x); }
static inline bool __all(bool p) { return p; }
static inline bool __any(bool p) { return p; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
