// Host-compilation shim used ONLY by gen_golden.cpp (this container, never shipped to the GPU box).
// It lets the reference's pure-arithmetic __device__ functions (svd, compute_stress, bspline_weight)
// be compiled as ordinary host C++ so that their outputs can be recorded as golden vectors.
// Nothing here is an implementation of CUDA: the qualifiers are emptied and the three rounding
// intrinsics are given their documented IEEE meaning (round-to-nearest, no contraction).
#pragma once
#include <cmath>
#include <cstdio>
#include <algorithm>
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __restrict__
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __frsqrt_rn(float a) { return (float)(1.0 / std::sqrt((double)a)); }
using std::max;
using std::min;
using std::isnan;
