// Shared helpers for libkoaf (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/koaf.h"

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef long long v4l __attribute__((ext_vector_type(4)));

void koaf_set_error(const char* fmt, ...);

#define KOAF_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            koaf_set_error(__VA_ARGS__);   \
            return KOAF_EINVAL;            \
        }                                  \
    } while (0)

static inline int koaf_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        koaf_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return KOAF_ELAUNCH;
    }
    return KOAF_OK;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
