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

// Raise the device scalar *slot (a non-negative float, zeroed beforehand) to the block's maximum of v: one atomic per BLOCK,
// and none at all once the slot already holds a value >= the block's (the slot only grows, so a stale read can cost a
// redundant atomic but never lose a maximum).  Float bits of non-negative values order like unsigned integers; a NaN
// compares above every finite value and poisons the slot on purpose.  All threads of the block must call it.
__device__ __forceinline__ void block_amax_raise(float v, float* slot) {
    __shared__ float koaf_amax_red[16];
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) koaf_amax_red[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = koaf_amax_red[0];
        for (int i = 1; i < nw; ++i) m = fmaxf(m, koaf_amax_red[i]);
        const unsigned bits = __float_as_uint(m);
        if (bits > __hip_atomic_load(reinterpret_cast<unsigned*>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(reinterpret_cast<unsigned*>(slot), bits);
    }
}
