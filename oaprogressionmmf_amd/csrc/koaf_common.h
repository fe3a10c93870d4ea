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

// |x| as an unsigned integer: orders like the magnitude for finite values, +Inf above every finite value and every NaN above
// +Inf -- an integer maximum over these bits PROPAGATES non-finite values, which fmaxf (IEEE maxNum: a NaN operand is dropped)
// does not.  All "largest magnitude of a tensor" reductions run on these bits, so a NaN / Inf anywhere in an operand reaches
// its amax scalar, and the GEMM that scales by that amax turns its whole output into NaN (koaf_gemm.hip) instead of clamping
// the value away.
__device__ __forceinline__ unsigned koaf_absbits(float x) { return __float_as_uint(x) & 0x7fffffffu; }
__device__ __forceinline__ bool koaf_bits_finite(unsigned absbits) { return absbits < 0x7f800000u; }
__device__ __forceinline__ unsigned wave_max_u(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned w = (unsigned)__shfl_xor((int)v, o, 64); v = v > w ? v : w; }
    return v;
}

// Raise the device scalar *slot (a non-negative float, zeroed beforehand) to the block's maximum of the magnitudes `bits`
// (koaf_absbits): one atomic per BLOCK, and none at all once the slot already holds a value >= the block's (the slot only
// grows, so a stale read can cost a redundant atomic but never lose a maximum).  A NaN / Inf compares above every finite value
// and poisons the slot on purpose.  All threads of the block must call it.
__device__ __forceinline__ void block_amax_raise_bits(unsigned bits, float* slot) {
    __shared__ unsigned koaf_amax_red[16];
    bits = wave_max_u(bits);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) koaf_amax_red[w] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = koaf_amax_red[0];
        for (int i = 1; i < nw; ++i) m = m > koaf_amax_red[i] ? m : koaf_amax_red[i];
        if (m > __hip_atomic_load(reinterpret_cast<unsigned*>(slot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(reinterpret_cast<unsigned*>(slot), m);
    }
}
__device__ __forceinline__ void block_amax_raise(float v, float* slot) { block_amax_raise_bits(koaf_absbits(v), slot); }

// ---- numerics status words (koaf.h koaf_set_status_buffer): device uint32[4] registered by the host, or NULL -------------
//   [0] activation-operand elements / tiles that left the fp16 range of the fixed activation scale (clamped) or were not finite
//   [1] non-finite operand scales (a NaN / Inf in a weight or gradient tensor) and non-finite BatchNorm coefficients
uint32_t* koaf_status_ptr();
__device__ __forceinline__ void koaf_status_add(uint32_t* st, int slot, unsigned n) {
    if (st != nullptr && n != 0u) atomicAdd(&st[slot], n);
}
