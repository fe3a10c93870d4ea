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

// ---- bf16 activation storage (koaf.h KoafGemm.act16 / the act16 argument of the element-wise entry points): forward
// activations may live in HBM as bf16; every kernel widens them on load (exact) and computes in fp32; producers round to
// nearest even on store.  The pointers stay typed float*; offsets count elements.
typedef __bf16 koaf_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4f widen_bf16x4(unsigned lo, unsigned hi) {
    return (v4f){__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
}
__device__ __forceinline__ uint2 round_bf16x4(v4f v) {
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    const unsigned a = __builtin_bit_cast(unsigned, __builtin_convertvector((v2f_){v[0], v[1]}, koaf_bf16x2));
    const unsigned b = __builtin_bit_cast(unsigned, __builtin_convertvector((v2f_){v[2], v[3]}, koaf_bf16x2));
    return make_uint2(a, b);
}
template <bool H>
__device__ __forceinline__ v4f load4_nt(const float* p, int64_t off) {          // streamed once (epilogue operands), widened
    if constexpr (!H) return __builtin_nontemporal_load((const v4f*)(p + off));
    else {
        const unsigned short* q = reinterpret_cast<const unsigned short*>(p) + off;
        const unsigned lo = __builtin_nontemporal_load((const unsigned*)q), hi = __builtin_nontemporal_load((const unsigned*)q + 1);
        return widen_bf16x4(lo, hi);
    }
}
template <bool H>
__device__ __forceinline__ void store4_nt(float* p, int64_t off, v4f v) {
    if constexpr (!H) __builtin_nontemporal_store(v, (v4f*)(p + off));
    else {
        const uint2 u = round_bf16x4(v);
        unsigned* q = (unsigned*)(reinterpret_cast<unsigned short*>(p) + off);
        __builtin_nontemporal_store(u.x, q);
        __builtin_nontemporal_store(u.y, q + 1);
    }
}

template <bool H>
__device__ __forceinline__ v4f load4(const float* p, int64_t off) {             // cached load, widened
    if constexpr (!H) return *(const v4f*)(p + off);
    else {
        const uint2 u = *(const uint2*)(reinterpret_cast<const unsigned short*>(p) + off);
        return widen_bf16x4(u.x, u.y);
    }
}
template <bool H>
__device__ __forceinline__ void store4(float* p, int64_t off, v4f v) {
    if constexpr (!H) *(v4f*)(p + off) = v;
    else *(uint2*)(reinterpret_cast<unsigned short*>(p) + off) = round_bf16x4(v);
}
