// (GPU box; standalone) What the matrix pipe delivers under SUSTAINED load with the k-loop shape of the 3x3 kernels -- per wave and step
// 16 ds_read_b128 of fragments (the next step's, under this step's MFMAs) and the three-term fp16 x 2 product of a 64 x 64 x 32 wave
// tile -- issued as 24 v_mfma_f32_32x32x16_f16 or as 48 v_mfma_f32_16x16x32_f16 (the same FLOPs, the same LDS bytes, the same
// issue cycles), on random fp16 data, 256 blocks x 8 waves (two waves per SIMD), about a second per run.  DESIGN 3.10: the step's
// matrix-heavy kernels are clock-managed (1.8-1.9 GHz); this measures whether the 16x16x32 shape is given a higher clock for the same work.
//     hipcc --offload-arch=gfx950 -O3 scripts/mfma_shapes.hip -o /tmp/mfma_shapes && /tmp/mfma_shapes [seconds per run = 1.0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Frags { v4i a[8]; v4i b[8]; };      // 64 rows x 32 k x two planes of A, the same of B: sixteen 16-B reads per lane

template <bool READS>
__device__ __forceinline__ void load(Frags& f, const v4i* lds, int it, int lane, int w) {
    if constexpr (READS) {
        // lane-linear 16-B reads (conflict-free), a different 1-KiB line per read and iteration
        const v4i* p = lds + ((it & 3) * 1024 + w * 16) % 3072 + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) { f.a[i] = p[64 * i]; f.b[i] = p[64 * (i + 8)]; }
    }
}

// 32x32x16: a[(g * 2 + i) * 2 + q], b[(g * 2 + j) * 2 + q]; k-groups g, M tiles i, N tiles j, planes q (0 = hi, 1 = lo)
__device__ __forceinline__ void mma32(v16f (&acc)[2][2], const Frags& f) {
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, f.a[(g * 2 + i) * 2 + PA[t]]),
                                                                        __builtin_bit_cast(h16x8, f.b[(g * 2 + j) * 2 + PB[t]]), acc[i][j], 0, 0, 0);
}
// 16x16x32: a[i * 2 + q] (four 16-row blocks, all 32 k), b[j * 2 + q]
__device__ __forceinline__ void mma16(v4f (&acc)[4][4], const Frags& f) {
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, f.a[i * 2 + PA[t]]),
                                                                    __builtin_bit_cast(h16x8, f.b[j * 2 + PB[t]]), acc[i][j], 0, 0, 0);
}

template <int SHAPE, bool READS>
__global__ void __launch_bounds__(512) loop_kernel(const v4i* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ v4i lds[4096];          // 64 KiB
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[(blockIdx.x & 7) * 4096 + i];
    __syncthreads();
    Frags F0, F1;
    load<true>(F0, lds, 0, lane, w);
    load<true>(F1, lds, 1, lane, w);
    float s = 0.f;
    if constexpr (SHAPE == 32) {
        v16f acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll 1
        for (int it = 0; it < iters; it += 2) {
            load<READS>(F1, lds, it + 1, lane, w);
            __builtin_amdgcn_sched_barrier(0);
            mma32(acc, F0);
            __builtin_amdgcn_sched_barrier(0);
            load<READS>(F0, lds, it + 2, lane, w);
            __builtin_amdgcn_sched_barrier(0);
            mma32(acc, F1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    } else {
        v4f acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
#pragma unroll 1
        for (int it = 0; it < iters; it += 2) {
            load<READS>(F1, lds, it + 1, lane, w);
            __builtin_amdgcn_sched_barrier(0);
            mma16(acc, F0);
            __builtin_amdgcn_sched_barrier(0);
            load<READS>(F0, lds, it + 2, lane, w);
            __builtin_amdgcn_sched_barrier(0);
            mma16(acc, F1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int SHAPE, bool READS>
double run(const v4i* src, float* out, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((loop_kernel<SHAPE, READS>), dim3(256), dim3(512), 0, 0, src, out, 64);     // warm-up
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((loop_kernel<SHAPE, READS>), dim3(256), dim3(512), 0, 0, src, out, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 1.0;
    std::vector<unsigned short> h(8 * 4096 * 8);
    unsigned x = 12345u;
    for (auto& v : h) {                       // random fp16 in +-[0.5, 2): sign, exponent 14..15, random mantissa
        x = x * 1664525u + 1013904223u;
        v = (unsigned short)(((x >> 31) << 15) | ((14u + ((x >> 30) & 1u)) << 10) | ((x >> 12) & 0x3ffu));
    }
    v4i* src; float* out;
    CHECK(hipMalloc(&src, h.size() * 2)); CHECK(hipMalloc(&out, 256 * 512 * 4));
    CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const double flop_it = 256.0 * 8 * 24 * 32768.0;        // per iteration of the whole grid (either shape)
    // calibrate the iteration count on the first shape
    double ms = run<32, true>(src, out, 20000);
    int iters = (int)(20000 * secs * 1e3 / ms) & ~1;
    printf("iterations per run: %d (%.2f s of 24 MFMAs 32x32x16 + 16 ds_read_b128 per wave and iteration)\n", iters, secs);
    for (int rep = 0; rep < 2; ++rep) {
        double a = run<32, true>(src, out, iters), b = run<16, true>(src, out, iters);
        double c = run<32, false>(src, out, iters), d = run<16, false>(src, out, iters);
        printf("rep %d  with fragment reads: 32x32x16 %8.1f ms %7.1f TFLOP/s | 16x16x32 %8.1f ms %7.1f TFLOP/s  (%.3f x)   "
               "MFMAs only: 32x32x16 %7.1f TFLOP/s | 16x16x32 %7.1f TFLOP/s  (%.3f x)\n", rep, a, flop_it * iters / a / 1e9, b,
               flop_it * iters / b / 1e9, a / b, flop_it * iters / c / 1e9, flop_it * iters / d / 1e9, c / d);
        fflush(stdout);
    }
    return 0;
}
