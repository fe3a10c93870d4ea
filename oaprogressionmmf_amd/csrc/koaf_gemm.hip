// koaf_gemm.hip -- the one MFMA GEMM under every dense contraction of the koafusion train step:
// implicit-GEMM conv forward / dgrad / wgrad (NHWC), nn.Linear forward / dgrad / wgrad and the
// attention contractions.  fp32 in, fp32 out, fp32 accumulate.
//
// Arithmetic: gfx950 has no TF32-class matrix mode and its fp32 MFMA runs at 1/16 of the bf16 rate, so the fp32 x fp32
// products are formed on the 16-bit matrix pipe from exact pieces of the operands, fp32 accumulate.  Two schemes
// (KoafGemm.fmt), both at fp32 rounding level against float64 (scripts/gemm_accuracy.py, tests/test_kernels_gpu.py):
//   fmt 0 "bf16 x 3": every operand value is cut by truncation into three bf16 pieces hi + mid + lo that together hold
//       all 24 significand bits; of the nine piece products the six of relative weight >= 2^-16 are issued as
//       v_mfma_f32_32x32x16_bf16 (each exact in fp32); the three dropped ones are < 2^-21 of the product.  Works for any
//       fp32 operand (bf16 has fp32's exponent range): linear layers, attention, anything without scale information.
//       Six 8-pass MFMAs replace eight 16-pass fp32 MFMAs per 16 k: matrix-pipe bound 2500 / 6 = 417 TFLOP/s.
//   fmt 1 "fp16 x 2": the operand is multiplied by a power of two that puts its largest magnitude near 2^14 (the
//       producer of the tensor leaves max |x| in device memory: KoafOperand.amax; activations behind BatchNorm use a
//       fixed factor) and cut into hi = fp16(x'), lo = fp16(x' - hi), round-to-nearest: x' = hi + lo to 2^-24 relative
//       (lo is signed: 22 explicit bits + sign) down to |x'| = 2^-2 and to 2^-25 ABSOLUTE below (fp16 subnormals, which
//       the MFMA does not flush), i.e. <= 2^-40 of the tensor's largest magnitude.  Three products hi*hi, hi*lo, lo*hi on
//       v_mfma_f32_32x32x16_f16 (11 x 11 bits: exact in fp32); the dropped lo*lo is <= 2^-24 of the product.  Half the
//       matrix instructions of fmt 0 for the same accuracy -- this is what the convolutions (97 % of the FLOPs) run:
//       the bf16 scheme sits at the chip's power limit (the clock falls under six MFMAs per product), so fewer matrix
//       instructions per product is the lever.  Bound 2500 / 3 = 833 TFLOP/s.
//   Inf operands become NaN (inf - inf in the split); NaN stays NaN.
//
// Block = 256 threads = 4 waves (2x2), block tile BM x BN x 32, wave tile (BM/2) x (BN/2) built from 32x32 MFMA
// tiles; 2 blocks per CU.  Operand tiles are staged global -> registers (fused BN+ReLU prologue, zero fill) ->
// split -> LDS; the next tile's global loads are in flight under the current tile's MFMAs.  LDS holds three
// (fmt 1: two) packed 16-bit plane images per operand (see plane_dwords()):
//   K-contiguous operand ("KC"): plane[row][32 k + 8 pad] -- ds_write_b64, fragments by ds_read_b128
//       (80-B rows: the 16 lanes of a b128 group hit 16 distinct 4-bank slots).
//   K-major operand ("KM"):      plane[k][ROWS + 32 pad]  -- ds_write_b64 of 4 rows, fragments by the transposing
//       ds_read_b64_tr_b16 (k-row stride = 16 mod 64 dwords: conflict-free).
// Both present the same k order to the MFMA (lane (r, h), element e: k = 16g + 8h + e), so any pairing of KC / KM
// operands works.  Accumulators live in VGPRs (built with -mllvm -amdgpu-mfma-vgpr-form, see the Makefile).
//   Pre-split operand ("PS", conv weights, fmt 1): the two planes are cut ONCE per optimizer step by koaf_wplanes_build into
//       fp16 plane images [plane][row][K] in HBM; the kernel moves them global -> LDS with global_load_lds_dwordx4 (no
//       VGPR staging, no split arithmetic in the k-loop) into a linear [row][32 k] image whose 16-B chunks are
//       XOR-swizzled (chunk ^ (row / 4 % 4), applied to the per-lane SOURCE address and to the ds_read_b128 address):
//       LDS-DMA writes are lane-linear, so rows cannot be padded, and the swizzle keeps the fragment reads conflict-free.
//       Double-buffered: the DMA of k-tile t+1 lands while tile t is multiplied.
#include "koaf_common.h"
#include <stdlib.h>
#include <type_traits>

// In-kernel phase stamps (diagnostic builds only: make STAMPS=1 -> libkoaf_stamps.so, scripts/stamps_*.py): thread 0 of every
// block adds the 100 MHz real-time counter differences between its phase boundaries to a device table.
#ifdef KOAF_STAMPS
// (64 replicas of the table, indexed by block id: the adds of ~10^5 tiles per launch must not queue on eight addresses; the
// stamps themselves are wave-uniform s_memrealtime reads kept in scalar registers, consumed only at the end of the tile)
__device__ unsigned long long koaf_stamp_tab[64][8];
#define KOAF_STAMP_DECL unsigned long long kst_[6] = {0, 0, 0, 0, 0, 0}
#define KOAF_STAMP(i) do { kst_[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define KOAF_STAMP_ADD(slot, a, b) do { if (threadIdx.x == 0 && kst_[b] >= kst_[a]) atomicAdd(&koaf_stamp_tab[blockIdx.x & 63][slot], kst_[b] - kst_[a]); } while (0)
#define KOAF_STAMP_ACC(slot, v) do { if (threadIdx.x == 0) atomicAdd(&koaf_stamp_tab[blockIdx.x & 63][slot], (unsigned long long)(v)); } while (0)
#define KOAF_STAMP_NOW() __builtin_amdgcn_s_memrealtime()
#else
#define KOAF_STAMP_NOW() 0ull
#define KOAF_STAMP_DECL
#define KOAF_STAMP(i)
#define KOAF_STAMP_ADD(slot, a, b)
#define KOAF_STAMP_ACC(slot, v)
#endif

namespace {

constexpr int BK = 32;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int v2i __attribute__((ext_vector_type(2)));

// The split runs once per element in the loader and LDS holds three packed-bf16 plane images
// per operand:  KC operand  plane[row][32 k + 8 pad]   (80-B rows: ds_read_b128 fragments, conflict-free)
//               KM operand  plane[k][ROWS + 32 pad]    (ds_write_b64 of 4 rows, fragments by the transposing
//                                                       ds_read_b64_tr_b16; k-row stride = 16 (mod 64) dwords)
// Lane (r, h) of a 32x32x16 MFMA holds k = 16g + 8h + e (e = 0..7) of its row in both images.
__host__ __device__ constexpr int plane_dwords(int rows, bool kc) { return kc ? rows * 20 : 32 * (rows / 2 + 16); }

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// four fp32 values -> three (hi, mid, lo) pairs of dwords holding 4 packed bf16 each
__device__ __forceinline__ void split3v(const v4f x, unsigned out[3][2]) {
    float r1[4], r2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r1[e] = x[e] - __uint_as_float(__float_as_uint(x[e]) & 0xffff0000u);
        r2[e] = r1[e] - __uint_as_float(__float_as_uint(r1[e]) & 0xffff0000u);
    }
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        out[0][d] = __builtin_amdgcn_perm(__float_as_uint(x[2 * d + 1]), __float_as_uint(x[2 * d]), 0x07060302u);
        out[1][d] = __builtin_amdgcn_perm(__float_as_uint(r1[2 * d + 1]), __float_as_uint(r1[2 * d]), 0x07060302u);
        out[2][d] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * d + 1]), __float_as_uint(r2[2 * d]), 0x07060302u);
    }
}

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// power of two that brings a tensor whose largest magnitude is amax to [2^14, 2^15) (1 for amax == 0); koaf_wplanes_build
// and the GEMM kernel both derive an operand's scale from the same device scalar with this function
__device__ __forceinline__ float scale_of_amax(float amax) {
    if (!(amax > 0.f)) return 1.f;
    const int e = min(max(__builtin_amdgcn_frexp_expf(amax), -100), 100);   // amax = m * 2^e, m in [0.5, 1)
    return __builtin_ldexpf(1.f, 15 - e);
}
__device__ __forceinline__ float operand_scale(const KoafOperand& o) {
    return o.amax ? scale_of_amax(*o.amax) : (o.fscale != 0.f ? o.fscale : 1.f);
}

// four fp32 values x' (already multiplied by the operand's scale and clamped to the fp16 range by the loader's finish()) ->
// (hi, lo) pairs of dwords holding 4 packed fp16 each: hi = fp16(x'), lo = fp16(x' - hi), both round-to-nearest, so
// x' = hi + lo to 2^-24 relative (lo carries a sign) down to |x'| = 2^-2 and to 2^-25 absolute below that (fp16 subnormal
// spacing 2^-24).  The residual is taken from the PACKED hi, so one v_cvt_pk_f16_f32 serves storage and residual.
__device__ __forceinline__ void split2h(const v4f x, unsigned out[2][2]) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float a = x[2 * d], b = x[2 * d + 1];
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector((v2f){a, b}, h16x2));
        out[0][d] = hb;
        // residuals x - float(hi) in ONE instruction each (v_fma_mix_f32 reads the fp16 half of hb directly: the exact difference,
        // rounded once -- the bits of v_cvt_f32_f16 + v_sub_f32, which hipcc emits for the C++ form, at half the vector issue)
        float ra, rb;
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hb), "v"(a));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hb), "v"(b));
        out[1][d] = __builtin_bit_cast(unsigned, __builtin_convertvector((v2f){ra, rb}, h16x2));
    }
}

// ---- bf16 ACTIVATION STORAGE (KoafGemm.act16) ------------------------------------------------------------------------------
// The forward activations of a trunk (conv outputs, block outputs) may live in HBM as bf16 instead of fp32: half the bytes on
// every HBM-bound call.  Arithmetic is unchanged: a loader widens the bf16 values to fp32 (exact), applies its transform and
// cuts the fp32 result into the same two fp16 pieces; accumulation, statistics and all gradients stay fp32; only the store of a
// forward output rounds (to nearest even).  Raw 16-bit loads are kept as bits in the loader slot and widened in finish(), so
// that the loads stay in flight under the MFMAs exactly like the fp32 ones.
// 4 consecutive elements at element offset `off` of a tensor stored as fp32 (H = false) or bf16 (H = true: the pointer is typed
// float* all the same); the 16-bit form returns the raw bits in lanes 0 / 1 (widen_bf16x4 later)
template <bool H>
__device__ __forceinline__ v4f load4_raw(const float* p, int64_t off) {
    if constexpr (!H) return *(const v4f*)(p + off);
    else {
        const uint2 u = *(const uint2*)(reinterpret_cast<const unsigned short*>(p) + off);
        return (v4f){__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f};
    }
}
// operand access modes (compile-time: the loaders are straight-line code, so hipcc can schedule their
// address arithmetic into the shadows of the MFMAs)
enum { M_KC = 0,     // K-contiguous rows, dense
       M_KC_G1 = 1,  // K-contiguous, conv forward gather (NHWC source)
       M_KC_G2 = 2,  // K-contiguous, transposed-conv (dgrad) gather
       M_KM = 3,     // K-major, dense
       M_KM_G1 = 4,  // K-major, conv gather on the k index (wgrad activations)
       M_KM_G3 = 5,  // K-major, tapped weights (dgrad)
       M_PS = 6,     // pre-split fp16 plane images, K-contiguous rows, optionally tapped (weights: forward and dgrad)
       M_PA1 = 7,    // pre-split fp16 plane images of an NHWC activation, conv forward gather (A operand)
       M_PA2 = 8,    // the same, transposed-conv (dgrad) gather
       M_PH = 9,     // the same images, 3x3 / stride 1 / pad 1: the tile's pixel rows + halo stay in LDS for all nine taps
       M_PK = 10,    // activation plane images read K-major (weight gradient: k = pixel, rows = channels), dense
       M_PKG = 11,   // the same with the conv gather on the k index and the filter tap in the column (wgrad activations)
       M_PT = 12,    // activation plane images, 3x3 / stride 1 / pad 1, 2-D pixel tiles (8 x 16) with a zero-filled halo in LDS (64 channels
                     // at a time: one filter tap x 64 channels per barrier)
       M_KS = 13     // K-contiguous dense rows (as M_KC), STREAMED: every wave loads, transforms and splits its OWN 32 rows, several
                     // k-tiles ahead in registers (StreamA) -- the 1x1 / stride-1 convolutions and their data gradients
};
__host__ __device__ constexpr bool mode_is_kc(int m) { return m < 3 || m == M_KS; }
__host__ __device__ constexpr bool mode_is_pa(int m) { return m == M_PA1 || m == M_PA2; }
// halo kernel (M_PH): widest image row kept in LDS (BM + 2 W + 2 pixels of 32 channels, two buffers) and the number of
// weight-tile stages, chosen per column-tile width so that everything fits 160 KiB
// Two shapes: 256 pixel rows / 8 waves / one block per CU with the halo double-buffered across channel chunks, and 128 rows /
// 4 waves with ONE halo buffer in under 80 KiB, so that two blocks share a CU and one's prologue, chunk switch and epilogue
// run under the other's MFMAs (the shallow-K layers: 64 channels = two chunks, where those phases outweigh the k-loop).
__host__ __device__ constexpr int halo_max_w(int bn, int bm = 256) { return bm == 256 ? (bn == 64 ? 96 : 64) : (bn == 64 ? 96 : 48); }
__host__ __device__ constexpr int halo_b_stages(int bn, int bm = 256) { return bm == 256 ? 3 : (bn == 64 ? 4 : 3); }

// TF = transform on load (KoafOperand.tf): 0 none; 1 relu(sc[c] * x + sh[c]) -- the producer's BatchNorm + ReLU; 2 the
// BatchNorm-BACKWARD apply dc = sc[c] * dz + sh[c] - sc2[c] * c_raw of TWO source tensors (x = dz at ptr, c_raw at ptr2, same
// layout): the gradient w.r.t. a conv output is formed in the loaders of the dgrad / wgrad GEMMs that consume it and never
// written to HBM.  (TF 2 needs the vector path.)
// F16: the operand feeds the fp16 scheme: finish() also multiplies by the operand's scale `fsc` (folded into the transform
// coefficients where there is a transform) and clamps to the fp16 range (relu and clamp are one v_med3 for TF 1).
// S16 / S2_16: the source tensor at ptr / ptr2 is stored as bf16 (activation storage mode; vector path only)
template <int ROWS, int MODE, int TF, bool VEC, bool F16, bool S16 = false, bool S2_16 = false>
struct TileLoader {
    static_assert(!(S16 || S2_16) || VEC, "bf16 sources need the vector path");
    static constexpr int NU = ROWS / 32;
    static constexpr bool KC = mode_is_kc(MODE);
    static constexpr int NU2 = (TF == 2 || TF == 3) ? NU : 1;
    static_assert(TF != 3 || (MODE == M_KC && VEC && F16), "the bottleneck-tail prologue (tf 3) serves the dense K-contiguous fp16-scheme loader");
    // registers of one k-tile in flight
    struct Slot {
        v4f r[NU];
        v4f r2[NU2];   // TF 2: the second source
        unsigned vm;   // validity bits: VEC 1 bit / unit, else 4 bits / unit
        v4f ts4, th4, tk4;  // transform coefficients of the tile (KC operands: they depend on k)
        v4f tq4;            // TF 3 with the identity's own affine (a downsample branch): its shift (tk4 = its scale)
    };
    Slot sa, sb;
    bool tail2 = false;      // TF 3: the identity is sc2[c] * x2 + sh2[c] (KoafOperand.sc2 / sh2 given)
    v4f kts4, kth4, ktk4;    // transform coefficients of this thread's columns (KM operands: fixed)
    float fsc;               // F16: operand scale (a power of two)
    unsigned satmax;         // F16, TF 1: packed maximum of the fp16 hi pieces stored so far (0x7bff = clamped at 65504)
    // KC state (ext-vector values, not arrays: arrays of per-unit state were left in scratch by hipcc and
    // every scratch reload drained the in-flight global loads through the in-order vmcnt)
    v4l base;          // element offset of each unit's row / image from the operand pointer
    v4i iy0, ix0;
    unsigned rvm;      // row-valid bits
    v4i toff;          // conv gathers: per-image element offset of the CURRENT filter tap (recomputed per tap,
    unsigned tvm;      //   not per k-step: a tap spans C/32 k-steps) and its validity bits
    int tap_cur;
    // KM state
    int col, cc, kh_, kw_;
    unsigned cvm;      // column-valid bits
    // Running decomposition of k.  issue() is called for k0 = kbeg, kbeg + BK, ... in order, so the filter tap of a
    // k-tile (u_*: wave-uniform) and the source pixel of every k row of a gathered K-major tile (g_*: per unit) are
    // carried from one call to the next by adds and single carries instead of being re-derived by integer divisions
    // and 64-bit multiplies -- those were a third of the vector instructions of the weight-gradient kernels.
    int u_tap, u_coff, u_kh, u_kw;
    v4i gsx, gsy;
    v4l goff;
    int g_cs, g_bs, g_pws, g_phs, g_sxlim, g_sylim;
    int64_t g_d0, g_d1, g_d2;

    __device__ __forceinline__ void init(const KoafOperand& op, int r0, int R, int z1, float scale) {
        const int t = threadIdx.x;
        fsc = scale;
        satmax = 0u;
        sa.vm = sb.vm = 0;
        sa.ts4 = sa.th4 = sb.ts4 = sb.th4 = kts4 = kth4 = sa.tk4 = sb.tk4 = ktk4 = sa.tq4 = sb.tq4 = (v4f){0.f, 0.f, 0.f, 0.f};
        if constexpr (TF == 3) tail2 = op.sc2 != nullptr;
        rvm = cvm = 0;
        base = (v4l){0, 0, 0, 0};
        iy0 = ix0 = toff = (v4i){0, 0, 0, 0};
        tvm = 0;
        tap_cur = -1;
        col = cc = kh_ = kw_ = 0;
        if constexpr (KC) {
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int row = r0 + (t >> 3) + 32 * i;
                rvm |= (row < R ? 1u : 0u) << i;
                if constexpr (MODE == M_KC) {
                    base[i] = (int64_t)row * op.ld + 4 * (t & 7);
                } else {
                    const int ppi = op.PH * op.PW;
                    const int n = row / ppi;
                    const int rem = row - n * ppi;
                    const int py = rem / op.PW;
                    const int px = rem - py * op.PW;
                    base[i] = (int64_t)n * op.H * op.W * op.CS + 4 * (t & 7);
                    if constexpr (MODE == M_KC_G1) {
                        iy0[i] = py * op.stride - op.pad;
                        ix0[i] = px * op.stride - op.pad_w;
                    } else {
                        iy0[i] = py + op.pad;
                        ix0[i] = px + op.pad_w;
                    }
                }
            }
        } else {
            constexpr int CV = ROWS / 4;
            col = r0 + 4 * (t % CV);
#pragma unroll
            for (int j = 0; j < 4; ++j) cvm |= ((col + j) < R ? 1u : 0u) << j;
            cc = col;
            if constexpr (MODE == M_KM_G1) {
                // this thread's own filter tap (its 4 columns lie in one tap: C % 4 == 0), so a tile may span taps
                const int tap = col / op.C;
                cc = col - tap * op.C;
                kh_ = tap / op.KW;
                kw_ = tap - kh_ * op.KW;
            }
            if constexpr (TF != 0) {
                const float* sc = op.sc + z1 * op.tf_bs;
                const float* sh = op.sh + z1 * op.tf_bs;
                if (VEC) {
                    if (cvm & 1u) {
                        kts4 = *(const v4f*)(sc + cc);
                        kth4 = *(const v4f*)(sh + cc);
                        if constexpr (TF == 2) ktk4 = *(const v4f*)(op.sc2 + z1 * op.tf_bs + cc);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((cvm >> j) & 1u) { kts4[j] = sc[cc + j]; kth4[j] = sh[cc + j]; }
                }
                if constexpr (F16) { kts4 *= fsc; kth4 *= fsc; ktk4 *= fsc; }
            }
        }
    }

    // Position the running k decomposition at k0 (the only place that divides); call once before the first issue().
    __device__ __forceinline__ void seek(const KoafOperand& op, int k0) {
        u_tap = u_coff = u_kh = u_kw = 0;
        gsx = gsy = (v4i){0, 0, 0, 0};
        goff = (v4l){0, 0, 0, 0};
        g_cs = g_bs = g_pws = g_phs = g_sxlim = g_sylim = 0;
        g_d0 = g_d1 = g_d2 = 0;
        if constexpr (MODE == M_KC_G1 || MODE == M_KC_G2 || MODE == M_KM_G3) {
            u_tap = k0 / op.C;
            u_coff = k0 - u_tap * op.C;
            u_kh = u_tap / op.KW;
            u_kw = u_tap - u_kh * op.KW;
        }
        if constexpr (MODE == M_KM_G1) {
            constexpr int CV = ROWS / 4;
            constexpr int RP = 256 / CV;
            const int ppi = op.PH * op.PW;
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int k = k0 + (int)threadIdx.x / CV + RP * i;
                const int n = k / ppi;
                const int rem = k - n * ppi;
                const int py = rem / op.PW;
                const int px = rem - py * op.PW;
                gsy[i] = py * op.stride - op.pad + kh_;
                gsx[i] = px * op.stride - op.pad_w + kw_;
                goff[i] = ((int64_t)(n * op.H + gsy[i]) * op.W + gsx[i]) * op.CS + cc;
            }
            // one k-step = BK rows further: BK = a * ppi + b * PW + c  (c < PW, b < PH: single carries below)
            const int a = BK / ppi, r = BK - a * ppi, b = r / op.PW, c = r - b * op.PW;
            const int64_t wcs = (int64_t)op.W * op.CS, hwcs = (int64_t)op.H * wcs;
            g_cs = c * op.stride;
            g_bs = b * op.stride;
            g_pws = op.PW * op.stride;
            g_phs = op.PH * op.stride;
            g_sxlim = g_pws - op.pad_w + kw_;      // px == PW  <=>  sx == sxlim
            g_sylim = g_phs - op.pad + kh_;
            g_d0 = a * hwcs + g_bs * wcs + (int64_t)g_cs * op.CS;
            g_d1 = (int64_t)op.stride * wcs - (int64_t)g_pws * op.CS;    // px wraps: next pixel row
            g_d2 = hwcs - g_phs * wcs;                                   // py wraps: next image
        }
    }

    // Issue the global loads of the k-tile [k0, k0+32): nothing here consumes a loaded value, so the
    // s_waitcnt lands in finish(), after the MFMAs of the tile currently in LDS.
    __device__ __forceinline__ void issue(Slot& s, const KoafOperand& op, const float* ptr, int k0, int kend, int z1) {
        const int t = threadIdx.x;
        [[maybe_unused]] const float* ptr2 = (TF == 2 || TF == 3) ? op.ptr2 + (ptr - op.ptr) : nullptr;   // (same batch offset; batches of one where the storage types differ)
        s.vm = 0;
        if constexpr (KC) {
            const int kk = k0 + 4 * (t & 7);
            const bool kok = kk < kend;
            int ch = kk, coff = k0;
            if constexpr (MODE != M_KC) {
                // a 32-wide k chunk lies inside one filter tap (C % 32 == 0); (tap, coff, kh, kw) of this k0 are carried
                const int tap = u_tap;
                coff = u_coff;
                ch = coff + 4 * (t & 7);
                const int kh = u_kh, kw = u_kw;
                u_coff += BK;
                if (u_coff >= op.C) {
                    u_coff -= op.C;
                    ++u_tap;
                    if (++u_kw == op.KW) { u_kw = 0; ++u_kh; }
                }
                if (tap != tap_cur) {          // wave-uniform: new tap -> new source pixel / bounds for every unit
                    tap_cur = tap;
                    tvm = 0;
#pragma unroll
                    for (int i = 0; i < NU; ++i) {
                        int sy, sx;
                        bool ok = (rvm >> i) & 1u;
                        if constexpr (MODE == M_KC_G1) {
                            sy = iy0[i] + kh;
                            sx = ix0[i] + kw;
                        } else {
                            const int ny = iy0[i] - kh, nx = ix0[i] - kw;
                            ok = ok && ny >= 0 && nx >= 0;
                            if (op.stride == 1) {
                                sy = ny;
                                sx = nx;
                            } else if (op.stride == 2) {
                                sy = ny >> 1;
                                sx = nx >> 1;
                                ok = ok && (((ny | nx) & 1) == 0);
                            } else {
                                sy = ny / op.stride;
                                sx = nx / op.stride;
                                ok = ok && (sy * op.stride == ny) && (sx * op.stride == nx);
                            }
                        }
                        ok = ok && (unsigned)sy < (unsigned)op.H && (unsigned)sx < (unsigned)op.W;
                        toff[i] = (sy * op.W + sx) * op.CS;   // per-image offset fits 32 bits
                        tvm |= (ok ? 1u : 0u) << i;
                    }
                }
            }
            if constexpr (TF != 0) {
                const float* sc = op.sc + z1 * op.tf_bs;
                const float* sh = op.sh + z1 * op.tf_bs;
                if (VEC) {
                    const int c = kok ? ch : 0;
                    s.ts4 = *(const v4f*)(sc + c);
                    s.th4 = *(const v4f*)(sh + c);
                    if constexpr (TF == 2) s.tk4 = *(const v4f*)(op.sc2 + z1 * op.tf_bs + c);
                    if constexpr (TF == 3) {
                        if (tail2) { s.tk4 = *(const v4f*)(op.sc2 + c); s.tq4 = *(const v4f*)(op.sh2 + c); }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = (kk + j < kend) ? ch + j : 0;
                        s.ts4[j] = sc[c];
                        s.th4[j] = sh[c];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const bool rok = (rvm >> i) & 1u;
                if constexpr (MODE == M_KC) {
                    if (VEC) {
                        const bool ok = rok && kok;
                        s.r[i] = load4_raw<S16>(ptr, ok ? base[i] + k0 : 0);
                        if constexpr (TF == 2 || TF == 3) s.r2[i] = load4_raw<S2_16>(ptr2, ok ? base[i] + k0 : 0);
                        s.vm |= (ok ? 1u : 0u) << i;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool ok = rok && (kk + j) < kend;
                            s.r[i][j] = ptr[ok ? base[i] + k0 + j : 0];
                            s.vm |= (ok ? 1u : 0u) << (4 * i + j);
                        }
                    }
                } else {
                    const bool ok = kok && ((tvm >> i) & 1u);
                    s.r[i] = load4_raw<S16>(ptr, ok ? base[i] + (toff[i] + coff) : 0);
                    if constexpr (TF == 2) s.r2[i] = load4_raw<S2_16>(ptr2, ok ? base[i] + (toff[i] + coff) : 0);
                    s.vm |= (ok ? 1u : 0u) << i;
                }
            }
        } else {
            constexpr int CV = ROWS / 4;
            constexpr int RP = 256 / CV;
            const int kr0 = t / CV;
            int c03 = 0, th3 = 0, tw3 = 0;
            if constexpr (MODE == M_KM_G3) {
                c03 = u_coff; th3 = u_kh; tw3 = u_kw;
                u_coff += BK;
                if (u_coff >= op.C) {
                    u_coff -= op.C;
                    if (++u_kw == op.KW) { u_kw = 0; ++u_kh; }
                }
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int k = k0 + kr0 + RP * i;
                bool ok = k < kend;
                int64_t off;
                if constexpr (MODE == M_KM) {
                    off = (int64_t)k * op.ld + col;
                } else if constexpr (MODE == M_KM_G3) {
                    // tapped weights: k = (tap, ck), tap = th*KW + tw; element at ck*ld + th*tap_stride_h + tw*tap_stride + col
                    off = (int64_t)(c03 + kr0 + RP * i) * op.ld + th3 * op.tap_stride_h + tw3 * op.tap_stride + col;
                } else {
                    const int sy = gsy[i], sx = gsx[i];
                    ok = ok && (unsigned)sy < (unsigned)op.H && (unsigned)sx < (unsigned)op.W;
                    off = goff[i];
                    // advance this unit's source pixel by BK rows of k
                    int nsx = sx + g_cs;
                    const bool c1 = nsx >= g_sxlim;
                    nsx -= c1 ? g_pws : 0;
                    int nsy = sy + g_bs + (c1 ? op.stride : 0);
                    const bool c2 = nsy >= g_sylim;
                    nsy -= c2 ? g_phs : 0;
                    gsx[i] = nsx;
                    gsy[i] = nsy;
                    goff[i] = off + g_d0 + (c1 ? g_d1 : (int64_t)0) + (c2 ? g_d2 : (int64_t)0);
                }
                if (VEC) {
                    ok = ok && (cvm & 1u);
                    s.r[i] = load4_raw<S16>(ptr, ok ? off : 0);
                    if constexpr (TF == 2) s.r2[i] = load4_raw<S2_16>(ptr2, ok ? off : 0);
                    s.vm |= (ok ? 1u : 0u) << i;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool okj = ok && ((cvm >> j) & 1u);
                        s.r[i][j] = ptr[okj ? off + j : 0];
                        s.vm |= (okj ? 1u : 0u) << (4 * i + j);
                    }
                }
            }
        }
    }

    // transform + zero-fill of the tile issued by issue(); first consumer of the loaded registers
    // TF 3 (the bottleneck tail formed on load): y = relu(sc[c] * x + sh[c] + x2) of the conv output x at ptr and the identity x2
    // at ptr2 -- the arithmetic of koaf_bn_add_relu, bit for bit; y itself is written to `side` (same layout as x) when this
    // block owns the column range (side != nullptr: the first column tile), at element offset base[i] + k0s.
    float* side = nullptr;
    int k0s = 0;
    // FULL: every element of this wave's slot is valid (interior tiles of the dense operands: the usual case) -- no zero-fill selects
    template <bool FULL>
    __device__ __forceinline__ void finish_unit(Slot& s, int i, v4f a, v4f b, v4f k, v4f q = (v4f){0.f, 0.f, 0.f, 0.f}) {
        constexpr float HMAX = 65504.f;
        if constexpr (S16) s.r[i] = widen_bf16x4(__float_as_uint(s.r[i][0]), __float_as_uint(s.r[i][1]));
        if constexpr (S2_16 && (TF == 2 || TF == 3)) s.r2[i < NU2 ? i : 0] = widen_bf16x4(__float_as_uint(s.r2[i < NU2 ? i : 0][0]), __float_as_uint(s.r2[i < NU2 ? i : 0][1]));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = FULL || (VEC ? ((s.vm >> i) & 1u) : ((s.vm >> (4 * i + j)) & 1u));
            float x = s.r[i][j];
            if constexpr (TF == 1) {
                x = fmaf(x, a[j], b[j]);
                x = F16 ? __builtin_amdgcn_fmed3f(x, 0.f, HMAX) : fmaxf(x, 0.f);
            } else if constexpr (TF == 2) {
                x = fmaf(a[j], x, fmaf(-k[j], s.r2[i < NU2 ? i : 0][j], b[j]));
                if constexpr (F16) x = __builtin_amdgcn_fmed3f(x, -HMAX, HMAX);
            } else if constexpr (TF == 3) {
                float idv = s.r2[i < NU2 ? i : 0][j];
                if (tail2) idv = fmaf(idv, k[j], q[j]);                  // (a downsample branch: its BatchNorm, as koaf_bn_add_relu's idsc / idsh)
                x = fmaxf(fmaf(x, a[j], b[j]) + idv, 0.f);      // (a, b unscaled here: y is stored as it is)
                if constexpr (S16) x = widen_bf16x4(round_bf16x4((v4f){x, 0.f, 0.f, 0.f}).x, 0u)[0];   // bf16 storage: everyone reads the ROUNDED y
                s.r2[i < NU2 ? i : 0][j] = x;
                x = fminf(x * fsc, HMAX);
            } else if constexpr (F16) {
                x = __builtin_amdgcn_fmed3f(x * fsc, -HMAX, HMAX);
            }
            s.r[i][j] = ok ? x : 0.f;
        }
        if constexpr (TF == 3) {
            if (side != nullptr && (FULL || ((s.vm >> i) & 1u))) {
                if constexpr (S16) store4<true>(side, base[i] + k0s, s.r2[i < NU2 ? i : 0]);
                else *(v4f*)(side + base[i] + k0s) = s.r2[i < NU2 ? i : 0];
            }
        }
    }
    __device__ __forceinline__ void finish(Slot& s) {
        v4f a = KC ? s.ts4 : kts4, b = KC ? s.th4 : kth4, k = KC ? s.tk4 : ktk4;
        if constexpr (F16 && KC && TF != 0 && TF != 3) { a *= fsc; b *= fsc; k *= fsc; }     // (KM coefficients were scaled once in init)
        // (wave-uniform: one ballot per tile; the ragged last tiles and the padded taps of gathers take the selecting form)
        constexpr unsigned ALLV = VEC ? ((NU >= 32) ? ~0u : ((1u << NU) - 1u)) : ((4 * NU >= 32) ? ~0u : ((1u << (4 * NU)) - 1u));
        if (__builtin_amdgcn_ballot_w64(s.vm != ALLV) == 0ull) {
#pragma unroll
            for (int i = 0; i < NU; ++i) finish_unit<true>(s, i, a, b, k, s.tq4);
        } else {
#pragma unroll
            for (int i = 0; i < NU; ++i) finish_unit<false>(s, i, a, b, k, s.tq4);
        }
    }
    // LDS dword offset (within a plane) of unit i of this thread
    __device__ __forceinline__ int plane_off(int i) const {
        const int t = threadIdx.x;
        if constexpr (KC) {
            return ((t >> 3) + 32 * i) * 20 + 2 * (t & 7);
        } else {
            constexpr int CV = ROWS / 4;
            constexpr int RP = 256 / CV;
            return (t / CV + RP * i) * (ROWS / 2 + 16) + 2 * (t % CV);
        }
    }

    template <int NPL>
    __device__ __forceinline__ void store(const Slot& s, float* Sf) {
        unsigned* S = (unsigned*)Sf;
        constexpr int P = plane_dwords(ROWS, KC);
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            unsigned pl[NPL][2];
            if constexpr (F16) split2h(s.r[i], pl);
            else split3v(s.r[i], pl);
            if constexpr (F16 && (TF == 1 || TF == 3) && KC) {
                // saturation watch of the fixed activation scale: behind the ReLU the hi pieces are non-negative fp16, whose
                // bits order like the values -- one packed 16-bit maximum per two elements; a tile that reached 65504 (0x7bff)
                // clamped something (koaf.h koaf_set_status_buffer)
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int d = 0; d < 2; ++d)
                    satmax = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, satmax),
                                                                                     __builtin_bit_cast(u16x2, pl[0][d])));
            }
            const int off = plane_off(i);
#pragma unroll
            for (int q = 0; q < NPL; ++q) *(uint2*)&S[q * P + off] = make_uint2(pl[q][0], pl[q][1]);
        }
    }
};


// ---- M_KS: the dense K-contiguous fp32 A operand, streamed per wave ------------------------------------------------------------
// The block-wide loader above keeps ONE k-tile in flight (issued at the top of a k-step, consumed at its end) and meets at two
// barriers per step: on the 1x1 convolutions of layer2-4 -- K = 128 .. 2048, 4 .. 64 steps of 24 MFMAs per wave -- every step
// then costs a memory latency (in-kernel stamps: 1.7 us per step against 0.4 us of matrix work; the same kernel fed from
// pre-split plane images by LDS-DMA, no conversion at all, is only 10 % faster).  Here the waves of a block are 4 x 1: wave w
// owns rows 32 w .. 32 w + 31 of the tile and ALL its columns, so the A image rows it writes are the rows it reads -- no block
// barrier on the A side, only the in-order LDS queue of the wave itself -- and it keeps SD k-tiles of its rows in flight in
// registers (16 per tile and source).  The flattened (tile, k-step) sequence of a persistent block is prefetched across tile
// boundaries: the first SD k-tiles of the next tile land under the epilogue of the current one.  Arithmetic, pieces and MFMA
// order per accumulator are those of TileLoader + the shared k-loop: bit-identical outputs (test_stream_kernel_is_bit_identical).
// Lane l of a wave: unit i (0..3) = row 8 i + l / 8 of the wave's band, k = 4 (l % 8) .. + 3 -- 128-B row segments per 8 lanes.
constexpr int STREAM_TAB_K = 1024;      // longest k range of a TF 1 call on the streamed path (8 KiB of LDS beside the operand images)
template <int TF, int SD>
struct StreamA {
    static constexpr bool TWO = (TF == 2 || TF == 3);
    struct Slot {
        v4f r[4];
        v4f r2[TWO ? 4 : 1];
        v4f ts, th, tk, tq;      // transform coefficients of the k-tile's 4 columns of this lane (sc, sh, sc2, sh2)
    };
    Slot sl[SD];
    const float* ptr;
    const float* ptr2;
    const float* sc;
    const float* sh;
    const float* sc2;
    const float* sh2;
    const float* tab;     // TF 1: LDS table [2][STREAM_TAB_K] of sc * fsc, sh * fsc (the coefficients depend on k only: read when a k-tile
                          // is consumed instead of riding in eight registers per tile in flight)
    int64_t ld;
    float fsc;
    bool tail2;
    bool once;        // the A operand is read by ONE column tile (N <= BN): non-temporal loads (KOAF_STREAM_NT=0: off)
    unsigned satmax;
    // issue cursor: the k-tile the next issue() fetches
    v4l ibase;        // element offset of each unit's (clamped) row + 4 (l % 8)
    int ik;
    // consumer side
    v4l cbase;        // TF 3: offsets of the side store (the tile being consumed)
    unsigned rvm;     // row-valid bits of the tile being consumed
    int kbeg, kend;

    __device__ __forceinline__ v4l bases(int m0, int M) const {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        v4l b;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = min(m0 + 32 * w + 8 * i + (lane >> 3), M - 1);     // (rows past M re-read the last row: zeroed in consume())
            b[i] = (int64_t)row * ld + 4 * (lane & 7);
        }
        return b;
    }
    __device__ __forceinline__ void init(const KoafOperand& op, const float* p, int m0, int M, int kb, int ke, float scale) {
        ptr = p;
        ptr2 = TWO ? op.ptr2 + (p - op.ptr) : nullptr;
        sc = op.sc; sh = op.sh; sc2 = op.sc2; sh2 = op.sh2;
        ld = op.ld;
        fsc = scale;
        tail2 = (TF == 3) && op.sc2 != nullptr;
        once = false;
        satmax = 0u;
        tab = nullptr;
        kbeg = kb; kend = ke;
        ik = kb;
        ibase = bases(m0, M);
        tile(m0, M);
    }
    // the consumer moves on to the tile at m0
    __device__ __forceinline__ void tile(int m0, int M) {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        rvm = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) rvm |= ((m0 + 32 * w + 8 * i + (lane >> 3)) < M ? 1u : 0u) << i;
        if constexpr (TF == 3) cbase = bases(m0, M);
    }
    // loads of the k-tile at the cursor into slot s; the cursor then advances by one k-tile and, at the end of the k range, to
    // the first row next() returns for the block's following tile (the cursor runs SD k-tiles ahead of the consumer, so with as few
    // as SD k-steps per tile it is a whole tile ahead: it keeps its own place in the block's tile sequence); next() < 0: no
    // further tile -- the k-loop issues no more loads then
    template <class NextFn>
    __device__ __forceinline__ void issue(Slot& s, NextFn next, int M) {
        const int lane = threadIdx.x & 63;
        const int c = ik + 4 * (lane & 7);
        if constexpr (TF > 1) {
            s.ts = *(const v4f*)(sc + c);
            s.th = *(const v4f*)(sh + c);
            if constexpr (TF == 2) s.tk = *(const v4f*)(sc2 + c);
            if constexpr (TF == 3) {
                if (tail2) { s.tk = *(const v4f*)(sc2 + c); s.tq = *(const v4f*)(sh2 + c); }
            }
        }
        if (once) {
            // (one column tile: every A byte is read exactly once by the whole grid -- streamed past the caches' replacement order)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s.r[i] = __builtin_nontemporal_load((const v4f*)(ptr + ibase[i] + ik));
                if constexpr (TWO) s.r2[i] = __builtin_nontemporal_load((const v4f*)(ptr2 + ibase[i] + ik));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s.r[i] = *(const v4f*)(ptr + ibase[i] + ik);
                if constexpr (TWO) s.r2[i] = *(const v4f*)(ptr2 + ibase[i] + ik);
            }
        }
        ik += BK;
        if (ik >= kend) {
            ik = kbeg;
            const int nm0 = next();
            if (nm0 >= 0) ibase = bases(nm0, M);
        }
    }
    // transform (TileLoader::finish_unit's arithmetic), split and store of slot s = the k-tile at k0 of the tile being consumed,
    // into this wave's rows of the block's A plane images S (plane_dwords(128, true) dwords per plane)
    template <bool FULL>
    __device__ __forceinline__ void consume_as(Slot& s, unsigned* S, int k0, float* side) {
        constexpr float HMAX = 65504.f;
        constexpr int P = plane_dwords(128, true);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        v4f a = s.ts, b = s.th, k = s.tk;
        if constexpr (TF == 2) { a *= fsc; b *= fsc; k *= fsc; }
        if constexpr (TF == 1) {
            a = *(const v4f*)(tab + (k0 - kbeg) + 4 * (lane & 7));
            b = *(const v4f*)(tab + STREAM_TAB_K + (k0 - kbeg) + 4 * (lane & 7));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = FULL || ((rvm >> i) & 1u);
            v4f x = s.r[i];
            [[maybe_unused]] v4f y = x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = x[j];
                if constexpr (TF == 1) {
                    v = fmaf(v, a[j], b[j]);
                    v = __builtin_amdgcn_fmed3f(v, 0.f, HMAX);
                } else if constexpr (TF == 2) {
                    v = fmaf(a[j], v, fmaf(-k[j], s.r2[i][j], b[j]));
                    v = __builtin_amdgcn_fmed3f(v, -HMAX, HMAX);
                } else if constexpr (TF == 3) {
                    float idv = s.r2[i][j];
                    if (tail2) idv = fmaf(idv, k[j], s.tq[j]);
                    v = fmaxf(fmaf(v, a[j], b[j]) + idv, 0.f);
                    y[j] = v;
                    v = fminf(v * fsc, HMAX);
                } else {
                    v = __builtin_amdgcn_fmed3f(v * fsc, -HMAX, HMAX);
                }
                x[j] = ok ? v : 0.f;
            }
            if constexpr (TF == 3) {
                if (side != nullptr && ok) *(v4f*)(side + cbase[i] + k0) = y;
            }
            unsigned pl[2][2];
            split2h(x, pl);
            if constexpr (TF == 1 || TF == 3) {
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int d = 0; d < 2; ++d)
                    satmax = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, satmax),
                                                                                     __builtin_bit_cast(u16x2, pl[0][d])));
            }
            const int off = (32 * w + 8 * i + (lane >> 3)) * 20 + 2 * (lane & 7);
#pragma unroll
            for (int q = 0; q < 2; ++q) *(uint2*)&S[q * P + off] = make_uint2(pl[q][0], pl[q][1]);
        }
    }
    __device__ __forceinline__ void consume(Slot& s, unsigned* S, int k0, float* side) {
        if (__builtin_amdgcn_ballot_w64(rvm != 15u) == 0ull) consume_as<true>(s, S, k0, side);
        else consume_as<false>(s, S, k0, side);
    }
};

typedef short v4s __attribute__((ext_vector_type(4)));

// bf16x8 MFMA fragment of plane image P: rows row0 .. row0+31, k = 16g + 8h + (0..7); lane = 32h + r
template <int ROWS, bool KC>
__device__ __forceinline__ v4i frag_load(const unsigned* P, int row0, int g, int lane) {
    if constexpr (KC) {
        return *(const v4i*)&P[(row0 + (lane & 31)) * 20 + 8 * g + 4 * (lane >> 5)];
    } else {
        // two transposed 4(k) x 16(rows) block reads; lane 4q+p of a 16-lane group addresses block row q, cols 4p..4p+3
        const int li = lane & 15, q = li >> 2, pp = li & 3;
        const int rb = row0 + 16 * ((lane >> 4) & 1) + 4 * pp;
        const int k0 = 16 * g + 8 * (lane >> 5) + q;
        constexpr int SK = ROWS / 2 + 16;
        typedef __attribute__((address_space(3))) v4s* lds_v4s;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(P + k0 * SK + rb / 2));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(P + (k0 + 4) * SK + rb / 2));
        const v2i l2 = __builtin_bit_cast(v2i, lo), h2 = __builtin_bit_cast(v2i, hi);
        return (v4i){l2[0], l2[1], h2[0], h2[1]};
    }
}

// v & m as four opaque v_and_b32 (written in C++, hipcc turns the masked fragment load into a branch around the ds_read --
// or, with a plain vector AND, fails in instruction selection on this kernel)
__device__ __forceinline__ v4i and_mask(v4i v, int m) {
    v4i r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int x;
        asm("v_and_b32 %0, %1, %2" : "=v"(x) : "v"(v[e]), "v"(m));
        r[e] = x;
    }
    return r;
}

// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to the 1 KiB at LDS byte address `lds_addr`
// (wave-uniform), lane-linear.
// Issued as inline assembly, not through __builtin_amdgcn_global_load_lds: the compiler's wait-count pass treats every
// LDS read after a builtin LDS-DMA as possibly aliasing it and puts s_waitcnt vmcnt(0) in front of the ds_reads of the k-loop
// -- which drains the prefetch of the NEXT tiles before the current one is multiplied and was the largest single stall
// of the DMA kernels.  The kernels order DMA against LDS reads themselves (counted s_waitcnt vmcnt + s_barrier); no
// compiler-tracked vector memory operation is in flight while these are (the loops hold only DMA, and they drain it
// before the epilogue).
// (m0 is a reserved register to clang, which warns that it does not preserve it around the statement: nothing else in
// these kernels lives in m0 -- gfx9 LDS instructions do not read it and there is no indirect register indexing.)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_addr) {
    const int la = __builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(la) : "memory", "m0");
}
// LDS byte address of a __shared__ array (taken ONCE, on the array itself, where the cast folds: converting the
// generic pointers computed later back to LDS addresses left a null check on the aperture register that hipcc could not select)
#define KOAF_LDS_ADDR(arr) ((unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(arr))
#pragma clang diagnostic pop

// Pre-split operand (M_PS): bf16 plane images [plane][row][K] cut in HBM by koaf_wplanes_build, moved global -> LDS by
// global_load_lds_dwordx4.  One wave instruction fills 1 KiB = 16 rows x 64 B of one plane; the LDS image is linear
// (DMA writes land at wave base + 16 * lane), so its 16-B chunks are XOR-swizzled through the SOURCE address: lane l
// fetches chunk (l & 3) ^ (row / 4 % 4) of row l / 4 of its piece, and frag_load_ps() applies the same XOR.
template <int ROWS>
struct PlaneLoader {
    static constexpr int NPIECE = ROWS / 16;       // 1-KiB pieces per plane
    static constexpr int PPW = NPIECE / 4;         // per wave
    static constexpr int PLANE_BYTES = ROWS * 64;
    int64_t src[PPW];      // element offset (bf16) of this lane's chunk at k = 0, plane 0
    int u_coff, u_kh, u_kw;

    __device__ __forceinline__ void init(const KoafOperand& op, int r0, int R) {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            // rows past R re-read the last row: finite values that only reach output columns >= N, which are never stored
            const int row = min(r0 + 16 * (w + 4 * j) + (lane >> 2), R - 1);
            src[j] = (int64_t)row * op.ld + 8 * ((lane & 3) ^ ((lane >> 4) & 3));
        }
        u_coff = u_kh = u_kw = 0;
    }
    __device__ __forceinline__ void seek(const KoafOperand& op, int k0) {
        const int tap = k0 / op.C;
        u_coff = k0 - tap * op.C;
        u_kh = tap / op.KW;
        u_kw = tap - u_kh * op.KW;
    }
    // DMA of the k-tile at the running position into the plane images at `lds` (one buffer = NPL * PLANE_BYTES)
    template <int NPL>
    __device__ __forceinline__ void issue(const KoafOperand& op, const unsigned short* planes, unsigned lds) {
        const int w = threadIdx.x >> 6;
        const int64_t koff = u_kh * op.tap_stride_h + u_kw * op.tap_stride + u_coff;
        u_coff += BK;
        if (u_coff >= op.C) {
            u_coff -= op.C;
            if (++u_kw == op.KW) { u_kw = 0; ++u_kh; }
        }
#pragma unroll
        for (int q = 0; q < NPL; ++q)
#pragma unroll
            for (int j = 0; j < PPW; ++j)
                lds_dma16(planes + q * op.plane_stride + src[j] + koff, lds + q * PLANE_BYTES + (w + 4 * j) * 1024);
    }
};

// fragment of a swizzled linear plane image written by PlaneLoader: rows of 16 dwords (32 k)
__device__ __forceinline__ v4i frag_load_ps(const unsigned* P, int row0, int g, int lane) {
    const int row = row0 + (lane & 31);
    return *(const v4i*)&P[row * 16 + 4 * ((2 * g + (lane >> 5)) ^ ((row >> 2) & 3))];
}

// Pre-split ACTIVATION operand (M_PA1 / M_PA2): the two fp16 piece planes [plane][pixel][CS] of an NHWC tensor, cut once
// by koaf_act_planes (BatchNorm + ReLU prologue or BatchNorm-backward apply included), gathered global -> LDS by
// global_load_lds_dwordx4 exactly like PlaneLoader: a lane moves the 16 B = 8 channels of ONE source pixel, so the im2col
// gather costs address arithmetic only (once per filter tap) -- no conversion, no split, no VGPR staging in the k-loop,
// where the fp32 loader redoes the split of every element for each of the KH*KW taps that touch it.  Padding taps and
// rows past M fetch the image's zero chunk (KoafOperand.zeros).  G = 1: conv forward gather; 2: transposed (dgrad).
template <int ROWS, int G>
struct PlaneGatherLoader {
    static constexpr int NPIECE = ROWS / 16;
    static constexpr int PPW = NPIECE / 4;
    static_assert(PPW >= 1 && PPW <= 4, "1..4 pieces per wave");
    static constexpr int PLANE_BYTES = ROWS * 64;
    v4l base;          // element offset of the piece row's image + this lane's swizzled chunk
    v4i iy0, ix0;
    v4i toff;          // per-image element offset of the current tap's source pixel
    unsigned rvm, tvm;
    int u_coff, u_kh, u_kw;
    bool fresh;

    __device__ __forceinline__ void init(const KoafOperand& op, int r0, int R) {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        base = (v4l){0, 0, 0, 0};
        iy0 = ix0 = toff = (v4i){0, 0, 0, 0};
        rvm = tvm = 0;
        const int ppi = op.PH * op.PW;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int row = r0 + 16 * (w + 4 * j) + (lane >> 2);
            rvm |= (row < R ? 1u : 0u) << j;
            const int n = row / ppi;
            const int rem = row - n * ppi;
            const int py = rem / op.PW;
            const int px = rem - py * op.PW;
            base[j] = (int64_t)n * op.H * op.W * op.CS + 8 * ((lane & 3) ^ ((lane >> 4) & 3));
            if constexpr (G == 1) {
                iy0[j] = py * op.stride - op.pad;
                ix0[j] = px * op.stride - op.pad_w;
            } else {
                iy0[j] = py + op.pad;
                ix0[j] = px + op.pad_w;
            }
        }
        u_coff = u_kh = u_kw = 0;
        fresh = true;
    }
    __device__ __forceinline__ void seek(const KoafOperand& op, int k0) {
        const int tap = k0 / op.C;
        u_coff = k0 - tap * op.C;
        u_kh = tap / op.KW;
        u_kw = tap - u_kh * op.KW;
        fresh = true;
    }
    __device__ __forceinline__ void issue(const KoafOperand& op, const unsigned short* planes, unsigned lds) {
        const int w = threadIdx.x >> 6;
        if (fresh || u_coff == 0) {        // wave-uniform: a new filter tap -> new source pixel / bounds of every piece row
            fresh = false;
            tvm = 0;
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                int sy, sx;
                bool ok = (rvm >> j) & 1u;
                if constexpr (G == 1) {
                    sy = iy0[j] + u_kh;
                    sx = ix0[j] + u_kw;
                } else {
                    const int ny = iy0[j] - u_kh, nx = ix0[j] - u_kw;
                    ok = ok && ny >= 0 && nx >= 0;
                    if (op.stride == 1) {
                        sy = ny;
                        sx = nx;
                    } else if (op.stride == 2) {
                        sy = ny >> 1;
                        sx = nx >> 1;
                        ok = ok && (((ny | nx) & 1) == 0);
                    } else {
                        sy = ny / op.stride;
                        sx = nx / op.stride;
                        ok = ok && (sy * op.stride == ny) && (sx * op.stride == nx);
                    }
                }
                ok = ok && (unsigned)sy < (unsigned)op.H && (unsigned)sx < (unsigned)op.W;
                toff[j] = (sy * op.W + sx) * op.CS;
                tvm |= (ok ? 1u : 0u) << j;
            }
        }
        const int coff = u_coff;
        u_coff += BK;
        if (u_coff >= op.C) {
            u_coff = 0;
            if (++u_kw == op.KW) { u_kw = 0; ++u_kh; }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                const bool ok = (tvm >> j) & 1u;
                const unsigned short* src = ok ? planes + q * op.plane_stride + base[j] + (toff[j] + coff) : op.zeros;
                lds_dma16(src, lds + q * PLANE_BYTES + (w + 4 * j) * 1024);
            }
    }
};

// K-major operands from activation plane images (M_PK / M_PKG: the weight gradient, k = pixel): one DMA instruction moves
// 1 KiB = KPI whole k-rows (pixels) of ROWS channels; the LDS image is plane[32 k][ROWS] fp16, linear, its 16-B chunks
// XOR-swizzled by k so that the transposing fragment reads (ds_read_b64_tr_b16: 4 k-rows x 32 B per 16-lane group, two
// groups per LDS cycle) hit 64 distinct banks: chunk ^ 4 (k & 3) for 256-B rows, chunk ^ 4 (k / 2 & 1) for 128-B rows.
__host__ __device__ constexpr int kmd_swz(int rows, int k) { return rows == 128 ? 4 * (k & 3) : 4 * ((k >> 1) & 1); }

template <int ROWS, bool GATHER>
struct PlaneKLoader {
    static_assert(ROWS == 128 || ROWS == 64, "tile rows");
    static constexpr int CPR = ROWS / 8;          // 16-B chunks per k-row
    static constexpr int KPI = 64 / CPR;          // k-rows per DMA instruction
    static constexpr int NPIECE = 32 / KPI;       // instructions per plane and k-tile
    static constexpr int PPW = NPIECE / 4;        // per wave
    static constexpr int PLANE_BYTES = ROWS * 64;
    int col, cc, kh_, kw_;      // first column of this lane's chunk; its channel and filter tap (gather)
    bool cok;
    v4i kk;                     // k of each piece of this lane (k-tile origin excluded)
    // running source pixel of each piece (gather), as in TileLoader's M_KM_G1: advanced by adds and single carries
    v4i gsx, gsy;
    v4l goff;
    int g_cs, g_bs, g_pws, g_phs, g_sxlim, g_sylim;
    int64_t g_d0, g_d1, g_d2;

    __device__ __forceinline__ void init(const KoafOperand& op, int r0, int R) {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int kl = lane / CPR, phys = lane % CPR;
        col = r0 + 8 * (phys ^ kmd_swz(ROWS, kl));       // (pieces start on multiples of KPI >= 4 | 8: the swizzle sees kl only)
        cok = col < R;                                   // R % 8 == 0
        cc = col; kh_ = kw_ = 0;
        if constexpr (GATHER) {
            const int tap = col / op.C;
            cc = col - tap * op.C;
            kh_ = tap / op.KW;
            kw_ = tap - kh_ * op.KW;
        }
        kk = (v4i){0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < PPW; ++j) kk[j] = KPI * (w + 4 * j) + kl;
        gsx = gsy = (v4i){0, 0, 0, 0};
        goff = (v4l){0, 0, 0, 0};
        g_cs = g_bs = g_pws = g_phs = g_sxlim = g_sylim = 0;
        g_d0 = g_d1 = g_d2 = 0;
    }
    __device__ __forceinline__ void seek(const KoafOperand& op, int k0) {
        if constexpr (GATHER) {
            const int ppi = op.PH * op.PW;
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                const int k = k0 + kk[j];
                const int n = k / ppi;
                const int rem = k - n * ppi;
                const int py = rem / op.PW;
                const int px = rem - py * op.PW;
                gsy[j] = py * op.stride - op.pad + kh_;
                gsx[j] = px * op.stride - op.pad_w + kw_;
                goff[j] = ((int64_t)(n * op.H + gsy[j]) * op.W + gsx[j]) * op.CS + cc;
            }
            const int a = BK / ppi, r = BK - a * ppi, b = r / op.PW, c = r - b * op.PW;
            const int64_t wcs = (int64_t)op.W * op.CS, hwcs = (int64_t)op.H * wcs;
            g_cs = c * op.stride;
            g_bs = b * op.stride;
            g_pws = op.PW * op.stride;
            g_phs = op.PH * op.stride;
            g_sxlim = g_pws - op.pad_w + kw_;
            g_sylim = g_phs - op.pad + kh_;
            g_d0 = a * hwcs + g_bs * wcs + (int64_t)g_cs * op.CS;
            g_d1 = (int64_t)op.stride * wcs - (int64_t)g_pws * op.CS;
            g_d2 = hwcs - g_phs * wcs;
        }
    }
    // DMA of the k-tile [k0, k0 + 32) into the two plane images at LDS byte address `lds`
    __device__ __forceinline__ void issue(const KoafOperand& op, const unsigned short* planes, int k0, int kend, unsigned lds) {
        const int w = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            bool ok = cok && (k0 + kk[j]) < kend;
            int64_t off;
            if constexpr (GATHER) {
                const int sy = gsy[j], sx = gsx[j];
                ok = ok && (unsigned)sy < (unsigned)op.H && (unsigned)sx < (unsigned)op.W;
                off = goff[j];
                int nsx = sx + g_cs;
                const bool c1 = nsx >= g_sxlim;
                nsx -= c1 ? g_pws : 0;
                int nsy = sy + g_bs + (c1 ? op.stride : 0);
                const bool c2 = nsy >= g_sylim;
                nsy -= c2 ? g_phs : 0;
                gsx[j] = nsx;
                gsy[j] = nsy;
                goff[j] = off + g_d0 + (c1 ? g_d1 : (int64_t)0) + (c2 ? g_d2 : (int64_t)0);
            } else {
                off = (int64_t)(k0 + kk[j]) * op.ld + col;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned short* src = ok ? planes + q * op.plane_stride + off : op.zeros;
                lds_dma16(src, lds + q * PLANE_BYTES + (w + 4 * j) * 1024);
            }
        }
    }
};

// fragment of a k-swizzled K-major plane image written by PlaneKLoader (cf. frag_load's K-major branch)
template <int ROWS>
__device__ __forceinline__ v4i frag_load_kmd(const unsigned* P, int row0, int g, int lane) {
    const int li = lane & 15, q = li >> 2, pp = li & 3;
    const int rb = row0 + 16 * ((lane >> 4) & 1) + 4 * pp;
    const int k0 = 16 * g + 8 * (lane >> 5) + q;         // (k0 + 4 has the same swizzle)
    const int boff = k0 * (ROWS * 2) + (((rb >> 3) ^ kmd_swz(ROWS, k0)) << 4) + ((rb & 7) << 1);
    typedef __attribute__((address_space(3))) v4s* lds_v4s;
    const char* Pb = reinterpret_cast<const char*>(P);
    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(Pb + boff));
    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(Pb + boff + 4 * (ROWS * 2)));
    const v2i l2 = __builtin_bit_cast(v2i, lo), h2 = __builtin_bit_cast(v2i, hi);
    return (v4i){l2[0], l2[1], h2[0], h2[1]};
}

// M_PT: LDS-DMA of the 10 x 18 pixel halo of 2-D tile `tm` (64 channels from channel 64 * chunk, both planes: 2880 granules of 16 B
// = 45 pieces; wave w moves pieces w, w + 4, ...) into the image at LDS byte address halo0.  Granule c of halo pixel (y, x) lands at
// ((18 y + x) * 8 + (c ^ (x / 2 % 8))) * 16; pixels outside the image fetch the operand's zero chunk.
__device__ __forceinline__ void t2d_issue_halo(const KoafOperand& A, const unsigned short* Apl, int tm, int chunk, unsigned halo0) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int Wd = A.W, Hd = A.H, CSa = A.CS;
    const int txn = Wd >> 4, tpi = (Hd >> 3) * txn;          // tiles per image row / per image
    const int img = tm / tpi, trem = tm - img * tpi, tyi = trem / txn, txi = trem - tyi * txn;
    // this lane's granule of its wave's first piece; every later piece is 256 granules = 32 halo pixels further on (one halo row and
    // 14 pixels), the second plane 1440 granules = 10 halo rows back: the pixel is carried, not re-derived by divisions
    int Gp = w * 64 + lane, q = 0;
    const int cs = Gp & 7;
    int y = (Gp >> 3) / 18, x = (Gp >> 3) - 18 * y;
    const int iy0 = tyi * 8 - 1, ix0 = txi * 16 - 1;
    const int64_t ibase = (int64_t)img * Hd * Wd;
#pragma unroll 1
    for (int pc = w; pc < 45; pc += 4) {
        const int c16 = cs ^ ((x >> 1) & 7);
        const int iy = iy0 + y, ix = ix0 + x;
        const bool ok = (unsigned)iy < (unsigned)Hd && (unsigned)ix < (unsigned)Wd;
        const unsigned short* src = ok ? Apl + q * A.plane_stride + (ibase + iy * Wd + ix) * CSa + (chunk * 64 + c16 * 8) : A.zeros;
        lds_dma16(src, halo0 + pc * 1024);
        Gp += 256; x += 14; y += 1;
        if (x >= 18) { x -= 18; y += 1; }
        if (q == 0 && Gp >= 1440) { Gp -= 1440; q = 1; y -= 10; }
    }
}

// Row loop of the vector epilogue for a FULL tile without row map, specialised on what is fused (residual, BatchNorm-
// backward mode, second BatchNorm) so that it is branch-free: the loads of four rows go out together before the first
// is consumed (the generic loop below tests every row and ends up with one load in flight at a time, which held the
// HBM-bound 1x1-dgrad epilogues at 2-3 TB/s).
// C16: the output tensor is stored as bf16; E16: the BatchNorm-backward operands (c / y / c2) are (KoafGemm.act16 1 / 2)
// KoafGemm.out_planes: the activation plane images of relu(out_sc * v + out_sh) * KOAF_ACT_SCALE for the four output elements v at
// element offset `off` -- koaf_act_planes' tf-1 arithmetic on the value as STORED (bf16 storage: the rounded one), bit for bit
template <bool C16>
__device__ __forceinline__ void epi_emit_planes(const KoafGemm& p, int64_t off, v4f v, v4f a, v4f b, unsigned& nsat) {
    constexpr float HMAX = 65504.f;
    if constexpr (C16) { const uint2 u = round_bf16x4(v); v = widen_bf16x4(u.x, u.y); }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float u = fmaf(v[j], a[j], b[j]);
        nsat += !(u <= HMAX) ? 1u : 0u;
        v[j] = __builtin_amdgcn_fmed3f(u, 0.f, HMAX);
    }
    unsigned pl[2][2];
    split2h(v, pl);
    *(uint2*)(p.out_planes + off) = make_uint2(pl[0][0], pl[0][1]);
    *(uint2*)(p.out_planes + p.out_ps + off) = make_uint2(pl[1][0], pl[1][1]);
}

// T2D: the tile's rows are an 8 x 16 pixel rectangle of one image (M_PT): row lr = pixel (lr / 16, lr % 16) of the tile whose first
// pixel is m0, image rows w2d pixels apart
// EMIT: the kernel instantiation that serves KoafGemm.out_planes (separate instantiations: the persistent 1x1 kernels carry the next
// tile's operand slot through this loop at the 256-register limit, and the emission arithmetic inline cost them 55-126 spilled registers)
template <int BM, int BN, int NT, bool HAS_R, int MODE, bool HAS_C2, bool C16, bool E16, bool T2D = false, bool EMIT = false>
__device__ __forceinline__ void epi_rows_full(const KoafGemm& p, const float* Cs, int ldcs, float* Cp, int64_t ldc,
                                              const float* Rp, int m0, int col, int c4, int rr, v4f bv, v4f mu, v4f is,
                                              v4f ms, v4f mh, v4f mu2, v4f is2, v4f& q1, v4f& q2, v4f& q3, v4f& qm, int w2d = 0) {
    constexpr int C4 = BN / 4, RPP = NT / C4, U = 4;
    static_assert((BM / RPP) % U == 0, "rows per thread must be a multiple of the batch");
    [[maybe_unused]] v4f ea = {0.f, 0.f, 0.f, 0.f}, eb = ea;      // KoafGemm.out_planes: this thread's columns of out_sc / out_sh, at the activation scale
    [[maybe_unused]] unsigned nsat = 0;
    if constexpr (EMIT && MODE == 0 && !HAS_R) {
        if (p.out_planes) { ea = *(const v4f*)(p.out_sc + col) * KOAF_ACT_SCALE; eb = *(const v4f*)(p.out_sh + col) * KOAF_ACT_SCALE; }
    }
#pragma unroll 1
    for (int row = rr; row < BM; row += RPP * U) {
        v4f rv[U], cv[U], yv[U], c2v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t orow;
            if constexpr (T2D) { const int lr_ = row + u * RPP; orow = m0 + (lr_ >> 4) * w2d + (lr_ & 15); }
            else orow = m0 + row + u * RPP;
            // (streamed once: non-temporal, like the stores below -- the tile's operands, not these, should stay in L2)
            if constexpr (HAS_R) rv[u] = __builtin_nontemporal_load((const v4f*)(Rp + orow * p.ldr + col));
            if constexpr (MODE != 0) cv[u] = load4_nt<E16>(p.bnb_c, orow * ldc + col);
            if constexpr (MODE == 1) yv[u] = load4_nt<E16>(p.bnb_y, orow * ldc + col);
            if constexpr (HAS_C2) c2v[u] = load4_nt<E16>(p.bnb2_c, orow * ldc + col);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t orow;
            if constexpr (T2D) { const int lr_ = row + u * RPP; orow = m0 + (lr_ >> 4) * w2d + (lr_ & 15); }
            else orow = m0 + row + u * RPP;
            v4f v = *(const v4f*)&Cs[(row + u * RPP) * ldcs + 4 * c4] + bv;
            if constexpr (HAS_R) v += rv[u];
            if constexpr (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = yv[u][j] > 0.f ? v[j] : 0.f;
            } else if constexpr (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (cv[u][j] * ms[j] + mh[j]) > 0.f ? v[j] : 0.f;
            }
            if constexpr (MODE != 0) {
                q1 += v;
                q2 += v * ((cv[u] - mu) * is);
                if constexpr (HAS_C2) q3 += v * ((c2v[u] - mu2) * is2);
#pragma unroll
                for (int j = 0; j < 4; ++j) qm[j] = __uint_as_float(max(__float_as_uint(qm[j]), koaf_absbits(v[j])));
            }
            store4_nt<C16>(Cp, orow * ldc + col, v);
            if constexpr (EMIT && MODE == 0 && !HAS_R) {
                if (p.out_planes) epi_emit_planes<C16>(p, orow * ldc + col, v, ea, eb, nsat);
            }
        }
    }
    if constexpr (EMIT && MODE == 0 && !HAS_R) koaf_status_add(p.status, 0, nsat);
}


// F16 = KoafGemm.fmt == 1 (two fp16 planes per operand, three products); else three bf16 planes, six products
// NT = threads per block: 256 (waves 2 x 2) or 512 (waves 4 x 2: the 256-row tiles of the halo kernel)
// PERSIST variants (see the kernel): the one-source A loaders only -- measured on the headline step, the forward 1x1
// convolutions gain 4-11 %, while the two-source (BatchNorm-backward apply) data-gradient kernels, whose second slot and
// fused-reduction epilogue already fill the register file, spill 40-250 B per lane and lose 8-20 %.
__host__ __device__ constexpr bool persist_mode(int am, int bmd, bool f16, int tfa) {
    return f16 && bmd == M_PS && am <= M_KC_G2 && tfa != 2 && tfa != 3;
}
// (the persistent variants carry the next tile's A slot through the epilogue: held to two waves per SIMD = 256 registers)
// ACT = KoafGemm.act16: which tensors of this call are bf16 ACTIVATIONS (0: none; 1 forward: A.ptr and C; 2 data gradient:
// A.ptr2 (the conv output c of a tf-2 apply) and the BatchNorm-backward operands of the epilogue; 3 weight gradient: A.ptr2 and B.ptr)
// SD (M_KS only): k-tiles of its rows a wave keeps in flight; the host picks one that divides the number of k-steps
template <int BM, int BN, int AM, int BMD, int TFA, int TFB, bool VEC, bool F16, int NT = 256, int ACT = 0, bool EMIT = false, int SD = 0>
__global__ void __launch_bounds__(NT, (persist_mode(AM, BMD, F16, TFA) || AM == M_PT || AM == M_KS) ? 2 : 1) koaf_gemm_kernel(const KoafGemm p) {
    static_assert(ACT == 0 || VEC, "bf16 activation storage needs the vector path");
    constexpr bool C16 = (ACT == 1), E16 = (ACT == 2);
    static_assert((TFA < 2 && TFB < 2) || VEC, "the two-source prologues need the vector path");
    constexpr int NPL = F16 ? 2 : 3;
    static_assert(BMD != M_PS || F16, "plane images are fp16");
    constexpr bool AS = (AM == M_KS);                // the streamed dense A operand: waves 4 x 1, each on its own 32 rows (StreamA)
    static_assert(!AS || (BM == 128 && NT == 256 && BMD == M_PS && F16 && VEC && ACT == 0 && (SD == 2 || SD == 4) && TFB == 0),
                  "the streamed A operand's one shape");
    constexpr int WGN = AS ? 1 : 2;
    constexpr int NW = NT / 64, WGM = NW / WGN;                      // waves: WGM along M x WGN along N
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr bool AKC = mode_is_kc(AM), BKC = mode_is_kc(BMD), BPS = (BMD == M_PS), APS = mode_is_pa(AM), AH = (AM == M_PH);
    constexpr bool AT = (AM == M_PT);                // 3x3 over plane images in 8 x 16 pixel tiles: halo in LDS, weight fragments in registers
    static_assert(!AT || (BM == 128 && BN == 64 && NT == 256 && BMD == M_PS && TFA == 0 && F16 && VEC), "the 2-D tile kernel's one shape");
    constexpr bool WPS = (AM == M_PK);               // weight gradient from plane images: both operands K-major by LDS-DMA
    static_assert(WPS == (BMD == M_PKG || BMD == M_PK), "K-major plane images come in pairs");
    static_assert(!(APS || AH || AT) || (BPS && TFA == 0), "a pre-split A pairs with a pre-split B and carries its transform in the image");
    static_assert(NT == 256 || AH, "only the halo kernel runs 512 threads (the fp32 loaders are laid out for 256)");
    static_assert(!AH || (BM == 256) == (NT == 512), "halo shapes: 256 rows x 512 threads, 128 rows x 256 threads");
    constexpr int HP_MAX = (BM + 2 * halo_max_w(BN, BM) + 2 + 15) / 16;  // 16-pixel (1 KiB) pieces of a halo plane
    constexpr bool HDB = (NT == 512);                // halo double-buffered across channel chunks (the 256-row shape)
    constexpr int A_PL = AH ? HP_MAX * 256 : ((APS || WPS) ? BM * 16 : plane_dwords(BM, AKC));
    constexpr int B_PL = (BPS || WPS) ? BN * 16 : plane_dwords(BN, BKC);
    constexpr int A_ELEMS = NPL * A_PL, B_ELEMS = NPL * B_PL;
    constexpr int NBA = (APS || (AH && HDB) || WPS) ? 2 : 1;                                  // LDS buffers per operand
    constexpr int NBB = AH ? halo_b_stages(BN, BM) : (AS ? 3 : ((BPS || WPS) ? 2 : 1));
    constexpr int LDC_S = BN + 4;                                    // epilogue staging row (floats)
    // the 256-row halo kernel multiplies in 16 x 16 x 32 MFMAs (the same FLOPs, LDS bytes and issue cycles as 32 x 32 x 16; the chip
    // clocks them higher under sustained load: scripts/mfma_shapes.hip, 1.12-1.14 x): its accumulators are NRB x NCB tiles of v4f
    constexpr bool M16 = (AM == M_PH) && (NT == 512);
    constexpr int NRB = M16 ? WM / 16 : 1, NCB = M16 ? WN / 16 : 1;
    constexpr int C_ELEMS = VEC ? BM * LDC_S : 0;
    constexpr int OPS = NBA * A_ELEMS + NBB * B_ELEMS;
    // M_PT: the epilogue's staging tile (which the two weight-tile stages of the k-loop share) and the halo image (180 pixels x 64
    // channels x two fp16 planes = 45 KiB) sit side by side: 80 960 B with the 64 B of block_amax_raise_bits -- two blocks per CU
    constexpr int T2D_HALO_BYTES = 2 * 180 * 128;
    constexpr int SMEM = AT ? (C_ELEMS + T2D_HALO_BYTES / 4) : ((OPS > C_ELEMS) ? OPS : C_ELEMS);
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    __shared__ __attribute__((aligned(16))) float s_tab[(AS && TFA == 1) ? 2 * STREAM_TAB_K : 4];     // M_KS, tf 1: see StreamA::tab

    KOAF_STAMP_DECL;
    KOAF_STAMP(0);
    const int ntn = (p.N + BN - 1) / BN;
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MiB L2): without a remap the ntn blocks that
    // share an A row tile land on ntn different L2s and the tile is fetched from beyond L2 ntn times.  Bijective remap:
    // XCD x works through one contiguous chunk of the tile order, so a row tile's blocks follow each other on one L2.
    // PERSIST (fp32 A loader + weight tiles by DMA: the 1x1 and stride-2 convolutions, whose k-loops are 2-32 steps): the
    // block walks the tiles vt = blockIdx.x, + gridDim.x, ... (the host launches 2 blocks per CU) and issues the NEXT tile's
    // first A loads before the epilogue of the current one, so their HBM latency runs under the staging / stores instead
    // of in front of the next k-loop.  All other variants run their single tile through the same loop.
    constexpr bool PERSIST = persist_mode(AM, BMD, F16, TFA) || (AS && TFA < 2 && SD == 2);      // (M_KS with four k-tiles in flight: one tile per block)
    const unsigned ntx = (unsigned)((p.M - p.m_base + BM - 1) / BM) * (unsigned)ntn;     // tiles of one (split, batch) slice
    auto decode = [&](unsigned v, int& tm_, int& tn_) {
        const unsigned q = ntx >> 3, rem = ntx & 7, x = v & 7, j = v >> 3;
        const unsigned b = x * q + (x < rem ? x : rem) + j;
        tn_ = (int)(b % (unsigned)ntn);
        tm_ = (int)(b / (unsigned)ntn);
    };
    unsigned vt = blockIdx.x;
    unsigned bid = blockIdx.x;
    int split = blockIdx.y;
    if (gridDim.y > 1 && (gridDim.y & 7) == 0) {
        // Split-K (weight gradients): the tiles of ONE k-range read the same pixels of both operands, so they should share an
        // L2 -- left alone, the handful of tiles of a split are dealt to different XCDs and every one of them fetches its
        // operands from HBM again (a 3x3 weight gradient re-read its inputs 5-9 times).  Dispatch order is x-fastest:
        // XCD = linear id % 8; XCD x takes the splits = x (mod 8), all tiles of a split in consecutive slots.
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, x = lin & 7, slot = lin >> 3;
        bid = slot % gridDim.x;
        split = (int)((slot / gridDim.x) * 8 + x);
    }
    int tn = bid % ntn, tm = bid / ntn;
    if (!(gridDim.y > 1 && (gridDim.y & 7) == 0)) decode(vt, tm, tn);
    int m0 = p.m_base + tm * BM, n0 = tn * BN;
    const int z0 = blockIdx.z / p.nb1, z1 = blockIdx.z - z0 * p.nb1;
    const int kchunk = (((p.K + p.splitk - 1) / p.splitk + BK - 1) / BK) * BK;
    const int kbeg = split * kchunk;
    const int kend = min(p.K, kbeg + kchunk);

    // operand scales of the fp16 scheme (powers of two; 1 otherwise): applied on load, divided out in the epilogue
    const float sca = F16 ? operand_scale(p.A) : 1.f;
    const float scb = F16 ? operand_scale(p.B) : 1.f;
    float alpha = F16 ? p.alpha / (sca * scb) : p.alpha;
    if constexpr (F16) {
        // a NaN / Inf anywhere in an operand reaches its amax scalar (the reductions propagate them, koaf_common.h); the pieces
        // themselves are clamped to the fp16 range, so the whole OUTPUT is made NaN here: a diverged run shows as one
        const bool bad = (p.A.amax && !koaf_bits_finite(koaf_absbits(*p.A.amax))) || (p.B.amax && !koaf_bits_finite(koaf_absbits(*p.B.amax)));
        if (bad) {
            alpha = __uint_as_float(0x7fc00000u);
            if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) koaf_status_add(p.status, 1, 1u);
        }
    }

    // (batch offsets count elements: a bf16 tensor behind a float-typed pointer advances by half the bytes)
    auto eoff = [](const float* q, int64_t elems, bool h16) { return h16 ? (const float*)((const unsigned short*)q + elems) : q + elems; };
    const float* Ap = (APS || AH || AT || WPS) ? nullptr : eoff(p.A.ptr, z0 * p.A.bs0 + z1 * p.A.bs1, ACT == 1);
    const unsigned short* Apl = (APS || AH || AT || WPS) ? p.A.planes + z0 * p.A.bs0 + z1 * p.A.bs1 : nullptr;
    const float* Bp = (BPS || WPS) ? nullptr : eoff(p.B.ptr, z0 * p.B.bs0 + z1 * p.B.bs1, ACT == 3);
    const unsigned short* Bpl = (BPS || WPS) ? p.B.planes + z0 * p.B.bs0 + z1 * p.B.bs1 : nullptr;

    // (the unused ones of the loaders are dead code to the compiler)
    TileLoader<(APS || AH || AT) ? 128 : BM, (APS || AH || AT || WPS || AS) ? M_KC : AM, AS ? 0 : TFA, VEC, F16, ACT == 1,
               ((ACT == 2 || ACT == 3) && TFA == 2) || (ACT == 1 && TFA == 3)> la;
    TileLoader<BN, (BPS || WPS) ? M_KC : BMD, TFB, VEC, F16, ACT == 3> lb;
    PlaneKLoader<WPS ? BM : 128, false> wka;
    PlaneKLoader<BN, BMD == M_PKG> wkb;
    PlaneLoader<BN> lp;
    PlaneGatherLoader<AH ? 128 : BM, AM == M_PA2 ? 2 : 1> lpa;
    StreamA<AS ? TFA : 0, AS ? SD : 2> st;
    if constexpr (AS) {
        // (set up below, once the first tile is known)
    } else if constexpr (WPS) {
        wka.init(p.A, m0, p.M);
        wka.seek(p.A, kbeg);
        wkb.init(p.B, n0, p.N);
        wkb.seek(p.B, kbeg);
    } else if constexpr (APS) {
        lpa.init(p.A, m0, p.M);
        lpa.seek(p.A, kbeg);
    } else if constexpr (!AH && !AT) {
        la.init(p.A, m0, p.M, z1, sca);
        la.seek(p.A, kbeg);
    }
    if constexpr (AH || AT || WPS) {
        // (the halo loop below addresses both operands itself; the K-major pair was set up above)
    } else if constexpr (BPS) {
        lp.init(p.B, n0, p.N);
        lp.seek(p.B, kbeg);
    } else {
        lb.init(p.B, n0, p.N, z1, scb);
        lb.seek(p.B, kbeg);
    }

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wm = w / WGN, wn = w % WGN;
    const int r = lane & 31, h = lane >> 5;
    if constexpr (PERSIST && !AS) {
        if (kbeg < kend) la.issue(la.sa, p.A, Ap, kbeg, kend, z1);      // the first tile's A loads
    }
    // M_KS: the tile after this one (the cursor of the A stream crosses into it SD k-steps before this tile's k-loop ends)
    [[maybe_unused]] bool s_has_next = false;
    [[maybe_unused]] int s_tm2 = 0, s_tn2 = 0, s_m0n = -1;
    auto stream_next = [&]() {
        s_has_next = PERSIST && (vt + gridDim.x) < ntx;
        s_m0n = -1;
        if (s_has_next) { decode(vt + gridDim.x, s_tm2, s_tn2); s_m0n = p.m_base + s_tm2 * BM; }
    };
    [[maybe_unused]] unsigned c_vt = blockIdx.x;      // the tile the A stream's cursor is on
    auto cursor_next = [&]() -> int {
        if (!PERSIST || c_vt + gridDim.x >= ntx) return -1;
        c_vt += gridDim.x;
        int a, b;
        decode(c_vt, a, b);
        return p.m_base + a * BM;
    };
    if constexpr (AS) {
        st.init(p.A, Ap, m0, p.M, kbeg, kend, sca);
        st.once = (ntn == 1) && p.prec != 55;      // (prec is informational to the kernels; the launch code writes 55 there for KOAF_STREAM_NT=0)
        if constexpr (TFA == 1) {
            for (int k = t; k < kend - kbeg; k += NT) { s_tab[k] = p.A.sc[kbeg + k] * sca; s_tab[STREAM_TAB_K + k] = p.A.sh[kbeg + k] * sca; }
            st.tab = s_tab;
            __syncthreads();
        }
        stream_next();
#pragma unroll
        for (int d = 0; d < SD; ++d) st.issue(st.sl[d], cursor_next, p.M);      // the first SD k-tiles of this block's first tile
    }
    for (;;) {      // the tiles of this block (one, unless PERSIST)
    KOAF_STAMP(0);
    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    [[maybe_unused]] v4f acc16[NRB][NCB];
#pragma unroll
    for (int i = 0; i < NRB; ++i)
#pragma unroll
        for (int j = 0; j < NCB; ++j) acc16[i][j] = (v4f){0.f, 0.f, 0.f, 0.f};

    float* const Bs0 = smem + NBA * A_ELEMS;
    const unsigned sm0 = KOAF_LDS_ADDR(smem), sb0 = sm0 + NBA * A_ELEMS * 4;     // LDS byte addresses of the A / B buffers
    [[maybe_unused]] int t2d_base = 0, t2d_w = 0;                               // M_PT: first pixel of the tile's rectangle, image row pitch
    // the MFMAs of one k-tile whose plane images sit at Au / Bu
    auto mma = [&](const unsigned* Au, const unsigned* Bu) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            v4i ap[TM][NPL], bp[NPL];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    if constexpr (APS) ap[i][q] = frag_load_ps(Au + q * A_PL, wm * WM + 32 * i, g, lane);
                    else if constexpr (WPS) ap[i][q] = frag_load_kmd<BM>(Au + q * A_PL, wm * WM + 32 * i, g, lane);
                    else ap[i][q] = frag_load<BM, AKC>(Au + q * A_PL, wm * WM + 32 * i, g, lane);
                }
#pragma unroll
            for (int jn = 0; jn < TN; ++jn) {
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    if constexpr (BPS) bp[q] = frag_load_ps(Bu + q * B_PL, wn * WN + 32 * jn, g, lane);
                    else if constexpr (WPS) bp[q] = frag_load_kmd<BN>(Bu + q * B_PL, wn * WN + 32 * jn, g, lane);
                    else bp[q] = frag_load<BN, BKC>(Bu + q * B_PL, wn * WN + 32 * jn, g, lane);
                }
                // piece products, smallest first.  bf16: the six of weight >= 2^-16.  fp16: lo*hi, hi*lo, hi*hi.
                constexpr int NTERM = F16 ? 3 : 6;
                constexpr int PA3[6] = {2, 0, 1, 1, 0, 0}, PB3[6] = {0, 2, 1, 0, 1, 0};
                constexpr int PAH[3] = {1, 0, 0}, PBH[3] = {0, 1, 0};
#pragma unroll
                for (int term = 0; term < NTERM; ++term) {
                    const int pa = F16 ? PAH[term] : PA3[term];
                    const int pb = F16 ? PBH[term] : PB3[term];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        if constexpr (F16)
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, ap[i][pa]),
                                                                                __builtin_bit_cast(h16x8, bp[pb]),
                                                                                acc[i][jn], 0, 0, 0);
                        else
                            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[i][pa]),
                                                                                 __builtin_bit_cast(bf16x8, bp[pb]),
                                                                                 acc[i][jn], 0, 0, 0);
                    }
                }
            }
        }
    };
    if constexpr (AH) {
        // 3x3 / stride 1 / pad 1 over activation plane images, both operands by LDS-DMA.  The tile's BM output pixels are
        // consecutive in raster order, so the source pixels of ALL nine taps lie in the contiguous range
        // [m0 - W - 1, m0 + BM + W + 1): that halo (32 channels of it) is fetched ONCE per channel chunk and tap (dy, dx)
        // is the same LDS image read W*dy + dx pixels further on -- the input tile travels L2 -> LDS 1.8 times instead of
        // nine.  Taps that fall off the image (the raster neighbour is then another row or image) are zeroed in the
        // fragment registers by per-row validity bits.  k runs (chunk, tap, channel); the weight tile of every (chunk, tap)
        // step is double-buffered as in the loops above, the next chunk's halo arrives in ninths under the nine tap steps.
        static_assert(!HDB || 2 * HP_MAX <= 14 * NW, "the next halo is spread over seven tap steps, at most two pieces per wave and step");
        constexpr int NPB = BN / 16;                        // 1-KiB pieces of a weight plane tile
        constexpr int BPW = (2 * NPB + NW - 1) / NW;        // weight pieces per wave and step
        const int Wd = p.A.W, Hd = p.A.H, CSa = p.A.CS, Ca = p.A.C;
        const bool flip = p.A.gather == 2;
        const int np2 = 2 * ((BM + 2 * Wd + 2 + 15) >> 4);  // halo pieces per chunk (two planes)
        const int64_t hbase = (int64_t)m0 - Wd - 1, plast = (int64_t)p.M - 1;
        const int nchunk = Ca / 32;
        const int swz = 8 * ((lane & 3) ^ ((lane >> 4) & 3));
        // validity bits of the nine taps (bit kh*3+kw) and halo pixel of the centre tap for this lane's row of each M tile
        constexpr int NVB = HDB ? WM / 16 : TM;             // row blocks of a wave: 16 rows (256-row shape: 16 x 16 x 32 MFMAs) or 32
        unsigned vb[NVB];
        int i0[NVB];
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int lrow = HDB ? wm * WM + 16 * i + (lane & 15) : wm * WM + 32 * i + r, row = m0 + lrow;
            i0[i] = lrow + Wd + 1;
            vb[i] = 0;
            if (row < p.M) {
                const int rem = row % (Hd * Wd), y = rem / Wd, x = rem - y * Wd;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int dy = flip ? 1 - tp / 3 : tp / 3 - 1, dx = flip ? 1 - tp % 3 : tp % 3 - 1;
                    const bool ok = (unsigned)(y + dy) < (unsigned)Hd && (unsigned)(x + dx) < (unsigned)Wd;
                    vb[i] |= (ok ? 1u : 0u) << tp;
                }
            }
        }
        int64_t bsrc[BPW];
#pragma unroll
        for (int j = 0; j < BPW; ++j) {
            const int idx = w + NW * j, piece = idx % NPB;
            const int row = min(n0 + 16 * piece + (lane >> 2), p.N - 1);     // (rows past N: finite values, never stored)
            bsrc[j] = (int64_t)(idx / NPB) * p.B.plane_stride + (int64_t)row * p.B.ld + swz;
        }
        auto issue_halo = [&](int idx, int chunk, unsigned buf) {
            const int piece = idx >> 1, q = idx & 1;
            int64_t gp = hbase + 16 * piece + (lane >> 2);
            gp = gp < 0 ? 0 : (gp > plast ? plast : gp);       // (pixels off the tensor are never valid taps)
            const unsigned short* src = Apl + q * p.A.plane_stride + gp * CSa + (chunk * 32 + swz);
            lds_dma16(src, buf + q * (A_PL * 4) + piece * 1024);
        };
        auto issue_b = [&](int tap, int chunk, unsigned buf) {
            const int koff = tap * Ca + chunk * 32;
#pragma unroll
            for (int j = 0; j < BPW; ++j) {
                const int idx = w + NW * j;
                if (idx < 2 * NPB)
                    lds_dma16(Bpl + bsrc[j] + koff, buf + (idx / NPB) * (B_PL * 4) + (idx % NPB) * 1024);
            }
        };
        if constexpr (HDB) {
        // 256-row shape.  Software pipeline over the steps s = (chunk, tap):
        //   * weight tiles: three LDS stages; tile s + 3 is issued at step s into the stage tile s leaves;
        //   * fragments: the registers of step s + 1 are read from LDS DURING the MFMAs of step s (two register sets,
        //     ping-pong), so the matrix pipe never waits for an LDS read burst -- with one barrier per step all eight waves
        //     used to read, then multiply, in lockstep, and the k-loop ran at half the matrix rate;
        //   * the next chunk's halo arrives under taps 0..6 of the current chunk (its first fragments are read at tap 8);
        //   * waits are counted (loads retire in order): barrier(s) needs tile s + 1, issued two steps earlier, and lets
        //     everything issued since stay in flight -- no drain at chunk boundaries.
        static_assert(NBB == 3, "three weight-tile stages");
        typedef const __attribute__((address_space(3))) v4i* lds_v4i;
        typedef const __attribute__((address_space(3))) unsigned* lds_u;
        constexpr int NST = 7;                              // taps that carry pieces of the next halo
        const int nh = (np2 + NW - 1) / NW;                 // halo pieces per wave and chunk
        const int hq = nh / NST, hr = nh - hq * NST;        // pieces at tap t < NST: hq + (t < hr)   (<= 2: np2 <= 14 NW)
        const int nstep = 9 * nchunk;
        // fragments of one step (32 channels of one tap) for 16 x 16 x 32 MFMAs: lane (r = l % 16, c = l / 16) holds k = 8 c .. 8 c + 7 of
        // row r of its block -- 16-B chunk c of the pixel / weight row, one read per block and plane
        struct Frags { v4i a[NRB][2]; v4i b[NCB][2]; };
        const int q4 = lane >> 4;
        // (shift: the tap's pixel offset in the halo, (kh - 1) W + (kw - 1), negated for the data gradient's flipped filter -- carried
        // by the loop, not derived from the tap)
        auto load_frags = [&](Frags& f, int shift, int hbuf, int stage) {
            const lds_u Ah = (lds_u)smem + hbuf * A_ELEMS;
            const lds_u Bu = (lds_u)smem + NBA * A_ELEMS + stage * B_ELEMS;
#pragma unroll
            for (int i = 0; i < NRB; ++i) {
                const int hp = i0[i] + shift;
                const int off = hp * 16 + 4 * (q4 ^ ((hp >> 2) & 3));
#pragma unroll
                for (int q = 0; q < 2; ++q) f.a[i][q] = *(lds_v4i)&Ah[q * A_PL + off];
            }
#pragma unroll
            for (int j = 0; j < NCB; ++j) {
                const int brow = wn * WN + 16 * j + (lane & 15);
#pragma unroll
                for (int q = 0; q < 2; ++q) f.b[j][q] = *(lds_v4i)&Bu[q * B_PL + brow * 16 + 4 * (q4 ^ ((brow >> 2) & 3))];
            }
        };
        for (int idx = w; idx < np2; idx += NW) issue_halo(idx, 0, sm0);
#pragma unroll
        for (int d = 0; d < 3; ++d) issue_b(d, 0, sb0 + d * (B_ELEMS * 4));       // (nstep >= 9)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * BPW) : "memory");             // halo 0 and tile 0 have landed
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        KOAF_STAMP(1);
        Frags F0, F1;
        const int shift0 = flip ? Wd + 1 : -Wd - 1, dshift = flip ? -1 : 1, rshift = flip ? 2 - Wd : Wd - 2;     // tap 0; to the next tap in a row / to the next row
        load_frags(F0, shift0, 0, 0);
        int nshift = shift0, nkw = 0;       // of tap + 1 (advanced below)
        int tap = 0, chunk = 0, sb = 0;     // this step; sb = stage of its weight tile
        int itap = 3, ich = 0;              // (tap, chunk) of tile s + 3
        int hk = 0, hprev = 0;              // halo pieces of the next chunk issued so far / loads per wave at the previous step
        [[maybe_unused]] unsigned long long kst_hw = 0, kst_hv = 0;
        // this wave's halo pieces inside the loop: slot hk of a chunk is piece 4 hk + w / 2 of plane w % 2 (idx = 8 hk + w above), so the
        // plane, the lane's pixel offset and its 64-bit base are per-tile constants and a piece costs a clamp and one multiply-add
        // (slots past the last piece repeat it)
        const int ws = __builtin_amdgcn_readfirstlane(w);
        const int h_npp = np2 >> 1, h_p0 = ws >> 1;
        const int h_px = (int)hbase + (lane >> 2), h_last = (int)plast;
        const unsigned short* const h_src = Apl + (ws & 1) * p.A.plane_stride + swz;
        const unsigned h_dst = (ws & 1) * (A_PL * 4);
        auto issue_halo_w = [&](int k, int chunk, unsigned buf) {
            const int piece = min(4 * k + h_p0, h_npp - 1);
            const int gp = min(max(h_px + 16 * piece, 0), h_last);
            lds_dma16(h_src + ((int64_t)gp * CSa + chunk * 32), buf + h_dst + piece * 1024);
        };
        auto vm_wait = [&](int h) {
            if (h == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BPW) : "memory");
            else if (h == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BPW + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BPW + 2) : "memory");
        };
        auto lgkm0_barrier = [&]() {
            // (the wait is the BUILTIN: hipcc's wait-count pass does not read inline assembly and would take the fragment registers
            // of step s for still in flight at their MFMAs)
            __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
            asm volatile("s_barrier" ::: "memory");
        };
        // One step = four (two: 64 columns) groups of twelve MFMAs (one 16-column block), each followed by a piece of everything else (the
        // fragment reads of step s + 1, the two halo pieces, the weight tile), pinned by scheduling barriers: a wave issues in
        // order, and the address arithmetic in one block in front of 24 back-to-back MFMAs ran with the matrix pipe idle in both
        // waves of the SIMD (the step barrier keeps them in lockstep).
        auto mask_all = [&](Frags& f) {       // (in place: the set is dead after its step)
#pragma unroll
            for (int i = 0; i < NRB; ++i) {
                const int okm = -(int)((vb[i] >> tap) & 1u);       // all ones / zero: the tap's validity as an AND mask
#pragma unroll
                for (int q = 0; q < 2; ++q) f.a[i][q] = and_mask(f.a[i][q], okm);
            }
        };
        auto mma_col = [&](Frags& f, int j) {      // column block j: 3 NRB MFMAs
            constexpr int PAH[3] = {1, 0, 0}, PBH[3] = {0, 1, 0};
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int i = 0; i < NRB; ++i)
                    acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, f.a[i][PAH[term]]),
                                                                          __builtin_bit_cast(h16x8, f.b[j][PBH[term]]), acc16[i][j], 0, 0, 0);
        };
        constexpr int NQ = NCB;             // MFMA groups of a step
        static_assert(NQ == 4 || NQ == 2, "the four pieces follow the MFMA groups in ones or twos");
        auto body = [&](Frags& cur, Frags& nxt) {
            [[maybe_unused]] const unsigned long long tw0 = KOAF_STAMP_NOW();
            // tile s + 1 (issued at step s - 2, the last loads of that step) has landed; what step s - 1 issued stays in flight
            vm_wait(hprev);
            lgkm0_barrier();
            [[maybe_unused]] const unsigned long long tw1 = KOAF_STAMP_NOW();
            const bool more_chunks = chunk + 1 < nchunk;
            const int hc = (more_chunks && tap < NST) ? hq + (tap < hr ? 1 : 0) : 0;
            const unsigned Anext = sm0 + ((chunk + 1) & 1) * (A_ELEMS * 4);
            hprev = hc;
            int ntap = tap + 1, nch2 = chunk;
            if (++nkw == 3) { nkw = 0; nshift += rshift; } else nshift += dshift;
            if (ntap == 9) { ntap = 0; ++nch2; nshift = shift0; }
            int sbn = sb + 1;
            if (sbn == 3) sbn = 0;
            auto piece = [&](int f) {
                __builtin_amdgcn_sched_barrier(0);
                // (past the last step the reads fetch a stage / halo nobody uses: unconditional, so that the two register sets stay two)
                if (f == 0) load_frags(nxt, nshift, nch2 & 1, sbn);
                if (f == 1 && hc > 0) { issue_halo_w(hk, chunk + 1, Anext); ++hk; }
                if (f == 2 && hc > 1) { issue_halo_w(hk, chunk + 1, Anext); ++hk; }
                // tile s + 3 into the stage tile s leaves (past the end: re-fetch the last tile there -- nobody reads it, the counts stay uniform)
                if (f == 3) issue_b(ich < nchunk ? itap : 8, ich < nchunk ? ich : nchunk - 1, sb0 + sb * (B_ELEMS * 4));
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll
            for (int m = 0; m < NQ; ++m) {
                if (m == 0) mask_all(cur);
                mma_col(cur, m);
#pragma unroll
                for (int f = m * (4 / NQ); f < (m + 1) * (4 / NQ); ++f) piece(f);
            }
            if (++itap == 9) { itap = 0; ++ich; }
            if (ntap == 0) hk = 0;
            sb = sbn;
            [[maybe_unused]] const unsigned long long tw2 = KOAF_STAMP_NOW();
            kst_hv += tw1 - tw0;
            kst_hw += tw2 - tw1;
            tap = ntap; chunk = nch2;
        };
#pragma unroll 1
        for (int s2 = 0; s2 + 1 < nstep; s2 += 2) {
            body(F0, F1);
            body(F1, F0);
        }
        if (nstep & 1) body(F0, F1);
        KOAF_STAMP_ACC(5, kst_hv);
        KOAF_STAMP_ACC(6, kst_hw);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the surplus fetches behind the last step)
        __syncthreads();       // the epilogue reuses the operand buffers
        } else {
        // Pipeline: the weight tile of step s + D (D = NBB - 1 steps ahead) and one halo piece of the NEXT chunk are issued
        // at step s; loads retire in order, so "the tile of step s has landed" is a counted wait that leaves the younger
        // loads in flight.  Only the first tap of a chunk drains everything (its halo was completed by the previous step).
        constexpr int D = NBB - 1;
        auto step_of = [&](int sidx, int& tp, int& ch) { ch = sidx / 9; tp = sidx - 9 * ch; };
        const int nstep = 9 * nchunk;
        for (int idx = w; idx < np2; idx += NW) issue_halo(idx, 0, sm0);
#pragma unroll
        for (int d = 0; d < D; ++d)
            if (d < nstep) { int tp, ch; step_of(d, tp, ch); issue_b(tp, ch, sb0 + d * (B_ELEMS * 4)); }
        int sb = 0;                 // stage holding the current step's weight tile
        int ntap = D % 9, nch = D / 9;     // (tap, chunk) of step s + D
        for (int chunk = 0; chunk < nchunk; ++chunk) {
            // (LDS pointers typed as such: left generic, hipcc could not always prove the address space of these reads)
            typedef const __attribute__((address_space(3))) v4i* lds_v4i;
            typedef const __attribute__((address_space(3))) unsigned* lds_u;
            const lds_u Ah = (lds_u)smem + (HDB ? (chunk & 1) : 0) * A_ELEMS;
            const unsigned Anext = sm0 + (HDB ? ((chunk + 1) & 1) : 0) * (A_ELEMS * 4);
            const bool more_chunks = chunk + 1 < nchunk;
#pragma unroll 1
            for (int tap = 0; tap < 9; ++tap) {
                // in flight behind this step's tile: D - 1 younger tiles (BPW loads each) and, inside a chunk, D halo pieces
                if (tap == 0) {
                    KOAF_STAMP(4);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                else if (HDB && more_chunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * BPW + (D < 9 ? D : 9)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * BPW) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (tap == 0) {
                    KOAF_STAMP(5);
                    KOAF_STAMP_ADD(5, 4, 5);          // exposed wait for a chunk's halo (+ barrier skew)
                    if (chunk == 0) KOAF_STAMP(1);
                }
                {
                    int sd = sb + D;
                    if (sd >= NBB) sd -= NBB;
                    // (past the last step: re-fetch the last tile into a stage nobody reads, which keeps the counts uniform)
                    issue_b(nch < nchunk ? ntap : 8, nch < nchunk ? nch : nchunk - 1, sb0 + sd * (B_ELEMS * 4));
                    if (++ntap == 9) { ntap = 0; ++nch; }
                }
                if (HDB && more_chunks) {
                    const int idx = tap * NW + w;
                    issue_halo(idx < np2 ? idx : np2 - 1, chunk + 1, Anext);    // (surplus slots repeat the last piece)
                }
                const int kh = tap / 3, kw = tap - 3 * kh;
                const int shift = flip ? (1 - kh) * Wd + (1 - kw) : (kh - 1) * Wd + (kw - 1);
                const lds_u Bu = (lds_u)smem + NBA * A_ELEMS + sb * B_ELEMS;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    v4i ap[TM][2], bp[2];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int hp = i0[i] + shift;
                        const int okm = -(int)((vb[i] >> tap) & 1u);       // all ones / zero: the tap's validity as an AND mask
                        const int off = hp * 16 + 4 * ((2 * g + h) ^ ((hp >> 2) & 3));
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const v4i v = *(lds_v4i)&Ah[q * A_PL + off];
                            ap[i][q] = and_mask(v, okm);
                        }
                    }
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn) {
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int brow = wn * WN + 32 * jn + (lane & 31);
                            bp[q] = *(lds_v4i)&Bu[q * B_PL + brow * 16 + 4 * ((2 * g + h) ^ ((brow >> 2) & 3))];   // = frag_load_ps
                        }
                        constexpr int PAH[3] = {1, 0, 0}, PBH[3] = {0, 1, 0};
#pragma unroll
                        for (int term = 0; term < 3; ++term)
#pragma unroll
                            for (int i = 0; i < TM; ++i)
                                acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, ap[i][PAH[term]]),
                                                                                    __builtin_bit_cast(h16x8, bp[PBH[term]]),
                                                                                    acc[i][jn], 0, 0, 0);
                    }
                }
                if (++sb == NBB) sb = 0;
            }
            if (!HDB && more_chunks) {
                // one halo buffer: every wave is done with this chunk, then the next one is fetched whole (the first tap of
                // the next chunk waits for it; the CU's other block computes meanwhile)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                for (int idx = w; idx < np2; idx += NW) issue_halo(idx, chunk + 1, Anext);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the surplus fetches behind the last step)
        __syncthreads();       // the epilogue reuses the operand buffers
        }
    } else if constexpr (AT) {
        // 3x3 / stride 1 / pad 1 over activation plane images, 8 x 16 pixel tiles of ONE image (H % 8 == 0, W % 16 == 0).
        //   * Halo: the 10 x 18 source pixels of the tile, 64 channels at a time, both fp16 planes, are fetched ONCE by LDS-DMA;
        //     pixels outside the image fetch the zero chunk, so the k-loop needs no validity masks.  A raster tile of 128 pixels
        //     (M_PH) needs 128 + 2 W + 2 halo pixels -- 2.5 tiles' worth at W = 96; the rectangle needs 1.4 -- and that is what
        //     lets a 64-channel halo AND two blocks share a CU: one step = one filter tap over all 64 channels = 24 MFMAs per
        //     wave (M_PH, 128 rows: 12 per barrier).
        //   * LDS image: granule (16 B = 8 channels) c of halo pixel (y, x) at ((18 y + x) * 8 + (c ^ (x / 2 % 8))) * 16: the 16 lanes
        //     of a ds_read_b128 group hold x = x0 + {0..3, 12..15} of one tile row and x0 + {4..11} of the next, i.e. every
        //     residue mod 16 once -- (x % 2, x / 2 % 8) are 16 distinct (bank half, 16-B slot) pairs: conflict-free.  The DMA
        //     writes lane-linear, so the permutation is applied to the per-lane SOURCE address.
        //   * Weights: the 64 x 64 tile of one (tap, chunk) step by LDS-DMA, double-buffered INSIDE the epilogue's staging region
        //     (the two never live at the same time), one barrier per step.  (Loading the fragments straight into registers --
        //     no barrier at all -- was measured first: 64 KB per step and CU through the vector memory path in 32-B segments
        //     was the limiter, 245 TFLOP/s.)
        //   * k runs (chunk of 64 channels, tap, channel); C = 128 reloads the halo once (two barriers).
        typedef const __attribute__((address_space(3))) v4i* lds_v4i;
        typedef const __attribute__((address_space(3))) char* lds_c;
        const int Wd = p.A.W, Hd = p.A.H, CSa = p.A.CS, Ca = p.A.C;
        const bool flip = p.A.gather == 2;
        const int txn = Wd >> 4, tpi = (Hd >> 3) * txn;          // tiles per image row / per image
        const int img = tm / tpi, trem = tm - img * tpi, tyi = trem / txn, txi = trem - tyi * txn;
        const int nchunk = Ca >> 6;
        const unsigned halo0 = KOAF_LDS_ADDR(smem) + C_ELEMS * 4;
        t2d_base = (img * Hd + tyi * 8) * Wd + txi * 16;
        t2d_w = Wd;
        // this lane's rows of the two M tiles: pixel (ly, lx) of the tile; halo pixel of tap (ky, kx) = (ly + ky, lx + kx)
        const int lx = r & 15;
        int hp0[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) hp0[i] = (4 * wm + 2 * i + (r >> 4)) * 18 + lx;
        // weight tile of one step: 64 output channels x 64 k x two planes = 16 KiB = 16 DMA pieces, four per wave; granule c of row
        // `row` at (row * 8 + (c ^ (row / 2 % 8))) * 16 (the halo image's conflict-free pattern); two stages in the staging region
        const unsigned bst0 = KOAF_LDS_ADDR(smem);
        int64_t bsrc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int G = (w + 4 * j) * 64 + lane, q = G >> 9, Gp = G & 511, row = Gp >> 3, cs = Gp & 7;
            const int brow = min(n0 + row, p.N - 1);                // (rows past N: finite values, never stored)
            bsrc[j] = (int64_t)q * p.B.plane_stride + (int64_t)brow * p.B.ld + 8 * (cs ^ ((row >> 1) & 7));
        }
        auto issue_b = [&](int tap, int chunk, int stage) {
            const int koff = tap * Ca + chunk * 64;
#pragma unroll
            for (int j = 0; j < 4; ++j) lds_dma16(Bpl + bsrc[j] + koff, bst0 + stage * 16384 + (w + 4 * j) * 1024);
        };
        // fragments of k-group g of filter tap `tap`: A of both M tiles, B of this wave's 32 columns, both planes
        struct FR { v4i a[TM][2]; v4i b[2]; };
        const int browl = wn * WN + r, bsw = (browl >> 1) & 7;
        auto read_f = [&](FR& f, int tap, int g, int stage) {
            const int kh = tap / 3, kw = tap - 3 * kh;
            const int ky = flip ? 2 - kh : kh, kx = flip ? 2 - kw : kw;
            const int sw = ((lx + kx) >> 1) & 7;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const unsigned off = (unsigned)((hp0[i] + ky * 18 + kx) * 128 + (((2 * g + h) ^ sw) << 4));
#pragma unroll
                for (int q = 0; q < 2; ++q) f.a[i][q] = *(lds_v4i)((lds_c)smem + (C_ELEMS * 4 + q * 23040) + off);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q)
                f.b[q] = *(lds_v4i)((lds_c)smem + (stage * 16384 + q * 8192 + browl * 128 + (((2 * g + h) ^ bsw) << 4)));
        };
        auto mma_g = [&](const FR& f) {
            constexpr int PAH[3] = {1, 0, 0}, PBH[3] = {0, 1, 0};
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, f.a[i][PAH[term]]),
                                                                        __builtin_bit_cast(h16x8, f.b[PBH[term]]),
                                                                        acc[i][0], 0, 0, 0);
        };
        static_assert(TM == 2 && TN == 1 && C_ELEMS * 4 >= 2 * 16384, "2 x 2 waves of 64 x 32; two weight stages inside the staging region");
        FR R0, R1;
        t2d_issue_halo(p.A, Apl, tm, 0, halo0);
        issue_b(0, 0, 0);
        const int nstep = 9 * nchunk;
        int tap = 0, chunk = 0;
        // One step = one filter tap over 64 channels = four k-groups of 6 MFMAs per wave, one barrier.  Pinned with scheduling
        // barriers (left alone, hipcc sinks every LDS read to just in front of its first use): the fragments of k-group g + 1
        // are read while group g is multiplied; the next step's weight tile lands under this step's MFMAs.
#pragma unroll 1
        for (int s_ = 0; s_ < nstep; ++s_) {
            const int stage = s_ & 1;
            int ntap = tap + 1, nch = chunk;
            if (ntap == 9) { ntap = 0; ++nch; }
            const bool has_next = s_ + 1 < nstep;
            // this step's weight tile (and, at s = 0, the halo) has landed; every wave is done with the other stage
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (s_ == 0) KOAF_STAMP(1);
            if (has_next && ntap != 0) issue_b(ntap, nch, stage ^ 1);
            read_f(R0, tap, 0, stage);
            __builtin_amdgcn_sched_barrier(0);
            read_f(R1, tap, 1, stage);
            __builtin_amdgcn_sched_barrier(0);
            mma_g(R0);
            __builtin_amdgcn_sched_barrier(0);
            read_f(R0, tap, 2, stage);
            __builtin_amdgcn_sched_barrier(0);
            mma_g(R1);
            __builtin_amdgcn_sched_barrier(0);
            read_f(R1, tap, 3, stage);
            __builtin_amdgcn_sched_barrier(0);
            mma_g(R0);
            __builtin_amdgcn_sched_barrier(0);
            mma_g(R1);
            __builtin_amdgcn_sched_barrier(0);
            if (has_next && ntap == 0) {
                // next 64 channels: every wave is done with this halo, then it is replaced together with the first weight tile
                // (the CU's other block computes meanwhile)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                t2d_issue_halo(p.A, Apl, tm, nch, halo0);
                issue_b(0, nch, stage ^ 1);
            }
            tap = ntap; chunk = nch;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();       // the epilogue's staging tile covers the weight stages
    } else if constexpr (AS) {
        // Streamed A (StreamA) x weight plane images by LDS-DMA through a three-stage ring.  Step s of the tile, per wave:
        //   consume A(s) -- the registers issued SD steps ago: transform, split, this wave's rows of the A image; wait for this
        //   wave's pieces of B(s) (issued at step s - 2: counted, what was issued since stays in flight) and meet the other waves:
        //   B(s) is complete and nobody still reads the stage of step s - 1, which B(s + 2) now overwrites; issue B(s + 2), then
        //   A(s + SD) into the registers A(s) left; fragments + MFMAs.
        // The A loads are ordinary loads (the compiler keeps their registers and waits for them itself); the LDS-DMA is inline
        // assembly it does not see, so its wait for A(s) also retires the (SD - 1) NB oldest operations issued after A(s) -- in this
        // order those are B(s - SD + 3), A(s + 1), B(s - SD + 4) ...: tiles already needed or needed next.  The manual waits
        // count only loads that are certainly issued (NLA data loads per A tile: coefficient loads and side stores make the true
        // count larger, which errs towards waiting longer).
        // (Measured and dropped: the transform + split of A(s + 1) cut into quarters between the MFMAs of step s -- pinned with
        // scheduling barriers, since hipcc otherwise puts every vector instruction behind the last MFMA -- was slower than this
        // order on every layer, 2002 against 1977 ms per step.)
        constexpr int NLA = (TFA == 2 || TFA == 3) ? 8 : 4;
        constexpr int NB = 2 * (BN / 64);                   // LDS-DMA instructions per wave and weight tile (two planes)
        const int nstep = (kend - kbeg) / BK;               // host: a multiple of SD
        unsigned* const Aim = (unsigned*)smem;
        float* const side = (TFA == 3 && n0 == 0) ? p.A.side : nullptr;
        lp.template issue<NPL>(p.B, Bpl, sb0);
        if (nstep > 1) lp.template issue<NPL>(p.B, Bpl, sb0 + (B_ELEMS * 4));
        // "at most n vector-memory operations of this wave still in flight", rounded down to an immediate of the ladder, + barrier
        auto wait_barrier = [&](int n) {
            if (n >= 2 * NLA + NB) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NLA + NB) : "memory");
            else if (n >= 2 * NLA) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * NLA) : "memory");
            else if (n >= NLA + NB) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NLA + NB) : "memory");
            else if (n >= NLA) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NLA) : "memory");
            else if (n >= NB) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        };
        // does step j issue an A tile?  (its target k-tile exists: later in this tile, or in the next tile of a persistent block)
        auto issues = [&](int j) { return j >= 0 && (j + SD < nstep || s_m0n >= 0); };
        int bst = 0;
        KOAF_STAMP(1);
        [[maybe_unused]] unsigned long long kst_c = 0, kst_w = 0;      // (stamps build: time in consume() incl. the wait for A; in the B wait + barrier)
        // one step; ISSUE: it fetches an A tile (compile-time: the steps that do and the steps that do not sit in two loops, because a
        // load issued on one path only makes hipcc count its waits for the path WITHOUT it -- every wait for A then retires nearly
        // everything in flight)
        auto sstep = [&](auto D, auto ISSUE, int ss) {
            constexpr int d = decltype(D)::value;
            [[maybe_unused]] const unsigned long long ta0 = KOAF_STAMP_NOW();
            st.consume(st.sl[d], Aim, kbeg + ss * BK, side);
            [[maybe_unused]] const unsigned long long ta1 = KOAF_STAMP_NOW();
            // younger than B(ss): the A tiles of steps ss - 2 and ss - 1, B(ss + 1)
            wait_barrier((issues(ss - 2) ? NLA : 0) + (issues(ss - 1) ? NLA : 0) + ((ss + 1 < nstep) ? NB : 0));
            kst_c += ta1 - ta0;
            kst_w += KOAF_STAMP_NOW() - ta1;
            if (ss + 2 < nstep) {
                int b2 = bst + 2;
                if (b2 >= 3) b2 -= 3;
                lp.template issue<NPL>(p.B, Bpl, sb0 + b2 * (B_ELEMS * 4));
            }
            if constexpr (decltype(ISSUE)::value) st.issue(st.sl[d], cursor_next, p.M);
            mma(Aim, (const unsigned*)(Bs0 + bst * B_ELEMS));
            if (++bst == 3) bst = 0;
        };
        static_assert(SD == 0 || SD == 2, "two steps per group");
        const int nmain = (s_m0n >= 0) ? nstep : nstep - SD;        // the steps that issue (all of them when a next tile follows)
        int s0 = 0;
#pragma unroll 1
        for (; s0 < nmain; s0 += SD) {
            sstep(std::integral_constant<int, 0>{}, std::true_type{}, s0);
            sstep(std::integral_constant<int, 1>{}, std::true_type{}, s0 + 1);
        }
        if (s0 < nstep) {
            sstep(std::integral_constant<int, 0>{}, std::false_type{}, s0);
            sstep(std::integral_constant<int, 1>{}, std::false_type{}, s0 + 1);
        }
        // every wave is done with the operand images (the epilogue's staging tile covers them); the next tile's A stays in flight
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        KOAF_STAMP_ACC(5, kst_w);
        KOAF_STAMP_ACC(6, kst_c);
        if constexpr (TFA == 1 || TFA == 3) {
            if (((st.satmax & 0xffffu) >= 0x7bffu) | ((st.satmax >> 16) >= 0x7bffu)) koaf_status_add(p.status, 0, 1u);
            st.satmax = 0u;
        }
    } else if constexpr (WPS) {
        // weight gradient: both K-major operands by LDS-DMA, double-buffered, one barrier per k-tile (as below)
        if (kbeg < kend) {
            wka.issue(p.A, Apl, kbeg, kend, sm0);
            wkb.issue(p.B, Bpl, kbeg, kend, sb0);
        }
        int cur = 0;
        KOAF_STAMP(1);
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if ((k0 + BK) < kend) {
                wka.issue(p.A, Apl, k0 + BK, kend, sm0 + (cur ^ 1) * (A_ELEMS * 4));
                wkb.issue(p.B, Bpl, k0 + BK, kend, sb0 + (cur ^ 1) * (B_ELEMS * 4));
            }
            mma((const unsigned*)(smem + cur * A_ELEMS), (const unsigned*)(Bs0 + cur * B_ELEMS));
            cur ^= 1;
        }
        __syncthreads();       // the epilogue reuses the operand buffers
    } else if constexpr (APS) {
        // both operands by LDS-DMA, double-buffered: one barrier per k-tile.  At the top of iteration t every wave waits for
        // its own pieces of tile t (issued one iteration ago, under the MFMAs of tile t-1) and meets the others: tile t is
        // complete and nobody still reads the buffers of tile t-1, which the DMA of tile t+1 now overwrites.
        if (kbeg < kend) {
            lpa.issue(p.A, Apl, sm0);
            lp.template issue<NPL>(p.B, Bpl, sb0);
        }
        int cur = 0;
        KOAF_STAMP(1);
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if ((k0 + BK) < kend) {
                lpa.issue(p.A, Apl, sm0 + (cur ^ 1) * (A_ELEMS * 4));
                lp.template issue<NPL>(p.B, Bpl, sb0 + (cur ^ 1) * (B_ELEMS * 4));
            }
            mma((const unsigned*)(smem + cur * A_ELEMS), (const unsigned*)(Bs0 + cur * B_ELEMS));
            cur ^= 1;
        }
        __syncthreads();       // the epilogue reuses the operand buffers
    } else {
    if (kbeg < kend) {
        if constexpr (BPS) lp.template issue<NPL>(p.B, Bpl, sb0);
        if constexpr (!PERSIST) la.issue(la.sa, p.A, Ap, kbeg, kend, z1);       // (PERSIST: in flight since the last tile's epilogue)
        if constexpr (!BPS) lb.issue(lb.sa, p.B, Bp, kbeg, kend, z1);
        if constexpr (TFA == 3) { la.side = (n0 == 0) ? p.A.side : nullptr; la.k0s = kbeg; }
        la.finish(la.sa);
        la.template store<NPL>(la.sa, smem);
        if constexpr (!BPS) {
            lb.finish(lb.sa);
            lb.template store<NPL>(lb.sa, Bs0);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's DMA pieces have landed
        }
    }
    __syncthreads();
    int cur = 0;
    KOAF_STAMP(1);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        // the next tile's global loads go out first: in flight under this tile's MFMAs (the DMA into the other B buffer,
        // which every wave stopped reading at the last barrier)
        if (more) {
            if constexpr (BPS) lp.template issue<NPL>(p.B, Bpl, sb0 + (cur ^ 1) * (B_ELEMS * 4));
            la.issue(la.sa, p.A, Ap, k0 + BK, kend, z1);
            if constexpr (!BPS) lb.issue(lb.sa, p.B, Bp, k0 + BK, kend, z1);
        }
        mma((const unsigned*)smem, (const unsigned*)(Bs0 + cur * B_ELEMS));
        // every wave is done reading the A image (and this B buffer).  With LDS-DMA in flight __syncthreads() would
        // drain vmcnt here, in the middle of the MFMA stream: a raw barrier behind the LDS-read wait keeps the next
        // tile's loads in flight until finish() needs them.
        if constexpr (BPS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
        if (more) {
            if constexpr (TFA == 3) la.k0s = k0 + BK;
            la.finish(la.sa);
            la.template store<NPL>(la.sa, smem);
            if constexpr (!BPS) {
                lb.finish(lb.sa);
                lb.template store<NPL>(lb.sa, Bs0);
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                cur ^= 1;
            }
            __syncthreads();
        }
    }
    }

    // the next tile of this block: ids, A loader state and its first loads -- before the epilogue, whose staging, loads
    // and stores they run under (the A slot registers are free here; the LDS is not: the staging tile covers the operand
    // buffers, so the weight tile's DMA has to wait for the end of the epilogue)
    bool has_next = false;
    int tm2 = 0, tn2 = 0;
    // (M_PT blocks are NOT persistent: requesting the next tile's halo under this tile's epilogue was measured -- the prologue
    // fell from 3.9 to 0.7 us per tile and the k-loops grew by as much: with two blocks per CU one block's prologue already runs
    // under the other's MFMAs)
    if constexpr (AS) {
        has_next = s_has_next; tm2 = s_tm2; tn2 = s_tn2;       // (its first SD k-tiles are already in flight)
    } else if constexpr (PERSIST) {
        has_next = (vt + gridDim.x) < ntx;
        if (has_next) {
            decode(vt + gridDim.x, tm2, tn2);
            la.init(p.A, p.m_base + tm2 * BM, p.M, z1, sca);
            la.seek(p.A, kbeg);
            if (kbeg < kend) la.issue(la.sa, p.A, Ap, kbeg, kend, z1);
        }
    }

    if constexpr (F16 && (TFA == 1 || TFA == 3) && AKC && !APS && !AH && !WPS && !AS) {
        if (((la.satmax & 0xffffu) >= 0x7bffu) | ((la.satmax >> 16) >= 0x7bffu)) koaf_status_add(p.status, 0, 1u);
        la.satmax = 0u;
    }
    // ---- epilogue ----
    KOAF_STAMP(2);
    float* Cp;
    int64_t ldc;
    const bool slab = p.splitk > 1;
    if (slab) {
        Cp = p.C + (int64_t)(blockIdx.z * p.splitk + split) * p.M * p.N;
        ldc = p.N;
    } else {
        Cp = const_cast<float*>(eoff(p.C, z0 * p.cbs0 + z1 * p.cbs1, C16));
        ldc = p.ldc;
    }
    const float* Rp = (p.residual && !slab) ? p.residual + z0 * p.rbs0 + z1 * p.rbs1 : nullptr;
    const float* bias = slab ? nullptr : p.bias;
    const bool do_stats = (p.stats != nullptr) && !slab;
    // (per 32-row band i of the wave's rows: the tile's sums are then the same tree whether its four bands sit in two waves or,
    // on the streamed kernels, in four -- band sums, lane halves, band pairs, pair of pairs)
    float s1[TM][TN], s2[TM][TN], kshift[TN];
    [[maybe_unused]] float t1[NCB], t2[NCB], kshift16[NCB];      // (M16: per 16-column block)
    if constexpr (M16) {
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            t1[j] = t2[j] = 0.f;
            const int scol = n0 + wn * WN + 16 * j + (lane & 15);
            kshift16[j] = (do_stats && p.stats_shift && scol < p.N) ? p.stats_shift[(int64_t)blockIdx.z * p.stats_bs + scol] : 0.f;
        }
    }
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
#pragma unroll
        for (int i = 0; i < TM; ++i) s1[i][jn] = s2[i][jn] = 0.f;
        // statistics are summed about a per-column shift (the BatchNorm's running mean): sum (v - k), sum (v - k)^2
        // lose nothing to cancellation when |mean| >> std, which sum v^2 - (sum v)^2 / n does
        const int scol = n0 + wn * WN + 32 * jn + r;
        kshift[jn] = (do_stats && p.stats_shift && scol < p.N) ? p.stats_shift[(int64_t)blockIdx.z * p.stats_bs + scol] : 0.f;
    }

    if constexpr (VEC) {
        // stage the accumulator tile through LDS so global stores (and residual / bias loads) are
        // 16 B per lane on full 512-B row segments instead of 4 B per lane.  (Measured alternatives: 4-B stores straight
        // from the accumulator registers -- two 128-B segments per wave store -- are 25-30 % slower on the output-bound 1x1
        // convolutions; staging in two 64-row halves to fit a third block per CU needs <= 168 VGPRs, which spills ~130
        // dwords per lane here and halves the speed.)
        float* Cs = smem;   // all waves passed the k-loop's last barrier: operand tiles are dead
        if constexpr (M16) {
            // 16 x 16 tiles: lane (c = l % 16, q = l / 16) holds column c, rows 4 q + e of its tile
#pragma unroll
            for (int i = 0; i < NRB; ++i)
#pragma unroll
                for (int j = 0; j < NCB; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = alpha * acc16[i][j][e];
                        t1[j] += v - kshift16[j];
                        t2[j] = fmaf(v - kshift16[j], v - kshift16[j], t2[j]);
                        Cs[(wm * WM + 16 * i + 4 * (lane >> 4) + e) * LDC_S + wn * WN + 16 * j + (lane & 15)] = v;
                    }
        } else
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float v = alpha * acc[i][jn][e];
                    s1[i][jn] += v - kshift[jn];
                    s2[i][jn] = fmaf(v - kshift[jn], v - kshift[jn], s2[i][jn]);     // (explicit: every instantiation rounds alike)
                    Cs[(wm * WM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC_S + wn * WN + 32 * jn + r] = v;
                }
        __syncthreads();
        KOAF_STAMP(3);
        constexpr int C4 = BN / 4;
        constexpr int RPP = NT / C4;           // rows per pass
        const int c4 = t % C4, rr = t / C4;
        const int col = n0 + 4 * c4;
        const bool bnb = (p.bnb_mode != 0) && !slab;
        v4f q1 = {0.f, 0.f, 0.f, 0.f}, q2 = q1, q3 = q1;   // fused BN-backward column sums of this thread's rows
        v4f qm = q1;                                       // and the largest |dz| it stored (KoafGemm.bnb_amax), as magnitude bits
        if (col < p.N) {                         // N % 4 == 0 on this path
            v4f bv = {0.f, 0.f, 0.f, 0.f};
            if (bias) bv = *(const v4f*)(bias + col);
            v4f mu = bv, is = bv, ms = bv, mh = bv, mu2 = bv, is2 = bv;
            if (bnb) {
                mu = *(const v4f*)(p.bnb_mean + col);
                is = *(const v4f*)(p.bnb_invstd + col);
                if (p.bnb_mode == 2) { ms = *(const v4f*)(p.bnb_sc + col); mh = *(const v4f*)(p.bnb_sh + col); }
                if (p.bnb2_c) { mu2 = *(const v4f*)(p.bnb2_mean + col); is2 = *(const v4f*)(p.bnb2_invstd + col); }
            }
            const bool full = (m0 + BM <= p.M) && !p.cmap;
            if (full) {
                const bool hr = Rp != nullptr, h2 = bnb && p.bnb2_c != nullptr;
                const int mode = bnb ? p.bnb_mode : 0;
#define KOAF_EPI(R_, M_, C2_) epi_rows_full<BM, BN, NT, R_, M_, C2_, C16, E16, AT, EMIT>(p, Cs, LDC_S, Cp, ldc, Rp, AT ? t2d_base : m0, col, c4, rr, bv, mu, is, \
                                                                 ms, mh, mu2, is2, q1, q2, q3, qm, t2d_w)
                if (mode == 0) { if (hr) KOAF_EPI(true, 0, false); else KOAF_EPI(false, 0, false); }
                else if (mode == 1) {
                    if (hr) { if (h2) KOAF_EPI(true, 1, true); else KOAF_EPI(true, 1, false); }
                    else { if (h2) KOAF_EPI(false, 1, true); else KOAF_EPI(false, 1, false); }
                } else {
                    if (hr) { if (h2) KOAF_EPI(true, 2, true); else KOAF_EPI(true, 2, false); }
                    else { if (h2) KOAF_EPI(false, 2, true); else KOAF_EPI(false, 2, false); }
                }
#undef KOAF_EPI
            } else
#pragma unroll 4
            for (int row = rr; row < BM; row += RPP) {
                const int grow = m0 + row;
                if (grow < p.M) {
                    int64_t orow = grow;
                    if (p.cmap) {
                        const int ppi = p.cm_PH * p.cm_PW;
                        const int n = grow / ppi;
                        const int rem = grow - n * ppi;
                        const int yy = rem / p.cm_PW;
                        const int xx = rem - yy * p.cm_PW;
                        orow = ((int64_t)n * p.cm_H + 2 * yy + p.cm_py) * p.cm_W + 2 * xx + p.cm_px;
                    }
                    v4f v = *(const v4f*)&Cs[row * LDC_S + 4 * c4] + bv;
                    if (Rp) v += *(const v4f*)(Rp + orow * p.ldr + col);
                    if (bnb) {
                        const v4f cv = load4_nt<E16>(p.bnb_c, orow * ldc + col);
                        if (p.bnb_mode == 1) {
                            const v4f yv = load4_nt<E16>(p.bnb_y, orow * ldc + col);
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = yv[j] > 0.f ? v[j] : 0.f;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = (cv[j] * ms[j] + mh[j]) > 0.f ? v[j] : 0.f;
                        }
                        q1 += v;
                        q2 += v * ((cv - mu) * is);
#pragma unroll
                        for (int j = 0; j < 4; ++j) qm[j] = __uint_as_float(max(__float_as_uint(qm[j]), koaf_absbits(v[j])));
                        if (p.bnb2_c) {
                            const v4f c2 = load4_nt<E16>(p.bnb2_c, orow * ldc + col);
                            q3 += v * ((c2 - mu2) * is2);
                        }
                    }
                    store4_nt<C16>(Cp, orow * ldc + col, v);
                    if (EMIT && p.out_planes && !bnb && !Rp) {
                        unsigned ns = 0;
                        epi_emit_planes<C16>(p, orow * ldc + col, v, *(const v4f*)(p.out_sc + col) * KOAF_ACT_SCALE,
                                             *(const v4f*)(p.out_sh + col) * KOAF_ACT_SCALE, ns);
                        koaf_status_add(p.status, 0, ns);
                    }
                }
            }
        }
        if (EMIT && p.out_planes && tm == 0 && tn == 0 && t == 0 && blockIdx.z == 0) *(uint4*)(p.out_planes + 2 * p.out_ps) = make_uint4(0u, 0u, 0u, 0u);   // the zero chunk
        if (bnb) {
            if (p.bnb_amax) block_amax_raise_bits(max(max(__float_as_uint(qm[0]), __float_as_uint(qm[1])), max(__float_as_uint(qm[2]), __float_as_uint(qm[3]))), p.bnb_amax);
            // column sums over the block's rows: RPP row-threads per column vector -> LDS -> one partial row
            __syncthreads();                     // Cs fully consumed
            v4f* red4 = reinterpret_cast<v4f*>(smem);   // [3][RPP][C4]
            red4[(0 * RPP + rr) * C4 + c4] = q1;
            red4[(1 * RPP + rr) * C4 + c4] = q2;
            red4[(2 * RPP + rr) * C4 + c4] = q3;
            __syncthreads();
            const int nsum = p.bnb2_c ? 3 : 2;
            if (rr < nsum && col < p.N) {
                v4f a = red4[(rr * RPP) * C4 + c4];
                for (int j = 1; j < RPP; ++j) a += red4[(rr * RPP + j) * C4 + c4];
                *(v4f*)(p.bnb_part + ((int64_t)(p.part_row0 + tm) * nsum + rr) * p.N + col) = a;
            }
        }
        if (do_stats) __syncthreads();           // Cs is about to be reused by the statistics reduction
    } else {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            const int col = n0 + wn * WN + 32 * jn + r;
            const bool cok = col < p.N;
            const float bv = (bias && cok) ? bias[col] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * WM + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
                float v = alpha * acc[i][jn][e];
                s1[i][jn] += v - kshift[jn];
                s2[i][jn] = fmaf(v - kshift[jn], v - kshift[jn], s2[i][jn]);     // (explicit: every instantiation rounds alike)
                if (cok && row < p.M) {
                    v += bv;
                    if (Rp) v += Rp[(int64_t)row * p.ldr + col];
                    Cp[(int64_t)row * ldc + col] = v;
                }
            }
        }
    }
    }
    if (do_stats) {
        // column sums over this block's BM rows: lanes (r,0)+(r,1), then the WGM M-waves via LDS
        float* red = smem;  // [WGM][2][BN]
        if constexpr (M16) {
            // the four lane quarters hold rows 4 q .. 4 q + 3 of every tile of the column
#pragma unroll
            for (int j = 0; j < NCB; ++j) {
                float a1 = t1[j] + __shfl_xor(t1[j], 16, 64), a2 = t2[j] + __shfl_xor(t2[j], 16, 64);
                a1 += __shfl_xor(a1, 32, 64);
                a2 += __shfl_xor(a2, 32, 64);
                if (lane < 16) {
                    red[(wm * 2 + 0) * BN + wn * WN + 16 * j + lane] = a1;
                    red[(wm * 2 + 1) * BN + wn * WN + 16 * j + lane] = a2;
                }
            }
        } else
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            float a1 = s1[0][jn] + __shfl_xor(s1[0][jn], 32, 64);
            float a2 = s2[0][jn] + __shfl_xor(s2[0][jn], 32, 64);
#pragma unroll
            for (int i = 1; i < TM; ++i) {
                a1 += s1[i][jn] + __shfl_xor(s1[i][jn], 32, 64);
                a2 += s2[i][jn] + __shfl_xor(s2[i][jn], 32, 64);
            }
            if (h == 0) {
                red[(wm * 2 + 0) * BN + wn * WN + 32 * jn + r] = a1;
                red[(wm * 2 + 1) * BN + wn * WN + 32 * jn + r] = a2;
            }
        }
        __syncthreads();
        if (t < BN && (n0 + t) < p.N) {
            float* st = p.stats + (int64_t)(p.part_row0 + tm) * 2 * p.stats_ld + (int64_t)blockIdx.z * p.stats_bs;
            float a1 = red[0 * BN + t], a2 = red[1 * BN + t];
            if constexpr (WGM == 4 && TM == 1) {      // (the streamed kernels' four one-band waves: pairs first, as two two-band waves add up)
                a1 = (a1 + red[2 * BN + t]) + (red[4 * BN + t] + red[6 * BN + t]);
                a2 = (a2 + red[3 * BN + t]) + (red[5 * BN + t] + red[7 * BN + t]);
            } else {
#pragma unroll
                for (int m = 1; m < WGM; ++m) { a1 += red[(2 * m) * BN + t]; a2 += red[(2 * m + 1) * BN + t]; }
            }
            if (p.stats_shift) {
                // rows of the tile past M were accumulated as zeros: each put (0 - k) and k^2 into the shifted sums
                const float k = p.stats_shift[(int64_t)blockIdx.z * p.stats_bs + n0 + t];
                const int ninv = max(0, m0 + BM - p.M);
                a1 += (float)ninv * k;
                a2 -= (float)ninv * k * k;
            }
            st[n0 + t] = a1;
            st[p.stats_ld + n0 + t] = a2;
        }
    }
    KOAF_STAMP(4);
    KOAF_STAMP_ADD(0, 0, 1);      // prologue (entry -> first k-step ready); only the halo loop sets stamp 1
    KOAF_STAMP_ADD(1, 1, 2);      // k-loop
    KOAF_STAMP_ADD(2, 2, 3);      // accumulators -> LDS staging
    KOAF_STAMP_ADD(3, 3, 4);      // stores / fused reductions / statistics
    KOAF_STAMP_ADD(4, 0, 4);      // whole tile
    KOAF_STAMP_ACC(7, 1);         // tiles
    if (!has_next) break;
    // on to this block's next tile (PERSIST only): its A loads are in flight; the LDS is free once every wave is here
    __syncthreads();
    vt += gridDim.x;
    tm = tm2; tn = tn2;
    m0 = p.m_base + tm * BM; n0 = tn * BN;
    if constexpr (PERSIST) {
        lp.init(p.B, n0, p.N);
        lp.seek(p.B, kbeg);
    }
    if constexpr (AS) {
        st.tile(m0, p.M);
        stream_next();
    }
    }
}

// out[i] = sum_s slabs[s][i]: block = 64 float4-columns x 4 slab groups (LDS tree), so small outputs (a 64x64
// weight gradient split 1024 ways) still spread over many waves instead of 4 blocks doing 1024 serial loads
// out[m][c] = sum_s slabs[s][m][c] + bias[c] + residual[m][c]   (split-K combine of a small-grid linear layer)
__global__ void __launch_bounds__(256) slab_reduce_epi_kernel(const float* __restrict__ slabs, int nslab, int M, int N,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ residual, int64_t ldr,
                                                              float* __restrict__ out, int64_t ldo) {
    const int N4 = N / 4;
    const int64_t total = (int64_t)M * N4;
    const int64_t n = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / N4), c = (int)(i - (int64_t)m * N4) * 4;
        v4f a = *(const v4f*)(slabs + (int64_t)m * N + c);
        for (int s = 1; s < nslab; ++s) a += *(const v4f*)(slabs + (int64_t)s * n + (int64_t)m * N + c);
        if (bias) a += *(const v4f*)(bias + c);
        if (residual) a += *(const v4f*)(residual + (int64_t)m * ldr + c);
        *(v4f*)(out + (int64_t)m * ldo + c) = a;
    }
}

__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* __restrict__ slabs, int nslab,
                                                          int64_t n, float* __restrict__ out) {
    __shared__ v4f red[4][64];
    const int cx = threadIdx.x & 63, gy = threadIdx.x >> 6;
    const int64_t i = ((int64_t)blockIdx.x * 64 + cx) * 4;
    const int s_per = (nslab + gridDim.y - 1) / gridDim.y;
    const int s0 = blockIdx.y * s_per, s1 = min(nslab, s0 + s_per);
    v4f a = {0.f, 0.f, 0.f, 0.f};
    if (i < n)
        for (int s = s0 + gy; s < s1; s += 4) a += *(const v4f*)(slabs + (int64_t)s * n + i);
    red[gy][cx] = a;
    __syncthreads();
    if (gy == 0 && i < n) {
        a = red[0][cx] + red[1][cx] + red[2][cx] + red[3][cx];
        if (gridDim.y == 1) *(v4f*)(out + i) = a;
        else *(v4f*)(out + (int64_t)blockIdx.y * n + i) = a;   // second-level slabs
    }
}

constexpr unsigned PERSIST_BLOCKS = 512;      // 2 per CU x 256 CUs; a multiple of 8 (virtual tile ids keep their XCD)

#ifdef KOAF_DEV_STREAM      // (development builds, with KOAF_DEV_T2D: the streamed-A instantiations alone -- 20 s -- for their register counts / ISA)
template __global__ void koaf_gemm_kernel<128, 128, M_KS, M_PS, 0, 0, true, true, 256, 0, false, 2>(const KoafGemm);
template __global__ void koaf_gemm_kernel<128, 128, M_KS, M_PS, 1, 0, true, true, 256, 0, false, 2>(const KoafGemm);
template __global__ void koaf_gemm_kernel<128, 128, M_KS, M_PS, 2, 0, true, true, 256, 0, false, 2>(const KoafGemm);
template __global__ void koaf_gemm_kernel<128, 128, M_KS, M_PS, 3, 0, true, true, 256, 0, false, 2>(const KoafGemm);
template __global__ void koaf_gemm_kernel<128, 64, M_KS, M_PS, 2, 0, true, true, 256, 0, false, 2>(const KoafGemm);
template __global__ void koaf_gemm_kernel<128, 64, M_KS, M_PS, 3, 0, true, true, 256, 0, true, 2>(const KoafGemm);
#endif


bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

bool operand_vec_ok(const KoafOperand& o, int R, int K) {
    if (o.kind == 3) {
        // K-major activation plane images: 16-B chunks of 8 channels along the rows
        return aligned16(o.planes) && aligned16(o.zeros) && o.zeros && !(o.plane_stride & 7) && !(o.bs0 & 7) && !(o.bs1 & 7) &&
               !(R & 7) && (o.gather ? (o.C > 0 && !(o.C & 7) && !(o.CS & 7) && o.KW > 0) : !(o.ld & 7));
    }
    if (o.kind == 2 && o.gather) {
        // activation plane images: 16-B chunks of 8 channels, k-tiles never straddle a tap
        return aligned16(o.planes) && aligned16(o.zeros) && o.zeros && !(o.plane_stride & 7) && !(o.bs0 & 7) && !(o.bs1 & 7) &&
               o.C > 0 && !(o.C & 31) && !(o.CS & 7) && o.KW > 0 && (K % BK) == 0;
    }
    if (o.kind == 2) {
        // pre-split plane images: rows and taps start on 16-B boundaries, k-tiles never straddle a tap
        if (!aligned16(o.planes) || (o.ld & 7) || (o.plane_stride & 7) || (o.bs0 & 7) || (o.bs1 & 7)) return false;
        if ((o.tap_stride & 7) || (o.tap_stride_h & 7) || o.C <= 0 || (o.C & 31) || o.KW <= 0) return false;
        return true;
    }
    if (!aligned16(o.ptr)) return false;
    if ((o.ld & 3) || (o.bs0 & 3) || (o.bs1 & 3)) return false;
    if (o.kind == 0) {
        if (K & 3) return false;
        if (o.gather && ((o.C & 31) || (o.CS & 3))) return false;
    } else {
        if (R & 3) return false;
        if (o.gather == 3 && ((o.C & 31) || (o.tap_stride & 3) || (o.tap_stride_h & 3))) return false;
        if (o.gather == 1 && ((o.C & 3) || (o.CS & 3))) return false;
    }
    if (o.tf && (!aligned16(o.sc) || !aligned16(o.sh))) return false;
    if (o.tf == 2 && (!aligned16(o.sc2) || !aligned16(o.ptr2))) return false;
    return true;
}

int operand_mode(const KoafOperand& o) {
    if (o.kind == 3) return o.gather ? M_PKG : M_PK;
    if (o.kind == 2) return o.gather == 0 ? M_PS : (o.gather == 1 ? M_PA1 : M_PA2);
    if (o.kind == 0) return o.gather == 0 ? M_KC : (o.gather == 1 ? M_KC_G1 : M_KC_G2);
    return o.gather == 0 ? M_KM : (o.gather == 1 ? M_KM_G1 : M_KM_G3);
}

#define KOAF_LAUNCH(AMODE, BMODE, TA, TB)                                                                        \
    hipLaunchKernelGGL((koaf_gemm_kernel<BM, BN, AMODE, BMODE, TA, TB, VEC, F16, 256, ACT>), grid, dim3(256), 0, s, g);      \
    return koaf_check_launch("koaf_gemm")
// the persistent variants (fp32 A loader + weight tiles by DMA): at most two blocks per CU, each walking its tiles
#define KOAF_LAUNCH_P(AMODE, BMODE, TA, TB)                                                                      \
    hipLaunchKernelGGL((koaf_gemm_kernel<BM, BN, AMODE, BMODE, TA, TB, VEC, F16, 256, ACT>), (persist_mode(AMODE, BMODE, F16, TA) ? pgrid : grid), dim3(256), 0, s, g);     \
    return koaf_check_launch("koaf_gemm")

// KoafGemm.out_planes (the epilogue also cuts the consumer's plane images): the instantiations with EMIT, for the calls that use it
// -- dense 1x1 forward convolutions with weight plane images (plain, BatchNorm-prologue and bottleneck-tail loaders)
#define KOAF_LAUNCH_E(AMODE, BMODE, TA, TB)                                                                      \
    hipLaunchKernelGGL((koaf_gemm_kernel<BM, BN, AMODE, BMODE, TA, TB, VEC, F16, 256, ACT, true>), grid, dim3(256), 0, s, g);      \
    return koaf_check_launch("koaf_gemm/emit")
#define KOAF_LAUNCH_PE(AMODE, BMODE, TA, TB)                                                                     \
    hipLaunchKernelGGL((koaf_gemm_kernel<BM, BN, AMODE, BMODE, TA, TB, VEC, F16, 256, ACT, true>), (persist_mode(AMODE, BMODE, F16, TA) ? pgrid : grid), dim3(256), 0, s, g);     \
    return koaf_check_launch("koaf_gemm/emit")

// the streamed dense A operand (M_KS, StreamA) in front of weight plane images: SD = 2 k-tiles in flight per wave
#define KOAF_LAUNCH_S(TA, EM)                                                                                    \
    hipLaunchKernelGGL((koaf_gemm_kernel<BM, BN, M_KS, M_PS, TA, 0, VEC, F16, 256, ACT, EM, 2>), ((TA) < 2 ? pgrid : grid), dim3(256), 0, s, g);      \
    return koaf_check_launch("koaf_gemm/stream")
// (SD = 4 -- four k-tiles in flight, one tile per block, for the one-source loaders with K >= 256 -- builds without spills (213
// registers) and was measured on the headline step: 1977.2 ms against 1974.4 ms with SD = 2 everywhere; not instantiated.)

int g_stream_mode = -1;     // the streamed A operand for the dense 1x1 kernels: -1 = read KOAF_STREAM once (default on)
// dense K-contiguous fp32 A (1x1 / stride-1 convolutions and their data gradients) in whole k-tiles, an even number of them
bool stream_ok(const KoafGemm& g) {
    if (g_stream_mode < 0) { const char* e = getenv("KOAF_STREAM"); g_stream_mode = (e && e[0] == '0') ? 0 : 1; }
    return g_stream_mode == 1 && g.A.kind == 0 && g.A.gather == 0 && g.K >= 2 * BK && (g.K % (2 * BK)) == 0 && g.splitk == 1 &&
           g.nb0 * g.nb1 == 1 && (g.A.tf != 1 || g.K <= STREAM_TAB_K);
}

// the operand-mode pairs the library uses: conv fwd (KC|KC_G1 x KC|PS), dgrad (KC|KC_G2 x PS, or KC x KM | KC_G2 x KM_G3
// on fp32 weights), wgrad (KM x KM|KM_G1), linear / attention (dense pairs).  tf only where a BatchNorm prologue exists.
// ACT (bf16 activation storage, KoafGemm.act16): only the pairs of its role are instantiated -- 1 forward convolutions,
// 2 data gradients, 3 weight gradients.
template <int BM, int BN, bool VEC, bool F16, int ACT>
int launch_modes(const KoafGemm& g, dim3 grid, hipStream_t s) {
    const int am = operand_mode(g.A), bm = operand_mode(g.B);
    const int ta = g.A.tf, tb = g.B.tf;
    dim3 pgrid = grid;
    if (grid.y == 1 && grid.x > PERSIST_BLOCKS) pgrid.x = PERSIST_BLOCKS;
    if constexpr (VEC && F16 && ACT == 0 && BM == 128) {
        if (am == M_KC && bm == M_PS && !tb && stream_ok(g)) {
            if (g.out_planes) {
                if (ta == 0) { KOAF_LAUNCH_S(0, true); }
                if (ta == 1) { KOAF_LAUNCH_S(1, true); }
                if (ta == 3) { KOAF_LAUNCH_S(3, true); }
            } else {
                if (ta == 0) { KOAF_LAUNCH_S(0, false); }
                if (ta == 1) { KOAF_LAUNCH_S(1, false); }
                if (ta == 2) { KOAF_LAUNCH_S(2, false); }
                if (ta == 3) { KOAF_LAUNCH_S(3, false); }
            }
        }
    }
    if (g.out_planes) {
        if constexpr (VEC && F16 && (ACT == 0 || ACT == 1)) {
            if (am == M_KC && bm == M_PS && ta == 3) { KOAF_LAUNCH_E(M_KC, M_PS, 3, 0); }
            if (am == M_KC && bm == M_PS && ta < 2) { if (ta == 1) { KOAF_LAUNCH_PE(M_KC, M_PS, 1, 0); } else { KOAF_LAUNCH_PE(M_KC, M_PS, 0, 0); } }
        }
        koaf_set_error("koaf_gemm: out_planes is built for dense K-contiguous A x weight plane images on the fp16 scheme (1x1 forward convolutions); "
                       "got operand modes (%d,%d) tf=%d fmt=%d act16=%d", am, bm, ta, (int)F16, ACT);
        return KOAF_EINVAL;
    }
    if constexpr (ACT == 0) {
        if (am == M_KC && bm == M_KC && !tb) { if (ta == 1) { KOAF_LAUNCH(M_KC, M_KC, 1, 0); } else if (!ta) { KOAF_LAUNCH(M_KC, M_KC, 0, 0); } }
        if (am == M_KC && bm == M_KM && !ta && !tb) { KOAF_LAUNCH(M_KC, M_KM, 0, 0); }
        if (am == M_KM && bm == M_KM && !ta) { if (tb == 1) { KOAF_LAUNCH(M_KM, M_KM, 0, 1); } else if (!tb) { KOAF_LAUNCH(M_KM, M_KM, 0, 0); } }
    }
    if constexpr (VEC) {
        if constexpr (ACT == 0 || (ACT == 1 && !F16)) {
            if (am == M_KC_G1 && bm == M_KC && !tb) { if (ta == 1) { KOAF_LAUNCH(M_KC_G1, M_KC, 1, 0); } else if (!ta) { KOAF_LAUNCH(M_KC_G1, M_KC, 0, 0); } }
        }
        if constexpr (ACT == 0) {
            if (am == M_KC_G2 && bm == M_KM_G3 && !ta && !tb) { KOAF_LAUNCH(M_KC_G2, M_KM_G3, 0, 0); }
        }
        if constexpr (ACT == 0 || (ACT == 3 && !F16)) {
            if (am == M_KM && bm == M_KM_G1 && !ta) { if (tb == 1) { KOAF_LAUNCH(M_KM, M_KM_G1, 0, 1); } else if (!tb) { KOAF_LAUNCH(M_KM, M_KM_G1, 0, 0); } }
        }
        if constexpr (F16) {
            if constexpr (ACT == 0 || ACT == 1) {
                if (am == M_KC && bm == M_PS && ta == 3) { KOAF_LAUNCH(M_KC, M_PS, 3, 0); }
                if (am == M_KC && bm == M_PS && ta < 2) { if (ta == 1) { KOAF_LAUNCH_P(M_KC, M_PS, 1, 0); } else { KOAF_LAUNCH_P(M_KC, M_PS, 0, 0); } }
                if (am == M_KC_G1 && bm == M_PS && ta < 2) { if (ta) { KOAF_LAUNCH_P(M_KC_G1, M_PS, 1, 0); } else { KOAF_LAUNCH_P(M_KC_G1, M_PS, 0, 0); } }
                if (am == M_PA1 && bm == M_PS) { KOAF_LAUNCH(M_PA1, M_PS, 0, 0); }
            }
            if constexpr (ACT == 0 || ACT == 2) {
                if (am == M_KC && bm == M_PS && ta == 2) { KOAF_LAUNCH(M_KC, M_PS, 2, 0); }
                if (am == M_KC_G2 && bm == M_PS && ta != 1) { if (ta) { KOAF_LAUNCH(M_KC_G2, M_PS, 2, 0); } else { KOAF_LAUNCH_P(M_KC_G2, M_PS, 0, 0); } }
                if (am == M_PA2 && bm == M_PS) { KOAF_LAUNCH(M_PA2, M_PS, 0, 0); }
            }
            if constexpr (ACT == 2) {       // (a data gradient whose dy is a tensor: only the epilogue's operands are bf16)
                if (am == M_KC && bm == M_PS && ta == 0) { KOAF_LAUNCH(M_KC, M_PS, 0, 0); }
            }
            if constexpr (ACT == 0) {
                if (am == M_PK && bm == M_PKG) { KOAF_LAUNCH(M_PK, M_PKG, 0, 0); }
                if (am == M_PK && bm == M_PK) { KOAF_LAUNCH(M_PK, M_PK, 0, 0); }
            }
            if constexpr (ACT == 0 || ACT == 3) {
                // weight gradient with the BatchNorm-backward apply formed in the A loader (dy = sc * dz + sh - sc2 * c)
                if (am == M_KM && bm == M_KM && ta == 2) { if (tb == 1) { KOAF_LAUNCH(M_KM, M_KM, 2, 1); } else if (!tb) { KOAF_LAUNCH(M_KM, M_KM, 2, 0); } }
                if (am == M_KM && bm == M_KM_G1 && ta == 2) { if (tb == 1) { KOAF_LAUNCH(M_KM, M_KM_G1, 2, 1); } else if (!tb) { KOAF_LAUNCH(M_KM, M_KM_G1, 2, 0); } }
            }
        }
    }
    koaf_set_error("koaf_gemm: operand mode pair (%d,%d) tf=(%d,%d) fmt=%d vec=%d act16=%d is not instantiated", am, bm, ta, tb,
                   (int)F16, (int)VEC, ACT);
    return KOAF_EINVAL;
}

}  // namespace

extern "C" int koaf_gemm_pick_tile(const KoafGemm* g, int32_t* bm, int32_t* bn) {
    int b_m = g->bm, b_n = g->bn;
    const int64_t batch = (int64_t)g->nb0 * g->nb1 * (g->splitk > 0 ? g->splitk : 1);
    if (b_n == 0) {
        b_n = (g->N >= 128) ? 128 : 64;
        // (a gathered K-major B tile may span filter taps: every thread derives the tap of its own 4 columns)
        if (g->B.kind == 1 && g->B.gather == 1 && (g->B.C % 64) != 0) b_n = 64;
    }
    if (b_m == 0) b_m = (g->M >= 128) ? 128 : 64;
    if (g->bm == 0 || g->bn == 0) {
        auto tiles = [&](int m, int n) { return cdiv64(g->M, m) * cdiv64(g->N, n) * batch; };
        // fill the 256 CUs: shrink the tile while the grid is under 384 blocks (1.5 per CU: a single 78 %-full round of
        // 128x128 tiles beats two rounds of the 1.25x costlier 64-row tiles; measured 512 / 384 / 256)
        constexpr int64_t fill = 384;
        if (g->bm == 0 && tiles(b_m, b_n) < fill && b_m == 128) b_m = 64;
        if (g->bn == 0 && tiles(b_m, b_n) < fill && b_n == 128) b_n = 64;
    }
    *bm = b_m;
    *bn = b_n;
    return KOAF_OK;
}

namespace {
struct TilePlan { int bm, bn; bool vec; int part_rows; bool halo; bool t2d; };

int g_halo_mode = 1;        // koaf_set_conv3x3_halo: 0 off, 1 pick the shape per layer, 2 always 256 rows, 3 always 128 rows
int g_t2d_mode = -1;        // the 2-D tile kernel (M_PT) for 64- / 128-channel layers whose image tiles evenly: -1 = read KOAF_CONV3_T2D once (default on)

// 3x3 / stride 1 / pad 1 over activation plane images with the whole pixel range as rows: the halo kernel (M_PH)
bool halo_ok(const KoafGemm& g) {
    const KoafOperand& a = g.A;
    return g_halo_mode != 0 && a.kind == 2 && (a.gather == 1 || a.gather == 2) && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 &&
           a.pad_w == 1 && a.PH == a.H && a.PW == a.W && a.W <= halo_max_w(g.N >= 128 ? 128 : 64, 256) && a.H * a.W > 0 && (g.M % (a.H * a.W)) == 0 &&
           g.nb0 * g.nb1 == 1 && g.m_base == 0 && g.K == 9 * a.C && g.B.kind == 2 && g.fmt == 1 && !g.cmap && g.splitk == 1;
}

bool gemm_vec_ok(const KoafGemm& g) {
    bool vec = operand_vec_ok(g.A, g.M, g.K) && operand_vec_ok(g.B, g.N, g.K);
    // vector epilogue: 16-B aligned rows of C / residual / bias
    if ((g.N & 3) || !aligned16(g.C) || (g.ldc & 3) || (g.cbs0 & 3) || (g.cbs1 & 3)) vec = false;
    if (g.residual && (!aligned16(g.residual) || (g.ldr & 3) || (g.rbs0 & 3) || (g.rbs1 & 3))) vec = false;
    if (g.bias && !aligned16(g.bias)) vec = false;
    return vec;
}

// g must already have its defaults filled (nb*, splitk, CS, stats_ld)
TilePlan plan_tiles(const KoafGemm& g) {
    TilePlan t;
    KoafGemm q = g;
    koaf_gemm_pick_tile(&q, &t.bm, &t.bn);
    t.vec = gemm_vec_ok(g);
    if (!t.vec) { t.bm = 64; t.bn = 64; }
    t.halo = t.vec && halo_ok(g);
    if (g_t2d_mode < 0) { const char* e = getenv("KOAF_CONV3_T2D"); g_t2d_mode = (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : 1); }
    // 64 channels in (one halo chunk), 64 / 128 out, images of whole 8 x 16 tiles: the rectangle-tile kernel (halo_ok: 3x3 / stride 1 /
    // pad 1 over plane images, one GEMM over all pixels, every pixel a row).  KOAF_CONV3_T2D=2 admits 128 input channels as well (two
    // chunks, the halo fetched twice per 64-column tile: measured no faster than the 128-row raster kernel there, 2.59 vs 2.55 ms).
    t.t2d = t.halo && g_t2d_mode >= 1 && g_halo_mode == 1 && (g.A.C == 64 || (g.A.C == 128 && g_t2d_mode == 2)) && (g.N == 64 || g.N == 128) &&
            (g.A.W % 16) == 0 && (g.A.H % 8) == 0 && g.A.zeros != nullptr;
    if (t.t2d) {
        t.halo = false;
        t.bm = 128;
        t.bn = 64;
    } else if (t.halo) {
        t.bn = g.N >= 128 ? 128 : 64;
        // 128 rows x two blocks per CU where the k-loop is short (few channel chunks) and the row fits its halo buffer
        const bool fits128 = g.A.W <= halo_max_w(t.bn, 128);
        t.bm = (g_halo_mode == 3 || (g_halo_mode == 1 && g.A.C <= 64)) && fits128 ? 128 : 256;
    }
    t.part_rows = (int)cdiv64(g.M - g.m_base, t.bm);
    return t;
}
}  // namespace

// defaults of the optional descriptor fields (shared by the launch and by the planning queries, which must agree)
static void fill_defaults(KoafGemm& g) {
    if (g.nb0 < 1) g.nb0 = 1;
    if (g.nb1 < 1) g.nb1 = 1;
    if (g.splitk < 1) g.splitk = 1;
    if (g.A.CS == 0) g.A.CS = g.A.C;
    if (g.B.CS == 0) g.B.CS = g.B.C;
    if (g.stats_ld == 0) g.stats_ld = g.N;
    if (g.B.kind == 2 && g.B.C == 0) { g.B.C = ((g.K + BK - 1) / BK) * BK; g.B.KW = 1; }   // dense: one "tap" spanning the padded K
}

namespace {
// all launches of one activation-storage role (KoafGemm.act16)
template <int ACT>
int launch_act(const KoafGemm& g, const TilePlan& tp, dim3 grid, hipStream_t s) {
    const bool vec = tp.vec;
    if (g.out_planes && (tp.t2d || tp.halo)) { koaf_set_error("koaf_gemm: out_planes is not built for the 3x3 plane-image kernels"); return KOAF_EINVAL; }
    if (tp.t2d) {
        if constexpr (ACT == 3) { koaf_set_error("koaf_gemm: 2-D tile kernel with act16 = 3"); return KOAF_EINVAL; }
        else {
            hipLaunchKernelGGL((koaf_gemm_kernel<128, 64, M_PT, M_PS, 0, 0, true, true, 256, ACT>), grid, dim3(256), 0, s, g);
            return koaf_check_launch("koaf_gemm/t2d");
        }
    }
#ifdef KOAF_DEV_T2D      // (development builds: only the kernels a 3x3 A/B needs are instantiated -- seconds instead of minutes)
    if (g.M == -12345) hipLaunchKernelGGL((koaf_gemm_kernel<128, 128, M_KC, M_PS, 1, 0, true, true, 256, 0>), grid, dim3(256), 0, s, g);   // (register-usage probe)
    if (g.M == -12346) hipLaunchKernelGGL((koaf_gemm_kernel<128, 128, M_KC, M_PS, 2, 0, true, true, 256, 0>), grid, dim3(256), 0, s, g);
#ifdef KOAF_DEV_H256
    if constexpr (ACT == 0) {
        if (tp.halo && tp.bm == 256 && tp.bn == 128) {
            hipLaunchKernelGGL((koaf_gemm_kernel<256, 128, M_PH, M_PS, 0, 0, true, true, 512, 0>), grid, dim3(512), 0, s, g);
            return koaf_check_launch("koaf_gemm/halo");
        }
    }
#endif
    if (!tp.halo || tp.bm != 128) { koaf_set_error("koaf_gemm: KOAF_DEV_T2D build"); return KOAF_EINVAL; }
    if constexpr (ACT != 0) { koaf_set_error("koaf_gemm: KOAF_DEV_T2D build"); return KOAF_EINVAL; }
    else {
        if (tp.bn == 128) hipLaunchKernelGGL((koaf_gemm_kernel<128, 128, M_PH, M_PS, 0, 0, true, true, 256, 0>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((koaf_gemm_kernel<128, 64, M_PH, M_PS, 0, 0, true, true, 256, 0>), grid, dim3(256), 0, s, g);
        return koaf_check_launch("koaf_gemm/halo128");
    }
#else
    if (tp.halo) {
        // (the halo kernels read plane images: only their epilogue sees the storage type -- forward: the output; data
        // gradient: the BatchNorm-backward operands)
        if constexpr (ACT == 3) { koaf_set_error("koaf_gemm: halo kernel with act16 = 3"); return KOAF_EINVAL; }
        else {
            if (tp.bm == 128) {
                if (tp.bn == 128) hipLaunchKernelGGL((koaf_gemm_kernel<128, 128, M_PH, M_PS, 0, 0, true, true, 256, ACT>), grid, dim3(256), 0, s, g);
                else hipLaunchKernelGGL((koaf_gemm_kernel<128, 64, M_PH, M_PS, 0, 0, true, true, 256, ACT>), grid, dim3(256), 0, s, g);
                return koaf_check_launch("koaf_gemm/halo128");
            }
            if (tp.bn == 128) hipLaunchKernelGGL((koaf_gemm_kernel<256, 128, M_PH, M_PS, 0, 0, true, true, 512, ACT>), grid, dim3(512), 0, s, g);
            else hipLaunchKernelGGL((koaf_gemm_kernel<256, 64, M_PH, M_PS, 0, 0, true, true, 512, ACT>), grid, dim3(512), 0, s, g);
            return koaf_check_launch("koaf_gemm/halo");
        }
    }
    if (g.fmt == 1) {
        KOAF_REQUIRE(vec, "koaf_gemm: the fp16 scheme needs the vector path (16-B aligned operands, K %% 4 == 0, N %% 4 == 0)");
        if (tp.bm == 128 && tp.bn == 128) return launch_modes<128, 128, true, true, ACT>(g, grid, s);
        if (tp.bm == 128 && tp.bn == 64) return launch_modes<128, 64, true, true, ACT>(g, grid, s);
        if (tp.bm == 64 && tp.bn == 128) return launch_modes<64, 128, true, true, ACT>(g, grid, s);
        return launch_modes<64, 64, true, true, ACT>(g, grid, s);
    }
    if (!vec) {
        if constexpr (ACT == 0) return launch_modes<64, 64, false, false, 0>(g, grid, s);
        else { koaf_set_error("koaf_gemm: bf16 activation storage needs the vector path"); return KOAF_EINVAL; }
    }
    if (tp.bm == 128 && tp.bn == 128) return launch_modes<128, 128, true, false, ACT>(g, grid, s);
    if (tp.bm == 128 && tp.bn == 64) return launch_modes<128, 64, true, false, ACT>(g, grid, s);
    if (tp.bm == 64 && tp.bn == 128) return launch_modes<64, 128, true, false, ACT>(g, grid, s);
    return launch_modes<64, 64, true, false, ACT>(g, grid, s);
#endif
}
}  // namespace

extern "C" int koaf_gemm_part_rows(const KoafGemm* gp) {
    KoafGemm g = *gp;
    fill_defaults(g);
    return plan_tiles(g).part_rows;
}

extern "C" int koaf_gemm(const KoafGemm* gp, void* stream) {
    KoafGemm g = *gp;
    KOAF_REQUIRE(g.M > 0 && g.N > 0 && g.K >= 0, "koaf_gemm: bad dims M=%d N=%d K=%d", g.M, g.N, g.K);
    KOAF_REQUIRE((g.A.kind >= 2 ? (const void*)g.A.planes : (const void*)g.A.ptr) &&
                 (g.B.kind >= 2 ? (const void*)g.B.planes : (const void*)g.B.ptr) && g.C, "koaf_gemm: null operand");
    fill_defaults(g);
    if (!g.status) g.status = koaf_status_ptr();
    KOAF_REQUIRE(!g.cmap || (g.splitk == 1 && !g.stats), "koaf_gemm: row map excludes split-K / stats");
    KOAF_REQUIRE(!g.bnb_mode || (g.splitk == 1 && !g.stats && g.nb0 * g.nb1 == 1 && g.bnb_c && g.bnb_mean &&
                                 g.bnb_invstd && g.bnb_part && (g.bnb_mode == 1 ? g.bnb_y != nullptr
                                                                                 : (g.bnb_sc && g.bnb_sh))),
                 "koaf_gemm: fused BN-backward epilogue needs c/mean/invstd/part (+y or sc/sh), no split-K/batch");
    KOAF_REQUIRE(g.splitk == 1 || (!g.bias && !g.residual && !g.stats),
                 "koaf_gemm: split-K writes raw slabs (no epilogue)");
    KOAF_REQUIRE((int64_t)g.nb0 * g.nb1 <= 65535 && g.splitk <= 65535, "koaf_gemm: batch/splitk too large");
    KOAF_REQUIRE(g.A.kind == 0 || (g.A.kind == 1 && g.A.gather == 0) || g.A.kind == 2 || (g.A.kind == 3 && g.A.gather == 0),
                 "koaf_gemm: A is K-contiguous fp32, K-major without gather, or activation plane images");
    KOAF_REQUIRE((g.A.kind == 3) == (g.B.kind == 3), "koaf_gemm: K-major plane images (kind 3) come as a pair");
    if (g.A.kind == 3)
        KOAF_REQUIRE(g.fmt == 1 && !g.A.tf && !g.B.tf && g.A.zeros && g.B.zeros && (g.B.gather == 0 || g.B.gather == 1),
                     "koaf_gemm: K-major plane images need fmt 1, no transform, the zero chunks; B.gather 0 | 1");
    if (g.A.kind == 2)
        KOAF_REQUIRE((g.A.gather == 1 || g.A.gather == 2) && g.B.kind == 2 && g.fmt == 1 && !g.A.tf && g.splitk == 1 && g.A.zeros,
                     "koaf_gemm: activation plane images (A.kind 2) need gather 1|2, a pre-split B, fmt 1, no transform, no split-K");
    KOAF_REQUIRE(!(g.A.kind == 0 && g.A.gather == 3) && !(g.B.kind == 0 && g.B.gather == 3),
                 "koaf_gemm: tapped gather needs a K-major operand");
    KOAF_REQUIRE(!(g.B.kind == 0 && g.B.gather), "koaf_gemm: K-contiguous B cannot be gathered");
    KOAF_REQUIRE(g.fmt == 0 || g.fmt == 1, "koaf_gemm: fmt must be 0 (bf16 x 3) or 1 (fp16 x 2)");
    KOAF_REQUIRE(g.A.tf >= 0 && g.A.tf <= 3 && g.B.tf >= 0 && g.B.tf <= 1, "koaf_gemm: tf is 0 | 1 (A, B) | 2 | 3 (A)");
    KOAF_REQUIRE(g.A.tf != 3 || (g.A.kind == 0 && g.A.gather == 0 && g.A.ptr2 && g.A.sc && g.A.sh && g.fmt == 1 && g.B.kind == 2 &&
                                 g.nb0 * g.nb1 == 1 && g.splitk == 1 && aligned16(g.A.ptr2) && (!g.A.side || aligned16(g.A.side)) &&
                                 (!g.A.sc2 || (g.A.sh2 && aligned16(g.A.sc2) && aligned16(g.A.sh2)))),
                 "koaf_gemm: the bottleneck-tail prologue (tf 3) needs a dense K-contiguous A, ptr2 / sc / sh, the fp16 scheme with a pre-split B");
    KOAF_REQUIRE(g.A.tf != 2 || (g.A.ptr2 && g.A.sc && g.A.sh && g.A.sc2 && g.fmt == 1),
                 "koaf_gemm: the two-source prologue needs ptr2 / sc / sh / sc2 and the fp16 scheme");
    if (g.B.kind == 2) {
        KOAF_REQUIRE(g.fmt == 1 && g.B.amax, "koaf_gemm: plane images are fp16 pieces of B * scale(*B.amax): fmt 1, amax required");
        KOAF_REQUIRE((g.A.kind == 0 || g.A.kind == 2) && !g.B.tf, "koaf_gemm: a pre-split B pairs with a K-contiguous A and takes no transform");
    }
    const TilePlan tp = plan_tiles(g);
    const bool vec = tp.vec;
    KOAF_REQUIRE(!g.out_planes || (tp.vec && g.out_sc && g.out_sh && g.ldc == g.N && !(g.N & 7) && g.nb0 * g.nb1 == 1 && g.splitk == 1 && !g.cmap &&
                                   !g.residual && !g.bnb_mode && g.out_ps == (int64_t)g.M * g.N && aligned16(g.out_planes) && aligned16(g.out_sc) &&
                                   aligned16(g.out_sh)),
                 "koaf_gemm: out_planes needs the vector epilogue of a plain forward call (ldc == N, N %% 8 == 0, no batch / split-K / row map / "
                 "residual / BatchNorm-backward), out_ps == M * N and 16-B aligned out_planes / out_sc / out_sh");
    KOAF_REQUIRE(((tp.bm == 64 || tp.bm == 128) && (tp.bn == 64 || tp.bn == 128)) || tp.halo || tp.t2d, "koaf_gemm: tile must be 64|128");
    if (g.A.gather || g.B.gather) KOAF_REQUIRE(vec, "koaf_gemm: gathered operands need aligned, C%%32==0 tensors");
    if (g.B.kind >= 2 || g.A.kind >= 2) KOAF_REQUIRE(vec, "koaf_gemm: pre-split operands need 16-B aligned images (and C %% 32 == 0 per tap)");
    if (g.B.kind == 1 && g.B.gather == 1)
        KOAF_REQUIRE(g.B.C % 4 == 0, "koaf_gemm: gathered K-major operand needs channels per tap (%d) %% 4 == 0", g.B.C);
    KOAF_REQUIRE(!g.cmap || vec, "koaf_gemm: row map needs the vector epilogue");
    KOAF_REQUIRE(!g.bnb_mode || vec, "koaf_gemm: fused BN-backward needs the vector epilogue");
    hipStream_t s = (hipStream_t)stream;
    // A/B switch KOAF_STREAM_NT=0: the streamed kernel's non-temporal A loads off.  It travels in `prec` (informational: no kernel
    // reads it otherwise) of this launch's private copy of the descriptor.
    { static int nt = -1; if (nt < 0) { const char* e = getenv("KOAF_STREAM_NT"); nt = (e && e[0] == '0') ? 0 : 1; } if (!nt) g.prec = 55; }
    g.bm = tp.bm;
    g.bn = tp.bn;
    const int64_t tiles = cdiv64(g.M - g.m_base, tp.bm) * cdiv64(g.N, tp.bn);
    if (tiles <= 0) return KOAF_OK;
    if (tiles >= (1ll << 31)) { koaf_set_error("koaf_gemm: grid too large"); return KOAF_EINVAL; }
    dim3 grid((unsigned)tiles, (unsigned)g.splitk, (unsigned)(g.nb0 * g.nb1));
    switch (g.act16) {
        case 0: return launch_act<0>(g, tp, grid, s);
        case 1: return launch_act<1>(g, tp, grid, s);
        case 2: return launch_act<2>(g, tp, grid, s);
        case 3: return launch_act<3>(g, tp, grid, s);
    }
    koaf_set_error("koaf_gemm: act16 must be 0 .. 3");
    return KOAF_EINVAL;
}

extern "C" int koaf_slab_reduce(const float* slabs, int32_t nslab, int64_t n, float* out, void* stream) {
    KOAF_REQUIRE(slabs && out && nslab >= 1 && n > 0 && (n & 3) == 0, "koaf_slab_reduce: bad args (n %% 4 == 0)");
    KOAF_REQUIRE((((uintptr_t)slabs | (uintptr_t)out) & 15) == 0, "koaf_slab_reduce: unaligned");
    const unsigned bx = (unsigned)cdiv64(n / 4, 64);
    // two-level when one level would leave the chip idle: nslab -> 16 partial slabs written BEHIND the input
    // slabs (the workspace holds (nslab + 16) * n floats), then 16 -> out
    hipStream_t s = (hipStream_t)stream;
    if (nslab >= 64 && bx < 256) {
        float* part = const_cast<float*>(slabs) + (int64_t)nslab * n;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(bx, 16), dim3(256), 0, s, slabs, nslab, n, part);
        int rc = koaf_check_launch("koaf_slab_reduce/1");
        if (rc != KOAF_OK) return rc;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(bx, 1), dim3(256), 0, s, part, 16, n, out);
        return koaf_check_launch("koaf_slab_reduce/2");
    }
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(bx, 1), dim3(256), 0, s, slabs, nslab, n, out);
    return koaf_check_launch("koaf_slab_reduce");
}

extern "C" int koaf_slab_reduce_epilogue(const float* slabs, int32_t nslab, int32_t M, int32_t N, const float* bias,
                                         const float* residual, int64_t ldr, float* out, int64_t ldo, void* stream) {
    KOAF_REQUIRE(slabs && out && nslab >= 1 && M > 0 && N > 0 && (N & 3) == 0 && (ldo & 3) == 0 && (ldr & 3) == 0,
                 "koaf_slab_reduce_epilogue: bad args");
    const int64_t total = (int64_t)M * (N / 4);
    int64_t blocks = cdiv64(total, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(slab_reduce_epi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, slabs, nslab, M,
                       N, bias, residual, ldr, out, ldo);
    return koaf_check_launch("koaf_slab_reduce_epilogue");
}


// ================================================================================================
// fusion attention, forward, ONE launch (reference: koafusion/models/_core_trf.py:170-180): per (batch, head) and 32-query tile
//   S = scale * Q K^T  (all n <= 512 keys; scores stay in LDS)  ->  softmax rows  ->  attn written ONCE  ->  O = P V
// with the products formed like every other fp32 contraction here (three bf16 pieces per operand, six MFMAs, KoafGemm.fmt 0).
// Replaces GEMM -> softmax kernel -> GEMM with two (B, h, n, n) round trips; the attention maps are still emitted (they are
// returned by the reference, :182).  Block = 4 waves; wave w owns columns [32 w, 32 w + 32) of each 128-wide column chunk.
// ================================================================================================
namespace {
constexpr int ATT_BM = 32, ATT_NMAX = 512, ATT_SP = ATT_NMAX + 4;

__global__ void __launch_bounds__(512) attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ attn,
                                                            float* __restrict__ out, int n, int h, int d, float scale) {
    constexpr int A_PL = plane_dwords(ATT_BM, true), BK_PL = plane_dwords(128, true), BV_PL = plane_dwords(128, false);
    constexpr int B_PL = BK_PL > BV_PL ? BK_PL : BV_PL;
    __shared__ __attribute__((aligned(16))) float Ss[ATT_BM * ATT_SP];
    __shared__ __attribute__((aligned(16))) float Aps[2][3 * A_PL];
    __shared__ __attribute__((aligned(16))) float Bps[2][3 * B_PL];
    // block = (256, 2): two groups of four waves, each with its own operand planes, take alternate column chunks -- the operand
    // split (vector ALU) of one group runs under the MFMAs of the other (one group per CU left every SIMD with a single wave)
    const int t = threadIdx.x, grp = threadIdx.y, lane = t & 63, w = t >> 6, r = lane & 31, hh = lane >> 5;
    float* const Ap = Aps[grp];
    float* const Bp = Bps[grp];
    const int b = blockIdx.y / h, head = blockIdx.y - b * h;
    const int m0 = blockIdx.x * ATT_BM;
    const int64_t ld = 3ll * h * d;
    const float* Q = qkv + (int64_t)b * n * ld + (int64_t)head * d;
    const float* K = Q + (int64_t)h * d;
    const float* V = Q + 2ll * h * d;
    KoafOperand op{};
    op.ld = ld;
    constexpr int PA3[6] = {2, 0, 1, 1, 0, 0}, PB3[6] = {0, 2, 1, 0, 1, 0};       // piece products, smallest first
    auto mma = [&](v16f& acc, bool bkc) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            v4i ap[3], bp[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                ap[q] = frag_load<ATT_BM, true>((const unsigned*)Ap + q * A_PL, 0, g, lane);
                bp[q] = bkc ? frag_load<128, true>((const unsigned*)Bp + q * BK_PL, 32 * w, g, lane)
                            : frag_load<128, false>((const unsigned*)Bp + q * BV_PL, 32 * w, g, lane);
            }
#pragma unroll
            for (int term = 0; term < 6; ++term)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ap[PA3[term]]),
                                                              __builtin_bit_cast(bf16x8, bp[PB3[term]]), acc, 0, 0, 0);
        }
    };
    // ---- S = scale * Q K^T, 128 keys at a time ----
    const int nchunk = (n + 127) / 128;
    for (int c = grp; c < ((nchunk + 1) & ~1); c += 2) {       // (both groups run the same number of barriers; a chunk past n is all zero rows)
        TileLoader<ATT_BM, M_KC, 0, true, false> la;
        TileLoader<128, M_KC, 0, true, false> lb;
        la.init(op, m0, n, 0, 1.f);
        lb.init(op, 128 * c, n, 0, 1.f);
        la.seek(op, 0);
        lb.seek(op, 0);
        v16f acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        // two k-steps of loads in flight (slots sa / sb alternate): with one block of four waves per CU nothing else hides
        // the load latency
        la.issue(la.sa, op, Q, 0, d, 0);
        lb.issue(lb.sa, op, K, 0, d, 0);
        if (BK < d) {
            la.issue(la.sb, op, Q, BK, d, 0);
            lb.issue(lb.sb, op, K, BK, d, 0);
        }
        la.finish(la.sa);
        la.template store<3>(la.sa, Ap);
        lb.finish(lb.sa);
        lb.template store<3>(lb.sa, Bp);
        __syncthreads();
        auto step = [&](int k0, auto& fa, auto& fb, auto& na, auto& nb) {     // f*: the slot this step's tile came from (free)
            if (k0 + 2 * BK < d) {
                la.issue(fa, op, Q, k0 + 2 * BK, d, 0);
                lb.issue(fb, op, K, k0 + 2 * BK, d, 0);
            }
            mma(acc, true);
            __syncthreads();
            if (k0 + BK < d) {
                la.finish(na);
                la.template store<3>(na, Ap);
                lb.finish(nb);
                lb.template store<3>(nb, Bp);
                __syncthreads();
            }
        };
        for (int k0 = 0; k0 < d; k0 += 2 * BK) {
            step(k0, la.sa, lb.sa, la.sb, lb.sb);
            if (k0 + BK < d) step(k0 + BK, la.sb, lb.sb, la.sa, lb.sa);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)      // (keys past n multiplied zero rows: their scores are 0 and the softmax skips them)
            if (c < nchunk) Ss[((e & 3) + 8 * (e >> 2) + 4 * hh) * ATT_SP + 128 * c + 32 * w + r] = scale * acc[e];
    }
    __syncthreads();
    // ---- softmax rows (the arithmetic of koaf_softmax_rows: one wave per row), attn written once ----
    for (int row = w + 4 * grp; row < ATT_BM; row += 8) {
        float* xr = Ss + row * ATT_SP;
        float m = -INFINITY;
        for (int i = lane; i < n; i += 64) m = fmaxf(m, xr[i]);
        m = wave_max(m);
        float sum = 0.f;
        for (int i = lane; i < n; i += 64) sum += expf(xr[i] - m);
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        const bool live = (m0 + row) < n;
        float* ar = attn + (((int64_t)blockIdx.y * n) + m0 + row) * n;
        for (int i = lane; i < n; i += 64) {
            const float pv = expf(xr[i] - m) * inv;
            xr[i] = pv;
            if (live) ar[i] = pv;
        }
    }
    __syncthreads();
    // ---- O = P V, 128 head-dimension columns at a time; the A tile comes from the scores in LDS ----
    for (int j = grp; j < (((d + 127) / 128 + 1) & ~1); j += 2) {
        TileLoader<128, M_KM, 0, true, false> lv;
        lv.init(op, 128 * j, d, 0, 1.f);
        lv.seek(op, 0);
        v16f acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        auto stage_p = [&](int k0) {          // P[32 rows][k0 .. k0 + 32) -> the three bf16 planes (TileLoader's K-contiguous image)
            const int row = t >> 3, kk = k0 + 4 * (t & 7);
            v4f x = *(const v4f*)&Ss[row * ATT_SP + kk];          // (columns past n hold zeros)
            unsigned pl[3][2];
            split3v(x, pl);
            unsigned* S = (unsigned*)Ap;
            const int off = row * 20 + 2 * (t & 7);
#pragma unroll
            for (int q = 0; q < 3; ++q) *(uint2*)&S[q * A_PL + off] = make_uint2(pl[q][0], pl[q][1]);
        };
        lv.issue(lv.sa, op, V, 0, n, 0);
        if (BK < n) lv.issue(lv.sb, op, V, BK, n, 0);
        stage_p(0);
        lv.finish(lv.sa);
        lv.template store<3>(lv.sa, Bp);
        __syncthreads();
        auto step = [&](int k0, auto& fv, auto& nv) {
            if (k0 + 2 * BK < n) lv.issue(fv, op, V, k0 + 2 * BK, n, 0);
            mma(acc, false);
            __syncthreads();
            if (k0 + BK < n) {
                stage_p(k0 + BK);
                lv.finish(nv);
                lv.template store<3>(nv, Bp);
                __syncthreads();
            }
        };
        for (int k0 = 0; k0 < n; k0 += 2 * BK) {
            step(k0, lv.sa, lv.sb);
            if (k0 + BK < n) step(k0 + BK, lv.sb, lv.sa);
        }
        const int col = 128 * j + 32 * w + r;
        if (col < d) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (row < n) out[((int64_t)b * n + row) * ((int64_t)h * d) + (int64_t)head * d + col] = acc[e];
            }
        }
        __syncthreads();
    }
}
}  // namespace

// returns KOAF_OK when the fused kernel took the call, 1 when the shape is outside it (the caller runs the three-launch path)
int koaf_attention_fwd_fused(const float* qkv, float* attn, float* out, int32_t B, int32_t n, int32_t h, int32_t d, float scale,
                             void* stream) {
    if (n > ATT_NMAX || (d & 3) || !aligned16(qkv) || (int64_t)B * h > 65535) return 1;
    hipLaunchKernelGGL(attention_fwd_kernel, dim3((unsigned)((n + ATT_BM - 1) / ATT_BM), (unsigned)(B * h)), dim3(256, 2), 0,
                       (hipStream_t)stream, qkv, attn, out, n, h, d, scale);
    return koaf_check_launch("koaf_attention_fwd");
}

// ================================================================================================
// weight plane images (the M_PS operand): cut once per optimizer step for every convolution weight of the model
// ================================================================================================
namespace {
// block -> (descriptor, tile): the last descriptor whose first tile is <= blockIdx.x; tile = 32 (rows) x 32 (k of one tap)
struct WTile { KoafWPlane d; int idx, rt, tap, ct; };
__device__ __forceinline__ WTile wtile_of_block(const KoafWPlane* __restrict__ tab, int ntab) {
    int lo = 0, hi = ntab - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].tile0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    WTile w;
    w.d = tab[lo];
    w.idx = lo;
    int tl = (int)((int64_t)blockIdx.x - w.d.tile0);
    const int nct = (w.d.C + 31) / 32;
    w.ct = tl % nct; tl /= nct;
    w.tap = tl % w.d.taps;
    w.rt = tl / w.d.taps;
    return w;
}
__device__ __forceinline__ v4f wtile_load(const float* __restrict__ base, const WTile& w, int r, int c) {
    const KoafWPlane& d = w.d;
    const int64_t K = (int64_t)d.taps * d.C;
    v4f x = {0.f, 0.f, 0.f, 0.f};
    if (r < d.R) {
        const float* s = base + d.src_off + (int64_t)r * K + (int64_t)w.tap * d.C + c;
        if (c + 3 < d.C && ((K | d.C) & 3) == 0) x = *(const v4f*)s;
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (c + j < d.C) x[j] = s[j];
        }
    }
    return x;
}

// pass 1: amax[i] = max |w| of weight i (amax zeroed beforehand; float bits of non-negative values order like integers)
__global__ void __launch_bounds__(256) wplanes_amax_kernel(const float* __restrict__ base, const KoafWPlane* __restrict__ tab,
                                                           int ntab, float* __restrict__ amax) {
    const WTile w = wtile_of_block(tab, ntab);
    const int t = threadIdx.x;
    const v4f x = wtile_load(base, w, w.rt * 32 + (t >> 3), w.ct * 32 + 4 * (t & 7));
    block_amax_raise_bits(max(max(koaf_absbits(x[0]), koaf_absbits(x[1])), max(koaf_absbits(x[2]), koaf_absbits(x[3]))), amax + w.idx);
}

// pass 2: the images of w * scale_of_amax(amax[i]) (split2h: bit-identical to the in-kernel split of the same operand)
//   F image [2][R][Kp]         (Kp = taps * C rounded up to 32; forward B operand: rows = output channels)
//   D image [2][C][taps * Rp]  (Rp = R rounded up to 32; the transposed weight, dgrad B operand: rows = input channels,
//                               k = (tap, output channel)); the tile is transposed through LDS.
// Both are zero-filled up to their padded extents.
__global__ void __launch_bounds__(256) wplanes_build_kernel(const float* __restrict__ base, unsigned short* __restrict__ planes,
                                                            const KoafWPlane* __restrict__ tab, int ntab,
                                                            const float* __restrict__ amax) {
    __shared__ unsigned short tile[2][32][36];      // [plane][c][r] (+4 pad)
    const WTile w = wtile_of_block(tab, ntab);
    const KoafWPlane& d = w.d;
    const int t = threadIdx.x, ty = t >> 3, tx = t & 7;
    const int r = w.rt * 32 + ty, c = w.ct * 32 + 4 * tx;
    unsigned pl[2][2];
    {
        v4f x = wtile_load(base, w, r, c);
        const float sc = scale_of_amax(amax[w.idx]);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = __builtin_amdgcn_fmed3f(x[j] * sc, -65504.f, 65504.f);
        split2h(x, pl);
    }
    if (d.f_off >= 0 && r < d.R) {
        // (c + 3 < Kp always: Kp and c are multiples of 4, the tile covers C rounded up to 32 only when taps == 1)
        unsigned short* f = planes + d.f_off + (int64_t)r * d.Kp + (int64_t)w.tap * d.C + c;
        const int64_t ps = (int64_t)d.R * d.Kp;
#pragma unroll
        for (int q = 0; q < 2; ++q) *(uint2*)(f + q * ps) = make_uint2(pl[q][0], pl[q][1]);
    }
    if (d.d_off < 0) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        tile[q][4 * tx + 0][ty] = (unsigned short)(pl[q][0] & 0xffffu);
        tile[q][4 * tx + 1][ty] = (unsigned short)(pl[q][0] >> 16);
        tile[q][4 * tx + 2][ty] = (unsigned short)(pl[q][1] & 0xffffu);
        tile[q][4 * tx + 3][ty] = (unsigned short)(pl[q][1] >> 16);
    }
    __syncthreads();
    const int cc = w.ct * 32 + ty;                   // this thread now owns input channel cc, rows rt*32 + 4tx .. +3
    if (cc < d.C) {
        const int64_t ldd = (int64_t)d.taps * d.Rp, ps = (int64_t)d.C * ldd;
        unsigned short* o = planes + d.d_off + (int64_t)cc * ldd + (int64_t)w.tap * d.Rp + w.rt * 32 + 4 * tx;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const unsigned short* sr = &tile[q][ty][4 * tx];
            *(uint2*)(o + q * ps) = make_uint2((unsigned)sr[0] | ((unsigned)sr[1] << 16), (unsigned)sr[2] | ((unsigned)sr[3] << 16));
        }
    }
}
}  // namespace

extern "C" int koaf_wplanes_build(const float* base, uint16_t* planes, float* amax, const KoafWPlane* table_dev, int32_t n,
                                  int64_t ntiles, void* stream) {
    KOAF_REQUIRE(base && planes && amax && table_dev && n > 0 && ntiles > 0 && ntiles < (1ll << 31), "koaf_wplanes_build: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(amax, 0, sizeof(float) * (size_t)n, s) != hipSuccess) {
        koaf_set_error("koaf_wplanes_build: memset failed");
        return KOAF_ELAUNCH;
    }
    hipLaunchKernelGGL(wplanes_amax_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, base, table_dev, n, amax);
    int rc = koaf_check_launch("koaf_wplanes_build/amax");
    if (rc != KOAF_OK) return rc;
    hipLaunchKernelGGL(wplanes_build_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, base, planes, table_dev, n, amax);
    return koaf_check_launch("koaf_wplanes_build");
}


// ================================================================================================
// activation plane images (the M_PA operand): the fp16 piece planes of an NHWC tensor, transform included
// ================================================================================================
namespace {
// TF as in TileLoader (0 none, 1 relu(sc*x+sh), 2 sc*x + sh - sc2*x2); the arithmetic is finish_unit()'s + split2h, so the
// images hold bit for bit what the fp32 loader of the same operand puts into LDS.
// X16: the activation among the sources is stored as bf16 (tf 0 / 1: x; tf 2: x2 = the conv output c)
template <int TF, bool X16>
__global__ void __launch_bounds__(256) act_planes_kernel(const float* __restrict__ x, const float* __restrict__ x2, int64_t n8,
                                                         int C, const float* __restrict__ sc, const float* __restrict__ sh,
                                                         const float* __restrict__ sc2, const float* __restrict__ amax,
                                                         float fscale, unsigned short* __restrict__ planes, int64_t ps,
                                                         uint32_t* status) {
    constexpr float HMAX = 65504.f;
    unsigned nsat = 0;      // elements beyond the fp16 range of the scale (clamped below) or not finite
    const float fsc = amax ? scale_of_amax(*amax) : (fscale != 0.f ? fscale : 1.f);
    if (blockIdx.x == 0 && threadIdx.x == 0) *(uint4*)(planes + 2 * ps) = make_uint4(0u, 0u, 0u, 0u);   // the zero chunk
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int c = (int)((i * 8) % C);
        unsigned pl[2][2][2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            v4f v = load4<X16 && TF != 2>(x, i * 8 + 4 * hf);
            if constexpr (TF == 1) {
                const v4f a = *(const v4f*)(sc + c + 4 * hf) * fsc, b = *(const v4f*)(sh + c + 4 * hf) * fsc;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float u = fmaf(v[j], a[j], b[j]);
                    nsat += !(u <= HMAX) ? 1u : 0u;
                    v[j] = __builtin_amdgcn_fmed3f(u, 0.f, HMAX);
                }
            } else if constexpr (TF == 2) {
                const v4f a = *(const v4f*)(sc + c + 4 * hf) * fsc, b = *(const v4f*)(sh + c + 4 * hf) * fsc;
                const v4f k = *(const v4f*)(sc2 + c + 4 * hf) * fsc;
                const v4f w = load4<X16>(x2, i * 8 + 4 * hf);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float u = fmaf(a[j], v[j], fmaf(-k[j], w[j], b[j]));
                    nsat += !(fabsf(u) <= HMAX) ? 1u : 0u;
                    v[j] = __builtin_amdgcn_fmed3f(u, -HMAX, HMAX);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float u = v[j] * fsc;
                    nsat += !(fabsf(u) <= HMAX) ? 1u : 0u;
                    v[j] = __builtin_amdgcn_fmed3f(u, -HMAX, HMAX);
                }
            }
            split2h(v, pl[hf]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
            *(uint4*)(planes + q * ps + i * 8) = make_uint4(pl[0][q][0], pl[0][q][1], pl[1][q][0], pl[1][q][1]);
    }
    koaf_status_add(status, 0, nsat);
}
}  // namespace

extern "C" int64_t koaf_act_planes_elems(int64_t npix, int32_t C) { return 2 * npix * C + 8; }

extern "C" int koaf_act_planes(const float* x, const float* x2, int64_t npix, int32_t C, int32_t tf, const float* sc,
                               const float* sh, const float* sc2, const float* amax, float fscale, uint16_t* planes,
                               int32_t act16, void* stream) {
    KOAF_REQUIRE(x && planes && npix > 0 && C > 0 && (C & 7) == 0 && tf >= 0 && tf <= 2, "koaf_act_planes: bad args (C %% 8 == 0)");
    KOAF_REQUIRE(tf == 0 || (sc && sh), "koaf_act_planes: tf needs sc / sh");
    KOAF_REQUIRE(tf != 2 || (x2 && sc2), "koaf_act_planes: tf 2 needs x2 / sc2");
    KOAF_REQUIRE(aligned16(x) && aligned16(planes) && (tf != 2 || aligned16(x2)) && (tf == 0 || (aligned16(sc) && aligned16(sh))),
                 "koaf_act_planes: unaligned");
    const int64_t ps = npix * C, n8 = ps / 8;
    int64_t blocks = cdiv64(n8, 256);
    if (blocks > 16384) blocks = 16384;
    hipStream_t s = (hipStream_t)stream;
#define KOAF_AP(TF_, X_) hipLaunchKernelGGL((act_planes_kernel<TF_, X_>), dim3((unsigned)blocks), dim3(256), 0, s, x, x2, n8, C, sc, sh, sc2, amax, fscale, planes, ps, koaf_status_ptr())
    if (tf == 0) { if (act16) KOAF_AP(0, true); else KOAF_AP(0, false); }
    else if (tf == 1) { if (act16) KOAF_AP(1, true); else KOAF_AP(1, false); }
    else { if (act16) KOAF_AP(2, true); else KOAF_AP(2, false); }
#undef KOAF_AP
    return koaf_check_launch("koaf_act_planes");
}

#ifdef KOAF_STAMPS
// out[8] = {prologue, k-loop, staging, stores, tile, chunk waits, -, tiles} summed over blocks, in 10 ns ticks; reset: zero the table
extern "C" int koaf_debug_stamps(unsigned long long* out, int reset) {
    static unsigned long long h[64][8];
    if (hipDeviceSynchronize() != hipSuccess) return KOAF_ELAUNCH;
    if (out) {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(koaf_stamp_tab), sizeof(h)) != hipSuccess) return KOAF_ELAUNCH;
        for (int k = 0; k < 8; ++k) { out[k] = 0; for (int r = 0; r < 64; ++r) out[k] += h[r][k]; }
    }
    if (reset) {
        for (int r = 0; r < 64; ++r) for (int k = 0; k < 8; ++k) h[r][k] = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(koaf_stamp_tab), h, sizeof(h)) != hipSuccess) return KOAF_ELAUNCH;
    }
    return KOAF_OK;
}
#endif

extern "C" int koaf_set_stream(int on) {
    if (g_stream_mode < 0) { const char* e = getenv("KOAF_STREAM"); g_stream_mode = (e && e[0] == '0') ? 0 : 1; }
    const int was = g_stream_mode;
    g_stream_mode = on ? 1 : 0;
    return was;
}

extern "C" int koaf_set_conv3x3_halo(int on) {
    const int was = g_halo_mode;
    g_halo_mode = (on >= 0 && on <= 3) ? on : 1;
    return was;
}
