// Stem forward on the matrix pipe: the 7x7 / stride-2 / pad-3 convolution of the single-channel image (reference:
// koafusion/models/_torchvision.py:170, with the 1 -> 3 channel repeat of _xrNmrMcP.py:211-213 folded into the weights) as
//     y[p][co] = sum_{kh, kw} x[2 oy + kh - 3][2 ox + kw - 3] * w1t[kh * 7 + kw][co],     k = (kh, kw) padded to 8 x 8 = 64.
// The vector kernel (koaf_conv.hip stem_fwd_kernel) spends 49 multiply-adds per output on the plain FMA pipe and runs at 70 %
// of that pipe's peak -- twice the time the 12 GB output takes to write.  Here every product is formed like the other fp32
// contractions of this library (KoafGemm.fmt 0): both operands cut into three bf16 pieces, six v_mfma_f32_32x32x16_bf16 per
// 16 k (no operand scale: bf16 keeps fp32's exponent range, so raw images of any magnitude are fine).
// Unit = one image x four output rows (a wave per row), walked in 192-column bands by persistent blocks (two per CU; the next
// band's rows are fetched into registers under the current band's matrix work); the band's 13 input rows sit in LDS as
// three bf16 planes; an A fragment (32 output pixels x 16 k = two filter rows of eight taps) is four ds_read_b32 per plane --
// lane r reads taps 2 r + kw, its left neighbour's pixels shifted by two -- and the weights' fragments stay in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include "koaf.h"
#include "koaf_common.h"

namespace {
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SM_TH = 4;                       // output rows per block (one wave each)
constexpr int SM_BAND = 192;                   // output columns per band (six 32-pixel tiles)
constexpr int SM_PH = 2 * SM_TH + 5;           // 13 input rows
constexpr int SM_PW = 2 * SM_BAND + 8;         // 392 input columns (taps 2 ox + 0..7)

// three bf16 pieces of x by truncation (koaf_gemm.hip split3v): x = p0 + p1 + p2 to 2^-24 relative
__device__ __forceinline__ void split3(float x, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned b0 = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(b0);
    const unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(b1);
    p0 = b0 >> 16;
    p1 = b1 >> 16;
    p2 = __float_as_uint(r2) >> 16;
}

template <bool Y16, bool ST>
__global__ void __launch_bounds__(256) stem_fwd_mma_kernel(const float* __restrict__ x, const float* __restrict__ w1t,
                                                           float* __restrict__ y, int N, int H, int W, int OH, int OW,
                                                           float* __restrict__ part, const float* __restrict__ shift) {
    __shared__ __attribute__((aligned(16))) unsigned short patch[3][SM_PH][SM_PW];
    __shared__ float sred[2][SM_TH][64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, r = lane & 31, hh = lane >> 5;
    const int ty = (OH + SM_TH - 1) / SM_TH;
    const int nunit = N * ty;                       // units = (image, band of four output rows); blocks are persistent over them

    // weight fragments: B operand lane (col = 32 jn + r, k = 16 g + 8 hh + j <-> kh = 2 g + hh, kw = j); kh = 7 / kw = 7: zero.
    // Built once per block (64 loads and splits per lane: a third of a unit's time when every unit was its own block)
    v4i bf[4][2][3];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int kh = 2 * g + hh;
            unsigned pc[3][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float wv = (kh < 7 && j < 7) ? w1t[(kh * 7 + j) * 64 + 32 * jn + r] : 0.f;
                split3(wv, pc[0][j], pc[1][j], pc[2][j]);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q)
                bf[g][jn][q] = (v4i){(int)(pc[q][0] | (pc[q][1] << 16)), (int)(pc[q][2] | (pc[q][3] << 16)),
                                     (int)(pc[q][4] | (pc[q][5] << 16)), (int)(pc[q][6] | (pc[q][7] << 16))};
        }
    float ksh[2] = {0.f, 0.f};
    if constexpr (ST) {
        if (shift) { ksh[0] = shift[r]; ksh[1] = shift[32 + r]; }
    }
    // the input rows of one (unit, column band) as registers: two columns per slot, PRE slots per thread
    constexpr int PAIRS = SM_PH * (SM_PW / 2), PRE = (PAIRS + 255) / 256;
    float pre0[PRE], pre1[PRE];
    auto fetch = [&](int u, int ob) {
        const int n = u / ty, by = u - n * ty;
        const int iy0 = 2 * by * SM_TH - 3, ix0 = 2 * ob - 3;
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const int i = t + 256 * k;
            const int pr = i / (SM_PW / 2), pc2 = i - pr * (SM_PW / 2);
            const int iy = iy0 + pr, ix = ix0 + 2 * pc2;
            float v0 = 0.f, v1 = 0.f;
            if (i < PAIRS && (unsigned)iy < (unsigned)H) {
                const float* row = x + ((int64_t)n * H + iy) * W;
                if ((unsigned)ix < (unsigned)W) v0 = row[ix];
                if ((unsigned)(ix + 1) < (unsigned)W) v1 = row[ix + 1];
            }
            pre0[k] = v0;
            pre1[k] = v1;
        }
    };
    auto stage = [&]() {                            // registers -> three bf16 planes in LDS (one dword per plane and slot)
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const int i = t + 256 * k;
            if (i < PAIRS) {
                const int pr = i / (SM_PW / 2), pc2 = i - pr * (SM_PW / 2);
                unsigned a0, a1, a2, b0, b1, b2;
                split3(pre0[k], a0, a1, a2);
                split3(pre1[k], b0, b1, b2);
                *(unsigned*)&patch[0][pr][2 * pc2] = a0 | (b0 << 16);
                *(unsigned*)&patch[1][pr][2 * pc2] = a1 | (b1 << 16);
                *(unsigned*)&patch[2][pr][2 * pc2] = a2 | (b2 << 16);
            }
        }
    };
    int u = blockIdx.x, ob = 0;
    if (u < nunit) fetch(u, 0);
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    while (u < nunit) {
        __syncthreads();                            // (every wave is done reading the previous band)
        stage();
        __syncthreads();
        // the next (unit, band): its loads fly under this band's matrix work
        int un = u, obn = ob + SM_BAND;
        if (obn >= OW) { un = u + gridDim.x; obn = 0; }
        if (un < nunit) fetch(un, obn);
        const int n = u / ty, by = u - n * ty;
        const int oy = by * SM_TH + w;
        if (oy < OH) {
            const int ntile = min(SM_BAND, OW - ob + 31) / 32;          // 32-pixel tiles of this band (the last may be ragged)
#pragma unroll 1
            for (int tl = 0; tl < ntile; ++tl) {
                v16f acc[2];
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[jn][e] = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // (kh = 7 does not exist: its weights are zero; the row read instead is the band's last, finite)
                    const int pr = 2 * w + min(2 * g + hh, 6);
                    v4i af[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const unsigned* src = (const unsigned*)&patch[q][pr][2 * (32 * tl + r)];
                        af[q] = (v4i){(int)src[0], (int)src[1], (int)src[2], (int)src[3]};
                    }
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};     // piece products, smallest first
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                        for (int term = 0; term < 6; ++term)
                            acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[PA[term]]),
                                                                              __builtin_bit_cast(bf16x8, bf[g][jn][PB[term]]),
                                                                              acc[jn], 0, 0, 0);
                }
                // lane (r, hh) holds channel 32 jn + r of pixels ox0 + (e & 3) + 8 (e >> 2): one base address per tile, constant offsets
                const int ox0 = ob + 32 * tl + 4 * hh;
                const int64_t base = (((int64_t)n * OH + oy) * OW + ox0) * 64 + r;
                auto put = [&](int jn, int e, bool guard) {
                    const int dx = (e & 3) + 8 * (e >> 2);
                    if (guard && ox0 + dx >= OW) return;
                    float v = acc[jn][e];
                    const int64_t idx = base + dx * 64 + 32 * jn;
                    if constexpr (Y16) {
                        const __bf16 rb = (__bf16)v;
                        reinterpret_cast<__bf16*>(y)[idx] = rb;
                        v = (float)rb;
                    } else y[idx] = v;
                    if constexpr (ST) { const float d = v - ksh[jn]; s1[jn] += d; s2[jn] += d * d; }
                };
                if (ob + 32 * tl + 32 <= OW) {
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                        for (int e = 0; e < 16; ++e) put(jn, e, false);
                } else {
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                        for (int e = 0; e < 16; ++e) put(jn, e, true);
                }
            }
        }
        if constexpr (ST) {
            if (obn == 0) {
                // the unit is complete: lanes r and r + 32 hold the same channels -- fold, then the four rows through LDS ->
                // part [unit][2][64]
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) {
                    s1[jn] += __shfl_xor(s1[jn], 32);
                    s2[jn] += __shfl_xor(s2[jn], 32);
                }
                if (hh == 0) {
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn) {
                        sred[0][w][32 * jn + r] = s1[jn];
                        sred[1][w][32 * jn + r] = s2[jn];
                    }
                }
                __syncthreads();
                if (t < 128) {
                    const int which = t >> 6, co = t & 63;
                    float a = 0.f;
#pragma unroll
                    for (int g = 0; g < SM_TH; ++g) a += sred[which][g][co];
                    part[((int64_t)u * 2 + which) * 64 + co] = a;
                }
                s1[0] = s1[1] = s2[0] = s2[1] = 0.f;
            }
        }
        u = un;
        ob = obn;
    }
}
}  // namespace

// rows of the statistics buffer (one per block) -- the same tiling as the vector kernel's (koaf_stem_stats_rows)
int koaf_stem_fwd_mma(const float* x, const float* w1t, float* y, int N, int H, int W, float* stats, const float* stats_shift,
                      int act16, void* stream) {
    static const bool off = [] { const char* e = getenv("KOAF_STEM_MMA"); return e && e[0] == '0'; }();
    if (off) return 1;
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    const int64_t units = (int64_t)N * ((OH + SM_TH - 1) / SM_TH);
    if (units >= (1ll << 31)) return 1;
    const dim3 grid((unsigned)(units < 512 ? units : 512));          // persistent: two blocks per CU
    hipStream_t st = (hipStream_t)stream;
    if (stats) {
        if (act16) hipLaunchKernelGGL((stem_fwd_mma_kernel<true, true>), grid, dim3(256), 0, st, x, w1t, y, N, H, W, OH, OW, stats, stats_shift);
        else hipLaunchKernelGGL((stem_fwd_mma_kernel<false, true>), grid, dim3(256), 0, st, x, w1t, y, N, H, W, OH, OW, stats, stats_shift);
    } else {
        if (act16) hipLaunchKernelGGL((stem_fwd_mma_kernel<true, false>), grid, dim3(256), 0, st, x, w1t, y, N, H, W, OH, OW, nullptr, nullptr);
        else hipLaunchKernelGGL((stem_fwd_mma_kernel<false, false>), grid, dim3(256), 0, st, x, w1t, y, N, H, W, OH, OW, nullptr, nullptr);
    }
    return koaf_check_launch("koaf_stem_fwd/mma");
}

// ================================================================================================
// Stem weight gradient on the matrix pipe:  dW1t[kh * 7 + kw][co] = sum_p x[2 oy + kh - 3][2 ox + kw - 3] * dc[p][co]
// as M = the 8 x 8 padded taps, N = 64 channels, K = output pixels (16 consecutive pixels of one output row per MFMA).
//   * A (taps x pixels): eight consecutive pixels of tap (kh, kw) are every OTHER input column -- the band's input rows are kept
//     in LDS split by column parity (column c = 2 ox + kw: parity kw & 1, half-index ox + (kw >> 1)), each parity twice, the
//     second copy shifted by one element, so that the eight values are four ds_read_b32 from a 4-byte aligned address whatever
//     the tap;
//   * B (pixels x channels): dc itself, formed on load from (dz, c, coef) like the vector kernel (KoafBnApply) or read as given,
//     cut into three bf16 pieces and written by each wave to its own [16 pixels][64 channels] image, read back transposed
//     (ds_read_b64_tr_b16) -- no block barrier inside a unit;
//   * every wave (one output row of the unit) accumulates the whole 64 x 64 gradient; the four are summed through LDS at the
//     end and the block writes one slab [49][64] (koaf_slab_reduce finishes).
// ================================================================================================
namespace {
typedef short v4s __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int SW_BAND = 96;                    // output columns per band of the weight gradient (two blocks per CU fit)
constexpr int SW_PW = 2 * SW_BAND + 8;         // input columns of a band
constexpr int SW_HW = SW_BAND + 8;             // half-columns per parity row (ox + (kw >> 1) + shift, ox < 96)
constexpr int SW_XP = 2 * 2 * 3 * SM_PH * SW_HW * 2, SW_DC = 4 * 3 * 16 * 64 * 2;       // bytes: x patch, dc images

template <bool APPLY, bool C16>
__global__ void __launch_bounds__(256) stem_wgrad_mma_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             float* __restrict__ slabs, int N, int H, int W, int OH, int OW,
                                                             const float* __restrict__ cc, const float* __restrict__ coef) {
    static_assert(SW_XP % 16 == 0 && SW_XP + SW_DC >= 3 * 64 * 64 * 4, "the final reduction reuses the operand buffers");
    __shared__ __attribute__((aligned(16))) unsigned char smem[SW_XP + SW_DC];
    // x patch: [parity][shift][piece][row][half-column] bf16; shift s stores half-index i at position i + s
    typedef unsigned short (*xp_t)[2][3][SM_PH][SW_HW];
    typedef unsigned short (*dci_t)[3][16][64];
    const xp_t xp = reinterpret_cast<xp_t>(smem);
    // dc images: per wave, [piece][16 pixels][64 channels] bf16, 16-B chunks swizzled by the pixel (chunk ^ 4 ((px >> 1) & 1))
    const dci_t dci = reinterpret_cast<dci_t>(smem + SW_XP);
    typedef __attribute__((address_space(3))) v4s* lds_v4s;
    typedef __attribute__((address_space(3))) unsigned char* lds_b;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, r = lane & 31, hh = lane >> 5;
    const int ty = (OH + SM_TH - 1) / SM_TH, nunit = N * ty;

    // A fragment geometry of this lane: tap (kh, kw) of M tile mi
    int arow[2], apar[2], ash[2], aoff[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int kidx = 32 * mi + r, kh = kidx >> 3, kw = kidx & 7;
        arow[mi] = min(kh, 6);                 // (kh = 7 / kw = 7 are padding rows of M: computed, never stored)
        apar[mi] = kw & 1;
        ash[mi] = (kw >> 1) & 1;
        aoff[mi] = (kw >> 1) + ash[mi];        // position of pixel ox in the copy = ox + aoff (even for ox even)
    }
    // B fragment geometry (transposing reads of the wave's dc image): lane -> (pixel row q4 of a 4-row group, 4-channel group)
    const int li = lane & 15, q4 = li >> 2, pp = li & 3;
    const int kb = 8 * hh + q4;                // + 4 for the second half
    unsigned boff[2][2];
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
        const int rb = 32 * nj + 16 * ((lane >> 4) & 1) + 4 * pp;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int k = kb + 4 * rd;
            boff[nj][rd] = (unsigned)(k * 128 + (((rb >> 3) ^ (4 * ((k >> 1) & 1))) << 4) + ((rb & 7) << 1));
        }
    }
    const lds_b dcb = (lds_b)(smem + SW_XP + w * (3 * 16 * 64 * 2));
    // this lane's share of a dc tile: pixel lane >> 2 (16 pixels), channels 16 (lane & 3) .. + 15 -> four 16-B stores per piece
    const int spx = lane >> 2, sch = 16 * (lane & 3);
    float k0[16], k2[16], k3[16];
    if constexpr (APPLY) {
#pragma unroll
        for (int j = 0; j < 16; ++j) { k0[j] = coef[sch + j]; k2[j] = coef[128 + sch + j]; k3[j] = coef[192 + sch + j]; }
    }
    v16f acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

    for (int u = blockIdx.x; u < nunit; u += gridDim.x) {
        const int n = u / ty, by = u - n * ty;
        const int oy = by * SM_TH + w;
        const int iy0 = 2 * by * SM_TH - 3;
        for (int ob = 0; ob < OW; ob += SW_BAND) {
            // ---- the band's input rows -> parity-split bf16 planes, two copies each ----
            __syncthreads();
            const int ix0 = 2 * ob - 3;
            for (int i = t; i < SM_PH * (SW_PW / 2); i += 256) {
                const int pr = i / (SW_PW / 2), hc = i - pr * (SW_PW / 2);       // column pair (2 hc, 2 hc + 1): half-index hc of each parity
                const int iy = iy0 + pr, ix = ix0 + 2 * hc;
                float v0 = 0.f, v1 = 0.f;
                if ((unsigned)iy < (unsigned)H) {
                    const float* row = x + ((int64_t)n * H + iy) * W;
                    if ((unsigned)ix < (unsigned)W) v0 = row[ix];
                    if ((unsigned)(ix + 1) < (unsigned)W) v1 = row[ix + 1];
                }
                unsigned a[3], b[3];
                split3(v0, a[0], a[1], a[2]);
                split3(v1, b[0], b[1], b[2]);
                if (hc < SW_HW - 1) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        xp[0][0][q][pr][hc] = (unsigned short)a[q];
                        xp[0][1][q][pr][hc + 1] = (unsigned short)a[q];
                        xp[1][0][q][pr][hc] = (unsigned short)b[q];
                        xp[1][1][q][pr][hc + 1] = (unsigned short)b[q];
                    }
                }
            }
            __syncthreads();
            if (oy >= OH) continue;
            const int npx = min(SW_BAND, OW - ob);
            const int64_t rowbase = (((int64_t)n * OH + oy) * OW + ob) * 64;
            // (the next 16 pixels' dz / c are fetched before this tile is multiplied: the counters showed the waves waiting on
            // these loads 62 % of the time)
            v4f g[4], c4[4];
            auto fetch_dc = [&](int p0, v4f (&gg)[4], v4f (&cq)[4]) {
                const bool ok = p0 + spx < npx;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int64_t o = rowbase + (int64_t)(p0 + spx) * 64 + sch + 4 * v;
                    gg[v] = ok ? *(const v4f*)(dy + o) : (v4f){0.f, 0.f, 0.f, 0.f};
                    if constexpr (APPLY) {
                        if constexpr (C16) {
                            const uint2 raw = ok ? *(const uint2*)(reinterpret_cast<const unsigned short*>(cc) + o) : make_uint2(0u, 0u);
                            cq[v] = (v4f){__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u),
                                          __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xffff0000u)};
                        } else cq[v] = ok ? *(const v4f*)(cc + o) : (v4f){0.f, 0.f, 0.f, 0.f};
                    } else cq[v] = (v4f){0.f, 0.f, 0.f, 0.f};
                }
            };
            fetch_dc(0, g, c4);
#pragma unroll 1
            for (int p0 = 0; p0 < npx; p0 += 16) {
                v4f gn[4], cn[4];
                if (p0 + 16 < npx) fetch_dc(p0 + 16, gn, cn);
                // ---- dc of 16 pixels x 64 channels -> the wave's three bf16 images ----
                {
                    const bool ok = p0 + spx < npx;
                    unsigned pk[3][8];
#pragma unroll
                    for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int j = 0; j < 4; j += 2) {
                            float d0 = g[v][j], d1 = g[v][j + 1];
                            if constexpr (APPLY) {
                                d0 = ok ? fmaf(k0[4 * v + j], d0, fmaf(-k2[4 * v + j], c4[v][j], k3[4 * v + j])) : 0.f;
                                d1 = ok ? fmaf(k0[4 * v + j + 1], d1, fmaf(-k2[4 * v + j + 1], c4[v][j + 1], k3[4 * v + j + 1])) : 0.f;
                            }
                            unsigned a[3], b[3];
                            split3(d0, a[0], a[1], a[2]);
                            split3(d1, b[0], b[1], b[2]);
#pragma unroll
                            for (int q = 0; q < 3; ++q) pk[q][2 * v + j / 2] = a[q] | (b[q] << 16);
                        }
                    // 16 channels = two 16-B chunks (2 (lane & 3), + 1) of pixel row spx, swizzled by the pixel
                    const int sw = 4 * ((spx >> 1) & 1);
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const int chunk = (2 * (lane & 3) + c) ^ sw;
                            *(v4i*)&dci[w][q][spx][8 * chunk] = (v4i){(int)pk[q][4 * c], (int)pk[q][4 * c + 1], (int)pk[q][4 * c + 2], (int)pk[q][4 * c + 3]};
                        }
                }
                // ---- fragments and products ----
                v4i af[2][3], bfr[2][3];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        // pixels p0 + 8 hh + j, j = 0..7: positions p0 + 8 hh + aoff .. + 7 of the (parity, shift) copy
                        const unsigned* src = (const unsigned*)&xp[apar[mi]][ash[mi]][q][2 * w + arow[mi]][p0 + 8 * hh + aoff[mi]];
                        af[mi][q] = (v4i){(int)src[0], (int)src[1], (int)src[2], (int)src[3]};
                    }
#pragma unroll
                for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(dcb + q * 2048 + boff[nj][0]));
                        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(dcb + q * 2048 + boff[nj][1]));
                        const v2i l2 = __builtin_bit_cast(v2i, lo), h2 = __builtin_bit_cast(v2i, hi);
                        bfr[nj][q] = (v4i){l2[0], l2[1], h2[0], h2[1]};
                    }
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                        for (int term = 0; term < 6; ++term)
                            acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[mi][PA[term]]),
                                                                                  __builtin_bit_cast(bf16x8, bfr[nj][PB[term]]),
                                                                                  acc[mi][nj], 0, 0, 0);
                if (p0 + 16 < npx) {
#pragma unroll
                    for (int v = 0; v < 4; ++v) { g[v] = gn[v]; c4[v] = cn[v]; }
                }
            }
        }
    }
    // ---- the four waves' sums -> slab [49][64] of this block (through the x patch's LDS) ----
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);           // [3 waves][64 taps][64 channels] floats = 48 KiB
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kidx = 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * hh, co = 32 * nj + r;
                if (w > 0) red[((w - 1) * 64 + kidx) * 64 + co] = acc[mi][nj][e];
            }
    __syncthreads();
    if (w == 0) {
        float* sl = slabs + (int64_t)blockIdx.x * 49 * 64;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kidx = 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * hh, co = 32 * nj + r;
                    const int kh = kidx >> 3, kw = kidx & 7;
                    if (kh < 7 && kw < 7)
                        sl[(kh * 7 + kw) * 64 + co] = acc[mi][nj][e] + red[kidx * 64 + co] + red[(64 + kidx) * 64 + co] + red[(128 + kidx) * 64 + co];
                }
    }
}
}  // namespace

// 1 = not taken (KOAF_STEM_MMA=0): the vector kernel of koaf_conv.hip; nb = the slab count both kernels use
int koaf_stem_wgrad_mma(const float* dy, const float* x, float* slabs, int nb, int N, int H, int W, const float* c,
                        const float* coef, int act16, void* stream) {
    static const bool off = [] { const char* e = getenv("KOAF_STEM_MMA"); return e && e[0] == '0'; }();
    if (off) return 1;
    const int OH = (H + 6 - 7) / 2 + 1, OW = (W + 6 - 7) / 2 + 1;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)nb);
    if (c) {
        if (act16) hipLaunchKernelGGL((stem_wgrad_mma_kernel<true, true>), grid, dim3(256), 0, st, dy, x, slabs, N, H, W, OH, OW, c, coef);
        else hipLaunchKernelGGL((stem_wgrad_mma_kernel<true, false>), grid, dim3(256), 0, st, dy, x, slabs, N, H, W, OH, OW, c, coef);
    } else
        hipLaunchKernelGGL((stem_wgrad_mma_kernel<false, false>), grid, dim3(256), 0, st, dy, x, slabs, N, H, W, OH, OW, nullptr, nullptr);
    return koaf_check_launch("koaf_stem_wgrad/mma");
}
