// koaf_conv.hip -- convolution / linear / attention entry points built on koaf_gemm, plus the two
// pieces that are not GEMM-shaped: the 1-channel 7x7 stem and the grouped-conv weight expansion.
#include <string.h>
#include "koaf_common.h"
#include <stdlib.h>

namespace {

inline void zero_gemm(KoafGemm* g) { *g = KoafGemm{}; g->alpha = 1.f; g->nb0 = g->nb1 = 1; g->splitk = 1; }

inline int conv_out(int H, int K, int s, int p) { return (H + 2 * p - K) / s + 1; }

inline int64_t rup32(int64_t v) { return (v + 31) / 32 * 32; }

// fp16 scheme with the activations at their fixed scale; B operand = the F plane image [2][R][Kp] of a weight [R][K]
// (koaf_wplanes_build) when the image is there, else the fp32 weight split in the kernel at the same scale
inline void set_fimg(KoafGemm* g, const KoafWImg* w, int R, int64_t K) {
    g->fmt = 1;
    g->A.fscale = KOAF_ACT_SCALE;
    g->B.amax = w->amax;
    if (!w->f) return;
    g->B.kind = 2; g->B.gather = 0;
    g->B.planes = w->f; g->B.ld = rup32(K); g->B.plane_stride = (int64_t)R * g->B.ld;
}

// A operand from activation plane images (koaf_act_planes): the transform is in the image
inline void set_aplanes(KoafOperand* a, const uint16_t* planes, int64_t plane_elems) {
    a->kind = 2; a->planes = planes; a->plane_stride = plane_elems; a->zeros = planes + 2 * plane_elems;
    a->ptr = nullptr; a->ptr2 = nullptr; a->tf = 0; a->sc = a->sh = a->sc2 = nullptr;
}

// split-K plan for weight gradients: M x N output, K = pixels.  ~1024 blocks, >= 512 k-rows per split.
struct WgradPlan { int bm, bn, splitk; };
inline WgradPlan wgrad_plan(int M, int N, int64_t K, int ctap, int batch) {
    WgradPlan p;
    p.bn = (N >= 128 && (ctap % 4) == 0) ? 128 : 64;      // (a B tile may span filter taps)
    p.bm = (M >= 128) ? 128 : 64;
    int64_t tiles = cdiv64(M, p.bm) * cdiv64(N, p.bn) * batch;
    int64_t sk = 1024 / tiles;
    int64_t kmax = cdiv64(K, 512);
    if (sk > kmax) sk = kmax;
    if (sk < 1) sk = 1;
    if (sk > 4096) sk = 4096;
    if (sk > 8) {
        // multiples of 8 (the kernel then keeps all tiles of a k-range on one XCD = one L2); among those between 3/4 and
        // twice the target, the count that fills whole rounds of the chip's 512 block slots best
        int64_t best = sk & ~7ll;
        double beff = 0.0;
        for (int64_t c = ((sk * 3 / 4) + 7) & ~7ll; c <= 2 * sk && c <= kmax && c <= 4096; c += 8) {
            const int64_t blocks = tiles * c;
            const double eff = (double)blocks / (double)(cdiv64(blocks, 512) * 512);
            if (eff > beff + 0.02) { beff = eff; best = c; }
        }
        sk = best;
    }
    p.splitk = (int)sk;
    return p;
}

// ---------------------------------------------------------------------------------------------
// slab reduce with batch: slabs [nb][ns][n] -> out [nb][n]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) slab_reduce_b_kernel(const float* __restrict__ slabs, int ns, int64_t n,
                                                            float* __restrict__ out) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const float* sb = slabs + (int64_t)blockIdx.y * ns * n;
    v4f a = *(const v4f*)(sb + i);
    for (int s = 1; s < ns; ++s) a += *(const v4f*)(sb + (int64_t)s * n + i);
    *(v4f*)(out + (int64_t)blockIdx.y * n + i) = a;
}

// ---------------------------------------------------------------------------------------------
// stem: 7x7 s2 p3, one input channel (the 3 repeated channels folded into the weights)
// block = 4 output rows x 16 output cols x 64 channels; thread = (channel, row)
// ---------------------------------------------------------------------------------------------
constexpr int ST_TH = 4, ST_TW = 16, ST_PH = 2 * ST_TH + 5, ST_PW = 2 * ST_TW + 5 + 3;  // patch 13 x 40

__device__ __forceinline__ void stem_load_patch(float (*patch)[ST_PW], const float* __restrict__ x, int n, int H,
                                                int W, int oy0, int ox0) {
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    for (int i = threadIdx.x; i < ST_PH * ST_PW; i += 256) {
        const int py = i / ST_PW, px = i - py * ST_PW;
        const int iy = iy0 + py, ix = ix0 + px;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[((int64_t)n * H + iy) * W + ix];
        patch[py][px] = v;
    }
}

// Y16: the output is stored as bf16 (activation storage mode); ST: per-block column sums and sums of squares of the (stored)
// output about `shift` go to part [block][2][64] -- the statistics of the BatchNorm behind the stem, without a pass over y
template <bool Y16, bool ST>
__global__ void __launch_bounds__(256) stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w1t,
                                                       float* __restrict__ y, int N, int H, int W, int OH, int OW,
                                                       float* __restrict__ part, const float* __restrict__ shift) {
    float s1 = 0.f, s2 = 0.f, kshift = 0.f;
    auto put = [&](int64_t idx, float v) {
        if constexpr (Y16) {
            const __bf16 r = (__bf16)v;
            reinterpret_cast<__bf16*>(y)[idx] = r;
            v = (float)r;
        } else y[idx] = v;
        if constexpr (ST) { const float d = v - kshift; s1 += d; s2 += d * d; }
    };
    __shared__ __attribute__((aligned(16))) float patch[ST_PH][ST_PW];
    const int co = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int tx = (OW + ST_TW - 1) / ST_TW, ty = (OH + ST_TH - 1) / ST_TH;
    // one block per (image, band of ST_TH output rows): it walks the band's column tiles with its 49 weights in registers
    // (a block per tile reloaded them -- 49 dependent loads in front of ~1.5 us of arithmetic -- 737 000 times per encoder)
    const int by = blockIdx.x % ty, n = blockIdx.x / ty;
    const int oy0 = by * ST_TH;
    float wr[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) wr[k] = w1t[k * 64 + co];
    if constexpr (ST) kshift = shift ? shift[co] : 0.f;
    const int oy = oy0 + pg;
#pragma unroll 1
    for (int bx = 0; bx < tx; ++bx) {
        const int ox0 = bx * ST_TW;
        __syncthreads();
        stem_load_patch(patch, x, n, H, W, oy0, ox0);
        __syncthreads();
#pragma unroll 1
        for (int q = 0; q < ST_TW / 2; ++q) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
                const float* pr = &patch[2 * pg + kh][4 * q];
                const v4f p0 = *(const v4f*)pr, p1 = *(const v4f*)(pr + 4);
                const float p8 = pr[8];
                const float pv[9] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3], p8};
#pragma unroll
                for (int kw = 0; kw < 7; ++kw) {
                    a0 += pv[kw] * wr[kh * 7 + kw];
                    a1 += pv[kw + 2] * wr[kh * 7 + kw];
                }
            }
            const int ox = ox0 + 2 * q;
            if (oy < OH) {
                if (ox < OW) put((((int64_t)n * OH + oy) * OW + ox) * 64 + co, a0);
                if (ox + 1 < OW) put((((int64_t)n * OH + oy) * OW + ox + 1) * 64 + co, a1);
            }
        }
    }
    if constexpr (ST) {
        __shared__ float sred[2][ST_TH][64];
        sred[0][pg][co] = s1;
        sred[1][pg][co] = s2;
        __syncthreads();
        if (pg < 2) {
            float a = 0.f;
#pragma unroll
            for (int g = 0; g < ST_TH; ++g) a += sred[pg][g][co];
            part[((int64_t)blockIdx.x * 2 + pg) * 64 + co] = a;
        }
    }
}

// APPLY: dy is formed on load from (dz = dy, c, coef [4][64]) as coef0*dz + coef3 - coef2*c -- the BatchNorm-backward apply
// (KoafBnApply, the arithmetic of the GEMM loaders' tf 2) -- so the stem's dc is never written; C16: c is stored as bf16
template <bool APPLY, bool C16>
__global__ void __launch_bounds__(256) stem_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         float* __restrict__ slabs, int N, int H, int W, int OH,
                                                         int OW, const float* __restrict__ cc, const float* __restrict__ coef) {
    __shared__ __attribute__((aligned(16))) float patch[ST_PH][ST_PW];
    __shared__ float red[3][64][49 + 1];
    const int co = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int tx = (OW + ST_TW - 1) / ST_TW, ty = (OH + ST_TH - 1) / ST_TH;
    const int ntile = N * ty * tx;
    float acc[49];
#pragma unroll
    for (int k = 0; k < 49; ++k) acc[k] = 0.f;
    float k0 = 1.f, k2 = 0.f, k3 = 0.f;
    if constexpr (APPLY) { k0 = coef[co]; k2 = coef[128 + co]; k3 = coef[192 + co]; }
    auto grad = [&](int64_t o) {
        float gv = dy[o];
        if constexpr (APPLY) {
            float cv;
            if constexpr (C16) cv = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(cc)[o] << 16);
            else cv = cc[o];
            gv = fmaf(k0, gv, fmaf(-k2, cv, k3));
        }
        return gv;
    };
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        int b = tile;
        const int bx = b % tx; b /= tx;
        const int by = b % ty;
        const int n = b / ty;
        const int oy0 = by * ST_TH, ox0 = bx * ST_TW;
        __syncthreads();
        stem_load_patch(patch, x, n, H, W, oy0, ox0);
        __syncthreads();
        const int oy = oy0 + pg;
#pragma unroll 1
        for (int q = 0; q < ST_TW / 2; ++q) {
            const int ox = ox0 + 2 * q;
            float g0 = 0.f, g1 = 0.f;
            if (oy < OH) {
                if (ox < OW) g0 = grad((((int64_t)n * OH + oy) * OW + ox) * 64 + co);
                if (ox + 1 < OW) g1 = grad((((int64_t)n * OH + oy) * OW + ox + 1) * 64 + co);
            }
#pragma unroll
            for (int kh = 0; kh < 7; ++kh) {
                const float* pr = &patch[2 * pg + kh][4 * q];
                const v4f p0 = *(const v4f*)pr, p1 = *(const v4f*)(pr + 4);
                const float p8 = pr[8];
                const float pv[9] = {p0[0], p0[1], p0[2], p0[3], p1[0], p1[1], p1[2], p1[3], p8};
#pragma unroll
                for (int kw = 0; kw < 7; ++kw) acc[kh * 7 + kw] += g0 * pv[kw] + g1 * pv[kw + 2];
            }
        }
    }
    // reduce the 4 row groups, write this block's slab [49][64]
    __syncthreads();
    if (pg > 0) {
#pragma unroll
        for (int k = 0; k < 49; ++k) red[pg - 1][co][k] = acc[k];
    }
    __syncthreads();
    if (pg == 0) {
        float* sl = slabs + (int64_t)blockIdx.x * 49 * 64;
#pragma unroll
        for (int k = 0; k < 49; ++k) sl[k * 64 + co] = acc[k] + red[0][co][k] + red[1][co][k] + red[2][co][k];
    }
}

// w [64][49][3] -> w1t [49][64] (sum over the 3 identical input channels)
__global__ void stem_fold_kernel(const float* __restrict__ w, float* __restrict__ w1t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 49) return;
    const int co = i / 49, k = i - co * 49;
    const float* s = w + (int64_t)i * 3;
    w1t[k * 64 + co] = s[0] + s[1] + s[2];
}
__global__ void stem_unfold_kernel(const float* __restrict__ dw1t, float* __restrict__ dw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 49) return;
    const int co = i / 49, k = i - co * 49;
    const float v = dw1t[k * 64 + co];
    dw[(int64_t)i * 3 + 0] = v;
    dw[(int64_t)i * 3 + 1] = v;
    dw[(int64_t)i * 3 + 2] = v;
}

// ---------------------------------------------------------------------------------------------
// grouped 3x3: weights [C][9][Cg] <-> block-diagonal 64-channel slabs [C/64][64][9][64]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gconv_expand_kernel(const float* __restrict__ w, float* __restrict__ wexp,
                                                           int C, int Cg, float* __restrict__ amax) {
    const int64_t total = (int64_t)C * 9 * 64;
    unsigned mb = 0u;          // largest magnitude seen by this thread, as bits (a NaN / Inf compares above every finite value)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ci = (int)(i & 63);
        int64_t r = i >> 6;
        const int tap = (int)(r % 9);
        const int co = (int)(r / 9);  // global output channel; slab = co/64
        const int col = co & 63;
        const int g0 = (col / Cg) * Cg;  // first slab-local input channel of co's group
        float v = 0.f;
        if (ci >= g0 && ci < g0 + Cg) v = w[((int64_t)co * 9 + tap) * Cg + (ci - g0)];
        wexp[i] = v;
        mb = max(mb, koaf_absbits(v));
    }
    if (amax != nullptr) block_amax_raise_bits(mb, amax);
}
__global__ void __launch_bounds__(256) gconv_compress_kernel(const float* __restrict__ dwexp, float* __restrict__ dw,
                                                             int C, int Cg) {
    const int64_t total = (int64_t)C * 9 * Cg;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cg = (int)(i % Cg);
        int64_t r = i / Cg;
        const int tap = (int)(r % 9);
        const int co = (int)(r / 9);
        const int g0 = ((co & 63) / Cg) * Cg;
        dw[i] = dwexp[((int64_t)co * 9 + tap) * 64 + g0 + cg];
    }
}

}  // namespace

#define STREAM ((hipStream_t)stream)

// ================================================================================================
// dense convolution
// ================================================================================================
extern "C" int koaf_conv2d_fwd(const float* x, const float* w, float* y, int32_t N, int32_t H, int32_t W, int32_t Cin,
                               int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad, const float* in_sc,
                               const float* in_sh, float* stats, int32_t* stats_rows, const float* stats_shift,
                               const KoafWImg* wimg, const uint16_t* x_planes, const KoafTail* tail, const KoafEmit* emit,
                               int32_t act16, void* stream) {
    KOAF_REQUIRE((x || x_planes) && w && y && N > 0 && Cin % 32 == 0 && Cout % 4 == 0, "koaf_conv2d_fwd: bad args (Cin=%d Cout=%d)",
                 Cin, Cout);
    KOAF_REQUIRE(!tail || (tail->idt && x && in_sc && in_sh && !x_planes && KH == 1 && KW == 1 && stride == 1 && pad == 0 && wimg &&
                           wimg->amax && wimg->f && (tail->idt_sc == nullptr) == (tail->idt_sh == nullptr)),
                 "koaf_conv2d_fwd: the fused bottleneck tail serves 1x1 / stride-1 convolutions with weight plane images, from x / in_sc / in_sh");
    KOAF_REQUIRE((in_sc == nullptr) == (in_sh == nullptr), "koaf_conv2d_fwd: in_sc/in_sh come together");
    const int OH = conv_out(H, KH, stride, pad), OW = conv_out(W, KW, stride, pad);
    const int64_t M = (int64_t)N * OH * OW;
    KOAF_REQUIRE(M < (1ll << 31), "koaf_conv2d_fwd: too many output pixels");
    KoafGemm g;
    zero_gemm(&g);
    g.A.ptr = x;
    g.A.kind = 0;
    if (KH == 1 && KW == 1 && stride == 1 && pad == 0 && !x_planes) {
        g.A.gather = 0;
        g.A.ld = Cin;
    } else {        // (plane images are always addressed as a gather, a 1x1 kernel being its one-tap case)
        g.A.gather = 1;
        g.A.H = H; g.A.W = W; g.A.C = Cin; g.A.CS = Cin;
        g.A.PH = OH; g.A.PW = OW;
        g.A.KH = KH; g.A.KW = KW; g.A.stride = stride; g.A.pad = pad; g.A.pad_w = pad;
    }
    if (in_sc) { g.A.tf = 1; g.A.sc = in_sc; g.A.sh = in_sh; }
    if (tail) { g.A.tf = 3; g.A.ptr2 = tail->idt; g.A.side = tail->y_out; g.A.sc2 = tail->idt_sc; g.A.sh2 = tail->idt_sc ? tail->idt_sh : nullptr; }
    g.B.ptr = w;
    g.B.kind = 0;
    g.B.ld = (int64_t)KH * KW * Cin;
    g.M = (int)M; g.N = Cout; g.K = KH * KW * Cin;
    if (wimg && wimg->amax) set_fimg(&g, wimg, Cout, g.K);
    if (x_planes) {
        KOAF_REQUIRE(g.A.gather == 1 && g.B.kind == 2, "koaf_conv2d_fwd: x_planes serve the gathered kernels and need wimg->f");
        set_aplanes(&g.A, x_planes, (int64_t)N * H * W * Cin);
    }
    g.C = y; g.ldc = Cout;
    g.stats = stats;
    g.stats_shift = stats ? stats_shift : nullptr;
    g.act16 = act16 ? 1 : 0;            // (x and y are bf16 activations)
    if (emit) {
        KOAF_REQUIRE(emit->planes && emit->sc && emit->sh && (Cout % 8) == 0, "koaf_conv2d_fwd: emit needs planes / sc / sh and Cout %% 8 == 0");
        g.out_planes = emit->planes; g.out_sc = emit->sc; g.out_sh = emit->sh; g.out_ps = M * Cout;
    }
    if (stats_rows) *stats_rows = koaf_gemm_part_rows(&g);
    return koaf_gemm(&g, stream);
}

// upper bound of the statistics rows koaf_conv2d_fwd writes for M output pixels (64-row tiles everywhere)
extern "C" int32_t koaf_conv2d_stats_rows(int64_t M, int32_t Cout) { return (int32_t)cdiv64(M, 64); }

// A operand = dy formed on load from (dz, c): dy = coef0 * dz + coef3 - coef2 * c (koaf_bn_bwd_finalize)
static void set_apply(KoafOperand* a, const KoafBnApply* ap, int C) {
    a->ptr = ap->dz; a->ptr2 = ap->c; a->tf = 2;
    a->sc = ap->coef; a->sc2 = ap->coef + 2 * (int64_t)C; a->sh = ap->coef + 3 * (int64_t)C;
    a->amax = ap->amax;
}

static void set_bnb(KoafGemm* g, const KoafBnb* b, float* part) {
    if (!b) return;
    g->bnb_amax = b->dz_amax;
    g->bnb_mode = b->mode;
    g->bnb_c = b->c; g->bnb_y = b->y; g->bnb_sc = b->sc; g->bnb_sh = b->sh;
    g->bnb_mean = b->mean; g->bnb_invstd = b->invstd;
    g->bnb2_c = b->c2; g->bnb2_mean = b->mean2; g->bnb2_invstd = b->invstd2;
    g->bnb_part = part;
}

extern "C" int32_t koaf_conv2d_dgrad_bnb_rows(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t stride) {
    // upper bound: 64-row tiles; stride 2 = four parity classes
    if (stride == 2) {
        int64_t r = 0;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) r += cdiv64((int64_t)N * ((H - py + 1) / 2) * ((W - px + 1) / 2), 64);
        return (int32_t)r;
    }
    return (int32_t)cdiv64((int64_t)N * H * W, 64);
}

extern "C" int koaf_conv2d_dgrad_bnb(const float* dy, const float* w, float* dx, int32_t N, int32_t H, int32_t W,
                                     int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                                     const float* residual, const KoafBnb* bnb, float* part, int32_t* part_rows,
                                     const KoafWImg* wimg, const float* dy_amax, const KoafBnApply* dy_apply,
                                     const uint16_t* dy_planes, int32_t act16, void* stream) {
    KOAF_REQUIRE(!dy_planes || (wimg && wimg->amax && wimg->d && (dy_amax || dy_apply) && KH * KW > 1),
                 "koaf_conv2d_dgrad: dy_planes need the weight's D plane image, dy_amax and a gathered (KH*KW > 1) kernel");
    KOAF_REQUIRE((dy || dy_apply || dy_planes) && w && dx && N > 0 && Cout % 32 == 0 && Cin % 4 == 0, "koaf_conv2d_dgrad: bad args");
    KOAF_REQUIRE(!dy_apply || (dy_apply->dz && dy_apply->c && dy_apply->coef && dy_apply->amax && wimg && wimg->amax && wimg->d),
                 "koaf_conv2d_dgrad: dy_apply needs dz / c / coef / amax and the weight's D plane image");
    if (dy_apply) { dy = dy_apply->dz; dy_amax = dy_apply->amax; }
    if (bnb && bnb->dz_amax && hipMemsetAsync(bnb->dz_amax, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) {
        koaf_set_error("koaf_conv2d_dgrad_bnb: memset failed");
        return KOAF_ELAUNCH;
    }
    // fp16 scheme when both operands' magnitudes are known; the weight tiles then come from the D image [2][Cin][KH*KW*Cout]
    // (rows = input channels, k = (tap, output channel): the k order of the gathered dy) if it is there
    const bool f16 = wimg && wimg->amax && dy_amax;
    const bool ps = f16 && wimg->d != nullptr;
    const uint16_t* w_dimg = ps ? wimg->d : nullptr;
    const int64_t dld = (int64_t)KH * KW * Cout, dps = (int64_t)Cin * dld;
    KOAF_REQUIRE(!bnb || (part && part_rows), "koaf_conv2d_dgrad_bnb: part / part_rows required");
    const int nsum = (bnb && bnb->c2) ? 3 : 2;
    int rows_done = 0;
    const int OH = conv_out(H, KH, stride, pad), OW = conv_out(W, KW, stride, pad);
    const int64_t M = (int64_t)N * H * W;
    KOAF_REQUIRE(M < (1ll << 31), "koaf_conv2d_dgrad: too many pixels");
    KoafGemm g;
    if (stride == 2) {
        // Parity decomposition: input pixel (y, x) only sees taps kh = (y+pad) mod 2 (+2, +4, ...), so the four
        // classes (y%2, x%2) are four stride-1 transposed gathers over their own tap subsets -- 4x less MFMA work
        // than multiplying the structural zeros of the plain gather.  Each class scatters its rows straight from
        // the epilogue (row map); a class with no tap (1x1 kernels) just writes residual / zero.
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                const int Hc = (H - py + 1) / 2, Wc = (W - px + 1) / 2;
                if (Hc <= 0 || Wc <= 0) continue;
                const int khs = (py + pad) & 1, kws = (px + pad) & 1;
                const int nkh = khs < KH ? (KH - khs + 1) / 2 : 0, nkw = kws < KW ? (KW - kws + 1) / 2 : 0;
                const int offy = (py + pad - khs) / 2, offx = (px + pad - kws) / 2;
                zero_gemm(&g);
                g.prec = 1;
                if (f16) { g.fmt = 1; g.A.amax = dy_amax; g.B.amax = wimg->amax; }
                g.A.ptr = dy; g.A.kind = 0; g.A.gather = 2;
                if (dy_apply) set_apply(&g.A, dy_apply, Cout);
                g.A.H = OH; g.A.W = OW; g.A.C = Cout; g.A.CS = Cout;
                g.A.PH = Hc; g.A.PW = Wc;
                g.A.KH = nkh > 0 ? nkh : 1; g.A.KW = nkw > 0 ? nkw : 1; g.A.stride = 1; g.A.pad = offy; g.A.pad_w = offx;
                g.B.ptr = w + ((int64_t)khs * KW + kws) * Cin; g.B.kind = 1; g.B.gather = 3;
                g.B.C = Cout; g.B.ld = (int64_t)KH * KW * Cin;
                g.B.KW = g.A.KW; g.B.tap_stride = 2ll * Cin; g.B.tap_stride_h = 2ll * KW * Cin;
                if (ps) {   // the class's tap subset of the D image: taps (khs + 2i, kws + 2j)
                    g.B.kind = 2; g.B.gather = 0;
                    g.B.planes = w_dimg + ((int64_t)khs * KW + kws) * Cout; g.B.ld = dld; g.B.plane_stride = dps;
                    g.B.tap_stride = 2ll * Cout; g.B.tap_stride_h = 2ll * KW * Cout;
                }
                if (dy_planes && nkh > 0 && nkw > 0) set_aplanes(&g.A, dy_planes, (int64_t)N * OH * OW * Cout);
                g.M = N * Hc * Wc; g.N = Cin; g.K = nkh * nkw * Cout;
                g.C = dx; g.ldc = Cin;
                g.residual = residual; g.ldr = Cin;
                g.cmap = 1; g.cm_PH = Hc; g.cm_PW = Wc; g.cm_H = H; g.cm_W = W; g.cm_py = py; g.cm_px = px;
                if (bnb) {
                    set_bnb(&g, bnb, part + (int64_t)rows_done * nsum * Cin);
                    rows_done += koaf_gemm_part_rows(&g);
                }
                g.act16 = act16 ? 2 : 0;
                int rc = koaf_gemm(&g, stream);
                if (rc != KOAF_OK) return rc;
            }
        if (part_rows) *part_rows = rows_done;
        return KOAF_OK;
    }
    zero_gemm(&g);
    g.prec = 1;
    if (f16) { g.fmt = 1; g.A.amax = dy_amax; g.B.amax = wimg->amax; }
    g.A.ptr = dy;
    g.A.kind = 0;
    if (dy_apply) set_apply(&g.A, dy_apply, Cout);
    g.B.ptr = w;
    g.B.kind = 1;
    if (KH == 1 && KW == 1 && stride == 1 && pad == 0) {
        g.A.gather = 0;
        g.A.ld = Cout;
        g.B.gather = 0;
        g.B.ld = Cin;  // element (cin, k=cout) at w + cout*Cin + cin
    } else {
        g.A.gather = 2;
        g.A.H = OH; g.A.W = OW; g.A.C = Cout; g.A.CS = Cout;
        g.A.PH = H; g.A.PW = W;
        g.A.KH = KH; g.A.KW = KW; g.A.stride = stride; g.A.pad = pad; g.A.pad_w = pad;
        g.B.gather = 3;
        g.B.C = Cout;
        g.B.ld = (int64_t)KH * KW * Cin;
        g.B.KW = KW;
        g.B.tap_stride = Cin;
        g.B.tap_stride_h = (int64_t)KW * Cin;
    }
    if (ps) {   // all taps in order: k is simply the row offset of the D image
        g.B.kind = 2; g.B.gather = 0; g.B.C = 0; g.B.tap_stride = g.B.tap_stride_h = 0;
        g.B.planes = w_dimg; g.B.ld = dld; g.B.plane_stride = dps;
    }
    if (dy_planes) set_aplanes(&g.A, dy_planes, (int64_t)N * OH * OW * Cout);
    g.M = (int)M; g.N = Cin; g.K = KH * KW * Cout;
    g.C = dx; g.ldc = Cin;
    g.residual = residual; g.ldr = Cin;
    if (bnb) {
        set_bnb(&g, bnb, part);
        *part_rows = koaf_gemm_part_rows(&g);
    }
    g.act16 = act16 ? 2 : 0;            // (dy_apply->c and the BatchNorm-backward operands c / y / c2 are bf16 activations)
    return koaf_gemm(&g, stream);
}

extern "C" int koaf_conv2d_dgrad(const float* dy, const float* w, float* dx, int32_t N, int32_t H, int32_t W,
                                 int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                                 const float* residual, const KoafWImg* wimg, const float* dy_amax,
                                 const KoafBnApply* dy_apply, const uint16_t* dy_planes, int32_t act16, void* stream) {
    return koaf_conv2d_dgrad_bnb(dy, w, dx, N, H, W, Cin, Cout, KH, KW, stride, pad, residual, nullptr, nullptr,
                                 nullptr, wimg, dy_amax, dy_apply, dy_planes, act16, stream);
}

// koaf_wgrad3.hip: 3x3 / stride 1 / pad 1 over plane images, both tensors walked once in padded raster order
bool koaf_wgrad3_ring_ok(int N, int H, int W, int Cin, int Cout);
int64_t koaf_wgrad3_ring_ws(int N, int H, int W, int Cin, int Cout);
int koaf_wgrad3_ring(const uint16_t* dy_planes, const uint16_t* x_planes, float* dw, float* slabs, const float* dy_amax,
                     float x_scale, int N, int H, int W, int Cin, int Cout, void* stream);
static inline bool wgrad3_ring_shape(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    return KH == 3 && KW == 3 && stride == 1 && pad == 1 && koaf_wgrad3_ring_ok(N, H, W, Cin, Cout);
}

extern "C" int64_t koaf_conv2d_wgrad_ws(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH,
                                        int32_t KW, int32_t stride, int32_t pad) {
    const int OH = conv_out(H, KH, stride, pad), OW = conv_out(W, KW, stride, pad);
    WgradPlan p = wgrad_plan(Cout, KH * KW * Cin, (int64_t)N * OH * OW, Cin, 1);
    int64_t ws = p.splitk > 1 ? (int64_t)(p.splitk + 16) * Cout * KH * KW * Cin : 0;   // +16: koaf_slab_reduce level-1 partials
    if (wgrad3_ring_shape(N, H, W, Cin, Cout, KH, KW, stride, pad)) {
        const int64_t w3 = koaf_wgrad3_ring_ws(N, H, W, Cin, Cout);
        if (w3 > ws) ws = w3;
    }
    return ws;
}

extern "C" int koaf_conv2d_wgrad(const float* dy, const float* x, float* dw, int32_t N, int32_t H, int32_t W,
                                 int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                                 const float* in_sc, const float* in_sh, float* slabs, const float* dy_amax,
                                 const KoafBnApply* dy_apply, const uint16_t* dy_planes, const uint16_t* x_planes,
                                 int32_t act16, void* stream) {
    KOAF_REQUIRE((dy_planes == nullptr) == (x_planes == nullptr) && (!dy_planes || dy_amax || dy_apply),
                 "koaf_conv2d_wgrad: dy_planes and x_planes come together, with dy_amax (or dy_apply)");
    KOAF_REQUIRE((dy || dy_apply || dy_planes) && (x || x_planes) && dw && N > 0 && Cin % 64 == 0 && Cout % 4 == 0, "koaf_conv2d_wgrad: bad args");
    KOAF_REQUIRE(!dy_apply || (dy_apply->dz && dy_apply->c && dy_apply->coef && dy_apply->amax),
                 "koaf_conv2d_wgrad: dy_apply needs dz / c / coef / amax");
    if (dy_apply) { dy = dy_apply->dz; dy_amax = dy_apply->amax; }
    KOAF_REQUIRE((in_sc == nullptr) == (in_sh == nullptr), "koaf_conv2d_wgrad: in_sc/in_sh come together");
    const int OH = conv_out(H, KH, stride, pad), OW = conv_out(W, KW, stride, pad);
    const int64_t P = (int64_t)N * OH * OW;
    KOAF_REQUIRE(P < (1ll << 31), "koaf_conv2d_wgrad: too many pixels");
    const int Ntot = KH * KW * Cin;
    if (dy_planes && slabs && wgrad3_ring_shape(N, H, W, Cin, Cout, KH, KW, stride, pad))
        return koaf_wgrad3_ring(dy_planes, x_planes, dw, slabs, dy_amax, KOAF_ACT_SCALE, N, H, W, Cin, Cout, stream);
    WgradPlan p = wgrad_plan(Cout, Ntot, P, Cin, 1);
    KOAF_REQUIRE(p.splitk == 1 || slabs, "koaf_conv2d_wgrad: workspace required");
    KoafGemm g;
    zero_gemm(&g);
    g.prec = 1;
    if (dy_amax) { g.fmt = 1; g.A.amax = dy_amax; g.B.fscale = KOAF_ACT_SCALE; }   // fp16 scheme: dy at its own scale, x fixed
    g.A.ptr = dy; g.A.kind = 1; g.A.ld = Cout;
    if (dy_apply) set_apply(&g.A, dy_apply, Cout);
    g.B.ptr = x; g.B.kind = 1;
    if (KH == 1 && KW == 1 && stride == 1 && pad == 0) {
        g.B.gather = 0;
        g.B.ld = Cin;
    } else {
        g.B.gather = 1;
        g.B.H = H; g.B.W = W; g.B.C = Cin; g.B.CS = Cin;
        g.B.PH = OH; g.B.PW = OW;
        g.B.KH = KH; g.B.KW = KW; g.B.stride = stride; g.B.pad = pad; g.B.pad_w = pad;
    }
    if (in_sc) { g.B.tf = 1; g.B.sc = in_sc; g.B.sh = in_sh; }
    if (dy_planes) {        // both operands from plane images, K-major (the 1x1 case as a one-tap gather)
        set_aplanes(&g.A, dy_planes, P * Cout);
        g.A.kind = 3; g.A.ld = Cout;
        set_aplanes(&g.B, x_planes, (int64_t)N * H * W * Cin);
        g.B.kind = 3; g.B.gather = 1;
        g.B.H = H; g.B.W = W; g.B.C = Cin; g.B.CS = Cin;
        g.B.PH = OH; g.B.PW = OW;
        g.B.KH = KH; g.B.KW = KW; g.B.stride = stride; g.B.pad = pad; g.B.pad_w = pad;
    }
    g.M = Cout; g.N = Ntot; g.K = (int)P;
    g.bm = p.bm; g.bn = p.bn; g.splitk = p.splitk;
    g.C = p.splitk > 1 ? slabs : dw;
    g.ldc = Ntot;
    g.act16 = (act16 && !dy_planes) ? 3 : 0;      // (x and dy_apply->c are bf16 activations; plane images carry no storage type)
    int rc = koaf_gemm(&g, stream);
    if (rc != KOAF_OK || p.splitk == 1) return rc;
    return koaf_slab_reduce(slabs, p.splitk, (int64_t)Cout * Ntot, dw, stream);
}

// ================================================================================================
// grouped 3x3 (ResNeXt) as 64-channel block-diagonal slabs through the same GEMM
// ================================================================================================
extern "C" int koaf_gconv_expand_w(const float* w, float* wexp, int32_t C, int32_t groups, float* amax, void* stream) {
    KOAF_REQUIRE(w && wexp && C % 64 == 0 && groups > 0 && C % groups == 0 && 64 % (C / groups) == 0,
                 "koaf_gconv_expand_w: C=%d groups=%d unsupported", C, groups);
    hipLaunchKernelGGL(gconv_expand_kernel, dim3(1024), dim3(256), 0, STREAM, w, wexp, C, C / groups, amax);
    return koaf_check_launch("koaf_gconv_expand_w");
}
extern "C" int koaf_gconv_compress_dw(const float* dwexp, float* dw, int32_t C, int32_t groups, void* stream) {
    KOAF_REQUIRE(dwexp && dw && C % 64 == 0 && groups > 0 && C % groups == 0 && 64 % (C / groups) == 0,
                 "koaf_gconv_compress_dw: C=%d groups=%d unsupported", C, groups);
    hipLaunchKernelGGL(gconv_compress_kernel, dim3(512), dim3(256), 0, STREAM, dwexp, dw, C, C / groups);
    return koaf_check_launch("koaf_gconv_compress_dw");
}

extern "C" int koaf_gconv3x3_fwd(const float* x, const float* wexp, float* y, int32_t N, int32_t H, int32_t W,
                                 int32_t C, int32_t stride, const float* in_sc, const float* in_sh, float* stats,
                                 int32_t* stats_rows, const float* stats_shift, const float* w_amax, int32_t act16, void* stream) {
    KOAF_REQUIRE(x && wexp && y && N > 0 && C % 64 == 0, "koaf_gconv3x3_fwd: bad args");
    const int OH = conv_out(H, 3, stride, 1), OW = conv_out(W, 3, stride, 1);
    const int64_t M = (int64_t)N * OH * OW;
    KOAF_REQUIRE(M < (1ll << 31), "koaf_gconv3x3_fwd: too many pixels");
    KoafGemm g;
    zero_gemm(&g);
    g.nb1 = C / 64;
    g.A.ptr = x; g.A.kind = 0; g.A.gather = 1; g.A.bs1 = 64;
    g.A.H = H; g.A.W = W; g.A.C = 64; g.A.CS = C; g.A.PH = OH; g.A.PW = OW;
    g.A.KH = 3; g.A.KW = 3; g.A.stride = stride; g.A.pad = 1; g.A.pad_w = 1;
    if (in_sc) { g.A.tf = 1; g.A.sc = in_sc; g.A.sh = in_sh; }
    g.B.ptr = wexp; g.B.kind = 0; g.B.ld = 576; g.B.bs1 = 64 * 576;
    g.M = (int)M; g.N = 64; g.K = 576;
    g.C = y; g.ldc = C; g.cbs1 = 64;
    g.stats = stats; g.stats_ld = C; g.stats_bs = 64;
    g.stats_shift = stats ? stats_shift : nullptr;
    g.bn = 64; g.bm = 128;
    if (stats_rows) *stats_rows = (int)cdiv64(M, g.bm);
    g.A.tf_bs = 64;
    g.act16 = act16 ? 1 : 0;
    if (w_amax && !act16) { g.fmt = 1; g.A.fscale = KOAF_ACT_SCALE; g.B.amax = w_amax; }      // (fp16 scheme: both magnitudes known)
    return koaf_gemm(&g, stream);
}

extern "C" int koaf_gconv3x3_dgrad(const float* dy, const float* wexp, float* dx, int32_t N, int32_t H, int32_t W,
                                   int32_t C, int32_t stride, const float* w_amax, const float* dy_amax, void* stream) {
    KOAF_REQUIRE(dy && wexp && dx && N > 0 && C % 64 == 0, "koaf_gconv3x3_dgrad: bad args");
    const int OH = conv_out(H, 3, stride, 1), OW = conv_out(W, 3, stride, 1);
    const int64_t M = (int64_t)N * H * W;
    KOAF_REQUIRE(M < (1ll << 31), "koaf_gconv3x3_dgrad: too many pixels");
    KoafGemm g;
    zero_gemm(&g);
    g.prec = 1;
    g.nb1 = C / 64;
    g.A.ptr = dy; g.A.kind = 0; g.A.gather = 2; g.A.bs1 = 64;
    g.A.H = OH; g.A.W = OW; g.A.C = 64; g.A.CS = C; g.A.PH = H; g.A.PW = W;
    g.A.KH = 3; g.A.KW = 3; g.A.stride = stride; g.A.pad = 1; g.A.pad_w = 1;
    g.B.ptr = wexp; g.B.kind = 1; g.B.gather = 3; g.B.C = 64; g.B.ld = 576; g.B.tap_stride = 64;
    g.B.tap_stride_h = 3 * 64; g.B.KW = 3;
    g.B.bs1 = 64 * 576;
    g.M = (int)M; g.N = 64; g.K = 576;
    g.C = dx; g.ldc = C; g.cbs1 = 64;
    g.bn = 64; g.bm = 128;
    if (w_amax && dy_amax) { g.fmt = 1; g.A.amax = dy_amax; g.B.amax = w_amax; }
    return koaf_gemm(&g, stream);
}

extern "C" int64_t koaf_gconv3x3_wgrad_ws(int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride) {
    const int OH = conv_out(H, 3, stride, 1), OW = conv_out(W, 3, stride, 1);
    WgradPlan p = wgrad_plan(64, 576, (int64_t)N * OH * OW, 64, C / 64);
    return (int64_t)p.splitk * C * 576;
}

// dwexp [C/64][64][9][64]
extern "C" int koaf_gconv3x3_wgrad(const float* dy, const float* x, float* dwexp, int32_t N, int32_t H, int32_t W,
                                   int32_t C, int32_t stride, const float* in_sc, const float* in_sh, float* slabs,
                                   const float* dy_amax, int32_t act16, void* stream) {
    KOAF_REQUIRE(dy && x && dwexp && slabs && N > 0 && C % 64 == 0, "koaf_gconv3x3_wgrad: bad args");
    const int OH = conv_out(H, 3, stride, 1), OW = conv_out(W, 3, stride, 1);
    const int64_t P = (int64_t)N * OH * OW;
    KOAF_REQUIRE(P < (1ll << 31), "koaf_gconv3x3_wgrad: too many pixels");
    const int nz = C / 64;
    WgradPlan p = wgrad_plan(64, 576, P, 64, nz);
    {
        // one launch: the C/64 slabs are the batch dimension (split-K slabs laid out [slab][split][64][576])
        KoafGemm g;
        zero_gemm(&g);
        g.prec = 1;
        g.A.ptr = dy; g.A.kind = 1; g.A.ld = C; g.A.bs1 = 64;
        g.B.ptr = x; g.B.kind = 1; g.B.gather = 1; g.B.bs1 = 64;
        g.B.H = H; g.B.W = W; g.B.C = 64; g.B.CS = C; g.B.PH = OH; g.B.PW = OW;
        g.B.KH = 3; g.B.KW = 3; g.B.stride = stride; g.B.pad = 1; g.B.pad_w = 1;
        if (in_sc) { g.B.tf = 1; g.B.sc = in_sc; g.B.sh = in_sh; g.B.tf_bs = 64; }
        g.M = 64; g.N = 576; g.K = (int)P;
        g.nb0 = 1; g.nb1 = nz;
        g.bm = 64; g.bn = 64; g.splitk = p.splitk;
        g.C = slabs;
        g.ldc = 576; g.cbs1 = 64 * 576;      // (unsplit case; split-K slabs are addressed by the kernel)
        g.act16 = act16 ? 3 : 0;
        if (dy_amax && !act16) { g.fmt = 1; g.A.amax = dy_amax; g.B.fscale = KOAF_ACT_SCALE; }
        int rc = koaf_gemm(&g, stream);
        if (rc != KOAF_OK) return rc;
    }
    const int64_t n = 64 * 576;
    hipLaunchKernelGGL(slab_reduce_b_kernel, dim3((unsigned)cdiv64(n / 4, 256), nz), dim3(256), 0, STREAM, slabs,
                       p.splitk, n, dwexp);
    return koaf_check_launch("koaf_gconv3x3_wgrad");
}

// ================================================================================================
// stem
// ================================================================================================
// koaf_stem.hip: the same convolution on the matrix pipe (three bf16 pieces per operand); 1 = not taken
int koaf_stem_fwd_mma(const float* x, const float* w1t, float* y, int N, int H, int W, float* stats, const float* stats_shift,
                      int act16, void* stream);
int koaf_stem_wgrad_mma(const float* dy, const float* x, float* slabs, int nb, int N, int H, int W, const float* c,
                        const float* coef, int act16, void* stream);
extern "C" int32_t koaf_stem_stats_rows(int32_t N, int32_t H) { return (int32_t)((int64_t)N * cdiv64(conv_out(H, 7, 2, 3), ST_TH)); }
extern "C" int koaf_stem_fwd(const float* x, const float* w1t, float* y, int32_t N, int32_t H, int32_t W,
                             float* stats, const float* stats_shift, int32_t act16, void* stream) {
    KOAF_REQUIRE(x && w1t && y && N > 0 && H > 0 && W > 0, "koaf_stem_fwd: bad args");
    const int OH = conv_out(H, 7, 2, 3), OW = conv_out(W, 7, 2, 3);
    const int64_t blocks = (int64_t)N * cdiv64(OH, ST_TH);          // one per (image, row band)
    KOAF_REQUIRE(blocks < (1ll << 31), "koaf_stem_fwd: grid too large");
    {
        const int rc = koaf_stem_fwd_mma(x, w1t, y, N, H, W, stats, stats_shift, act16, stream);
        if (rc <= 0) return rc;          // taken (or failed); 1: the vector kernel below (KOAF_STEM_MMA=0)
    }
    const dim3 grid((unsigned)blocks);
    if (stats) {
        if (act16) hipLaunchKernelGGL((stem_fwd_kernel<true, true>), grid, dim3(256), 0, STREAM, x, w1t, y, N, H, W, OH, OW, stats, stats_shift);
        else hipLaunchKernelGGL((stem_fwd_kernel<false, true>), grid, dim3(256), 0, STREAM, x, w1t, y, N, H, W, OH, OW, stats, stats_shift);
    } else {
        if (act16) hipLaunchKernelGGL((stem_fwd_kernel<true, false>), grid, dim3(256), 0, STREAM, x, w1t, y, N, H, W, OH, OW, nullptr, nullptr);
        else hipLaunchKernelGGL((stem_fwd_kernel<false, false>), grid, dim3(256), 0, STREAM, x, w1t, y, N, H, W, OH, OW, nullptr, nullptr);
    }
    return koaf_check_launch("koaf_stem_fwd");
}
static inline int stem_wgrad_blocks(int N, int H, int W) {
    const int OH = conv_out(H, 7, 2, 3), OW = conv_out(W, 7, 2, 3);
    int64_t tiles = (int64_t)N * cdiv64(OH, ST_TH) * cdiv64(OW, ST_TW);
    return (int)(tiles < 1024 ? tiles : 1024);
}
extern "C" int64_t koaf_stem_wgrad_ws(int32_t N, int32_t H, int32_t W) {
    return (int64_t)(stem_wgrad_blocks(N, H, W) + 16) * 49 * 64;
}
extern "C" int koaf_stem_wgrad(const float* dy, const float* x, float* dw1t, int32_t N, int32_t H, int32_t W,
                               float* slabs, const KoafBnApply* dy_apply, int32_t act16, void* stream) {
    KOAF_REQUIRE((dy || dy_apply) && x && dw1t && slabs && N > 0, "koaf_stem_wgrad: bad args");
    KOAF_REQUIRE(!dy_apply || (dy_apply->dz && dy_apply->c && dy_apply->coef), "koaf_stem_wgrad: dy_apply needs dz / c / coef");
    const int OH = conv_out(H, 7, 2, 3), OW = conv_out(W, 7, 2, 3);
    const int nb = stem_wgrad_blocks(N, H, W);
    {
        const int rc = koaf_stem_wgrad_mma(dy_apply ? dy_apply->dz : dy, x, slabs, nb, N, H, W, dy_apply ? dy_apply->c : nullptr,
                                           dy_apply ? dy_apply->coef : nullptr, act16, stream);
        if (rc < 0) return rc;
        if (rc == 0) return koaf_slab_reduce(slabs, nb, 49 * 64, dw1t, stream);       // (1: the vector kernel below, KOAF_STEM_MMA=0)
    }
    if (dy_apply) {
        if (act16) hipLaunchKernelGGL((stem_wgrad_kernel<true, true>), dim3(nb), dim3(256), 0, STREAM, dy_apply->dz, x, slabs, N, H, W, OH, OW,
                                      dy_apply->c, dy_apply->coef);
        else hipLaunchKernelGGL((stem_wgrad_kernel<true, false>), dim3(nb), dim3(256), 0, STREAM, dy_apply->dz, x, slabs, N, H, W, OH, OW,
                                dy_apply->c, dy_apply->coef);
    } else
        hipLaunchKernelGGL((stem_wgrad_kernel<false, false>), dim3(nb), dim3(256), 0, STREAM, dy, x, slabs, N, H, W, OH, OW, nullptr, nullptr);
    int rc = koaf_check_launch("koaf_stem_wgrad");
    if (rc != KOAF_OK) return rc;
    return koaf_slab_reduce(slabs, nb, 49 * 64, dw1t, stream);
}
extern "C" int koaf_stem_fold_w(const float* w, float* w1t, void* stream) {
    KOAF_REQUIRE(w && w1t, "koaf_stem_fold_w: bad args");
    hipLaunchKernelGGL(stem_fold_kernel, dim3((64 * 49 + 255) / 256), dim3(256), 0, STREAM, w, w1t);
    return koaf_check_launch("koaf_stem_fold_w");
}
extern "C" int koaf_stem_unfold_dw(const float* dw1t, float* dw, void* stream) {
    KOAF_REQUIRE(dw1t && dw, "koaf_stem_unfold_dw: bad args");
    hipLaunchKernelGGL(stem_unfold_kernel, dim3((64 * 49 + 255) / 256), dim3(256), 0, STREAM, dw1t, dw);
    return koaf_check_launch("koaf_stem_unfold_dw");
}

// ================================================================================================
// nn.Linear
// ================================================================================================
// split-K plan for a linear layer with few rows: the 64x64-tile grid of M x N is only a few hundred blocks with
// K/32 serial k-steps each (latency-bound at ~1 block per CU); splitting K 2-8 ways fills the chip.
struct LinTile { int bm, bn; };
static LinTile linear_tile(int M, int N) {
    // 128x128 tiles (half the loader / split work per FLOP of 64x64) once both dimensions offer a few of them; with
    // few rows (200-256 token rows of the per-MRI aggregators) 64x128: the wide N still halves the A traffic per FLOP
    static const int forced = [] { const char* e = getenv("KOAF_LIN_TILE"); return e ? atoi(e) : 0; }();
    if (forced == 64 || forced == 128) return {forced, forced};
    if (M >= 512 && N >= 512) return {128, 128};
    if (forced == 1 || N < 1024) return {64, 64};
    return {64, 128};
}
static int linear_splitk(int M, int N, int K) {
    if ((N & 3) || K < 512) return 1;
    const LinTile t = linear_tile(M, N);
    const int64_t tiles = cdiv64(M, t.bm) * cdiv64(N, t.bn);
    const int64_t want = t.bm == 128 ? 512 : (t.bn == 128 ? 768 : 1024);
    if (tiles >= want) return 1;
    int sk = (int)(want / tiles);
    if (sk > 8) sk = 8;
    while (sk > 1 && K / sk < 256) --sk;
    return sk;
}
extern "C" int64_t koaf_linear_ws(int32_t M, int32_t N, int32_t K) {
    const int sk = linear_splitk(M, N, K);
    return sk > 1 ? (int64_t)sk * M * N : 0;
}

// ---- narrow heads (N <= 8 outputs: the 2-class heads) ------------------------------------------------------------
// A 64x64-tile GEMM spends 64 serial k-steps on a handful of useful outputs (81 us per call on the native step);
// these three direct kernels do the same sums on the vector ALUs in a few microseconds, in plain fp32.
__global__ void __launch_bounds__(256) head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, const float* __restrict__ res,
                                                       float* __restrict__ y, int M, int N, int K) {
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;      // one wave per output
    if (o >= M * N) return;
    const int m = o / N, n = o - m * N;
    float a = 0.f;
    for (int k = lane; k < K; k += 64) a += x[(int64_t)m * K + k] * w[(int64_t)n * K + k];
    a = wave_sum(a);
    if (lane == 0) y[o] = a + (b ? b[n] : 0.f) + (res ? res[o] : 0.f);
}
__global__ void __launch_bounds__(256) head_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                         const float* __restrict__ res, float* __restrict__ dx, int M,
                                                         int N, int K) {
    const int64_t total = (int64_t)M * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / K), k = (int)(i - (int64_t)m * K);
        float a = res ? res[i] : 0.f;
        for (int n = 0; n < N; ++n) a += dy[m * N + n] * w[(int64_t)n * K + k];
        dx[i] = a;
    }
}
__global__ void __launch_bounds__(256) head_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                         float* __restrict__ dw, int M, int N, int K) {
    const int64_t total = (int64_t)N * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i - (int64_t)n * K);
        float a = 0.f;
        for (int m = 0; m < M; ++m) a += dy[m * N + n] * x[(int64_t)m * K + k];
        dw[i] = a;
    }
}
static inline bool narrow_head(int M, int N, int K) { return N <= 8 && (int64_t)M * N <= 4096 && K >= 64; }

extern "C" int koaf_linear_fwd(const float* x, const float* w, const float* b, const float* residual, float* y,
                               float* ws, int32_t M, int32_t N, int32_t K, void* stream) {
    KOAF_REQUIRE(x && w && y && M > 0 && N > 0 && K > 0, "koaf_linear_fwd: bad args");
    if (narrow_head(M, N, K)) {
        hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)cdiv64((int64_t)M * N, 4)), dim3(256), 0, STREAM, x, w, b, residual,
                           y, M, N, K);
        return koaf_check_launch("koaf_linear_fwd");
    }
    KoafGemm g;
    zero_gemm(&g);
    g.A.ptr = x; g.A.kind = 0; g.A.ld = K;
    g.B.ptr = w; g.B.kind = 0; g.B.ld = K;
    g.M = M; g.N = N; g.K = K;
    const int sk = ws ? linear_splitk(M, N, K) : 1;
    if (sk > 1) {
        g.splitk = sk; g.bm = linear_tile(M, N).bm; g.bn = linear_tile(M, N).bn;
        g.C = ws; g.ldc = N;
        int rc = koaf_gemm(&g, stream);
        if (rc != KOAF_OK) return rc;
        return koaf_slab_reduce_epilogue(ws, sk, M, N, b, residual, N, y, N, stream);
    }
    g.C = y; g.ldc = N;
    g.bias = b;
    g.residual = residual; g.ldr = N;
    return koaf_gemm(&g, stream);
}
extern "C" int koaf_linear_dgrad(const float* dy, const float* w, const float* residual, float* dx, float* ws,
                                 int32_t M, int32_t N, int32_t K, void* stream) {
    KOAF_REQUIRE(dy && w && dx && M > 0 && N > 0 && K > 0, "koaf_linear_dgrad: bad args");
    if (narrow_head(M, N, K)) {
        hipLaunchKernelGGL(head_dgrad_kernel, dim3((unsigned)cdiv64((int64_t)M * K, 256)), dim3(256), 0, STREAM, dy, w,
                           residual, dx, M, N, K);
        return koaf_check_launch("koaf_linear_dgrad");
    }
    KoafGemm g;
    zero_gemm(&g);
    g.prec = 1;
    g.A.ptr = dy; g.A.kind = 0; g.A.ld = N;
    g.B.ptr = w; g.B.kind = 1; g.B.ld = K;  // element (r = k_in, kk = n_out) at w + n_out*K + k_in
    g.M = M; g.N = K; g.K = N;
    const int sk = ws ? linear_splitk(M, K, N) : 1;
    if (sk > 1) {
        g.splitk = sk; g.bm = linear_tile(M, K).bm; g.bn = linear_tile(M, K).bn;
        g.C = ws; g.ldc = K;
        int rc = koaf_gemm(&g, stream);
        if (rc != KOAF_OK) return rc;
        return koaf_slab_reduce_epilogue(ws, sk, M, K, nullptr, residual, K, dx, K, stream);
    }
    g.C = dx; g.ldc = K;
    g.residual = residual; g.ldr = K;
    return koaf_gemm(&g, stream);
}
extern "C" int koaf_linear_wgrad(const float* dy, const float* x, float* dw, float* db, float* ws, int32_t M, int32_t N,
                                 int32_t K, void* stream) {
    KOAF_REQUIRE(dy && x && dw && M > 0 && N > 0 && K > 0, "koaf_linear_wgrad: bad args");
    if (narrow_head(M, N, K)) {
        hipLaunchKernelGGL(head_wgrad_kernel, dim3((unsigned)cdiv64((int64_t)N * K, 256)), dim3(256), 0, STREAM, dy, x, dw, M,
                           N, K);
        int rc = koaf_check_launch("koaf_linear_wgrad");
        if (rc != KOAF_OK || !db) return rc;
        return koaf_colsum(dy, db, M, N, ws, stream);
    }
    KoafGemm g;
    zero_gemm(&g);
    g.prec = 1;
    g.A.ptr = dy; g.A.kind = 1; g.A.ld = N;
    g.B.ptr = x; g.B.kind = 1; g.B.ld = K;
    g.M = N; g.N = K; g.K = M;
    g.C = dw; g.ldc = K;
    int rc = koaf_gemm(&g, stream);
    if (rc != KOAF_OK || !db) return rc;
    return koaf_colsum(dy, db, M, N, ws, stream);
}

// ================================================================================================
// attention core: S = scale*Q K^T -> softmax -> P V, and its backward (all on koaf_gemm, batched
// over (b, head) with strided operands straight out of the fused qkv buffer)
// ================================================================================================
int koaf_attention_fwd_fused(const float* qkv, float* attn, float* out, int32_t B, int32_t n, int32_t h, int32_t d, float scale,
                             void* stream);       // koaf_gemm.hip: ONE launch for n <= 512

extern "C" int koaf_attention_fwd(const float* qkv, float* attn, float* out, int32_t B, int32_t n, int32_t h,
                                  int32_t d, float scale, void* stream) {
    KOAF_REQUIRE(qkv && attn && out && B > 0 && n > 0 && h > 0 && d > 0, "koaf_attention_fwd: bad args");
    {
        static const bool off = [] { const char* e = getenv("KOAF_ATTN_FUSED"); return e && e[0] == '0'; }();    // (A/B switch)
        const int rc = off ? 1 : koaf_attention_fwd_fused(qkv, attn, out, B, n, h, d, scale, stream);
        if (rc <= 0) return rc;               // taken (or failed); 1 = shape outside the fused kernel: three launches below
    }
    const int64_t ld = 3ll * h * d;
    KoafGemm g;
    zero_gemm(&g);
    g.nb0 = B; g.nb1 = h;
    g.A.ptr = qkv; g.A.kind = 0; g.A.ld = ld; g.A.bs0 = n * ld; g.A.bs1 = d;
    g.B.ptr = qkv + (int64_t)h * d; g.B.kind = 0; g.B.ld = ld; g.B.bs0 = n * ld; g.B.bs1 = d;
    g.M = n; g.N = n; g.K = d;
    g.C = attn; g.ldc = n; g.cbs0 = (int64_t)h * n * n; g.cbs1 = (int64_t)n * n;
    g.alpha = scale;
    g.bm = 64; g.bn = 64;
    int rc = koaf_gemm(&g, stream);
    if (rc != KOAF_OK) return rc;
    rc = koaf_softmax_rows(attn, (int64_t)B * h * n, n, stream);
    if (rc != KOAF_OK) return rc;
    zero_gemm(&g);
    g.nb0 = B; g.nb1 = h;
    g.A.ptr = attn; g.A.kind = 0; g.A.ld = n; g.A.bs0 = (int64_t)h * n * n; g.A.bs1 = (int64_t)n * n;
    g.B.ptr = qkv + 2ll * h * d; g.B.kind = 1; g.B.ld = ld; g.B.bs0 = n * ld; g.B.bs1 = d;
    g.M = n; g.N = d; g.K = n;
    g.C = out; g.ldc = (int64_t)h * d; g.cbs0 = (int64_t)n * h * d; g.cbs1 = d;
    g.bm = 64; g.bn = 64;
    return koaf_gemm(&g, stream);
}

extern "C" int koaf_attention_bwd(const float* dout, const float* qkv, const float* attn, float* dqkv, float* ws,
                                  int32_t B, int32_t n, int32_t h, int32_t d, float scale, void* stream) {
    KOAF_REQUIRE(dout && qkv && attn && dqkv && ws && B > 0 && n > 0 && h > 0 && d > 0, "koaf_attention_bwd: bad args");
    const int64_t ld = 3ll * h * d, hd = (int64_t)h * d;
    const int64_t pb0 = (int64_t)h * n * n, pb1 = (int64_t)n * n;
    KoafGemm g;
    int rc;
    // dV[j,dd] = sum_i P[i,j] dO[i,dd]
    zero_gemm(&g);
    g.prec = 1;
    g.nb0 = B; g.nb1 = h; g.bm = 64; g.bn = 64;
    g.A.ptr = attn; g.A.kind = 1; g.A.ld = n; g.A.bs0 = pb0; g.A.bs1 = pb1;
    g.B.ptr = dout; g.B.kind = 1; g.B.ld = hd; g.B.bs0 = n * hd; g.B.bs1 = d;
    g.M = n; g.N = d; g.K = n;
    g.C = dqkv + 2 * hd; g.ldc = ld; g.cbs0 = n * ld; g.cbs1 = d;
    if ((rc = koaf_gemm(&g, stream)) != KOAF_OK) return rc;
    // dP[i,j] = sum_dd dO[i,dd] V[j,dd]
    zero_gemm(&g);
    g.prec = 1;
    g.nb0 = B; g.nb1 = h; g.bm = 64; g.bn = 64;
    g.A.ptr = dout; g.A.kind = 0; g.A.ld = hd; g.A.bs0 = n * hd; g.A.bs1 = d;
    g.B.ptr = qkv + 2 * hd; g.B.kind = 0; g.B.ld = ld; g.B.bs0 = n * ld; g.B.bs1 = d;
    g.M = n; g.N = n; g.K = d;
    g.C = ws; g.ldc = n; g.cbs0 = pb0; g.cbs1 = pb1;
    if ((rc = koaf_gemm(&g, stream)) != KOAF_OK) return rc;
    // dS = P * (dP - rowsum(dP*P)) * scale
    if ((rc = koaf_softmax_bwd_rows(ws, attn, (int64_t)B * h * n, n, scale, stream)) != KOAF_OK) return rc;
    // dQ[i,dd] = sum_j dS[i,j] K[j,dd]
    zero_gemm(&g);
    g.prec = 1;
    g.nb0 = B; g.nb1 = h; g.bm = 64; g.bn = 64;
    g.A.ptr = ws; g.A.kind = 0; g.A.ld = n; g.A.bs0 = pb0; g.A.bs1 = pb1;
    g.B.ptr = qkv + hd; g.B.kind = 1; g.B.ld = ld; g.B.bs0 = n * ld; g.B.bs1 = d;
    g.M = n; g.N = d; g.K = n;
    g.C = dqkv; g.ldc = ld; g.cbs0 = n * ld; g.cbs1 = d;
    if ((rc = koaf_gemm(&g, stream)) != KOAF_OK) return rc;
    // dK[j,dd] = sum_i dS[i,j] Q[i,dd]
    zero_gemm(&g);
    g.prec = 1;
    g.nb0 = B; g.nb1 = h; g.bm = 64; g.bn = 64;
    g.A.ptr = ws; g.A.kind = 1; g.A.ld = n; g.A.bs0 = pb0; g.A.bs1 = pb1;
    g.B.ptr = qkv; g.B.kind = 1; g.B.ld = ld; g.B.bs0 = n * ld; g.B.bs1 = d;
    g.M = n; g.N = d; g.K = n;
    g.C = dqkv + hd; g.ldc = ld; g.cbs0 = n * ld; g.cbs1 = d;
    return koaf_gemm(&g, stream);
}
