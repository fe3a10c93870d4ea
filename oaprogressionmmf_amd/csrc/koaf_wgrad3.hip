// Weight gradient of the 3x3 / stride-1 / pad-1 convolution over activation plane images (reference: the autograd of
// F.conv2d in koafusion/models/_torchvision.py:103-136's conv2), k = pixel:
//     dW[co][kh][kw][ci] = sum_p dy[p][co] * x[p + (kh-1, kw-1)][ci]
// The generic K-major GEMM (koaf_gemm.hip, M_PK x M_PKG) fetches every x pixel nine times (once per tap) and every dy pixel
// once per column tile through L2 -> LDS, and that path, not the matrix pipe, set its time (64 -> 64 at 96 x 96: 42 GB of LDS-DMA
// per call, 6.6 ms).  Here both tensors are walked ONCE per (co tile, ci tile) in PADDED raster order -- every image as
// (H + 2) x (W + 2) positions whose border holds zeros -- so that tap (kh, kw) of position P is simply position
// P + (kh - 1)(W + 2) + (kw - 1): no per-tap gather, no validity masks (dy is zero on the border, x is zero on the border).
// x lives in an LDS ring of 32-position chunks, dy in three 32-position stages; all nine taps of a 64 x 64 (co x ci) tile are
// accumulated by the twelve waves of the block (wave = (kh, co half, ci half), three kw accumulators each).
// Products as everywhere on the fp16 scheme (KoafGemm.fmt 1): operands are two fp16 pieces of value x scale, three MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include "koaf.h"
#include "koaf_common.h"

namespace {
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

struct Wg3Params {
    const unsigned short* xpl;   // x plane images [2][npix][Cin] (+ 16 B of zeros behind them)
    const unsigned short* dypl;  // dy plane images [2][npix][Cout]
    const unsigned short* xzero;
    const unsigned short* dyzero;
    int64_t xps, dyps;           // plane strides (elements)
    float* slabs;                // [nk][Cout][9][Cin] partial sums
    const float* dy_amax;        // scale of the dy images = scale_of_amax(*dy_amax)
    float x_scale;               // scale of the x images (fixed)
    uint32_t* status;
    int N, H, W, Cin, Cout;
    int nk, nchunk;              // k-ranges; 32-position chunks in all
};

__device__ __forceinline__ float wg3_scale_of_amax(float amax) {      // = koaf_gemm.hip scale_of_amax
    if (!(amax > 0.f)) return 1.f;
    const int e = min(max(__builtin_amdgcn_frexp_expf(amax), -100), 100);
    return __builtin_ldexpf(1.f, 15 - e);
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// one LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to the 1 KiB at LDS byte address lds_addr (see koaf_gemm.hip)
__device__ __forceinline__ void wg3_dma16(const void* gsrc, unsigned lds_addr) {
    const int la = __builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(la) : "memory", "m0");
}
#pragma clang diagnostic pop

// LDS images: [position][64 channels] fp16, 128 B per position, 16-B chunks XOR-swizzled by the position
// (chunk ^ 4 ((pos >> 1) & 1)): the transposing fragment reads (ds_read_b64_tr_b16) of four consecutive positions then
// hit distinct banks wherever they start (koaf_gemm.hip PlaneKLoader / frag_load_kmd, 128-B rows).
template <int RC>
__global__ void __launch_bounds__(768) wgrad3x3_ring_kernel(Wg3Params p) {
    constexpr int XPLANE = RC * 4096;            // bytes of one x ring plane (RC chunks x 32 positions x 128 B)
    constexpr int DYPLANE = 4096, DYSTAGE = 2 * DYPLANE;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * XPLANE + 3 * DYSTAGE];
    typedef __attribute__((address_space(3))) unsigned char* lds_b;
    typedef __attribute__((address_space(3))) v4s* lds_v4s;
    const lds_b sbase = (lds_b)smem;
    const unsigned sx0 = (unsigned)(uintptr_t)sbase, sdy0 = sx0 + 2 * XPLANE;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int Wp = p.W + 2, Hp = p.H + 2, Pp = Hp * Wp;
    const int D = (Wp + 1 + 31) >> 5;            // chunks on either side of the current one that a tap can reach
    // block -> (k-range z, co tile, ci tile); the tiles of one k-range are 8 block ids apart: one XCD, one L2
    const int ncomb = (p.Cout >> 6) * (p.Cin >> 6);
    const int b = blockIdx.x;
    const int grp = b / (8 * ncomb), rem = b - grp * 8 * ncomb;
    const int comb = rem >> 3, z = grp * 8 + (rem & 7);
    if (z >= p.nk) return;
    const int cot = comb / (p.Cin >> 6), cit = comb - cot * (p.Cin >> 6);
    const int per = (p.nchunk + p.nk - 1) / p.nk;
    const int c0 = z * per, c1 = min(p.nchunk, c0 + per);

    float alpha = 1.f / (p.x_scale * wg3_scale_of_amax(*p.dy_amax));
    if (!koaf_bits_finite(koaf_absbits(*p.dy_amax))) {
        alpha = __uint_as_float(0x7fc00000u);        // (a diverged dy: the whole gradient is NaN, as koaf_gemm does)
        if (b == 0 && t == 0) koaf_status_add(p.status, 1, 1u);
    }

    // ---- loader role (waves 0..3: x piece w of every chunk; 4..7: dy piece w - 4): both planes of 8 positions x 128 B ----
    const bool loader = w < 8, isx = w < 4;
    const int pj = w & 3, kl = lane >> 3, phys = lane & 7;
    const int lch = 8 * (phys ^ (4 * ((kl >> 1) & 1)));          // logical channel offset this lane fetches (swizzle by position)
    int ln = 0, lyy = 0, lxx = 0, lm = 0;                        // running padded position of this lane's next load: chunk lm
    auto seek = [&](int m) {
        lm = m;
        int P = 32 * m + 8 * pj + kl;
        ln = 0;
        if (P < 0) { P += Pp; ln = -1; }                         // (|P| < Pp: D * 32 < Pp is checked on the host)
        const int n = P / Pp;
        ln += n;
        P -= n * Pp;
        lyy = P / Wp;
        lxx = P - lyy * Wp;
    };
    auto issue = [&]() {
        const bool ok = ln >= 0 && ln < p.N && lyy >= 1 && lyy <= p.H && lxx >= 1 && lxx <= p.W;
        const int64_t pix = ((int64_t)ln * p.H + (lyy - 1)) * p.W + (lxx - 1);
        if (isx) {
            const unsigned dst = sx0 + (unsigned)(lm & (RC - 1)) * 4096u + (unsigned)pj * 1024u;
            const unsigned short* s0 = ok ? p.xpl + pix * p.Cin + (cit * 64 + lch) : p.xzero;
            const unsigned short* s1 = ok ? s0 + p.xps : p.xzero;
            wg3_dma16(s0, dst);
            wg3_dma16(s1, dst + XPLANE);
        } else {
            const unsigned dst = sdy0 + (unsigned)(((lm % 3) + 3) % 3) * DYSTAGE + (unsigned)pj * 1024u;
            const unsigned short* s0 = ok ? p.dypl + pix * p.Cout + (cot * 64 + lch) : p.dyzero;
            const unsigned short* s1 = ok ? s0 + p.dyps : p.dyzero;
            wg3_dma16(s0, dst);
            wg3_dma16(s1, dst + DYPLANE);
        }
        // next chunk: 32 positions on
        ++lm;
        lxx += 32;
#pragma unroll
        for (int it = 0; it < 3; ++it)
            if (lxx >= Wp) { lxx -= Wp; ++lyy; }                 // (Wp >= 11: at most three row wraps per 32 positions)
        if (lyy >= Hp) { lyy -= Hp; ++ln; }
    };
    if (loader) {
        if (isx) {
            seek(c0 - D);
            for (int m = c0 - D; m <= c0 + D + 1; ++m) issue();
        } else {
            seek(c0);
            issue();
            issue();
        }
    }

    // ---- compute role: wave = (kh, co half i, ci half jn) ----
    const int kh = w >> 2, mi = (w >> 1) & 1, nj = w & 1;
    const int li = lane & 15, q4 = li >> 2, pp = li & 3;
    const int k0 = 8 * (lane >> 5) + q4;                          // + 16 g (+ 4 for the second half of a fragment)
    const int rba = 32 * mi + 16 * ((lane >> 4) & 1) + 4 * pp, rbb = 32 * nj + 16 * ((lane >> 4) & 1) + 4 * pp;
    unsigned aoff[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int k = 16 * g + k0;
        aoff[g] = (unsigned)(k * 128 + (((rba >> 3) ^ (4 * ((k >> 1) & 1))) << 4) + ((rba & 7) << 1));
    }
    unsigned xo[3][2][2];                                         // ring byte offsets of the x fragments of the CURRENT chunk
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int tt = 16 * g + k0 + 4 * rd + (kh - 1) * Wp + (kw - 1);
                const unsigned cofs = (unsigned)((((rbb >> 3) ^ (4 * ((tt >> 1) & 1))) << 4) + ((rbb & 7) << 1));
                xo[kw][g][rd] = (((unsigned)((32 * c0 + tt) * 128)) & (unsigned)(XPLANE - 1)) | cofs;
            }
    v16f acc[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[kw][e] = 0.f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int st = ((c0 % 3) + 3) % 3;                                  // dy stage of the current chunk
#pragma unroll 1
    for (int c = c0; c < c1; ++c) {
        if (loader) issue();                                      // x chunk c + D + 2 / dy chunk c + 2 (their slots were last read at step c - 1)
        const unsigned dyb = 2 * XPLANE + st * DYSTAGE;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            v4i af[2], bf[3][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(sbase + dyb + q * DYPLANE + aoff[g]));
                const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(sbase + dyb + q * DYPLANE + aoff[g] + 512));
                const v2i l2 = __builtin_bit_cast(v2i, lo), h2 = __builtin_bit_cast(v2i, hi);
                af[q] = (v4i){l2[0], l2[1], h2[0], h2[1]};
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(sbase + q * XPLANE + xo[kw][g][0]));
                    const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(sbase + q * XPLANE + xo[kw][g][1]));
                    const v2i l2 = __builtin_bit_cast(v2i, lo), h2 = __builtin_bit_cast(v2i, hi);
                    bf[kw][q] = (v4i){l2[0], l2[1], h2[0], h2[1]};
                }
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};     // lo*hi, hi*lo, hi*hi: small terms first
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    acc[kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, af[PA[term]]),
                                                                     __builtin_bit_cast(h16x8, bf[kw][PB[term]]), acc[kw], 0, 0, 0);
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int rd = 0; rd < 2; ++rd) xo[kw][g][rd] = (xo[kw][g][rd] + 4096u) & (unsigned)(XPLANE - 1);   // (the low 7 bits -- chunk and channel -- ride along)
        if (++st == 3) st = 0;
        // the loads issued one step ago (x chunk c + D + 1, dy chunk c + 1) have landed; this step's two stay in flight
        if (loader) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the surplus fetches behind the last step)

    // ---- partial tile -> slab z ----
    float* out = p.slabs + (int64_t)z * p.Cout * 9 * p.Cin;
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = cot * 64 + 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * hh;
            const int ci = cit * 64 + 32 * nj + r;
            out[((int64_t)co * 9 + (kh * 3 + kw)) * p.Cin + ci] = alpha * acc[kw][e];
        }
}
}  // namespace

// k-ranges of the ring kernel for this layer: about four rounds of one block per CU (the weight gradients run on a low-priority
// stream beside the critical path: blocks of ~0.5 ms give the CUs back often enough; 256 long blocks cost 15 ms per step), whole
// multiples of 8 (an XCD each)
static int wg3_nk(int64_t nchunk, int Cin, int Cout) {
    static const int target = [] { const char* e = getenv("KOAF_WGRAD3_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
    const int ncomb = (Cout / 64) * (Cin / 64);
    int64_t nk = target / ncomb;
    if (nk < 8) nk = 8;
    nk &= ~7ll;
    if (nk > nchunk) nk = nchunk;
    return (int)(nk < 1 ? 1 : nk);
}

// can the ring kernel take this 3x3 weight gradient?  (geometry only; the caller checks that plane images are given)
bool koaf_wgrad3_ring_ok(int N, int H, int W, int Cin, int Cout) {
    static const bool off = [] { const char* e = getenv("KOAF_WGRAD3_RING"); return e && e[0] == '0'; }();
    if (off) return false;
    const int Wp = W + 2, Hp = H + 2;
    const int D = (Wp + 1 + 31) >> 5;
    return Cin % 64 == 0 && Cout % 64 == 0 && W >= 16 &&      // (narrower images: the zero border is > 25 % of the positions)
           2 * D + 3 <= 16 && (int64_t)D * 32 + 32 < (int64_t)Hp * Wp &&
           (int64_t)N * Hp * Wp < (1ll << 31) - 4096 && (int64_t)(Cout / 64) * (Cin / 64) <= 64;
}

int64_t koaf_wgrad3_ring_ws(int N, int H, int W, int Cin, int Cout) {
    const int64_t nchunk = ((int64_t)N * (H + 2) * (W + 2) + 31) / 32;
    return (int64_t)(wg3_nk(nchunk, Cin, Cout) + 16) * Cout * 9 * Cin;        // +16: koaf_slab_reduce level-1 partials
}

int koaf_wgrad3_ring(const uint16_t* dy_planes, const uint16_t* x_planes, float* dw, float* slabs, const float* dy_amax,
                     float x_scale, int N, int H, int W, int Cin, int Cout, void* stream) {
    KOAF_REQUIRE(dy_planes && x_planes && dw && slabs && dy_amax && koaf_wgrad3_ring_ok(N, H, W, Cin, Cout), "koaf_wgrad3_ring: bad args");
    Wg3Params p;
    const int64_t npix = (int64_t)N * H * W;
    p.xpl = x_planes; p.dypl = dy_planes;
    p.xps = npix * Cin; p.dyps = npix * Cout;
    p.xzero = x_planes + 2 * p.xps; p.dyzero = dy_planes + 2 * p.dyps;
    p.slabs = slabs; p.dy_amax = dy_amax; p.x_scale = x_scale;
    p.status = koaf_status_ptr();
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    const int64_t nchunk = ((int64_t)N * (H + 2) * (W + 2) + 31) / 32;
    p.nchunk = (int)nchunk;
    p.nk = wg3_nk(nchunk, Cin, Cout);
    const int ncomb = (Cout / 64) * (Cin / 64);
    const int groups = (p.nk + 7) / 8;
    const int D = (W + 2 + 1 + 31) >> 5;
    const dim3 grid((unsigned)(groups * 8 * ncomb));
    if (2 * D + 3 <= 8) hipLaunchKernelGGL(wgrad3x3_ring_kernel<8>, grid, dim3(768), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wgrad3x3_ring_kernel<16>, grid, dim3(768), 0, (hipStream_t)stream, p);
    int rc = koaf_check_launch("koaf_wgrad3_ring");
    if (rc != KOAF_OK) return rc;
    return koaf_slab_reduce(slabs, p.nk, (int64_t)Cout * 9 * Cin, dw, stream);
}
