// koaf_elem.hip -- HBM-bound kernels of the koafusion train step: BatchNorm statistics / backward,
// bottleneck tail, max-pool, GAP, LayerNorm, softmax, GELU, dropout, focal loss, Adam, layout moves.
// All are float4-vectorised along the channel (fastest) axis and sized for >= 8 blocks per CU.
#include <stdarg.h>
#include "koaf_common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void koaf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* koaf_last_error(void) { return g_err; }
static uint32_t* g_status = nullptr;
uint32_t* koaf_status_ptr() { return g_status; }
extern "C" int koaf_set_status_buffer(uint32_t* dev4) { g_status = dev4; return KOAF_OK; }
extern "C" int koaf_version(void) { return 180; }   // 1.8: koaf_set_stream (streamed kernel of the dense 1x1 convolutions, KoafGemm A mode M_KS); 1.7: KoafEmit / KoafGemm.out_planes (epilogue cuts the consumer's plane images), loss labels outside [0, C); 1.6: KoafTail.idt_sc / idt_sh (tails behind a downsample branch), koaf_stem_fwd statistics, koaf_stem_wgrad dy_apply, koaf_bn_bwd_reduce_pool

namespace {

constexpr int EB = 256;  // elementwise block

// grid cap of the grid-stride element-wise kernels, in blocks per CU (KOAF_EW_BLOCKS_PER_CU, default 32)
inline int ew_blocks_per_cu() {
    static const int v = [] { const char* e = getenv("KOAF_EW_BLOCKS_PER_CU"); int n = e ? atoi(e) : 32; return n < 1 ? 1 : n; }();
    return v;
}
inline unsigned ew_grid(int64_t nvec) {
    int64_t b = cdiv64(nvec, EB);
    if (b > 256 * ew_blocks_per_cu()) b = 256 * ew_blocks_per_cu();
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ------------------------------------------------------------------------------------------------
// column partial sums over a [rows][C] tensor.  Block (256 thr) owns a chunk of <= 1024 columns and
// `rpb` rows; thread (cvx, ry) walks rows ry, ry+RP, ...; LDS tree over ry; writes part[blk][k][C].
// ------------------------------------------------------------------------------------------------
struct ColGeom {
    int CW;      // columns per chunk
    int nchunk;  // column chunks
    int CV;      // column vectors per chunk (CW/4)
    int RP;      // rows per pass (256/CV)
    int rpb;     // rows per block
    int nblk;    // row blocks
};
inline bool col_geom(int64_t rows, int C, int max_blk, ColGeom* g) {
    if (C % 4) return false;
    int CW = C > 1024 ? 1024 : C;
    if (C % CW) return false;
    int CV = CW / 4;
    if (256 % CV) return false;
    g->CW = CW;
    g->nchunk = C / CW;
    g->CV = CV;
    g->RP = 256 / CV;
    int64_t rpb = cdiv64(rows, max_blk);
    rpb = cdiv64(rpb, g->RP) * g->RP;
    if (rpb < g->RP * 4) rpb = g->RP * 4;
    g->rpb = (int)rpb;
    g->nblk = (int)cdiv64(rows, rpb);
    return true;
}

template <int NS>
__device__ __forceinline__ void col_block_reduce(v4f (&s)[NS], float* part, int blk, int C, int c0, int CV,
                                                 int RP) {
    __shared__ v4f red[256];
    const int t = threadIdx.x;
    const int cvx = t % CV, ry = t / CV;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        __syncthreads();
        red[t] = s[k];
        __syncthreads();
        if (ry == 0) {
            v4f a = red[cvx];
            for (int j = 1; j < RP; ++j) a += red[j * CV + cvx];
            *(v4f*)&part[((int64_t)blk * NS + k) * C + c0 + 4 * cvx] = a;
        }
    }
}

template <bool H>
__global__ void __launch_bounds__(256) colstats_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                       ColGeom g, float* __restrict__ part, int sq,
                                                       const float* __restrict__ shift) {
    const int t = threadIdx.x, cvx = t % g.CV, ry = t / g.CV;
    const int c0 = blockIdx.y * g.CW;
    const int64_t rbeg = (int64_t)blockIdx.x * g.rpb;
    const int64_t rend = min(rows, rbeg + (int64_t)g.rpb);
    v4f s[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    v4f k = {0, 0, 0, 0};
    if (shift) k = *(const v4f*)&shift[c0 + 4 * cvx];       // sums about the shift (see KoafGemm.stats_shift)
    for (int64_t r = rbeg + ry; r < rend; r += g.RP) {
        v4f v = load4<H>(x, r * C + c0 + 4 * cvx) - k;
        s[0] += v;
        s[1] += v * v;
    }
    if (sq) col_block_reduce<2>(s, part, blockIdx.x, C, c0, g.CV, g.RP);
    else {
        v4f s1[1] = {s[0]};
        col_block_reduce<1>(s1, part, blockIdx.x, C, c0, g.CV, g.RP);
    }
}

// generic tiny-C column sum (C not a multiple of 4, e.g. the 2-class head bias)
__global__ void colsum_small_kernel(const float* __restrict__ x, int rows, int C, float* __restrict__ out) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float a = 0.f;
    for (int r = 0; r < rows; ++r) a += x[(int64_t)r * C + c];
    out[c] = a;
}

// partial [rows][NS][C] -> out [NS][C]; block = 64 columns x 16 row groups
template <int NS>
__global__ void __launch_bounds__(1024) colfinal_kernel(const float* __restrict__ part, int rows, int C,
                                                        float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ double red[NS][16][64];
    const int cx = threadIdx.x & 63, gy = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double a[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) a[k] = 0.0;
    if (c < C)
        for (int r = gy; r < rows; r += 16)
#pragma unroll
            for (int k = 0; k < NS; ++k) a[k] += (double)part[((int64_t)r * NS + k) * C + c];
#pragma unroll
    for (int k = 0; k < NS; ++k) red[k][gy][cx] = a[k];
    __syncthreads();
    if (gy == 0 && c < C) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            double s = 0.0;
            for (int j = 0; j < 16; ++j) s += red[k][j][cx];
            (k == 0 ? out0 : out1)[c] = (float)s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm finalize (train: from partial column sums; eval: running stats)
// ------------------------------------------------------------------------------------------------
// Stage 1 of the per-channel finalisations when there are many partial rows (one per 128-row GEMM tile: 6400 for a
// layer-1 activation of the native batch): src [rows][nsum][C] fp32 -> ws [S][2][C] fp64, block (bx, s) sums row
// slice s of sums (0, i1) for 64 channels.  Fixed slice boundaries and summation order: deterministic.
__global__ void __launch_bounds__(1024) part_reduce_kernel(const float* __restrict__ src, int rows, int C, int nsum,
                                                           int i1, int chunk, double* __restrict__ ws) {
    __shared__ double red[2][16][64];
    const int cx = threadIdx.x & 63, gy = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * chunk, r1 = min(rows, r0 + chunk);
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int r = r0 + gy; r < r1; r += 16) {
            a += (double)src[((int64_t)r * nsum + 0) * C + c];
            b += (double)src[((int64_t)r * nsum + i1) * C + c];
        }
    red[0][gy][cx] = a;
    red[1][gy][cx] = b;
    __syncthreads();
    if (gy < 2 && c < C) {
        double t = 0.0;
        for (int j = 0; j < 16; ++j) t += red[gy][j][cx];
        ws[((int64_t)blockIdx.y * 2 + gy) * C + c] = t;
    }
}

// slices of the two-stage reduction: 0 = single stage
static inline int part_slices(int rows) { return rows > 128 ? (rows >= 4096 ? 64 : (rows + 63) / 64) : 0; }

template <typename T>
__global__ void __launch_bounds__(1024) bn_finalize_kernel(const T* __restrict__ stats, int rows, int C,
                                                           double inv_count, double unbias,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* running_mean,
                                                           float* running_var, int64_t* nbt, float momentum,
                                                           float eps, int train, float* mean, float* invstd,
                                                           float* sc, float* sh, const float* __restrict__ shift,
                                                           uint32_t* status) {
    __shared__ double red[2][16][64];
    const int cx = threadIdx.x & 63, gy = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    if (train) {
        double a = 0.0, b = 0.0;
        if (c < C)
            for (int r = gy; r < rows; r += 16) {
                a += (double)stats[((int64_t)r * 2 + 0) * C + c];
                b += (double)stats[((int64_t)r * 2 + 1) * C + c];
            }
        red[0][gy][cx] = a;
        red[1][gy][cx] = b;
        __syncthreads();
    }
    if (gy == 0 && c < C) {
        float m, var;
        if (train) {
            double s1 = 0.0, s2 = 0.0;
            for (int j = 0; j < 16; ++j) { s1 += red[0][j][cx]; s2 += red[1][j][cx]; }
            double dm = s1 * inv_count;                 // mean of (x - k)
            double dv = s2 * inv_count - dm * dm;
            if (dv < 0.0) dv = 0.0;
            if (shift) dm += (double)shift[c];          // (read before running_mean, possibly the same buffer, is updated)
            m = (float)dm;
            var = (float)dv;
            // momentum < 0 = nn.BatchNorm2d(momentum=None): cumulative moving average, factor 1 / num_batches_tracked (which
            // the entry point has already incremented for this batch: torch increments first, torch/nn/modules/batchnorm.py)
            const float f = momentum < 0.f ? 1.f / (float)(*nbt) : momentum;
            running_mean[c] = (1.f - f) * running_mean[c] + f * m;
            running_var[c] = (1.f - f) * running_var[c] + f * (float)(dv * unbias);
        } else {
            m = running_mean[c];
            var = running_var[c];
        }
        float is = 1.0f / sqrtf(var + eps);
        float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        mean[c] = m;
        invstd[c] = is;
        sc[c] = g * is;
        sh[c] = b - m * g * is;
        // a NaN / Inf in the conv output reaches the statistics, and from there every element of the channel: say so (the
        // consumers' fp16 clamp would turn relu(NaN * x + NaN) into 0)
        if (!koaf_bits_finite(koaf_absbits(g * is)) || !koaf_bits_finite(koaf_absbits(b - m * g * is))) koaf_status_add(status, 1, 1u);
    }
    if (train && nbt && momentum >= 0.f && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

// y = relu(sc*c+sh [+ identity])   (H: c, idt and y are bf16 activations)
template <bool H>
__global__ void __launch_bounds__(256) bn_add_relu_kernel(const float* __restrict__ c, const float* __restrict__ sc,
                                                          const float* __restrict__ sh, const float* __restrict__ idt,
                                                          const float* __restrict__ idsc,
                                                          const float* __restrict__ idsh, float* __restrict__ y,
                                                          int64_t nvec, int C4, uint32_t* status) {
    unsigned nsat = 0;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * EB) {
        const int cv = (int)(i % C4) * 4;
        v4f v = load4_nt<H>(c, i * 4);       // (streams: read / written once, kept out of L2's way)
        // (explicit fused multiply-adds: the loader that forms this tail on load -- KoafOperand.tf 3 -- rounds exactly alike)
        const v4f s4 = *(const v4f*)&sc[cv], h4 = *(const v4f*)&sh[cv];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], s4[j], h4[j]);
        if (idt) {
            v4f d = load4_nt<H>(idt, i * 4);
            if (idsc) {
                const v4f is4 = *(const v4f*)&idsc[cv], ih4 = *(const v4f*)&idsh[cv];
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] = fmaf(d[j], is4[j], ih4[j]);
            }
            v += d;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // (this tensor feeds convolutions at the fixed activation scale: |y| * KOAF_ACT_SCALE beyond the fp16 range is
            // clamped there -- counted here, where the check is free; a NaN counts too and stays a NaN in y)
            nsat += !(v[j] * KOAF_ACT_SCALE <= 65504.f) ? 1u : 0u;
            v[j] = v[j] != v[j] ? v[j] : fmaxf(v[j], 0.f);
        }
        store4_nt<H>(y, i * 4, v);
    }
    koaf_status_add(status, 0, nsat);
}

// BN backward pass 1: masked gradient + column partials of dz and dz*xhat   (H: c and ymask are bf16 activations)
// POOL: g is not a tensor of `rows` rows but the gradient of the 3x3 / stride-2 / pad-1 max-pool that follows this BatchNorm
// (+ReLU): pool_g [N][OH][OW][C] with the window positions pool_am recorded (koaf_maxpool_fwd); the gradient of input pixel
// (n, iy, ix) is gathered here -- the arithmetic of maxpool_bwd_kernel -- instead of being written by it and read back
struct PoolGeom { const float* g; const uint8_t* am; int H, W, OH, OW; };
template <bool H, bool POOL>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ c,
                                                            const float* __restrict__ ymask,
                                                            const float* __restrict__ sc, const float* __restrict__ sh,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, int mask_mode,
                                                            float* __restrict__ dz_out, int64_t rows, int C,
                                                            ColGeom geo, float* __restrict__ part, float* dz_amax, PoolGeom pg) {
    const int t = threadIdx.x, cvx = t % geo.CV, ry = t / geo.CV;
    unsigned am = 0u;       // largest |dz| as magnitude bits (a NaN / Inf wins: koaf_common.h)
    const int c0 = blockIdx.y * geo.CW + 4 * cvx;
    const int64_t rbeg = (int64_t)blockIdx.x * geo.rpb;
    const int64_t rend = min(rows, rbeg + (int64_t)geo.rpb);
    const v4f mu = *(const v4f*)&mean[c0], is = *(const v4f*)&invstd[c0];
    v4f s4 = {0, 0, 0, 0}, h4 = {0, 0, 0, 0};
    if (mask_mode == 2) { s4 = *(const v4f*)&sc[c0]; h4 = *(const v4f*)&sh[c0]; }
    v4f s[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    // POOL: (image, row, column) of this thread's pixel, carried from pass to pass (three 64-bit divisions per pixel cost more
    // than the gather itself)
    int ix = 0, iy = 0, n = 0;
    if constexpr (POOL) {
        int64_t pp = rbeg + ry;
        ix = (int)(pp % pg.W);
        pp /= pg.W;
        iy = (int)(pp % pg.H);
        n = (int)(pp / pg.H);
    }
    for (int64_t r = rbeg + ry; r < rend; r += geo.RP) {
        const int64_t o = r * C + c0;
        v4f gv;
        if constexpr (POOL) {
            gv = (v4f){0.f, 0.f, 0.f, 0.f};
            // output windows covering (iy, ix): oy in {iy / 2, (iy + 1) / 2} (one window row when iy is even), likewise ox.  All four
            // candidates are fetched at once from clamped addresses and masked (a loop with early exits kept one pair of loads in
            // flight per lane: 2.9 TB/s)
            const int oya = iy >> 1, oyb = (iy + 1) >> 1, oxa = ix >> 1, oxb = (ix + 1) >> 1;
            const bool vy[2] = {oya < pg.OH, oyb != oya && oyb < pg.OH}, vx[2] = {oxa < pg.OW, oxb != oxa && oxb < pg.OW};
            const int oys[2] = {oya, oyb}, oxs[2] = {oxa, oxb};
            uint32_t a4[4];
            v4f gg[4];
#pragma unroll
            for (int wy = 0; wy < 2; ++wy)
#pragma unroll
                for (int wx = 0; wx < 2; ++wx) {
                    const bool ok = vy[wy] && vx[wx];
                    const int64_t po = ok ? ((int64_t)(n * pg.OH + oys[wy]) * pg.OW + oxs[wx]) * C + c0 : (int64_t)c0;
                    a4[2 * wy + wx] = *(const uint32_t*)&pg.am[po];
                    gg[2 * wy + wx] = *(const v4f*)&pg.g[po];
                }
#pragma unroll
            for (int wy = 0; wy < 2; ++wy)
#pragma unroll
                for (int wx = 0; wx < 2; ++wx) {
                    const bool ok = vy[wy] && vx[wx];
                    const uint32_t want = (uint32_t)((iy - (oys[wy] * 2 - 1)) * 3 + (ix - (oxs[wx] * 2 - 1)));
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (ok && ((a4[2 * wy + wx] >> (8 * j)) & 0xffu) == want) gv[j] += gg[2 * wy + wx][j];
                }
            ix += geo.RP;
            while (ix >= pg.W) { ix -= pg.W; if (++iy == pg.H) { iy = 0; ++n; } }
        } else {
            gv = *(const v4f*)&g[o];
        }
        v4f cvv = load4<H>(c, o);
        if (mask_mode == 1) {
            v4f yv = load4<H>(ymask, o);
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = yv[j] > 0.f ? gv[j] : 0.f;
        } else if (mask_mode == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = (cvv[j] * s4[j] + h4[j]) > 0.f ? gv[j] : 0.f;
        }
        if (dz_out) *(v4f*)&dz_out[o] = gv;
        s[0] += gv;
        s[1] += gv * ((cvv - mu) * is);
#pragma unroll
        for (int j = 0; j < 4; ++j) am = max(am, koaf_absbits(gv[j]));
    }
    col_block_reduce<2>(s, part, blockIdx.x, C, blockIdx.y * geo.CW, geo.CV, geo.RP);
    if (dz_amax) block_amax_raise_bits(am, dz_amax);
}

template <typename T>
__global__ void __launch_bounds__(1024) bn_bwd_finalize_kernel(const T* __restrict__ part, int rows, int C,
                                                               double inv_count, const float* __restrict__ sc,
                                                               const float* __restrict__ invstd, float* dgamma,
                                                               float* dbeta, float* coef, int nsum, int i1,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ dz_amax, double sqrt_nm1,
                                                               float* amax) {
    __shared__ double red[2][16][64];
    float bound = 0.f;
    const int cx = threadIdx.x & 63, gy = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int r = gy; r < rows; r += 16) {
            a += (double)part[((int64_t)r * nsum + 0) * C + c];
            b += (double)part[((int64_t)r * nsum + i1) * C + c];
        }
    red[0][gy][cx] = a;
    red[1][gy][cx] = b;
    __syncthreads();
    if (gy == 0 && c < C) {
        double s1 = 0.0, s2 = 0.0;
        for (int j = 0; j < 16; ++j) { s1 += red[0][j][cx]; s2 += red[1][j][cx]; }
        if (dbeta) dbeta[c] = (float)s1;
        if (dgamma) dgamma[c] = (float)s2;
        const float k0 = sc[c], k1 = (float)(s1 * inv_count), k2 = (float)((double)sc[c] * (double)invstd[c] * s2 * inv_count);
        coef[c] = k0;
        coef[C + c] = k1;
        coef[2 * C + c] = k2;
        if (mean) {
            // dc = k0 * (dz - k1) - k2 * (x - mean) = k0 * dz + k3 - k2 * x: the form the GEMM loaders evaluate (KoafOperand.tf 2)
            coef[3 * C + c] = k2 * mean[c] - k0 * k1;
            // |dc| <= |k0| (max|dz| + |k1|) + |k2| max|x - mean|, and no sample of n lies further than sqrt(n - 1) standard
            // deviations from its mean (Samuelson): a guaranteed bound of the tensor's largest magnitude without a pass
            if (amax) bound = fabsf(k0) * ((dz_amax ? *dz_amax : 0.f) + fabsf(k1)) + fabsf(k2) * (float)(sqrt_nm1 / (double)invstd[c]);
        }
    }
    if (amax && mean) block_amax_raise(bound, amax);
}

template <bool H>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ c,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ coef, float* __restrict__ dc,
                                                           int64_t nvec, int C, float* __restrict__ amax) {
    const int C4 = C / 4;
    unsigned m = 0u;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * EB) {
        const int cv = (int)(i % C4) * 4;
        v4f z = *(const v4f*)&dz[i * 4];
        v4f x = load4<H>(c, i * 4);
        v4f k0 = *(const v4f*)&coef[cv], k1 = *(const v4f*)&coef[C + cv], k2 = *(const v4f*)&coef[2 * C + cv];
        v4f mu = *(const v4f*)&mean[cv];
        const v4f o = k0 * (z - k1) - k2 * (x - mu);
        *(v4f*)&dc[i * 4] = o;
#pragma unroll
        for (int j = 0; j < 4; ++j) m = max(m, koaf_absbits(o[j]));
    }
    // max |dc| of the tensor: the scale of dc as an operand of the fp16 contraction scheme
    if (amax) block_amax_raise_bits(m, amax);
}

// ------------------------------------------------------------------------------------------------
// input pipeline on the device (koafusion/preproc/_pt.py:75-99 PTToUnitRange, :257-345 PTRotate3DInSlice / PTRotate2D,
// :203-232 PTGammaCorrection, :101-135 PTNormalize -- applied per sample by the reference's CPU data-loader workers)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) minmax_part_kernel(const float* __restrict__ x, int64_t n, int nblk,
                                                          float* __restrict__ part) {
    const int b = blockIdx.y;
    const float* xb = x + (int64_t)b * n;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) {
        const float v = xb[i];
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    lo = -wave_max(-lo);
    hi = wave_max(hi);
    __shared__ float red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lo; red[1][threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((int64_t)b * nblk + blockIdx.x) * 2 + 0] = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
        part[((int64_t)b * nblk + blockIdx.x) * 2 + 1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    }
}
__global__ void __launch_bounds__(64) minmax_final_kernel(const float* __restrict__ part, int nblk, float* __restrict__ mm) {
    const int b = blockIdx.x;
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nblk; i += 64) {
        lo = fminf(lo, part[((int64_t)b * nblk + i) * 2 + 0]);
        hi = fmaxf(hi, part[((int64_t)b * nblk + i) * 2 + 1]);
    }
    lo = -wave_max(-lo);
    hi = wave_max(hi);
    if (threadIdx.x == 0) { mm[2 * b] = lo; mm[2 * b + 1] = hi; }
}

// One thread per V consecutive slices of one (sample, row, column): the slice index is the fastest-varying one in
// memory, so consecutive lanes read consecutive addresses (V = 4 when S % 4 == 0, else 1; radiographs: S = 1, lanes run
// along the columns).  The four bilinear source positions depend on (row, column) only and are recomputed per lane.
template <int V>
__global__ void __launch_bounds__(256) augment_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      const float* __restrict__ mm, const float* __restrict__ prm,
                                                      int B, int R, int C, int S, float mean, float stdv) {
    const int SV = S / V;
    const int64_t total = (int64_t)B * R * C * SV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int sv = (int)(i % SV);
        const int64_t pix = i / SV;
        const int c = (int)(pix % C);
        const int r = (int)((pix / C) % R);
        const int b = (int)(pix / ((int64_t)R * C));
        const float mn = mm[2 * b], den = mm[2 * b + 1] - mn;
        const float cs = prm[4 * b], sn = prm[4 * b + 1], ex = prm[4 * b + 2];
        const bool rot = prm[4 * b + 3] != 0.f;
        const float* xb = x + (int64_t)b * R * C * S + V * sv;
        float v[V];
        if (rot) {
            // F.affine_grid(theta, align_corners=False) then F.grid_sample(bilinear, zeros, align_corners=False)
            const float xn = (2.f * c + 1.f) / C - 1.f, yn = (2.f * r + 1.f) / R - 1.f;
            const float gx = cs * xn - sn * yn, gy = sn * xn + cs * yn;
            const float ix = ((gx + 1.f) * C - 1.f) * 0.5f, iy = ((gy + 1.f) * R - 1.f) * 0.5f;
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy;
            const float tx = ix - fx, ty = iy - fy;
            const float w[4] = {(1.f - tx) * (1.f - ty), tx * (1.f - ty), (1.f - tx) * ty, tx * ty};
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
                if ((unsigned)xx < (unsigned)C && (unsigned)yy < (unsigned)R) {
                    const float* src = xb + (int64_t)(yy * C + xx) * S;
                    if constexpr (V == 4) {
                        const v4f q = *(const v4f*)src;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += ((q[j] - mn) / den) * w[k];
                    } else {
                        v[0] += ((src[0] - mn) / den) * w[k];
                    }
                }
            }
        } else {
            const float* src = xb + (int64_t)(r * C + c) * S;
            if constexpr (V == 4) {
                const v4f q = *(const v4f*)src;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (q[j] - mn) / den;
            } else {
                v[0] = (src[0] - mn) / den;
            }
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if (ex != 0.f) v[j] = powf(v[j], ex);
            v[j] = (v[j] - mean) / stdv;
        }
        float* dst = y + pix * S + V * sv;
        if constexpr (V == 4) *(v4f*)dst = (v4f){v[0], v[1], v[2], v[3]};
        else dst[0] = v[0];
    }
}

// ------------------------------------------------------------------------------------------------
// max-pool 3x3 s2 p1 over relu(sc*c+sh); GAP
// ------------------------------------------------------------------------------------------------
template <bool B16>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const float* __restrict__ c, const float* __restrict__ sc,
                                                          const float* __restrict__ sh, float* __restrict__ y,
                                                          uint8_t* __restrict__ am, int N, int H, int W, int C,
                                                          int OH, int OW, uint32_t* status) {
    const int C4 = C / 4;
    unsigned nsat = 0;
    const int64_t total = (int64_t)N * OH * OW * C4;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int cv = (int)(i % C4) * 4;
        int64_t p = i / C4;
        const int ox = (int)(p % OW);
        p /= OW;
        const int oy = (int)(p % OH);
        const int n = (int)(p / OH);
        const v4f s4 = *(const v4f*)&sc[cv], h4 = *(const v4f*)&sh[cv];
        v4f best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int iy = oy * 2 - 1 + kh;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ix = ox * 2 - 1 + kw;
                if ((unsigned)ix >= (unsigned)W) continue;
                v4f v = load4<B16>(c, ((int64_t)(n * H + iy) * W + ix) * C + cv);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = fmaxf(v[j] * s4[j] + h4[j], 0.f);
                    if (a > best[j]) { best[j] = a; bi[j] = kh * 3 + kw; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) nsat += !(best[j] * KOAF_ACT_SCALE <= 65504.f) ? 1u : 0u;    // (as in bn_add_relu_kernel)
        store4<B16>(y, i * 4, best);
        *(uint32_t*)&am[i * 4] = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
    }
    koaf_status_add(status, 0, nsat);
}

__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ am,
                                                          float* __restrict__ da, int N, int H, int W, int C, int OH,
                                                          int OW) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int cv = (int)(i % C4) * 4;
        int64_t p = i / C4;
        const int ix = (int)(p % W);
        p /= W;
        const int iy = (int)(p % H);
        const int n = (int)(p / H);
        v4f acc = {0, 0, 0, 0};
        // output windows covering (iy, ix): oy*2-1 <= iy <= oy*2+1
        for (int oy = (iy) / 2; oy <= (iy + 1) / 2; ++oy) {
            if (oy >= OH) continue;
            const int kh = iy - (oy * 2 - 1);
            if (kh < 0 || kh > 2) continue;
            for (int ox = (ix) / 2; ox <= (ix + 1) / 2; ++ox) {
                if (ox >= OW) continue;
                const int kw = ix - (ox * 2 - 1);
                if (kw < 0 || kw > 2) continue;
                const int64_t o = ((int64_t)(n * OH + oy) * OW + ox) * C + cv;
                const uint32_t a4 = *(const uint32_t*)&am[o];
                const v4f g = *(const v4f*)&dy[o];
                const uint32_t want = (uint32_t)(kh * 3 + kw);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (((a4 >> (8 * j)) & 0xffu) == want) acc[j] += g[j];
            }
        }
        *(v4f*)&da[i * 4] = acc;
    }
}

template <bool H>
__global__ void __launch_bounds__(256) gap_fwd_kernel(const float* __restrict__ y, float* __restrict__ out, int N,
                                                      int HW, int C) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)N * C4;
    const float inv = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int cv = (int)(i % C4) * 4;
        const int64_t n = i / C4;
        v4f a = {0, 0, 0, 0};
        for (int p = 0; p < HW; ++p) a += load4<H>(y, (n * HW + p) * C + cv);
        *(v4f*)&out[n * C + cv] = a * inv;
    }
}
__global__ void __launch_bounds__(256) gap_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dy, int N,
                                                      int HW, int C) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)N * HW * C4;
    const float inv = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        const int cv = (int)(i % C4) * 4;
        const int64_t n = i / ((int64_t)C4 * HW);
        *(v4f*)&dy[i * 4] = *(const v4f*)&dout[n * C + cv] * inv;
    }
}

// ------------------------------------------------------------------------------------------------
// layout moves
// ------------------------------------------------------------------------------------------------
// x [B][P][S] -> out [B][S][P]   (P = R*C pixels), 32x32 LDS tiles
__global__ void __launch_bounds__(256) slice_fold_kernel(const float* __restrict__ x, float* __restrict__ out, int P,
                                                         int S) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, s0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 8 rows per pass
    const float* xb = x + (int64_t)b * P * S;
    float* ob = out + (int64_t)b * P * S;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int p = p0 + ty + 8 * j, s = s0 + tx;
        tile[ty + 8 * j][tx] = (p < P && s < S) ? xb[(int64_t)p * S + s] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int s = s0 + ty + 8 * j, p = p0 + tx;
        if (p < P && s < S) ob[(int64_t)s * P + p] = tile[tx][ty + 8 * j];
    }
}

// 2x average pooling == F.interpolate(scale 0.5, align_corners=False, linear modes)
__global__ void __launch_bounds__(256) downscale2_kernel(const float* __restrict__ x, float* __restrict__ out, int B,
                                                         int R, int Cc, int S, int fs) {
    const int OR = R / 2, OC = Cc / 2, OS = S / fs;
    const int64_t total = (int64_t)B * OR * OC * OS;
    const float inv = 1.f / (float)(4 * fs);
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        int64_t p = i;
        const int os = (int)(p % OS); p /= OS;
        const int oc = (int)(p % OC); p /= OC;
        const int orr = (int)(p % OR);
        const int b = (int)(p / OR);
        float a = 0.f;
        for (int dr = 0; dr < 2; ++dr)
            for (int dc = 0; dc < 2; ++dc)
                for (int ds = 0; ds < fs; ++ds)
                    a += x[(((int64_t)b * R + 2 * orr + dr) * Cc + 2 * oc + dc) * S + os * fs + ds];
        out[i] = a * inv;
    }
}

// F.interpolate(x, scale_factor, mode = linear | bilinear | trilinear, align_corners=False, recompute_scale_factor=True) for
// ANY scale (preproc/_pt.py:175-192): x [BC][I0][I1][I2] -> out [BC][O0][O1][O2] (absent dimensions have size 1).  torch's rule:
// source coordinate = (in / out) * (dst + 0.5) - 0.5 clamped at 0, its two neighbours (the upper one clamped at in - 1)
// weighted linearly.
struct ResizeGeom { int I[3], O[3]; float rs[3]; };
__device__ __forceinline__ void resize_axis(int dst, int in, float rs, int& i0, int& i1, float& w1) {
    float src = rs * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = min((int)src, in - 1);
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    w1 = src - (float)i0;
}
__global__ void __launch_bounds__(256) resize_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t BC,
                                                     ResizeGeom g) {
    const int64_t on = (int64_t)g.O[0] * g.O[1] * g.O[2], in = (int64_t)g.I[0] * g.I[1] * g.I[2];
    const int64_t total = BC * on;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < total; i += (int64_t)gridDim.x * EB) {
        int64_t p = i;
        const int o2 = (int)(p % g.O[2]); p /= g.O[2];
        const int o1 = (int)(p % g.O[1]); p /= g.O[1];
        const int o0 = (int)(p % g.O[0]);
        const int64_t bc = p / g.O[0];
        int a0, a1, b0, b1, c0, c1;
        float wa, wb, wc;
        resize_axis(o0, g.I[0], g.rs[0], a0, a1, wa);
        resize_axis(o1, g.I[1], g.rs[1], b0, b1, wb);
        resize_axis(o2, g.I[2], g.rs[2], c0, c1, wc);
        const float* xb = x + bc * in;
        auto at = [&](int a, int b, int c) { return xb[((int64_t)a * g.I[1] + b) * g.I[2] + c]; };
        const float v00 = at(a0, b0, c0) * (1.f - wc) + at(a0, b0, c1) * wc, v01 = at(a0, b1, c0) * (1.f - wc) + at(a0, b1, c1) * wc;
        const float v10 = at(a1, b0, c0) * (1.f - wc) + at(a1, b0, c1) * wc, v11 = at(a1, b1, c0) * (1.f - wc) + at(a1, b1, c1) * wc;
        out[i] = (v00 * (1.f - wb) + v01 * wb) * (1.f - wa) + (v10 * (1.f - wb) + v11 * wb) * wa;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * D;
    const int D4 = D / 4;
    float s = 0.f;
    for (int i = lane; i < D4; i += 64) { v4f v = *(const v4f*)&xr[i * 4]; s += v[0] + v[1] + v[2] + v[3]; }
    const float m = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int i = lane; i < D4; i += 64) {
        v4f v = *(const v4f*)&xr[i * 4] - m;
        q += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
    for (int i = lane; i < D4; i += 64) {
        v4f v = (*(const v4f*)&xr[i * 4] - m) * rs;
        v = v * *(const v4f*)&gamma[i * 4] + *(const v4f*)&beta[i * 4];
        *(v4f*)&y[(int64_t)row * D + i * 4] = v;
    }
    if (lane == 0) { mean[row] = m; rstd[row] = rs; }
}

// dx per row (one wave per row) ...
__global__ void __launch_bounds__(256) layernorm_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ dx,
                                                               int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (int64_t)row * D;
    const float* gr = dy + (int64_t)row * D;
    const int D4 = D / 4;
    const float m = mean[row], rs = rstd[row];
    float a = 0.f, b = 0.f;
    for (int i = lane; i < D4; i += 64) {
        v4f g = *(const v4f*)&gr[i * 4] * *(const v4f*)&gamma[i * 4];
        v4f xh = (*(const v4f*)&xr[i * 4] - m) * rs;
        a += g[0] + g[1] + g[2] + g[3];
        b += g[0] * xh[0] + g[1] * xh[1] + g[2] * xh[2] + g[3] * xh[3];
    }
    a = wave_sum(a) / (float)D;
    b = wave_sum(b) / (float)D;
    for (int i = lane; i < D4; i += 64) {
        v4f g = *(const v4f*)&gr[i * 4] * *(const v4f*)&gamma[i * 4];
        v4f xh = (*(const v4f*)&xr[i * 4] - m) * rs;
        *(v4f*)&dx[(int64_t)row * D + i * 4] = (g - a - xh * b) * rs;
    }
}
// ... and the parameter gradients as column partials: part[blk][0] = sum dy*xhat, [1] = sum dy
__global__ void __launch_bounds__(256) layernorm_bwd_param_kernel(const float* __restrict__ dy,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, int64_t rows, int D,
                                                                  ColGeom geo, float* __restrict__ part) {
    const int t = threadIdx.x, cvx = t % geo.CV, ry = t / geo.CV;
    const int c0 = blockIdx.y * geo.CW + 4 * cvx;
    const int64_t rbeg = (int64_t)blockIdx.x * geo.rpb;
    const int64_t rend = min(rows, rbeg + (int64_t)geo.rpb);
    v4f s[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int64_t r = rbeg + ry; r < rend; r += geo.RP) {
        v4f g = *(const v4f*)&dy[r * D + c0];
        v4f xh = (*(const v4f*)&x[r * D + c0] - mean[r]) * rstd[r];
        s[0] += g * xh;
        s[1] += g;
    }
    col_block_reduce<2>(s, part, blockIdx.x, D, blockIdx.y * geo.CW, geo.CV, geo.RP);
}

// ------------------------------------------------------------------------------------------------
// softmax rows (attention), in place; one wave per row
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) softmax_rows_kernel(float* __restrict__ x, int64_t rows, int n) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* xr = x + row * n;
    float m = -INFINITY;
    for (int i = lane; i < n; i += 64) m = fmaxf(m, xr[i]);
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += expf(xr[i] - m);
    s = wave_sum(s);
    const float inv = 1.f / s;
    for (int i = lane; i < n; i += 64) xr[i] = expf(xr[i] - m) * inv;
}
// ds = p * (dp - sum(dp*p)) * scale, in place on dp
__global__ void __launch_bounds__(256) softmax_bwd_rows_kernel(float* __restrict__ dp, const float* __restrict__ p,
                                                               int64_t rows, int n, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* dr = dp + row * n;
    const float* pr = p + row * n;
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += dr[i] * pr[i];
    s = wave_sum(s);
    for (int i = lane; i < n; i += 64) dr[i] = pr[i] * (dr[i] - s) * scale;
}

// ------------------------------------------------------------------------------------------------
// pointwise
// ------------------------------------------------------------------------------------------------
enum { PW_GELU_F, PW_GELU_B, PW_RELU_F, PW_RELU_B, PW_ADD };
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_d(float x) {
    const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}
template <int OP>
__global__ void __launch_bounds__(256) pointwise_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ out, int64_t n) {
    const int64_t nvec = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * EB) {
        v4f x = *(const v4f*)&a[i * 4], y = {0, 0, 0, 0}, o;
        if (OP == PW_GELU_B || OP == PW_RELU_B || OP == PW_ADD) y = *(const v4f*)&b[i * 4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (OP == PW_GELU_F) o[j] = gelu_f(x[j]);
            else if (OP == PW_GELU_B) o[j] = x[j] * gelu_d(y[j]);       // a = dy, b = x
            else if (OP == PW_RELU_F) o[j] = fmaxf(x[j], 0.f);
            else if (OP == PW_RELU_B) o[j] = y[j] > 0.f ? x[j] : 0.f;   // a = dy, b = y
            else o[j] = x[j] + y[j];
        }
        *(v4f*)&out[i * 4] = o;
    }
    // tail
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = nvec * 4 + threadIdx.x;
        float x = a[i], y = (OP == PW_GELU_B || OP == PW_RELU_B || OP == PW_ADD) ? b[i] : 0.f, o;
        if (OP == PW_GELU_F) o = gelu_f(x);
        else if (OP == PW_GELU_B) o = x * gelu_d(y);
        else if (OP == PW_RELU_F) o = fmaxf(x, 0.f);
        else if (OP == PW_RELU_B) o = y > 0.f ? x : 0.f;
        else o = x + y;
        out[i] = o;
    }
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// `epoch` (nullable): a device-resident step counter folded into the seed, so that a HIP-graph replay of a captured step
// (whose `seed` argument is frozen) still draws fresh masks every step; forward and backward of one step read the same value
__device__ __forceinline__ uint64_t step_seed(uint64_t seed, const int64_t* epoch) {
    return epoch ? seed ^ mix64(0x9E3779B97F4A7C15ull * (uint64_t)(*epoch + 1)) : seed;
}

__global__ void __launch_bounds__(256) dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                      float p, float inv_keep, uint64_t seed0, const int64_t* epoch) {
    const uint64_t seed = step_seed(seed0, epoch);
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const uint64_t h = mix64(seed ^ mix64((uint64_t)i));
        const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
        y[i] = (u >= p) ? x[i] * inv_keep : 0.f;
    }
}

// nn.Dropout2d on an NHWC feature map: one Bernoulli draw per (image, channel), index n*C + c -- the same draw the
// element-wise kernel makes on the (N, C) pooled output, so pooled and spatial encoders share their masks
__global__ void __launch_bounds__(256) dropout2d_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                        int64_t hwc, int C, float p, float inv_keep, uint64_t seed0,
                                                        const int64_t* epoch) {
    const uint64_t seed = step_seed(seed0, epoch);
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
        const int64_t idx = (i / hwc) * C + (i % C);
        const uint64_t h = mix64(seed ^ mix64((uint64_t)idx));
        const float u = (float)(h >> 40) * (1.0f / 16777216.0f);
        y[i] = (u >= p) ? x[i] * inv_keep : 0.f;
    }
}

__global__ void counter_add_kernel(int64_t* ctr, int64_t delta) { *ctr += delta; }

// integer volumes as they come off the disk (uint8 radiographs, uint16 / int16 MRI) -> fp32, 16 elements per thread step
template <typename T>
__global__ void __launch_bounds__(256) widen_kernel(const T* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < nv; i += (int64_t)gridDim.x * EB) {
        T v[4];
        __builtin_memcpy(v, x + i * 4, sizeof(v));        // one 4- or 8-byte load (4-element alignment checked by the caller)
        *(v4f*)&y[i * 4] = (v4f){(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) y[nv * 4 + threadIdx.x] = (float)x[nv * 4 + threadIdx.x];
}

__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ p, float v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// losses (tiny: one block)
// ------------------------------------------------------------------------------------------------
// logits [B][C][S] (S = product of the spatial dims of a (b, ch, d0, d1, ...) input; 1 for (b, ch)), target [B][S], cw = class
// weights [C] or NULL.  F.cross_entropy(reduction='none', weight=cw) per element: ce = -cw[t] * log_softmax(x)[t];
//   focal (FocalLoss, _losses.py:101-108):  logpt = -ce, pt = exp(logpt), l = -(1 - pt)^gamma * logpt, plain mean | sum over elements
//   else  (nn.CrossEntropyLoss(weight=cw), _losses.py:36,49):  sum ce / sum cw[t]   (weighted mean)
// Targets outside [0, C): -100 is F.cross_entropy's default ignore_index -- such an element has zero loss and zero gradient, and
// the (weighted) mean of the cross-entropy leaves it out of its denominator, while the focal loss, which takes the mean of the
// per-element values itself (koafusion/various/_losses.py:101-108: reduction 'none', then .mean()), still divides by every
// element, as the reference does.  Any other out-of-range label (torch raises a device assert there) is treated the same way
// and counted in the numerics status word [1] instead of indexing out of bounds.
__global__ void __launch_bounds__(256) focal_loss_kernel(const float* __restrict__ logits,
                                                         const int64_t* __restrict__ target, const float* __restrict__ cw,
                                                         float* loss, float* __restrict__ dlogits, int B, int C, int64_t S,
                                                         float gamma, int mean, int focal, uint32_t* status) {
    __shared__ float red[256];
    const int64_t n = (int64_t)B * S;
    float wsum = (float)n;
    unsigned nbad = 0;
    if (!focal) {       // the (weighted) mean's denominator first: the elements that count
        float a = 0.f;
        for (int64_t i = threadIdx.x; i < n; i += 256) {
            const int64_t tg = target[i];
            if (tg >= 0 && tg < C) a += cw ? cw[(int)tg] : 1.f;
        }
        red[threadIdx.x] = a;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        wsum = red[0];
        __syncthreads();
    }
    float acc = 0.f;
    const float wgt = mean ? 1.f / wsum : 1.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const int64_t b = i / S, sp = i - b * S;
        const float* x = logits + b * C * S + sp;        // class j at x[j * S]
        float* d = dlogits + b * C * S + sp;
        const int64_t tg64 = target[i];
        if (tg64 < 0 || tg64 >= C) {
            nbad += (tg64 != -100) ? 1u : 0u;
            for (int j = 0; j < C; ++j) d[j * S] = 0.f;
            continue;
        }
        float m = -INFINITY;
        for (int j = 0; j < C; ++j) m = fmaxf(m, x[j * S]);
        float s = 0.f;
        for (int j = 0; j < C; ++j) s += expf(x[j * S] - m);
        const float lse = m + logf(s);
        const int tg = (int)tg64;
        const float w = cw ? cw[tg] : 1.f;
        const float logpt = w * (x[tg * S] - lse);
        const float pt = expf(logpt);
        float li, dl;  // loss_i, d loss_i / d logpt
        if (focal) {
            const float om = 1.f - pt;
            const float pw = powf(om, gamma);
            li = -pw * logpt;
            const float pw1 = (gamma == 0.f) ? 0.f : gamma * powf(om, gamma - 1.f);
            dl = -pw + pw1 * pt * logpt;
        } else {
            li = -logpt;
            dl = -1.f;
        }
        acc += li;
        for (int j = 0; j < C; ++j) {
            const float pj = expf(x[j * S] - lse);
            d[j * S] = dl * w * ((j == tg ? 1.f : 0.f) - pj) * wgt;
        }
    }
    koaf_status_add(status, 1, nbad);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = red[0] * wgt;
}

// Segmentation-sized logits (B * S elements beyond one block's reach): the same arithmetic on a grid, reduced in two fixed-order
// stages -- block k owns the elements [k * chunk, (k + 1) * chunk), its 256 lanes stride through them and meet in one LDS tree;
// the per-block partials ws[k] are then summed by ONE block in index order, so the result does not depend on scheduling.
// LOSS_PASS 0: ws[k] = the block's share of the (weighted) mean's denominator (cross-entropy only);
// LOSS_PASS 1: gradients + ws[k] = the block's loss sum; wden = device scalar of the denominator (NULL: n, or 1 for a sum)
template <int LOSS_PASS>
__global__ void __launch_bounds__(256) loss_grid_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                        const float* __restrict__ cw, float* __restrict__ dlogits, float* __restrict__ ws,
                                                        const float* __restrict__ wden, int B, int C, int64_t S, int64_t chunk, float gamma,
                                                        int mean, int focal, uint32_t* status) {
    __shared__ float red[256];
    const int64_t n = (int64_t)B * S;
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < n) ? lo + chunk : n;
    float acc = 0.f;
    unsigned nbad = 0;
    if constexpr (LOSS_PASS == 0) {
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
            const int64_t tg = target[i];
            if (tg >= 0 && tg < C) acc += cw ? cw[(int)tg] : 1.f;
        }
    } else {
        const float wgt = mean ? 1.f / (wden ? *wden : (float)n) : 1.f;
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
            const int64_t b = i / S, sp = i - b * S;
            const float* x = logits + b * C * S + sp;
            float* d = dlogits + b * C * S + sp;
            const int64_t tg64 = target[i];
            if (tg64 < 0 || tg64 >= C) {
                nbad += (tg64 != -100) ? 1u : 0u;
                for (int j = 0; j < C; ++j) d[j * S] = 0.f;
                continue;
            }
            float m = -INFINITY;
            for (int j = 0; j < C; ++j) m = fmaxf(m, x[j * S]);
            float se = 0.f;
            for (int j = 0; j < C; ++j) se += expf(x[j * S] - m);
            const float lse = m + logf(se);
            const int tg = (int)tg64;
            const float w = cw ? cw[tg] : 1.f;
            const float logpt = w * (x[tg * S] - lse);
            const float pt = expf(logpt);
            float li, dl;
            if (focal) {
                const float om = 1.f - pt;
                const float pw = powf(om, gamma);
                li = -pw * logpt;
                const float pw1 = (gamma == 0.f) ? 0.f : gamma * powf(om, gamma - 1.f);
                dl = -pw + pw1 * pt * logpt;
            } else {
                li = -logpt;
                dl = -1.f;
            }
            acc += li;
            for (int j = 0; j < C; ++j) {
                const float pj = expf(x[j * S] - lse);
                d[j * S] = dl * w * ((j == tg ? 1.f : 0.f) - pj) * wgt;
            }
        }
        koaf_status_add(status, 1, nbad);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) ws[blockIdx.x] = red[0];
}
// out = (sum_k ws[k]) * scale, k in index order (lane l takes k = l, l + 256, ...; one LDS tree); scale: 1 / *wden, 1 / nden or 1
__global__ void __launch_bounds__(256) loss_grid_sum_kernel(const float* __restrict__ ws, int nblk, float* __restrict__ out,
                                                            const float* __restrict__ wden, float nden) {
    __shared__ float red[256];
    float a = 0.f;
    for (int k = threadIdx.x; k < nblk; k += 256) a += ws[k];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0] * (wden ? 1.f / *wden : (nden > 0.f ? 1.f / nden : 1.f));
}

// ------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam single-tensor update rule, coupled L2; adamw: decoupled)
// ------------------------------------------------------------------------------------------------
// ++step; hyper = {lr, lr / (1 - b1^step), sqrt(1 - b2^step)}: what koaf_adam_step derives on the host from (lr, step), derived
// on the device so that a captured (HIP-graph) optimizer step advances from replay to replay
__global__ void adam_hyper_kernel(int32_t* step, const float* lr, double b1, double b2, float* hyper) {
    const int st = *step + 1;
    *step = st;
    const double bc1 = 1.0 - pow(b1, (double)st), bc2 = 1.0 - pow(b2, (double)st);
    hyper[0] = *lr;
    hyper[1] = (float)((double)*lr / bc1);
    hyper[2] = (float)sqrt(bc2);
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float omb1, float b2, float omb2, float eps, float wd, float step_size,
                                                   float bc2_sqrt, int adamw, const float* __restrict__ hyper,
                                                   float* __restrict__ vmax) {
    // (omb1 = 1 - beta1 and omb2 = 1 - beta2 arrive rounded from the DOUBLE differences, as torch forms them: 1.f - 0.999f is
    // 1.3e-5 away from float(1 - 0.999), which showed in the second moments)
    if (hyper) { lr = hyper[0]; step_size = hyper[1]; bc2_sqrt = hyper[2]; }   // device-resident step state (koaf_adam_hyper)
    const int64_t nvec = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * EB) {
        v4f pv = *(const v4f*)&p[i * 4], gv = *(const v4f*)&g[i * 4];
        v4f mv = *(const v4f*)&m[i * 4], vv = *(const v4f*)&v[i * 4];
        v4f xv = {0.f, 0.f, 0.f, 0.f};
        if (vmax) xv = *(const v4f*)&vmax[i * 4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gj = gv[j], pj = pv[j];
            if (adamw) pj *= (1.f - lr * wd);
            else gj += wd * pj;
            const float mj = mv[j] + (gj - mv[j]) * omb1;
            const float vj = vv[j] * b2 + omb2 * gj * gj;
            float vd = vj;
            if (vmax) { vd = fmaxf(xv[j], vj); xv[j] = vd; }     // amsgrad: the running maximum of the second moment
            const float denom = sqrtf(vd) / bc2_sqrt + eps;
            pv[j] = pj - step_size * (mj / denom);
            mv[j] = mj;
            vv[j] = vj;
        }
        *(v4f*)&p[i * 4] = pv;
        *(v4f*)&m[i * 4] = mv;
        *(v4f*)&v[i * 4] = vv;
        if (vmax) *(v4f*)&vmax[i * 4] = xv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = nvec * 4 + threadIdx.x;
        float gj = g[i], pj = p[i];
        if (adamw) pj *= (1.f - lr * wd);
        else gj += wd * pj;
        const float mj = m[i] + (gj - m[i]) * omb1;
        const float vj = v[i] * b2 + omb2 * gj * gj;
        float vd = vj;
        if (vmax) { vd = fmaxf(vmax[i], vj); vmax[i] = vd; }
        p[i] = pj - step_size * (mj / (sqrtf(vd) / bc2_sqrt + eps));
        m[i] = mj;
        v[i] = vj;
    }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
#define STREAM ((hipStream_t)stream)

extern "C" int64_t koaf_bn_reduce_ws(int32_t rows, int32_t C) {
    const int S = part_slices(rows);
    return S ? (int64_t)S * 2 * C * (int64_t)sizeof(double) : 0;
}

// stage 1 when it pays; returns the slice count (0: the caller reads the fp32 rows itself)
static int part_reduce(const float* src, int rows, int C, int nsum, int i1, double* ws, hipStream_t st) {
    const int S = ws ? part_slices(rows) : 0;
    if (!S) return 0;
    const int chunk = (rows + S - 1) / S;
    hipLaunchKernelGGL(part_reduce_kernel, dim3((C + 63) / 64, S), dim3(1024), 0, st, src, rows, C, nsum, i1, chunk, ws);
    return S;
}

extern "C" int koaf_bn_finalize(const float* stats, int32_t rows, int32_t C, int64_t count, const float* gamma,
                                const float* beta, float* running_mean, float* running_var,
                                int64_t* num_batches_tracked, float momentum, float eps, int32_t train, float* mean,
                                float* invstd, float* sc, float* sh, const float* shift, double* ws, void* stream) {
    KOAF_REQUIRE(C > 0 && mean && invstd && sc && sh && running_mean && running_var, "koaf_bn_finalize: bad args");
    KOAF_REQUIRE(!train || (stats && rows > 0 && count > 0), "koaf_bn_finalize: train mode needs stats");
    const double inv = train ? 1.0 / (double)count : 0.0;
    const double unbias = (train && count > 1) ? (double)count / (double)(count - 1) : 1.0;
    if (train && momentum < 0.f) {
        // cumulative average (momentum None): the factor is 1 / (the count INCLUDING this batch); the blocks of the kernel
        // below all read it, so the increment is its own stream-ordered launch in front of them
        KOAF_REQUIRE(num_batches_tracked, "koaf_bn_finalize: momentum < 0 (cumulative average) needs num_batches_tracked");
        int rc = koaf_counter_add(num_batches_tracked, 1, stream);
        if (rc != KOAF_OK) return rc;
    }
    const int S = train ? part_reduce(stats, rows, C, 2, 1, ws, STREAM) : 0;
    if (S)
        hipLaunchKernelGGL(bn_finalize_kernel<double>, dim3((C + 63) / 64), dim3(1024), 0, STREAM, ws, S, C, inv, unbias,
                           gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, train, mean,
                           invstd, sc, sh, train ? shift : nullptr, koaf_status_ptr());
    else
        hipLaunchKernelGGL(bn_finalize_kernel<float>, dim3((C + 63) / 64), dim3(1024), 0, STREAM, stats, rows, C, inv,
                           unbias, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, train,
                           mean, invstd, sc, sh, train ? shift : nullptr, koaf_status_ptr());
    return koaf_check_launch("koaf_bn_finalize");
}

extern "C" int koaf_bn_add_relu(const float* c, const float* sc, const float* sh, const float* idt, const float* idsc,
                                const float* idsh, float* y, int64_t rows, int32_t C, int32_t act16, void* stream) {
    KOAF_REQUIRE(c && sc && sh && y && rows > 0 && C > 0 && C % 4 == 0, "koaf_bn_add_relu: bad args");
    KOAF_REQUIRE(al16(c) && al16(y) && al16(sc) && al16(sh) && (!idt || al16(idt)), "koaf_bn_add_relu: unaligned");
    KOAF_REQUIRE((idsc == nullptr) == (idsh == nullptr), "koaf_bn_add_relu: idsc/idsh come together");
    const int64_t nvec = rows * (C / 4);
    if (act16) hipLaunchKernelGGL(bn_add_relu_kernel<true>, dim3(ew_grid(nvec)), dim3(EB), 0, STREAM, c, sc, sh, idt, idsc, idsh, y,
                                  nvec, C / 4, koaf_status_ptr());
    else hipLaunchKernelGGL(bn_add_relu_kernel<false>, dim3(ew_grid(nvec)), dim3(EB), 0, STREAM, c, sc, sh, idt, idsc, idsh, y,
                            nvec, C / 4, koaf_status_ptr());
    return koaf_check_launch("koaf_bn_add_relu");
}
extern "C" int koaf_bn_relu(const float* c, const float* sc, const float* sh, float* y, int64_t rows, int32_t C,
                            int32_t act16, void* stream) {
    return koaf_bn_add_relu(c, sc, sh, nullptr, nullptr, nullptr, y, rows, C, act16, stream);
}

extern "C" int koaf_colstats(const float* x, int64_t rows, int32_t C, float* part, int32_t* part_rows,
                             const float* shift, int32_t act16, void* stream) {
    ColGeom g;
    KOAF_REQUIRE(x && part && part_rows && rows > 0, "koaf_colstats: bad args");
    KOAF_REQUIRE(col_geom(rows, C, 1024, &g), "koaf_colstats: unsupported C=%d", C);
    if (act16) hipLaunchKernelGGL(colstats_kernel<true>, dim3(g.nblk, g.nchunk), dim3(256), 0, STREAM, x, rows, C, g, part, 1, shift);
    else hipLaunchKernelGGL(colstats_kernel<false>, dim3(g.nblk, g.nchunk), dim3(256), 0, STREAM, x, rows, C, g, part, 1, shift);
    *part_rows = g.nblk;
    return koaf_check_launch("koaf_colstats");
}
extern "C" int32_t koaf_colpart_rows(int64_t rows, int32_t C) {
    ColGeom g;
    if (!col_geom(rows, C, 1024, &g)) return -1;
    return g.nblk;
}

extern "C" int koaf_bn_bwd_reduce(const float* g, const float* c, const float* ymask, const float* sc, const float* sh,
                                  const float* mean, const float* invstd, int32_t mask_mode, float* dz_out, float* part,
                                  int32_t* part_rows, int64_t rows, int32_t C, float* dz_amax, int32_t act16, void* stream) {
    ColGeom geo;
    KOAF_REQUIRE(g && c && mean && invstd && part && part_rows && rows > 0, "koaf_bn_bwd_reduce: bad args");
    KOAF_REQUIRE(mask_mode != 1 || ymask, "koaf_bn_bwd_reduce: mask_mode 1 needs ymask");
    KOAF_REQUIRE(mask_mode != 2 || (sc && sh), "koaf_bn_bwd_reduce: mask_mode 2 needs sc/sh");
    KOAF_REQUIRE(col_geom(rows, C, 1024, &geo), "koaf_bn_bwd_reduce: unsupported C=%d", C);
    if (dz_amax && hipMemsetAsync(dz_amax, 0, sizeof(float), STREAM) != hipSuccess) {
        koaf_set_error("koaf_bn_bwd_reduce: memset failed");
        return KOAF_ELAUNCH;
    }
    const PoolGeom nopool{nullptr, nullptr, 0, 0, 0, 0};
    if (act16) hipLaunchKernelGGL((bn_bwd_reduce_kernel<true, false>), dim3(geo.nblk, geo.nchunk), dim3(256), 0, STREAM, g, c, ymask, sc, sh,
                                  mean, invstd, mask_mode, dz_out, rows, C, geo, part, dz_amax, nopool);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<false, false>), dim3(geo.nblk, geo.nchunk), dim3(256), 0, STREAM, g, c, ymask, sc, sh,
                            mean, invstd, mask_mode, dz_out, rows, C, geo, part, dz_amax, nopool);
    *part_rows = geo.nblk;
    return koaf_check_launch("koaf_bn_bwd_reduce");
}

extern "C" int koaf_bn_bwd_reduce_pool(const float* pool_g, const uint8_t* pool_argmax, const float* c, const float* sc,
                                       const float* sh, const float* mean, const float* invstd, float* dz_out, float* part,
                                       int32_t* part_rows, int32_t N, int32_t H, int32_t W, int32_t C, float* dz_amax,
                                       int32_t act16, void* stream) {
    ColGeom geo;
    KOAF_REQUIRE(pool_g && pool_argmax && c && sc && sh && mean && invstd && dz_out && part && part_rows && N > 0 && H > 0 && W > 0,
                 "koaf_bn_bwd_reduce_pool: bad args");
    const int64_t rows = (int64_t)N * H * W;
    KOAF_REQUIRE(col_geom(rows, C, 1024, &geo), "koaf_bn_bwd_reduce_pool: unsupported C=%d", C);
    if (dz_amax && hipMemsetAsync(dz_amax, 0, sizeof(float), STREAM) != hipSuccess) {
        koaf_set_error("koaf_bn_bwd_reduce_pool: memset failed");
        return KOAF_ELAUNCH;
    }
    const PoolGeom pg{pool_g, pool_argmax, H, W, (H + 2 - 3) / 2 + 1, (W + 2 - 3) / 2 + 1};
    if (act16) hipLaunchKernelGGL((bn_bwd_reduce_kernel<true, true>), dim3(geo.nblk, geo.nchunk), dim3(256), 0, STREAM, nullptr, c, nullptr,
                                  sc, sh, mean, invstd, 2, dz_out, rows, C, geo, part, dz_amax, pg);
    else hipLaunchKernelGGL((bn_bwd_reduce_kernel<false, true>), dim3(geo.nblk, geo.nchunk), dim3(256), 0, STREAM, nullptr, c, nullptr,
                            sc, sh, mean, invstd, 2, dz_out, rows, C, geo, part, dz_amax, pg);
    *part_rows = geo.nblk;
    return koaf_check_launch("koaf_bn_bwd_reduce_pool");
}
extern "C" int koaf_bn_bwd_finalize(const float* part, int32_t part_rows, int32_t C, int64_t count, const float* sc,
                                    const float* invstd, float* dgamma, float* dbeta, float* coef, int32_t nsum,
                                    int32_t i1, double* ws, const float* mean, const float* dz_amax, float* amax,
                                    void* stream) {
    KOAF_REQUIRE(part && part_rows > 0 && C > 0 && count > 0 && sc && invstd && coef, "koaf_bn_bwd_finalize: bad args");
    KOAF_REQUIRE(nsum >= 2 && i1 >= 1 && i1 < nsum, "koaf_bn_bwd_finalize: bad (nsum, i1)");
    KOAF_REQUIRE(!amax || mean, "koaf_bn_bwd_finalize: amax needs mean (coef gets its fourth row)");
    if (amax && hipMemsetAsync(amax, 0, sizeof(float), STREAM) != hipSuccess) {
        koaf_set_error("koaf_bn_bwd_finalize: memset failed");
        return KOAF_ELAUNCH;
    }
    const double sq = sqrt((double)(count > 1 ? count - 1 : 1));
    const int S = part_reduce(part, part_rows, C, nsum, i1, ws, STREAM);
    if (S)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<double>, dim3((C + 63) / 64), dim3(1024), 0, STREAM, ws, S, C,
                           1.0 / (double)count, sc, invstd, dgamma, dbeta, coef, 2, 1, mean, dz_amax, sq, amax);
    else
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<float>, dim3((C + 63) / 64), dim3(1024), 0, STREAM, part, part_rows, C,
                           1.0 / (double)count, sc, invstd, dgamma, dbeta, coef, nsum, i1, mean, dz_amax, sq, amax);
    return koaf_check_launch("koaf_bn_bwd_finalize");
}
extern "C" int koaf_bn_bwd_apply(const float* dz, const float* c, const float* mean, const float* coef, float* dc,
                                 int64_t rows, int32_t C, float* amax, int32_t act16, void* stream) {
    KOAF_REQUIRE(dz && c && mean && coef && dc && rows > 0 && C % 4 == 0, "koaf_bn_bwd_apply: bad args");
    const int64_t nvec = rows * (C / 4);
    if (act16) hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(ew_grid(nvec)), dim3(EB), 0, STREAM, dz, c, mean, coef, dc, nvec, C, amax);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(ew_grid(nvec)), dim3(EB), 0, STREAM, dz, c, mean, coef, dc, nvec, C, amax);
    return koaf_check_launch("koaf_bn_bwd_apply");
}

extern "C" int koaf_maxpool_fwd(const float* c, const float* sc, const float* sh, float* y, uint8_t* argmax, int32_t N,
                                int32_t H, int32_t W, int32_t C, int32_t act16, void* stream) {
    KOAF_REQUIRE(c && sc && sh && y && argmax && N > 0 && H > 0 && W > 0 && C % 4 == 0, "koaf_maxpool_fwd: bad args");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const int64_t nvec = (int64_t)N * OH * OW * (C / 4);
    if (act16) hipLaunchKernelGGL(maxpool_fwd_kernel<true>, dim3(ew_grid(nvec)), dim3(EB), 0, STREAM, c, sc, sh, y, argmax, N, H, W, C,
                                  OH, OW, koaf_status_ptr());
    else hipLaunchKernelGGL(maxpool_fwd_kernel<false>, dim3(ew_grid(nvec)), dim3(EB), 0, STREAM, c, sc, sh, y, argmax, N, H, W, C,
                            OH, OW, koaf_status_ptr());
    return koaf_check_launch("koaf_maxpool_fwd");
}
extern "C" int koaf_maxpool_bwd(const float* dy, const uint8_t* argmax, float* da, int32_t N, int32_t H, int32_t W,
                                int32_t C, void* stream) {
    KOAF_REQUIRE(dy && argmax && da && N > 0 && C % 4 == 0, "koaf_maxpool_bwd: bad args");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const int64_t nvec = (int64_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(nvec)), dim3(EB), 0, STREAM, dy, argmax, da, N, H, W, C, OH,
                       OW);
    return koaf_check_launch("koaf_maxpool_bwd");
}
extern "C" int koaf_gap_fwd(const float* y, float* out, int32_t N, int32_t HW, int32_t C, int32_t act16, void* stream) {
    KOAF_REQUIRE(y && out && N > 0 && HW > 0 && C % 4 == 0, "koaf_gap_fwd: bad args");
    if (act16) hipLaunchKernelGGL(gap_fwd_kernel<true>, dim3(ew_grid((int64_t)N * C / 4)), dim3(EB), 0, STREAM, y, out, N, HW, C);
    else hipLaunchKernelGGL(gap_fwd_kernel<false>, dim3(ew_grid((int64_t)N * C / 4)), dim3(EB), 0, STREAM, y, out, N, HW, C);
    return koaf_check_launch("koaf_gap_fwd");
}
extern "C" int koaf_gap_bwd(const float* dout, float* dy, int32_t N, int32_t HW, int32_t C, void* stream) {
    KOAF_REQUIRE(dout && dy && N > 0 && HW > 0 && C % 4 == 0, "koaf_gap_bwd: bad args");
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(ew_grid((int64_t)N * HW * C / 4)), dim3(EB), 0, STREAM, dout, dy, N, HW,
                       C);
    return koaf_check_launch("koaf_gap_bwd");
}

extern "C" int koaf_slice_fold(const float* x, float* out, int32_t B, int32_t R, int32_t Cc, int32_t S, void* stream) {
    KOAF_REQUIRE(x && out && B > 0 && R > 0 && Cc > 0 && S > 0 && B <= 65535, "koaf_slice_fold: bad args");
    const int P = R * Cc;
    dim3 grid((P + 31) / 32, (S + 31) / 32, B);
    hipLaunchKernelGGL(slice_fold_kernel, grid, dim3(256), 0, STREAM, x, out, P, S);
    return koaf_check_launch("koaf_slice_fold");
}
extern "C" int koaf_downscale2(const float* x, float* out, int32_t B, int32_t R, int32_t Cc, int32_t S, int32_t fs,
                               void* stream) {
    KOAF_REQUIRE(x && out && B > 0 && R % 2 == 0 && Cc % 2 == 0 && (fs == 1 || fs == 2) && S % fs == 0,
                 "koaf_downscale2: needs even R,C (and S if fs==2)");
    const int64_t total = (int64_t)B * (R / 2) * (Cc / 2) * (S / fs);
    hipLaunchKernelGGL(downscale2_kernel, dim3(ew_grid(total)), dim3(EB), 0, STREAM, x, out, B, R, Cc, S, fs);
    return koaf_check_launch("koaf_downscale2");
}

extern "C" int koaf_resize(const float* x, float* out, int64_t BC, int32_t ndim, const int32_t* in_size, const int32_t* out_size,
                           void* stream) {
    KOAF_REQUIRE(x && out && BC > 0 && ndim >= 1 && ndim <= 3 && in_size && out_size, "koaf_resize: bad args (1..3 spatial dims)");
    ResizeGeom g;
    for (int d = 0; d < 3; ++d) {
        const int k = d - (3 - ndim);         // leading absent dimensions have size 1
        g.I[d] = k >= 0 ? in_size[k] : 1;
        g.O[d] = k >= 0 ? out_size[k] : 1;
        KOAF_REQUIRE(g.I[d] > 0 && g.O[d] > 0, "koaf_resize: empty dimension");
        g.rs[d] = (float)g.I[d] / (float)g.O[d];
    }
    const int64_t total = BC * g.O[0] * g.O[1] * g.O[2];
    hipLaunchKernelGGL(resize_kernel, dim3(ew_grid(total)), dim3(EB), 0, STREAM, x, out, BC, g);
    return koaf_check_launch("koaf_resize");
}

extern "C" int64_t koaf_minmax_ws(int64_t n) {
    int64_t nb = cdiv64(n, 256 * 16);
    return (nb < 1 ? 1 : (nb > 256 ? 256 : nb)) * 2;      // floats per sample
}
extern "C" int koaf_minmax(const float* x, int32_t B, int64_t n, float* mm, float* ws, void* stream) {
    KOAF_REQUIRE(x && mm && ws && B > 0 && n > 0, "koaf_minmax: bad args");
    const int nblk = (int)(koaf_minmax_ws(n) / 2);
    hipLaunchKernelGGL(minmax_part_kernel, dim3(nblk, B), dim3(256), 0, STREAM, x, n, nblk, ws);
    hipLaunchKernelGGL(minmax_final_kernel, dim3(B), dim3(64), 0, STREAM, ws, nblk, mm);
    return koaf_check_launch("koaf_minmax");
}
extern "C" int koaf_augment(const float* x, float* y, const float* mm, const float* params, int32_t B, int32_t R,
                            int32_t C, int32_t S, float mean, float stdv, void* stream) {
    KOAF_REQUIRE(x && y && mm && params && B > 0 && R > 0 && C > 0 && S > 0, "koaf_augment: bad args");
    KOAF_REQUIRE((int64_t)R * C * S < (1ll << 31), "koaf_augment: sample too large");
    const bool v4 = (S % 4 == 0) && al16(x) && al16(y);
    const int64_t total = (int64_t)B * R * C * (v4 ? S / 4 : S);
    if (v4)
        hipLaunchKernelGGL(augment_kernel<4>, dim3(ew_grid(total)), dim3(EB), 0, STREAM, x, y, mm, params, B, R, C, S, mean, stdv);
    else
        hipLaunchKernelGGL(augment_kernel<1>, dim3(ew_grid(total)), dim3(EB), 0, STREAM, x, y, mm, params, B, R, C, S, mean, stdv);
    return koaf_check_launch("koaf_augment");
}

extern "C" int koaf_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                  float* rstd, int32_t rows, int32_t D, float eps, void* stream) {
    KOAF_REQUIRE(x && gamma && beta && y && mean && rstd && rows > 0 && D % 4 == 0, "koaf_layernorm_fwd: bad args");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, STREAM, x, gamma, beta, y, mean, rstd,
                       rows, D, eps);
    return koaf_check_launch("koaf_layernorm_fwd");
}
extern "C" int64_t koaf_layernorm_bwd_ws(int32_t rows, int32_t D) {
    ColGeom g;
    if (!col_geom(rows, D, 256, &g)) return -1;
    return (int64_t)g.nblk * 2 * D;
}
extern "C" int koaf_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                                  const float* rstd, float* dx, float* dgamma, float* dbeta, float* part, int32_t rows,
                                  int32_t D, void* stream) {
    ColGeom g;
    KOAF_REQUIRE(dy && x && gamma && mean && rstd && dx && dgamma && dbeta && part && rows > 0,
                 "koaf_layernorm_bwd: bad args");
    KOAF_REQUIRE(col_geom(rows, D, 256, &g), "koaf_layernorm_bwd: unsupported D=%d", D);
    hipLaunchKernelGGL(layernorm_bwd_dx_kernel, dim3((rows + 3) / 4), dim3(256), 0, STREAM, dy, x, gamma, mean, rstd,
                       dx, rows, D);
    hipLaunchKernelGGL(layernorm_bwd_param_kernel, dim3(g.nblk, g.nchunk), dim3(256), 0, STREAM, dy, x, mean, rstd,
                       (int64_t)rows, D, g, part);
    hipLaunchKernelGGL(colfinal_kernel<2>, dim3((D + 63) / 64), dim3(1024), 0, STREAM, part, g.nblk, D, dgamma, dbeta);
    return koaf_check_launch("koaf_layernorm_bwd");
}

extern "C" int koaf_softmax_rows(float* x, int64_t rows, int32_t n, void* stream) {
    KOAF_REQUIRE(x && rows > 0 && n > 0, "koaf_softmax_rows: bad args");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, STREAM, x, rows, n);
    return koaf_check_launch("koaf_softmax_rows");
}
extern "C" int koaf_softmax_bwd_rows(float* dp, const float* p, int64_t rows, int32_t n, float scale, void* stream) {
    KOAF_REQUIRE(dp && p && rows > 0 && n > 0, "koaf_softmax_bwd_rows: bad args");
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, STREAM, dp, p, rows, n,
                       scale);
    return koaf_check_launch("koaf_softmax_bwd_rows");
}

#define PW_LAUNCH(OP, a, b, out, n, name)                                                                      \
    KOAF_REQUIRE((a) && (out) && (n) > 0, name ": bad args");                                                  \
    KOAF_REQUIRE(al16(a) && al16(out) && (!(b) || al16(b)), name ": unaligned");                               \
    hipLaunchKernelGGL(pointwise_kernel<OP>, dim3(ew_grid(((n) + 3) / 4)), dim3(EB), 0, STREAM, a, b, out, n); \
    return koaf_check_launch(name)

extern "C" int koaf_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
    PW_LAUNCH(PW_GELU_F, x, (const float*)nullptr, y, n, "koaf_gelu_fwd");
}
extern "C" int koaf_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
    KOAF_REQUIRE(x, "koaf_gelu_bwd: bad args");
    PW_LAUNCH(PW_GELU_B, dy, x, dx, n, "koaf_gelu_bwd");
}
extern "C" int koaf_relu_fwd(const float* x, float* y, int64_t n, void* stream) {
    PW_LAUNCH(PW_RELU_F, x, (const float*)nullptr, y, n, "koaf_relu_fwd");
}
extern "C" int koaf_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    KOAF_REQUIRE(y, "koaf_relu_bwd: bad args");
    PW_LAUNCH(PW_RELU_B, dy, y, dx, n, "koaf_relu_bwd");
}
extern "C" int koaf_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    KOAF_REQUIRE(b, "koaf_add: bad args");
    PW_LAUNCH(PW_ADD, a, b, out, n, "koaf_add");
}
extern "C" int koaf_widen(const void* x, int32_t dtype, float* y, int64_t n, void* stream) {
    KOAF_REQUIRE(x && y && n > 0 && dtype >= 1 && dtype <= 3, "koaf_widen: bad args (dtype 1 = uint8, 2 = uint16, 3 = int16)");
    KOAF_REQUIRE(al16(y) && (((uintptr_t)x) & 7) == 0, "koaf_widen: unaligned");
    const unsigned grid = ew_grid(n / 4 + 1);
    if (dtype == 1) hipLaunchKernelGGL(widen_kernel<uint8_t>, dim3(grid), dim3(EB), 0, STREAM, (const uint8_t*)x, y, n);
    else if (dtype == 2) hipLaunchKernelGGL(widen_kernel<uint16_t>, dim3(grid), dim3(EB), 0, STREAM, (const uint16_t*)x, y, n);
    else hipLaunchKernelGGL(widen_kernel<int16_t>, dim3(grid), dim3(EB), 0, STREAM, (const int16_t*)x, y, n);
    return koaf_check_launch("koaf_widen");
}
extern "C" int koaf_counter_add(int64_t* counter, int64_t delta, void* stream) {
    KOAF_REQUIRE(counter, "koaf_counter_add: null counter");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, STREAM, counter, delta);
    return koaf_check_launch("koaf_counter_add");
}
extern "C" int koaf_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, const int64_t* epoch, void* stream) {
    KOAF_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f, "koaf_dropout: bad args");
    hipLaunchKernelGGL(dropout_kernel, dim3(ew_grid(n)), dim3(EB), 0, STREAM, x, y, n, p, 1.f / (1.f - p), seed, epoch);
    return koaf_check_launch("koaf_dropout");
}
extern "C" int koaf_dropout2d(const float* x, float* y, int32_t N, int32_t HW, int32_t C, float p, uint64_t seed,
                              const int64_t* epoch, void* stream) {
    KOAF_REQUIRE(x && y && N > 0 && HW > 0 && C > 0 && p >= 0.f && p < 1.f, "koaf_dropout2d: bad args");
    const int64_t n = (int64_t)N * HW * C;
    hipLaunchKernelGGL(dropout2d_kernel, dim3(ew_grid(n)), dim3(EB), 0, STREAM, x, y, n, (int64_t)HW * C, C, p,
                       1.f / (1.f - p), seed, epoch);
    return koaf_check_launch("koaf_dropout2d");
}
extern "C" int koaf_fill(float* p, float value, int64_t n, void* stream) {
    KOAF_REQUIRE(p && n > 0, "koaf_fill: bad args");
    hipLaunchKernelGGL(fill_kernel, dim3(ew_grid(n)), dim3(EB), 0, STREAM, p, value, n);
    return koaf_check_launch("koaf_fill");
}

extern "C" int koaf_colsum(const float* x, float* out, int32_t rows, int32_t C, float* part, void* stream) {
    KOAF_REQUIRE(x && out && rows > 0 && C > 0, "koaf_colsum: bad args");
    ColGeom g;
    if (part && al16(x) && col_geom(rows, C, 256, &g)) {
        hipLaunchKernelGGL(colstats_kernel<false>, dim3(g.nblk, g.nchunk), dim3(256), 0, STREAM, x, (int64_t)rows, C, g, part,
                           0, (const float*)nullptr);
        hipLaunchKernelGGL(colfinal_kernel<1>, dim3((C + 63) / 64), dim3(1024), 0, STREAM, part, g.nblk, C, out,
                           (float*)nullptr);
    } else {
        hipLaunchKernelGGL(colsum_small_kernel, dim3((C + 255) / 256), dim3(256), 0, STREAM, x, rows, C, out);
    }
    return koaf_check_launch("koaf_colsum");
}
extern "C" int64_t koaf_colsum_ws(int32_t rows, int32_t C) {
    ColGeom g;
    if (!col_geom(rows, C, 256, &g)) return 0;
    return (int64_t)g.nblk * C;
}

// elements one block of focal_loss_kernel covers in a few passes; beyond it the grid form runs (when the caller brought a workspace)
static const int64_t LOSS_ONE_BLOCK = 8192;
static const int64_t LOSS_CHUNK = 4096;        // elements per block of the grid form
extern "C" int64_t koaf_loss_ws(int32_t B, int64_t S) {
    const int64_t n = (int64_t)B * S;
    return n > LOSS_ONE_BLOCK ? 2 * ((n + LOSS_CHUNK - 1) / LOSS_CHUNK) + 4 : 0;
}
static int loss_grid(const float* logits, const int64_t* target, const float* cw, float* loss, float* dlogits, int B, int C, int64_t S,
                     float gamma, int mean, int focal, float* ws, hipStream_t st) {
    const int64_t n = (int64_t)B * S;
    const int nblk = (int)((n + LOSS_CHUNK - 1) / LOSS_CHUNK);
    float* wl = ws;                 // [nblk] loss partials
    float* wd = ws + nblk;          // [nblk] denominator partials, then the denominator itself at wd[nblk]
    const float* wden = nullptr;
    if (!focal) {                   // cross-entropy: the weighted mean divides by the weights of the elements that count
        hipLaunchKernelGGL((loss_grid_kernel<0>), dim3(nblk), dim3(256), 0, st, logits, target, cw, dlogits, wd, (const float*)nullptr, B, C, S,
                           LOSS_CHUNK, gamma, mean, focal, koaf_status_ptr());
        hipLaunchKernelGGL(loss_grid_sum_kernel, dim3(1), dim3(256), 0, st, wd, nblk, wd + nblk, (const float*)nullptr, 0.f);
        wden = wd + nblk;
    }
    hipLaunchKernelGGL((loss_grid_kernel<1>), dim3(nblk), dim3(256), 0, st, logits, target, cw, dlogits, wl, wden, B, C, S, LOSS_CHUNK, gamma,
                       mean, focal, koaf_status_ptr());
    hipLaunchKernelGGL(loss_grid_sum_kernel, dim3(1), dim3(256), 0, st, wl, nblk, loss, wden, (mean && !wden) ? (float)n : 0.f);
    return koaf_check_launch("koaf_loss (grid form)");
}
extern "C" int koaf_focal_loss(const float* logits, const int64_t* target, const float* class_weight, float* loss,
                               float* dlogits, int32_t B, int32_t C, int64_t S, float gamma, int32_t reduction_mean,
                               float* ws, void* stream) {
    KOAF_REQUIRE(logits && target && loss && dlogits && B > 0 && C > 0 && S > 0, "koaf_focal_loss: bad args");
    if (ws && (int64_t)B * S > LOSS_ONE_BLOCK)
        return loss_grid(logits, target, class_weight, loss, dlogits, B, C, S, gamma, reduction_mean, 1, ws, STREAM);
    hipLaunchKernelGGL(focal_loss_kernel, dim3(1), dim3(256), 0, STREAM, logits, target, class_weight, loss, dlogits, B, C, S,
                       gamma, reduction_mean, 1, koaf_status_ptr());
    return koaf_check_launch("koaf_focal_loss");
}
extern "C" int koaf_ce_loss(const float* logits, const int64_t* target, const float* class_weight, float* loss,
                            float* dlogits, int32_t B, int32_t C, int64_t S, float* ws, void* stream) {
    KOAF_REQUIRE(logits && target && loss && dlogits && B > 0 && C > 0 && S > 0, "koaf_ce_loss: bad args");
    if (ws && (int64_t)B * S > LOSS_ONE_BLOCK)
        return loss_grid(logits, target, class_weight, loss, dlogits, B, C, S, 0.f, 1, 0, ws, STREAM);
    hipLaunchKernelGGL(focal_loss_kernel, dim3(1), dim3(256), 0, STREAM, logits, target, class_weight, loss, dlogits, B, C, S,
                       0.f, 1, 0, koaf_status_ptr());
    return koaf_check_launch("koaf_ce_loss");
}

extern "C" int koaf_adam_hyper(int32_t* step, const float* lr, double beta1, double beta2, float* hyper, void* stream) {
    KOAF_REQUIRE(step && lr && hyper, "koaf_adam_hyper: bad args");
    hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(1), 0, STREAM, step, lr, beta1, beta2, hyper);
    return koaf_check_launch("koaf_adam_hyper");
}
extern "C" int koaf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, double beta1,
                              double beta2, float eps, float weight_decay, int32_t step, int32_t adamw,
                              const float* hyper, float* vmax, void* stream) {
    if (hyper) step = 1;      // (lr / step come from the device; the host values are ignored)
    KOAF_REQUIRE(p && g && m && v && n > 0 && step >= 1, "koaf_adam_step: bad args");
    KOAF_REQUIRE(al16(p) && al16(g) && al16(m) && al16(v) && (!vmax || al16(vmax)), "koaf_adam_step: unaligned");
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4 + 1)), dim3(EB), 0, STREAM, p, g, m, v, n, lr, (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), eps, weight_decay, step_size, bc2_sqrt, adamw, hyper, vmax);
    return koaf_check_launch("koaf_adam_step");
}
