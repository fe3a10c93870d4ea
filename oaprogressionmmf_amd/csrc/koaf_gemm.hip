// koaf_gemm.hip -- the one MFMA GEMM under every dense contraction of the koafusion train step:
// implicit-GEMM conv forward / dgrad / wgrad (NHWC), nn.Linear forward / dgrad / wgrad and the
// attention contractions.  fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32 (exact fp32 fma
// chain, 157 TFLOP/s dense peak on MI355X).
//
// Block = 256 threads = 4 waves (2x2), block tile BM x BN x 32, wave tile (BM/2) x (BN/2) built from
// 32x32 MFMA tiles.  Operand tiles are staged global -> registers -> LDS (single LDS buffer, the
// next tile's global loads are in flight under the current tile's MFMAs; an fp32 MFMA k-step is
// 4096 cycles per wave at 128x128, so HBM latency is covered with one block per SIMD set).
//   K-contiguous operand ("KC"): LDS image [row][32+4] -- fragment = one ds_read_b128 per 4 MFMAs,
//       conflict-free (row stride 36 dwords: 16 consecutive rows hit 16 distinct 4-bank slots).
//   K-major operand ("KM"):      LDS image [k][rows]   -- ds_write_b128 rows, ds_read_b32 fragments
//       (consecutive lanes, consecutive dwords).
// Both images present the same k order to the MFMA: lane (r, h), step j of k-group g uses
// k = 8g + 4h + j for A and for B, so any pairing of KC/KM operands works.
#include "koaf_common.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = BK + 4;

template <int ROWS, int KIND, bool VEC>
struct TileLoader {
    static constexpr int NU = ROWS / 32;
    v4f r[NU];
    unsigned vm;       // validity bits of the tile in flight: VEC 1 bit / unit, else 4 bits / unit
    v4f ts4, th4;      // transform coefficients of the tile in flight (KC) / of this thread's columns (KM)
    // KC state (ext-vector values, not arrays: arrays of per-unit state were left in scratch by hipcc and
    // every scratch reload drained the in-flight global loads through the in-order vmcnt)
    v4l base;          // element offset of each unit's row / image from the operand pointer
    v4i iy0, ix0;
    unsigned rvm;      // row-valid bits
    // KM state
    int col, cc, kh_, kw_;
    unsigned cvm;      // column-valid bits

    __device__ __forceinline__ void init(const KoafOperand& op, const float* ptr, int r0, int R, int z1) {
        const int t = threadIdx.x;
        vm = 0;
        ts4 = th4 = (v4f){0.f, 0.f, 0.f, 0.f};
        if constexpr (KIND == 0) {
            rvm = 0;
            base = (v4l){0, 0, 0, 0};
            iy0 = ix0 = (v4i){0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                int row = r0 + (t >> 3) + 32 * i;
                rvm |= (row < R ? 1u : 0u) << i;
                if (op.gather == 0) {
                    base[i] = (int64_t)row * op.ld;
                } else {
                    int ppi = op.PH * op.PW;
                    int n = row / ppi;
                    int rem = row - n * ppi;
                    int py = rem / op.PW;
                    int px = rem - py * op.PW;
                    base[i] = (int64_t)n * op.H * op.W * op.CS;
                    if (op.gather == 1) {
                        iy0[i] = py * op.stride - op.pad;
                        ix0[i] = px * op.stride - op.pad;
                    } else {
                        iy0[i] = py + op.pad;
                        ix0[i] = px + op.pad;
                    }
                }
            }
        } else {
            constexpr int CV = ROWS / 4;
            col = r0 + 4 * (t % CV);
            cvm = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) cvm |= ((col + j) < R ? 1u : 0u) << j;
            cc = col;
            kh_ = kw_ = 0;
            if (op.gather == 1) {
                int tap = r0 / op.C;
                cc = col - tap * op.C;
                kh_ = tap / op.KW;
                kw_ = tap - kh_ * op.KW;
            }
            if (op.tf) {
                const float* sc = op.sc + z1 * op.tf_bs;
                const float* sh = op.sh + z1 * op.tf_bs;
                if (VEC) {
                    if (cvm & 1u) { ts4 = *(const v4f*)(sc + cc); th4 = *(const v4f*)(sh + cc); }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((cvm >> j) & 1u) { ts4[j] = sc[cc + j]; th4[j] = sh[cc + j]; }
                }
            }
        }
    }

    // Issue the global loads of the k-tile [k0, k0+32) -- nothing here consumes a loaded value, so the
    // s_waitcnt lands in finish(), i.e. after the MFMAs of the tile currently in LDS.
    __device__ __forceinline__ void issue(const KoafOperand& op, const float* ptr, int k0, int kend, int z1) {
        const int t = threadIdx.x;
        vm = 0;
        if constexpr (KIND == 0) {
            const int kv = t & 7;
            const int kk = k0 + 4 * kv;
            int ch = kk;
            int kh = 0, kw = 0;
            if (op.gather != 0) {
                // conv gather: a 32-wide k chunk lies inside one filter tap (C % 32 == 0)
                const int tap = k0 / op.C;
                const int c0 = k0 - tap * op.C;
                kh = tap / op.KW;
                kw = tap - kh * op.KW;
                ch = c0 + 4 * kv;
            }
            if (op.tf) {
                const float* sc = op.sc + z1 * op.tf_bs;
                const float* sh = op.sh + z1 * op.tf_bs;
                if (VEC) {
                    const int c = (kk < kend) ? ch : 0;
                    ts4 = *(const v4f*)(sc + c);
                    th4 = *(const v4f*)(sh + c);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = (kk + j < kend) ? ch + j : 0;
                        ts4[j] = sc[c];
                        th4[j] = sh[c];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                if (op.gather == 0) {
                    if (VEC) {
                        const bool ok = ((rvm >> i) & 1u) && kk < kend;
                        r[i] = *(const v4f*)(ok ? ptr + base[i] + kk : ptr);
                        vm |= (ok ? 1u : 0u) << i;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool ok = ((rvm >> i) & 1u) && (kk + j) < kend;
                            r[i][j] = *(ok ? ptr + base[i] + kk + j : ptr);
                            vm |= (ok ? 1u : 0u) << (4 * i + j);
                        }
                    }
                } else {
                    int sy, sx;
                    bool ok = ((rvm >> i) & 1u) && kk < kend;
                    if (op.gather == 1) {
                        sy = iy0[i] + kh;
                        sx = ix0[i] + kw;
                    } else {
                        int ny = iy0[i] - kh, nx = ix0[i] - kw;
                        ok = ok && ny >= 0 && nx >= 0;
                        sy = ny / op.stride;
                        sx = nx / op.stride;
                        ok = ok && (sy * op.stride == ny) && (sx * op.stride == nx);
                    }
                    ok = ok && (unsigned)sy < (unsigned)op.H && (unsigned)sx < (unsigned)op.W;
                    r[i] = *(const v4f*)(ok ? ptr + base[i] + ((int64_t)(sy * op.W + sx) * op.CS + ch) : ptr);
                    vm |= (ok ? 1u : 0u) << i;
                }
            }
        } else {
            constexpr int CV = ROWS / 4;
            constexpr int RP = 256 / CV;
            const int kr0 = t / CV;
            int tap3 = 0, c03 = 0;
            if (op.gather == 3) { tap3 = k0 / op.C; c03 = k0 - tap3 * op.C; }
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int k = k0 + kr0 + RP * i;
                bool ok = k < kend;
                const float* src;
                if (op.gather == 0) {
                    src = ptr + (int64_t)k * op.ld + col;
                } else if (op.gather == 3) {
                    // tapped weights: k = (tap, ck); element at ck*ld + tap*tap_stride + col
                    src = ptr + (int64_t)(c03 + kr0 + RP * i) * op.ld + (int64_t)tap3 * op.tap_stride + col;
                } else {
                    int ppi = op.PH * op.PW;
                    int n = k / ppi;
                    int rem = k - n * ppi;
                    int py = rem / op.PW;
                    int px = rem - py * op.PW;
                    int sy = py * op.stride - op.pad + kh_;
                    int sx = px * op.stride - op.pad + kw_;
                    ok = ok && (unsigned)sy < (unsigned)op.H && (unsigned)sx < (unsigned)op.W;
                    src = ptr + ((int64_t)(n * op.H + sy) * op.W + sx) * op.CS + cc;
                }
                if (VEC) {
                    ok = ok && (cvm & 1u);
                    r[i] = *(const v4f*)(ok ? src : ptr);
                    vm |= (ok ? 1u : 0u) << i;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool okj = ok && ((cvm >> j) & 1u);
                        r[i][j] = *(okj ? src + j : ptr);
                        vm |= (okj ? 1u : 0u) << (4 * i + j);
                    }
                }
            }
        }
    }

    // transform + zero-fill of the tile issued by issue(); first consumer of the loaded registers
    __device__ __forceinline__ void finish(const KoafOperand& op) {
#pragma unroll
        for (int i = 0; i < NU; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = VEC ? ((vm >> i) & 1u) : ((vm >> (4 * i + j)) & 1u);
                float x = r[i][j];
                if (op.tf) x = fmaxf(x * ts4[j] + th4[j], 0.f);
                r[i][j] = ok ? x : 0.f;
            }
        }
    }

    __device__ __forceinline__ void store(float* S) const {
        const int t = threadIdx.x;
        if constexpr (KIND == 0) {
            const int kv = t & 7;
#pragma unroll
            for (int i = 0; i < NU; ++i) *(v4f*)&S[((t >> 3) + 32 * i) * LDK + 4 * kv] = r[i];
        } else {
            constexpr int CV = ROWS / 4;
            constexpr int RP = 256 / CV;
#pragma unroll
            for (int i = 0; i < NU; ++i) *(v4f*)&S[(t / CV + RP * i) * ROWS + 4 * (t % CV)] = r[i];
        }
    }
};

template <int BM, int BN, int AK, int BKD, bool VEC>
__global__ void __launch_bounds__(256) koaf_gemm_kernel(const KoafGemm p) {
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    constexpr int A_ELEMS = (AK == 0) ? BM * LDK : BK * BM;
    constexpr int B_ELEMS = (BKD == 0) ? BN * LDK : BK * BN;
    __shared__ __attribute__((aligned(16))) float smem[A_ELEMS + B_ELEMS];
    float* As = smem;
    float* Bs = smem + A_ELEMS;

    const int ntn = (p.N + BN - 1) / BN;
    const int tn = blockIdx.x % ntn;
    const int tm = blockIdx.x / ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int z0 = blockIdx.z / p.nb1, z1 = blockIdx.z - z0 * p.nb1;
    const int split = blockIdx.y;
    const int kchunk = (((p.K + p.splitk - 1) / p.splitk + BK - 1) / BK) * BK;
    const int kbeg = split * kchunk;
    const int kend = min(p.K, kbeg + kchunk);

    const float* Ap = p.A.ptr + z0 * p.A.bs0 + z1 * p.A.bs1;
    const float* Bp = p.B.ptr + z0 * p.B.bs0 + z1 * p.B.bs1;

    TileLoader<BM, AK, VEC> la;
    TileLoader<BN, BKD, VEC> lb;
    la.init(p.A, Ap, m0, p.M, z1);
    lb.init(p.B, Bp, n0, p.N, z1);

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int t = threadIdx.x;
    const int lane = t & 63, w = t >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int r = lane & 31, h = lane >> 5;

    if (kbeg < kend) {
        la.issue(p.A, Ap, kbeg, kend, z1);
        lb.issue(p.B, Bp, kbeg, kend, z1);
        la.finish(p.A);
        lb.finish(p.B);
        la.store(As);
        lb.store(Bs);
    }
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = (k0 + BK) < kend;
        if (more) {
            la.issue(p.A, Ap, k0 + BK, kend, z1);
            lb.issue(p.B, Bp, k0 + BK, kend, z1);
        }
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            v4f a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if constexpr (AK == 0) {
                    a[i] = *(const v4f*)&As[(wm * WM + 32 * i + r) * LDK + 8 * kg + 4 * h];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[i][j] = As[(8 * kg + 4 * h + j) * BM + wm * WM + 32 * i + r];
                }
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                if constexpr (BKD == 0) {
                    b[i] = *(const v4f*)&Bs[(wn * WN + 32 * i + r) * LDK + 8 * kg + 4 * h];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[i][j] = Bs[(8 * kg + 4 * h + j) * BN + wn * WN + 32 * i + r];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jn = 0; jn < TN; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], b[jn][j], acc[i][jn], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            la.finish(p.A);
            lb.finish(p.B);
            la.store(As);
            lb.store(Bs);
            __syncthreads();
        }
    }

    // ---- epilogue ----
    float* Cp;
    int64_t ldc;
    const bool slab = p.splitk > 1;
    if (slab) {
        Cp = p.C + (int64_t)(blockIdx.z * p.splitk + split) * p.M * p.N;
        ldc = p.N;
    } else {
        Cp = p.C + z0 * p.cbs0 + z1 * p.cbs1;
        ldc = p.ldc;
    }
    const float* Rp = (p.residual && !slab) ? p.residual + z0 * p.rbs0 + z1 * p.rbs1 : nullptr;
    const float* bias = slab ? nullptr : p.bias;
    const bool do_stats = (p.stats != nullptr) && !slab;
    float s1[TN], s2[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) s1[jn] = s2[jn] = 0.f;

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
                float v = p.alpha * acc[i][jn][e];
                s1[jn] += v;
                s2[jn] += v * v;
                if (cok && row < p.M) {
                    v += bv;
                    if (Rp) v += Rp[(int64_t)row * p.ldr + col];
                    Cp[(int64_t)row * ldc + col] = v;
                }
            }
        }
    }
    if (do_stats) {
        // column sums over this block's BM rows: lanes (r,0)+(r,1), then the two M-waves via LDS
        float* red = smem;  // [2 wm][2][BN]
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
            float a1 = s1[jn] + __shfl_xor(s1[jn], 32, 64);
            float a2 = s2[jn] + __shfl_xor(s2[jn], 32, 64);
            if (h == 0) {
                red[(wm * 2 + 0) * BN + wn * WN + 32 * jn + r] = a1;
                red[(wm * 2 + 1) * BN + wn * WN + 32 * jn + r] = a2;
            }
        }
        __syncthreads();
        if (t < BN && (n0 + t) < p.N) {
            float* st = p.stats + (int64_t)tm * 2 * p.stats_ld + (int64_t)blockIdx.z * p.stats_bs;
            st[n0 + t] = red[0 * BN + t] + red[2 * BN + t];
            st[p.stats_ld + n0 + t] = red[1 * BN + t] + red[3 * BN + t];
        }
    }
}

__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* __restrict__ slabs, int nslab,
                                                          int64_t n, float* __restrict__ out) {
    int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n && (n & 3) == 0) {
        v4f a = *(const v4f*)(slabs + i);
        for (int s = 1; s < nslab; ++s) a += *(const v4f*)(slabs + (int64_t)s * n + i);
        *(v4f*)(out + i) = a;
    } else {
        for (int j = 0; j < 4 && i + j < n; ++j) {
            float a = slabs[i + j];
            for (int s = 1; s < nslab; ++s) a += slabs[(int64_t)s * n + i + j];
            out[i + j] = a;
        }
    }
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

bool operand_vec_ok(const KoafOperand& o, int R, int K) {
    if (!aligned16(o.ptr)) return false;
    if ((o.ld & 3) || (o.bs0 & 3) || (o.bs1 & 3)) return false;
    if (o.kind == 0) {
        if (K & 3) return false;
        if (o.gather && ((o.C & 31) || (o.CS & 3))) return false;
    } else {
        if (R & 3) return false;
        if (o.gather == 3 && ((o.C & 31) || (o.tap_stride & 3))) return false;
        if (o.gather == 1 && ((o.C & 3) || (o.CS & 3))) return false;
    }
    if (o.tf && (!aligned16(o.sc) || !aligned16(o.sh))) return false;
    return true;
}

template <int BM, int BN, bool VEC>
int launch_kinds(const KoafGemm& g, dim3 grid, hipStream_t s) {
    const int ak = g.A.kind, bk = g.B.kind;
    if (ak == 0 && bk == 0) hipLaunchKernelGGL((koaf_gemm_kernel<BM, BN, 0, 0, VEC>), grid, dim3(256), 0, s, g);
    else if (ak == 0 && bk == 1) hipLaunchKernelGGL((koaf_gemm_kernel<BM, BN, 0, 1, VEC>), grid, dim3(256), 0, s, g);
    else if (ak == 1 && bk == 0) hipLaunchKernelGGL((koaf_gemm_kernel<BM, BN, 1, 0, VEC>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((koaf_gemm_kernel<BM, BN, 1, 1, VEC>), grid, dim3(256), 0, s, g);
    return koaf_check_launch("koaf_gemm");
}

}  // namespace

extern "C" int koaf_gemm_pick_tile(const KoafGemm* g, int32_t* bm, int32_t* bn) {
    int b_m = g->bm, b_n = g->bn;
    const int64_t batch = (int64_t)g->nb0 * g->nb1 * (g->splitk > 0 ? g->splitk : 1);
    if (b_n == 0) {
        b_n = (g->N >= 128) ? 128 : 64;
        // a gathered K-major B tile must stay inside one filter tap
        if (g->B.kind == 1 && g->B.gather == 1 && (g->B.C % b_n) != 0) b_n = 64;
    }
    if (b_m == 0) b_m = (g->M >= 128) ? 128 : 64;
    if (g->bm == 0 || g->bn == 0) {
        auto tiles = [&](int m, int n) { return cdiv64(g->M, m) * cdiv64(g->N, n) * batch; };
        // fill the 256 CUs: shrink the tile while the grid is under ~2 blocks per CU
        if (g->bm == 0 && tiles(b_m, b_n) < 512 && b_m == 128) b_m = 64;
        if (g->bn == 0 && tiles(b_m, b_n) < 512 && b_n == 128) b_n = 64;
    }
    *bm = b_m;
    *bn = b_n;
    return KOAF_OK;
}

extern "C" int koaf_gemm(const KoafGemm* gp, void* stream) {
    KoafGemm g = *gp;
    KOAF_REQUIRE(g.M > 0 && g.N > 0 && g.K >= 0, "koaf_gemm: bad dims M=%d N=%d K=%d", g.M, g.N, g.K);
    KOAF_REQUIRE(g.A.ptr && g.B.ptr && g.C, "koaf_gemm: null operand");
    if (g.nb0 < 1) g.nb0 = 1;
    if (g.nb1 < 1) g.nb1 = 1;
    if (g.splitk < 1) g.splitk = 1;
    if (g.A.CS == 0) g.A.CS = g.A.C;
    if (g.B.CS == 0) g.B.CS = g.B.C;
    if (g.stats_ld == 0) g.stats_ld = g.N;
    KOAF_REQUIRE(g.splitk == 1 || (!g.bias && !g.residual && !g.stats),
                 "koaf_gemm: split-K writes raw slabs (no epilogue)");
    KOAF_REQUIRE((int64_t)g.nb0 * g.nb1 <= 65535 && g.splitk <= 65535, "koaf_gemm: batch/splitk too large");
    KOAF_REQUIRE(g.A.kind == 0 || g.A.gather == 0, "koaf_gemm: K-major A cannot be gathered");
    KOAF_REQUIRE(!(g.A.kind == 0 && g.A.gather == 3) && !(g.B.kind == 0 && g.B.gather == 3),
                 "koaf_gemm: tapped gather needs a K-major operand");
    KOAF_REQUIRE(!(g.B.kind == 0 && g.B.gather), "koaf_gemm: K-contiguous B cannot be gathered");
    int bm, bn;
    koaf_gemm_pick_tile(&g, &bm, &bn);
    KOAF_REQUIRE((bm == 64 || bm == 128) && (bn == 64 || bn == 128), "koaf_gemm: tile must be 64|128");
    bool vec = operand_vec_ok(g.A, g.M, g.K) && operand_vec_ok(g.B, g.N, g.K);
    if (g.A.gather || g.B.gather) KOAF_REQUIRE(vec, "koaf_gemm: gathered operands need aligned, C%%32==0 tensors");
    if (g.B.kind == 1 && g.B.gather == 1)
        KOAF_REQUIRE(g.B.C % bn == 0, "koaf_gemm: wgrad tile (%d) must divide channels per tap (%d)", bn, g.B.C);
    if (!vec) { bm = 64; bn = 64; }
    g.bm = bm;
    g.bn = bn;
    const int64_t tiles = cdiv64(g.M, bm) * cdiv64(g.N, bn);
    KOAF_REQUIRE(tiles < (1ll << 31), "koaf_gemm: grid too large");
    dim3 grid((unsigned)tiles, (unsigned)g.splitk, (unsigned)(g.nb0 * g.nb1));
    hipStream_t s = (hipStream_t)stream;
    if (!vec) return launch_kinds<64, 64, false>(g, grid, s);
    if (bm == 128 && bn == 128) return launch_kinds<128, 128, true>(g, grid, s);
    if (bm == 128 && bn == 64) return launch_kinds<128, 64, true>(g, grid, s);
    if (bm == 64 && bn == 128) return launch_kinds<64, 128, true>(g, grid, s);
    return launch_kinds<64, 64, true>(g, grid, s);
}

extern "C" int koaf_slab_reduce(const float* slabs, int32_t nslab, int64_t n, float* out, void* stream) {
    KOAF_REQUIRE(slabs && out && nslab >= 1 && n > 0, "koaf_slab_reduce: bad args");
    KOAF_REQUIRE((((uintptr_t)slabs | (uintptr_t)out) & 15) == 0, "koaf_slab_reduce: unaligned");
    int64_t blocks = cdiv64(cdiv64(n, 4), 256);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, slabs,
                       nslab, n, out);
    return koaf_check_launch("koaf_slab_reduce");
}
