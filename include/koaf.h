/*
 * koaf.h -- C ABI of libkoaf.so: the hand-written gfx950 (MI355X / CDNA4) kernels under the
 * koafusion train-step hot path.
 *
 * The reference (imedslab/OAProgressionMMF) has no FFI: its hot path is stock torch.nn called from
 *   koafusion/models/_torchvision.py:118-138,141-246   (conv / BatchNorm2d / ReLU / MaxPool / GAP)
 *   koafusion/models/_core_trf.py:118-205               (Linear / LayerNorm / GELU / attention)
 *   koafusion/various/_losses.py:89-108                 (focal softmax-CE)
 *   koafusion/preproc/_pt.py:75-345                     (F.interpolate x0.5 downscale; the per-sample tensor transforms)
 *   torch.optim.Adam via koafusion/various/_optimizers.py:47-52
 * Each entry point below names the reference call site it replaces.  All pointers are DEVICE
 * pointers to fp32 (unless stated), all tensors are dense; activations are NHWC ("(n,h,w,c)",
 * c fastest).  Nothing allocates; every call is asynchronous on `stream` (a hipStream_t passed as
 * void*).  Return value: 0 = ok, negative = error (text via koaf_last_error()).
 *
 * Not thread-safe on one stream; re-entrant across devices/streams.
 */
#ifndef KOAF_H
#define KOAF_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define KOAF_OK 0
#define KOAF_EINVAL (-1)
#define KOAF_ELAUNCH (-2)

int koaf_version(void);          /* 100 * major + 10 * minor: 180 = this header */
const char* koaf_last_error(void);
/* Numerics status words: a device uint32[4] (zeroed by the caller; NULL = off, the default) that kernels bump with atomics when
 *   [0] an activation operand left the fp16 range of the fixed activation scale and was CLAMPED (KOAF_ACT_SCALE: |x| > 4094), or
 *       was not finite: counted per element by koaf_act_planes / koaf_bn_add_relu / koaf_bn_relu / koaf_maxpool_fwd (the
 *       producers of every tensor the convolutions read at that scale), and per GEMM tile by the fp32 loader that applies
 *       relu(sc*x+sh) on load (KoafOperand.tf 1; there a NaN becomes 0 and is not counted -- the BatchNorm coefficients are, [1]);
 *   [1] a NaN / Inf reached an operand's amax scalar (weights, gradients: the GEMM then returns NaN everywhere, see KoafGemm)
 *       or a BatchNorm's sc / sh came out non-finite (koaf_bn_finalize).
 * Process-wide (one process per GPU); read it back with a device-to-host copy whenever convenient (e.g. once per epoch). */
int koaf_set_status_buffer(uint32_t* dev4);

/* ---------------------------------------------------------------------------------------------
 * Generic MFMA GEMM  C[M,N] = alpha * sum_k A(m,k) B(n,k)  (+bias[n]) (+residual[m,n])
 * fp32 in / fp32 out / fp32 accumulate.  Products are formed on the bf16 matrix pipe from exact bf16 pieces of the
 * fp32 operands (KoafGemm.fmt: three bf16 pieces / six products, or two scaled fp16 pieces / three products): error at
 * fp32 rounding level either way.  Non-finite operands: fmt 0 -- an Inf operand yields NaN, NaN stays NaN.  fmt 1 -- the
 * scaled pieces are clamped to +-65504 (a NaN piece becomes the lower clamp bound), so a non-finite ELEMENT does not
 * propagate by itself; instead (a) operands scaled by a device amax (weights, gradients) carry their NaN / Inf into that
 * scalar (all amax reductions propagate non-finite values) and a GEMM that finds a non-finite amax returns NaN in EVERY
 * output element, and (b) activation operands at the fixed scale are watched by the numerics status words
 * (koaf_set_status_buffer): clamped / non-finite elements are counted, non-finite BatchNorm coefficients too.
 * Each operand is either K-contiguous ("KC": element (r,k) at ptr + r*ld + k) or K-major
 * ("KM": element (r,k) at ptr + k*ld + r).  Operands can be gathered on the fly from an NHWC
 * tensor (implicit-GEMM convolution) and transformed on load with relu(sc[c]*x+sh[c]) -- the
 * previous layer's BatchNorm+ReLU, so normalised activations never touch HBM.
 * ------------------------------------------------------------------------------------------- */
typedef struct KoafOperand {
    const float* ptr;
    int64_t ld;       /* KC: stride between rows; KM: stride between k-rows */
    int64_t bs0, bs1; /* batch strides (elements) for batch index z = z0*nb1 + z1 */
    int64_t tap_stride;   /* gather 3 (K-major weights [C][taps][rows]): offset between taps in a tap row */
    int64_t tap_stride_h; /* gather 3: offset between tap rows; tap index = th*KW + tw */
    int32_t kind;     /* 0 = KC, 1 = KM, 2 = pre-split fp16 plane images (B: weights, see `planes`; A: activations, see `zeros`),
                         3 = activation plane images read K-major (both operands of a weight gradient, see `zeros`) */
    int32_t gather;   /* 0 none; 1 conv forward gather; 2 transposed-conv (dgrad) gather;
                         3 (KM only) tapped weights: k = (tap, c), element at c*ld + tap*tap_stride + r */
    int32_t H, W, C;  /* source NHWC tensor dims for a gathered operand (C = channels per tap) */
    int32_t CS;       /* channels per source pixel (0 = C); > C when the GEMM sees a channel slab */
    int32_t PH, PW;   /* pixel grid the GEMM rows (KC) or the k index (KM) enumerate: (n,py,px) */
    int32_t KH, KW, stride, pad; /* pad = row (y) padding */
    int32_t pad_w;               /* column (x) padding (set = pad for square padding) */
    int32_t _pad1;
    int32_t tf;       /* transform on load (c = source channel): 0 none; 1 relu(sc[c]*x + sh[c]); 2 (A operand, fmt 1, vector
                         path) sc[c]*x + sh[c] - sc2[c]*x2 with x2 read from ptr2 (same layout and strides as ptr): the
                         BatchNorm-backward apply formed on load, see koaf_bn_bwd_finalize; 3 (A operand, dense K-contiguous,
                         fmt 1 with a pre-split B) y = relu(sc[c]*x + sh[c] + x2): the BOTTLENECK TAIL relu(bn3(c3) + identity)
                         (_torchvision.py:132-136) formed in the loader of the next block's conv1 -- koaf_bn_add_relu's
                         arithmetic bit for bit -- with y written once to `side` by the blocks of the first column tile; with
                         sc2 / sh2 the identity is sc2[c]*x2 + sh2[c] (a downsample branch's raw conv output and BatchNorm) */
    int32_t tf_bs;    /* channel offset of sc/sh per batch index z1 (grouped-conv slabs) */
    const float* sc;
    const float* sh;
    /* kind 2 (B operand, KoafGemm.fmt 1): `planes` -> plane 0 of the two fp16 plane images (hi, lo) of a K-contiguous matrix
       scaled by scale(*amax) (cut by koaf_wplanes_build): element (r, k) of plane q at planes + q*plane_stride + r*ld + k,
       every row zero-filled up to a multiple of 32 k.  All strides (ld, plane_stride, bs0, bs1, tap_stride, tap_stride_h)
       count 16-bit elements and are multiples of 8.  Tapped addressing as for conv weights: with C > 0 the GEMM's
       k = (tap, c), tap = th*KW + tw, lives at th*tap_stride_h + tw*tap_stride + c of the row (C % 32 == 0); C == 0: k is
       the row offset.  The kernel moves these planes global -> LDS directly (global_load_lds): no split arithmetic in
       the k-loop. */
    const uint16_t* planes;
    int64_t plane_stride;
    /* fp16 scheme (KoafGemm.fmt == 1): the operand is multiplied by a power of two before it is cut into fp16 pieces:
       amax != NULL: scale(*amax) = 2^e with *amax * 2^e in [2^14, 2^15) (1 for *amax == 0) -- *amax = max |element| of the
       tensor, written on the device by its producer; amax == NULL: fscale (0 = 1).  Scaled magnitudes are clamped to 65504. */
    const float* amax;
    float fscale;
    int32_t _pad4;
    const float* ptr2;  /* tf 2: second source tensor */
    const float* sc2;   /* tf 2: its per-channel coefficient */
    /* kind 2 with gather 1 | 2 (A operand): ACTIVATION plane images cut by koaf_act_planes -- the two fp16 piece planes
       [pixel][CS] of an NHWC tensor (plane q at planes + q*plane_stride), transform already applied (tf must be 0), scaled
       by the operand's scale (amax / fscale as above).  Gathered like the fp32 operand of the same `gather`; padding taps
       and rows past M read `zeros` (16 B of zeros, 16-B aligned: koaf_act_planes leaves them at planes + 2*plane_stride).
       C % 32 == 0, CS % 8 == 0; needs a pre-split B (kind 2), fmt 1, K % 32 == 0, no split-K. */
    /* kind 3 (A and B together, fmt 1): the same images read K-MAJOR -- k = pixel, rows = channels: element (k, r) of plane q
       at planes + q*plane_stride + k*ld + r (A, gather 0; ld = channels per pixel), or gathered like a kind-1 gather-1 operand
       (B: column = tap*C + c, k = output pixel -> source pixel by H/W/PH/PW/KH/KW/stride/pad; C % 8 == 0).  Rows % 8 == 0. */
    const uint16_t* zeros;
    float* side;        /* tf 3: where the loader stores y (same layout as ptr; nullable: y is then not kept) */
    const float* sh2;   /* tf 3 with sc2 != NULL: the identity is sc2[c]*x2 + sh2[c] (the BatchNorm of a downsample branch) */
} KoafOperand;

typedef struct KoafGemm {
    KoafOperand A, B;
    int32_t M, N, K;
    int32_t nb0, nb1; /* batch = nb0*nb1 (>=1 each) */
    int32_t splitk;   /* >=1; >1: C is a slab buffer [splitk][M][N] (ldc == N), no epilogue */
    int32_t bm, bn;   /* block tile (64|128); 0 = pick */
    float* C;
    int64_t ldc, cbs0, cbs1;
    float alpha;
    int32_t prec;            /* 0: a forward contraction; 1: a data / weight gradient contraction (informational) */
    const float* bias;     /* [N] or NULL */
    const float* residual; /* [M][ldr] (+batch strides) or NULL */
    int64_t ldr, rbs0, rbs1;
    float* stats;          /* [ceil(M/bm)][2][stats_ld] per-M-tile column sum / sum of squares, or NULL */
    int64_t stats_ld;      /* 0 = N */
    int64_t stats_bs;      /* column offset per batch index (grouped conv slabs) */
    /* output row map (stride-2 dgrad parity classes): GEMM row (n, y', x') over a cm_PH x cm_PW grid is
       written to pixel row (n*cm_H + 2y'+cm_py)*cm_W + 2x'+cm_px of C / residual.  cmap = 0: identity. */
    int32_t cmap, cm_PH, cm_PW, cm_H, cm_W, cm_py, cm_px, _pad2;
    /* Fused BatchNorm(+ReLU)-backward reduction (dgrad epilogue).  The GEMM output g (after bias/residual) is the
       gradient w.r.t. relu(bn(c)); the epilogue masks it (bnb_mode 1: bnb_y > 0; 2: bnb_sc*c+bnb_sh > 0), stores
       dz instead of g, and emits per-M-tile column sums  bnb_part[tile][k][N]:  k=0: sum dz, k=1: sum dz*xhat with
       xhat = (c-mean)*invstd, k=2 (if bnb2_c): sum dz*xhat2 for a second BatchNorm fed by the same dz (the
       downsample branch).  c / y / c2 are laid out like C (same ldc, same row map).  bnb_mode 0 = off. */
    int32_t bnb_mode;
    int32_t fmt;             /* how the fp32 x fp32 products are formed on the matrix pipe:
                              * 0: three exact bf16 pieces per operand, the six piece products of weight >= 2^-16
                              *    (v_mfma_f32_32x32x16_bf16): any fp32 operand, no scale information needed;
                              * 1: two fp16 pieces of operand * 2^e (hi = fp16(x'), lo = fp16(x' - hi): x' = hi + lo to 2^-24
                              *    relative down to |x'| = 2^-2, 2^-25 absolute below) and the three products hi*hi, hi*lo,
                              *    lo*hi (v_mfma_f32_32x32x16_f16; each exact in fp32, the dropped lo*lo <= 2^-24 of the
                              *    product): the same fp32-level accuracy from half the matrix instructions, for operands
                              *    whose magnitude is known (KoafOperand.amax / fscale).  Needs the vector path. */
    const float* bnb_c;
    const float* bnb_y;
    const float* bnb_sc;
    const float* bnb_sh;
    const float* bnb_mean;
    const float* bnb_invstd;
    const float* bnb2_c;
    const float* bnb2_mean;
    const float* bnb2_invstd;
    float* bnb_part;
    float* bnb_amax;       /* nullable: device scalar raised (atomic max; zero it beforehand) to the largest |dz| stored */
    /* row-space origin of this launch (mixed-height tiling: one GEMM = a 128-row-tile launch over rows
       [0, M1) + a 64-row-tile launch over [M1, M)); partial-statistics rows continue at part_row0 */
    int32_t m_base, part_row0;
    /* per-column shift k[N] (+ stats_bs per batch index) of the statistics: stats rows hold sum (v - k) and
       sum (v - k)^2 (NULL: k = 0).  With k near the column mean -- the BatchNorm's running mean -- the variance
       E[(v-k)^2] - E[v-k]^2 keeps its digits when |mean| >> std. */
    const float* stats_shift;
    uint32_t* status;      /* numerics status words (koaf_set_status_buffer); NULL = the registered buffer */
    /* bf16 ACTIVATION STORAGE: which tensors of this call are forward activations kept in HBM as bf16 instead of fp32 (the
       pointers stay typed float*; strides / offsets count elements).  0: none.  1 (forward convolution): A.ptr and C -- the
       loader widens (exact), the epilogue rounds the output to nearest even; statistics come from the fp32 accumulators.
       2 (data gradient): A.ptr2 (the conv output c of a tf-2 apply) and bnb_c / bnb_y / bnb2_c.  3 (weight gradient): A.ptr2
       and B.ptr.  Gradients, weights, statistics and all arithmetic stay fp32; needs the vector path. */
    int32_t act16;
    int32_t _pad5;
    /* Epilogue side output (nullable; forward convolutions whose OUTPUT BatchNorm is already known -- eval mode, stages rebuilt in
       backward): besides C, the epilogue writes the activation plane images of relu(out_sc[n] * C + out_sh[n]) at the scale
       KOAF_ACT_SCALE -- exactly what koaf_act_planes (tf 1) would cut from C in a pass of its own: [2][M][N] fp16 pieces,
       plane stride out_ps = M * N elements, the 16-B zero chunk behind them.  Needs the vector epilogue, ldc == N, no batch, no
       split-K, no row map. */
    uint16_t* out_planes;
    const float* out_sc;
    const float* out_sh;
    int64_t out_ps;
} KoafGemm;

int koaf_gemm(const KoafGemm* g, void* stream);
/* block tile koaf_gemm picks for (M, N, batch) when bm = bn = 0 */
int koaf_gemm_pick_tile(const KoafGemm* g, int32_t* bm, int32_t* bn);
/* number of per-tile partial rows (stats / bnb_part) koaf_gemm writes for this descriptor (mixed-height tiling
 * included); callers size / slice their buffers with it */
int koaf_gemm_part_rows(const KoafGemm* g);
/* out[i] = sum_s slabs[s][i], i < n, n % 4 == 0 (deterministic split-K combine).  The slab workspace must
 * hold (nslab + 16) * n floats: large counts are folded in two levels through the 16 trailing slabs. */
int koaf_slab_reduce(const float* slabs, int32_t nslab, int64_t n, float* out, void* stream);
/* out[m][c] = sum_s slabs[s][m][c] + bias[c] + residual[m][c]  (split-K combine with the linear epilogue) */
int koaf_slab_reduce_epilogue(const float* slabs, int32_t nslab, int32_t M, int32_t N, const float* bias,
                              const float* residual, int64_t ldr, float* out, int64_t ldo, void* stream);

/* ---- Weight plane images ---------------------------------------------------------------------
 * Every convolution weight of the model ([Cout][KH*KW][Cin] packed) is cut once per optimizer step into the two fp16
 * piece planes of w * scale(amax(w)) that the MFMA kernel multiplies (KoafGemm.fmt 1), in two arrangements:
 *   F image [2][R][Kp]          Kp = taps*C rounded up to 32: B operand of the forward GEMM (rows = output channels)
 *   D image [2][C][taps*Rp]     Rp = R rounded up to 32: the transposed weight, B operand of the data-gradient GEMM
 * Two launches handle a whole table of descriptors (device memory): max |w| per weight -> amax[i], then the images.
 * tile0 = running sum of ceil(R/32) * taps * ceil(C/32) over the preceding entries, `ntiles` the total.  taps > 1 needs
 * C % 32 == 0.  src_off counts floats from `base`, f_off / d_off 16-bit elements from `planes` (multiples of 8; -1 =
 * image not wanted). */
typedef struct KoafWPlane {
    int64_t src_off, f_off, d_off, tile0;
    int32_t R, taps, C, Kp, Rp, _pad;
} KoafWPlane;
int koaf_wplanes_build(const float* base, uint16_t* planes, float* amax, const KoafWPlane* table_dev, int32_t n,
                       int64_t ntiles, void* stream);
/* Activation plane images (KoafOperand.kind 2 on the A side): planes[q][pixel][C], q = 0 (hi), 1 (lo), of
 *   tf 0: x          tf 1: relu(sc[c]*x + sh[c])          tf 2: sc[c]*x + sh[c] - sc2[c]*x2
 * times the operand scale (scale(*amax) if amax != NULL, else fscale), clamped to +-65504, cut like the GEMM's own loader
 * cuts them (bit-identical), followed by the 16-B zero chunk.  `planes` holds koaf_act_planes_elems(npix, C) 16-bit
 * elements.  One HBM pass (read 4 or 8 B, write 4 B per element) that saves the convolution's k-loop the KH*KW-fold
 * conversion of every element. */
/* Gathered 3x3 / stride 1 / pad 1 convolutions over activation plane images (image rows up to 96 pixels) run the HALO
 * kernel: 256-pixel tiles in raster order whose source pixels for all nine taps are one contiguous range kept in LDS, so
 * the input tile is fetched 1.8 times instead of nine (k runs (channel chunk, tap, channel): the sums are reassociated
 * against the per-tap gather kernel, same accuracy).  Two shapes: 256 rows / 8 waves / one block per CU (halo double-
 * buffered across channel chunks) and 128 rows / 4 waves / two blocks per CU (one halo buffer; the 64-channel layers).
 * koaf_set_conv3x3_halo(mode): 0 sends them through the gather kernel instead, 1 (default) picks the shape per layer,
 * 2 / 3 force the 256- / 128-row shape where it fits (tests / A-B measurements); returns the previous mode.  Process-wide;
 * not meant to be flipped while other threads launch. */
int koaf_set_conv3x3_halo(int on);
/* Dense K-contiguous fp32 A operands in front of weight plane images -- the 1x1 / stride-1 convolutions (plain, BatchNorm-prologue,
 * bottleneck-tail loaders: koafusion/models/_torchvision.py:118-138) and their data gradients with the BatchNorm-backward apply --
 * with 128-row tiles and K a multiple of 64 run the STREAMED kernel: the four waves of a block each load, transform and split their
 * own 32 rows, two k-tiles ahead in registers, the weight tiles arrive through a three-stage LDS ring, and a persistent block
 * prefetches across tile boundaries.  Same pieces, same MFMA order per accumulator: bit-identical to the block-wide loader.
 * koaf_set_stream(0) sends them through the block-wide loader instead (tests / A-B measurements; environment KOAF_STREAM=0 does the
 * same for a whole process); returns the previous setting.  Process-wide; not meant to be flipped while other threads launch. */
int koaf_set_stream(int on);
int64_t koaf_act_planes_elems(int64_t npix, int32_t C);
int koaf_act_planes(const float* x, const float* x2, int64_t npix, int32_t C, int32_t tf, const float* sc, const float* sh,
                    const float* sc2, const float* amax, float fscale, uint16_t* planes, int32_t act16, void* stream);
/* what the convolution entry points take for a weight whose images are current (all device pointers) */
typedef struct KoafWImg {
    const uint16_t* f;      /* F image or NULL */
    const uint16_t* d;      /* D image or NULL */
    const float* amax;      /* max |w| the images were scaled by */
} KoafWImg;
/* dy given as its BatchNorm-backward apply (KoafOperand.tf 2): dy = coef0*dz + coef3 - coef2*c with coef [4][Cout] and amax
 * from koaf_bn_bwd_finalize; dz and c are [N,OH,OW,Cout] like dy.  Needs the fp16 scheme (amax; for dgrad also wimg). */
typedef struct KoafBnApply {
    const float* dz;
    const float* c;
    const float* coef;
    const float* amax;
} KoafBnApply;

/* ---- bf16 ACTIVATION STORAGE (`act16` of the entry points below; BASELINE.json config 2 "bf16") -------------------
 * A trunk may keep its FORWARD ACTIVATIONS -- stem / conv outputs, block outputs, max-pool output -- in HBM as bf16 instead of
 * fp32: half the bytes on every HBM-bound call, half the memory saved for backward.  act16 != 0 says that the activation tensors
 * among a call's arguments (named at each entry point) are bf16 behind their float* type; everything else stays fp32: weights,
 * BatchNorm statistics and coefficients, every gradient, all accumulation and all arithmetic -- a kernel widens the bf16 values
 * on load (exact), computes exactly as in the fp32 mode, and only a producer's store rounds (to nearest even).  It is a storage
 * mode, selected per model (config key `activation_storage: bf16`, default fp32), reported by bench.py as a named secondary
 * with its measured error against the fp32 mode; the parity gate (1e-3 of the reference) is the fp32 mode's.
 * Activation arguments per entry point: koaf_conv2d_fwd / koaf_gconv3x3_fwd x, y; koaf_conv2d_dgrad(_bnb) dy_apply->c and
 * bnb->c / y / c2; koaf_conv2d_wgrad / koaf_gconv3x3_wgrad x and dy_apply->c; koaf_stem_fwd y; koaf_colstats x; koaf_bn_add_relu /
 * koaf_bn_relu c, idt, y; koaf_bn_bwd_reduce c, ymask; koaf_bn_bwd_apply c; koaf_maxpool_fwd c, y; koaf_gap_fwd y; koaf_act_planes
 * x (tf 0 / 1) or x2 (tf 2). */

/* ---- Convolution (nn.Conv2d, bias-free; _torchvision.py:23-31) as implicit GEMM on NHWC -------
 * x [N,H,W,Cin], w packed [Cout,KH,KW,Cin] (the memory of a channels_last (Cout,Cin,KH,KW)
 * parameter), y [N,OH,OW,Cout].  in_sc/in_sh (nullable): fused BatchNorm+ReLU of the producer
 * applied to x on load.  stats (nullable): per-M-tile column sums / sums of squares of y for the
 * following BatchNorm (train mode), *stats_rows rows of [2][Cout], summed about stats_shift[Cout] (nullable = 0; pass
 * that BatchNorm's running_mean and hand the same pointer to koaf_bn_finalize).  wimg (nullable): plane images of w
 * (koaf_wplanes_build; they must be current): the contraction then runs on the fp16 scheme (KoafGemm.fmt 1) with the
 * activations at the fixed scale KOAF_ACT_SCALE and the weight tiles DMA'd from wimg->f (wimg->f NULL: weight split in
 * the kernel with the same scale -- bit-identical, slower).  x_planes (nullable; needs wimg->f, Cin % 32 == 0): activation
 * plane images of the TRANSFORMED input (koaf_act_planes: tf 1 with in_sc / in_sh, or tf 0; fscale KOAF_ACT_SCALE): the
 * gathered input tiles are then DMA'd as well and x / in_sc / in_sh are not read -- bit-identical, and what the 3x3
 * convolutions use: their fp32 loader converts every element nine times.  */
#define KOAF_ACT_SCALE 16.0f   /* activations are O(1) behind BatchNorm: |x| * 16 clamps at 65504, 2^-29 absolute resolution */
/* tail (nullable; 1x1 / stride 1 convolutions on the fp16 scheme with wimg->f): x is the raw output c3 of the PREVIOUS block's
 * last convolution, in_sc / in_sh its BatchNorm's coefficients, tail->idt that block's identity: the convolution's input
 * y = relu(in_sc*x + in_sh + idt) -- the bottleneck tail, _torchvision.py:132-136 -- is formed on load (KoafOperand.tf 3) and
 * written once to tail->y_out [N,H,W,Cin] (nullable), so the element-wise tail pass (koaf_bn_add_relu: 12 B per element) and
 * this convolution's own read of y (4 B) become one read of c3 + idt and one write of y.  idt_sc / idt_sh (nullable pair): the
 * block had a downsample branch -- idt is that branch's raw convolution output and these its BatchNorm coefficients. */
typedef struct KoafTail {
    const float* idt;
    float* y_out;
    const float* idt_sc;   /* nullable pair: the identity is idt_sc[c]*idt + idt_sh[c] -- the raw output of a downsample */
    const float* idt_sh;   /* convolution and its BatchNorm coefficients (koaf_bn_add_relu's idsc / idsh)             */
} KoafTail;
/* emit (nullable): the BatchNorm behind THIS convolution is already known (eval mode; a stage rebuilt in backward from its saved
 * statistics: _torchvision.py:118-138 with running / saved statistics): the epilogue also cuts the activation plane images of
 * relu(sc * y + sh) -- what the following 3x3 convolution's koaf_act_planes pre-pass would read y back for -- into `planes`
 * (koaf_act_planes_elems(N * OH * OW, Cout) elements, bit-identical to that pass). */
typedef struct KoafEmit {
    uint16_t* planes;
    const float* sc;
    const float* sh;
} KoafEmit;
int koaf_conv2d_fwd(const float* x, const float* w, float* y, int32_t N, int32_t H, int32_t W,
                    int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                    const float* in_sc, const float* in_sh, float* stats, int32_t* stats_rows,
                    const float* stats_shift, const KoafWImg* wimg, const uint16_t* x_planes, const KoafTail* tail,
                    const KoafEmit* emit, int32_t act16, void* stream);
/* rows of the stats buffer koaf_conv2d_fwd writes for M output pixels */
int32_t koaf_conv2d_stats_rows(int64_t M, int32_t Cout);
/* dx [N,H,W,Cin] = conv_transpose(dy [N,OH,OW,Cout], w) (+residual: the other branch's gradient);
 * w is read K-major in place (no re-packed copy).  With wimg AND dy_amax (device scalar: max |dy|, e.g. from
 * koaf_bn_bwd_apply) the contraction runs on the fp16 scheme, the weight tiles DMA'd from wimg->d.  dy_apply (nullable, needs
 * wimg->d; dy may then be NULL): dy is formed on load from (dz, c), see KoafBnApply.  dy_planes (nullable; needs wimg->d,
 * dy_amax, Cout % 32 == 0): activation plane images of dy (koaf_act_planes with amax = dy_amax; tf 2 for an applied dy):
 * dy / dy_apply are then not read.  */
int koaf_conv2d_dgrad(const float* dy, const float* w, float* dx, int32_t N, int32_t H, int32_t W,
                      int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride,
                      int32_t pad, const float* residual, const KoafWImg* wimg, const float* dy_amax,
                      const KoafBnApply* dy_apply, const uint16_t* dy_planes, int32_t act16, void* stream);
/* Same, with the BatchNorm(+ReLU) backward reduction of the layer that PRODUCED x fused into the epilogue (see
 * KoafGemm.bnb_*): dx receives the masked gradient dz; part [*part_rows][nsum][Cin] (nsum = 2, or 3 with c2) feeds
 * koaf_bn_bwd_finalize.  koaf_conv2d_dgrad_bnb_rows() bounds *part_rows for sizing. */
typedef struct KoafBnb {
    int32_t mode, _pad;     /* 1: mask y > 0; 2: mask sc*c+sh > 0 */
    float* dz_amax;         /* nullable: device scalar set to max |dz| (zeroed by the call) -> koaf_bn_bwd_finalize */
    const float* c;
    const float* y;
    const float* sc;
    const float* sh;
    const float* mean;
    const float* invstd;
    const float* c2;        /* optional second BatchNorm on the same dz */
    const float* mean2;
    const float* invstd2;
} KoafBnb;
int32_t koaf_conv2d_dgrad_bnb_rows(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t stride);
int koaf_conv2d_dgrad_bnb(const float* dy, const float* w, float* dx, int32_t N, int32_t H, int32_t W,
                          int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                          const float* residual, const KoafBnb* bnb, float* part, int32_t* part_rows,
                          const KoafWImg* wimg, const float* dy_amax, const KoafBnApply* dy_apply,
                          const uint16_t* dy_planes, int32_t act16, void* stream);
/* dw packed [Cout,KH,KW,Cin] = sum_pixels dy^T x, x optionally transformed on load.  Deterministic
 * split-K: slabs = workspace of koaf_conv2d_wgrad_ws() floats (0 = none needed).  dy_amax (nullable): max |dy| on the
 * device -> fp16 scheme.  dy_planes + x_planes (nullable, together; need dy_amax or dy_apply, Cin % 8 == 0, Cout % 8 == 0):
 * activation plane images of dy (amax = dy_amax; tf 2 for an applied dy) and of the transformed x (fscale KOAF_ACT_SCALE):
 * both operand tiles are then DMA'd, K-major, and dy / x / in_sc / in_sh are not read.  */
int64_t koaf_conv2d_wgrad_ws(int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                             int32_t KH, int32_t KW, int32_t stride, int32_t pad);
int koaf_conv2d_wgrad(const float* dy, const float* x, float* dw, int32_t N, int32_t H, int32_t W,
                      int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride,
                      int32_t pad, const float* in_sc, const float* in_sh, float* slabs,
                      const float* dy_amax, const KoafBnApply* dy_apply, const uint16_t* dy_planes,
                      const uint16_t* x_planes, int32_t act16, void* stream);

/* ---- Grouped 3x3 convolution (ResNeXt 32x4d; _torchvision.py:110,327-330) -------------------
 * Runs on the same MFMA GEMM as 64-channel block-diagonal slabs: packed weights [C][3][3][C/groups]
 * are expanded to wexp [C/64][64][9][64] (zeros off the group blocks), gradients compressed back.
 * Contraction scheme (KoafGemm.fmt): with the operands' largest magnitudes known the three calls run on the fp16 scheme (two
 * scaled fp16 pieces, three products: half the matrix instructions of the bf16 scheme, same accuracy) -- w_amax = device scalar
 * max |w| (koaf_gconv_expand_w leaves it when given `amax`, which the caller zeroes beforehand), dy_amax = device scalar max |dy|
 * (koaf_bn_bwd_apply leaves it); activations x use the fixed activation scale KOAF_ACT_SCALE.  NULL scalars (and every call with
 * act16 != 0): the bf16 scheme. */
int koaf_gconv_expand_w(const float* w, float* wexp, int32_t C, int32_t groups, float* amax, void* stream);
int koaf_gconv_compress_dw(const float* dwexp, float* dw, int32_t C, int32_t groups, void* stream);
int koaf_gconv3x3_fwd(const float* x, const float* wexp, float* y, int32_t N, int32_t H, int32_t W,
                      int32_t C, int32_t stride, const float* in_sc, const float* in_sh,
                      float* stats, int32_t* stats_rows, const float* stats_shift, const float* w_amax, int32_t act16, void* stream);
int koaf_gconv3x3_dgrad(const float* dy, const float* wexp, float* dx, int32_t N, int32_t H,
                        int32_t W, int32_t C, int32_t stride, const float* w_amax, const float* dy_amax, void* stream);
int64_t koaf_gconv3x3_wgrad_ws(int32_t N, int32_t H, int32_t W, int32_t C, int32_t stride);
int koaf_gconv3x3_wgrad(const float* dy, const float* x, float* dwexp, int32_t N, int32_t H,
                        int32_t W, int32_t C, int32_t stride, const float* in_sc,
                        const float* in_sh, float* slabs, const float* dy_amax, int32_t act16, void* stream);

/* ---- Stem: 7x7 s2 p3 conv on the 1->3 channel-repeated image (_torchvision.py:170; the
 * `repeat "b ch r c -> b (k ch) r c", k=3` of _xrNmrMcP.py:211-213 is folded: w1t = sum_c w[:,c]).
 * x [N,H,W] (single channel), w1t [49][64], y [N,OH,OW,64].  */
/* stats (nullable): koaf_stem_stats_rows(N, H) rows of [2][64] -- column sums and sums of squares of y about stats_shift[64]
 * (nullable = 0), the partials koaf_bn_finalize takes for the BatchNorm behind the stem (as koaf_conv2d_fwd's stats) */
int32_t koaf_stem_stats_rows(int32_t N, int32_t H);
int koaf_stem_fwd(const float* x, const float* w1t, float* y, int32_t N, int32_t H, int32_t W,
                  float* stats, const float* stats_shift, int32_t act16, void* stream);
int64_t koaf_stem_wgrad_ws(int32_t N, int32_t H, int32_t W);
/* dy_apply (nullable; dy may then be NULL): dy is formed on load from (dz, c, coef) as coef0*dz + coef3 - coef2*c -- the
 * BatchNorm-backward apply of the stem's BatchNorm (koaf_bn_bwd_finalize with mean) -- and never written; act16: c is bf16 */
int koaf_stem_wgrad(const float* dy, const float* x, float* dw1t, int32_t N, int32_t H, int32_t W,
                    float* slabs, const KoafBnApply* dy_apply, int32_t act16, void* stream);
/* w [64,7,7,3] packed -> w1t [49][64] (sum over the 3 channels); gradient un-fold (copy x3) */
int koaf_stem_fold_w(const float* w, float* w1t, void* stream);
int koaf_stem_unfold_dw(const float* dw1t, float* dw, void* stream);

/* ---- BatchNorm2d (nn.BatchNorm2d; _torchvision.py:172,121-131) ------------------------------- */
/* per-block column sums / sums of squares of x [rows][C] -> part [*part_rows][2][C]
 * (koaf_colpart_rows(rows, C) rows), summed about shift[C] (nullable = 0); for producers without a GEMM epilogue
 * (stem). */
int koaf_colstats(const float* x, int64_t rows, int32_t C, float* part, int32_t* part_rows,
                  const float* shift, int32_t act16, void* stream);
int32_t koaf_colpart_rows(int64_t rows, int32_t C);
/* Bytes of the fp64 workspace `ws` the two finalisations below use to spread a long list of partial rows over
 * the chip (two-stage, fixed-order reduction); 0 = not needed for this row count.  ws may always be NULL
 * (single-stage). */
int64_t koaf_bn_reduce_ws(int32_t rows, int32_t C);
/* stats [rows][2][C] -> mean, invstd, sc = gamma*invstd, sh = beta - mean*sc; train: updates
 * running_mean/var (momentum, unbiased var) and ++num_batches_tracked (int64).  eval (train==0):
 * stats ignored, uses running stats.  shift (nullable): the per-channel shift k the statistics were summed about
 * (KoafGemm.stats_shift / koaf_colstats): mean = k + E[x-k], var = E[(x-k)^2] - E[x-k]^2; it may alias running_mean
 * (read before the update).  */
int koaf_bn_finalize(const float* stats, int32_t rows, int32_t C, int64_t count,
                     const float* gamma, const float* beta, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                     int32_t train, float* mean, float* invstd, float* sc, float* sh,
                     const float* shift, double* ws, void* stream);
/* y = relu(sc*c + sh + identity-term); identity-term = idt (materialised) or idsc*idt+idsh
 * (downsample branch BN folded).  Bottleneck tail, _torchvision.py:132-136.  */
int koaf_bn_add_relu(const float* c, const float* sc, const float* sh, const float* idt,
                     const float* idsc, const float* idsh, float* y, int64_t rows, int32_t C,
                     int32_t act16, void* stream);
/* y = relu(sc*c+sh) materialised */
int koaf_bn_relu(const float* c, const float* sc, const float* sh, float* y, int64_t rows,
                 int32_t C, int32_t act16, void* stream);
/* backward reduce: dz = g * mask, partial sums of dz and dz*(c-mean)*invstd.
 * mask_mode 0: none; 1: y>0 from tensor `ymask`; 2: sc*c+sh>0 recomputed.  If dz_out != NULL
 * writes the masked gradient.  part [*part_rows][2][C] (koaf_colpart_rows rows).  dz_amax (nullable): device scalar set to
 * max |dz| (for koaf_bn_bwd_finalize's bound). */
int koaf_bn_bwd_reduce(const float* g, const float* c, const float* ymask, const float* sc,
                       const float* sh, const float* mean, const float* invstd, int32_t mask_mode,
                       float* dz_out, float* part, int32_t* part_rows, int64_t rows, int32_t C,
                       float* dz_amax, int32_t act16, void* stream);
/* the same reduction for a BatchNorm(+ReLU) that is followed by the 3x3 / stride-2 / pad-1 max-pool (the stem,
 * _torchvision.py:172-174): the upstream gradient is gathered from the POOL's gradient pool_g [N,OH,OW,C] and the window
 * positions pool_argmax of koaf_maxpool_fwd (the arithmetic of koaf_maxpool_bwd), masked by sc*c+sh > 0 and written to dz_out
 * [N,H,W,C] -- the pool's input gradient is never written and read back */
int koaf_bn_bwd_reduce_pool(const float* pool_g, const uint8_t* pool_argmax, const float* c, const float* sc,
                            const float* sh, const float* mean, const float* invstd, float* dz_out, float* part,
                            int32_t* part_rows, int32_t N, int32_t H, int32_t W, int32_t C, float* dz_amax,
                            int32_t act16, void* stream);
/* part [rows][nsum][C] -> dgamma (= sum index i1), dbeta (= sum index 0), and apply coefficients
 * coef [3][C] = {sc, dbeta/M, sc*invstd*dgamma/M}.  (nsum, i1) = (2, 1) for the plain layout.
 * With mean: coef is [4][C], the fourth row = coef2*mean - coef0*coef1, so that dc = coef0*dz + coef3 - coef2*c -- the form
 * the GEMM loaders evaluate on load (KoafOperand.tf 2: the dgrad / wgrad convolutions then read dz and c and dc is never
 * written).  amax (nullable, needs mean): device scalar set to a guaranteed bound of max |dc|, the scale of that operand on
 * the fp16 scheme: max_c |coef0| (dz_amax + |coef1|) + |coef2| sqrt(M - 1) / invstd  (dz_amax: device scalar max |dz| from
 * koaf_bn_bwd_reduce / KoafBnb.dz_amax; no sample lies further than sqrt(M - 1) standard deviations from its mean). */
int koaf_bn_bwd_finalize(const float* part, int32_t part_rows, int32_t C, int64_t count,
                         const float* sc, const float* invstd, float* dgamma, float* dbeta,
                         float* coef, int32_t nsum, int32_t i1, double* ws, const float* mean,
                         const float* dz_amax, float* amax, void* stream);
/* dc = coef0*(dz - coef1) - coef2*(c - mean), materialised (consumers that are not GEMMs: the stem's weight gradient; GEMMs
 * without scale information).  amax (nullable): device scalar raised to max |dc| (atomic max; zero it beforehand). */
int koaf_bn_bwd_apply(const float* dz, const float* c, const float* mean, const float* coef,
                      float* dc, int64_t rows, int32_t C, float* amax, int32_t act16, void* stream);

/* ---- input pipeline on the device (koafusion/preproc/_pt.py; applied per sample by the reference's CPU loader
 * workers, koafusion/datasets/_data_provider.py:295-335) ------------------------------------------------------- */
/* Integer volumes as stored on disk (dtype 1 = uint8: radiograph PNGs, 2 = uint16, 3 = int16: MRI NIfTI) -> fp32.  The loader
 * ships the raw integers over PCIe (4x / 2x fewer bytes than the fp32 tensors the reference's workers produce,
 * _data_provider.py:460-498) and the device widens them ahead of koaf_minmax / koaf_augment. */
int koaf_widen(const void* x, int32_t dtype, float* y, int64_t n, void* stream);
/* per-sample minimum and maximum of x [B][n] -> mm [B][2]; ws: B * koaf_minmax_ws(n) floats.  PTToUnitRange :75-99 */
int64_t koaf_minmax_ws(int64_t n);
int koaf_minmax(const float* x, int32_t B, int64_t n, float* mm, float* ws, void* stream);
/* y = ((gamma(rotate(unit(x)))) - mean) / std on a batch x [B][R][C][S] (S = 1: radiographs):
 *   unit(v) = (v - min_b) / (max_b - min_b)                                   PTToUnitRange        :75-99
 *   rotate: in-plane (R, C) bilinear resampling, zero padding, F.affine_grid + F.grid_sample with
 *           align_corners=False and [[cos,-sin,0],[sin,cos,0]]                PTRotate3DInSlice / PTRotate2D :257-345
 *   gamma(v) = pow(v, e)                                                      PTGammaCorrection    :203-232
 * params [B][4] = {cos(theta), sin(theta), e (0 = no gamma), rotate flag (0 = copy)}.  */
int koaf_augment(const float* x, float* y, const float* mm, const float* params, int32_t B, int32_t R, int32_t C,
                 int32_t S, float mean, float stdv, void* stream);

/* ---- MaxPool2d 3x3 s2 p1 over relu(sc*c+sh) (_torchvision.py:173-174), GAP (:182) ------------ */
int koaf_maxpool_fwd(const float* c, const float* sc, const float* sh, float* y, uint8_t* argmax,
                     int32_t N, int32_t H, int32_t W, int32_t C, int32_t act16, void* stream);
/* da [N,H,W,C] (gradient wrt relu(bn(c))) gathered from dy via argmax (every element written) */
int koaf_maxpool_bwd(const float* dy, const uint8_t* argmax, float* da, int32_t N, int32_t H,
                     int32_t W, int32_t C, void* stream);
int koaf_gap_fwd(const float* y, float* out, int32_t N, int32_t HW, int32_t C, int32_t act16, void* stream);
int koaf_gap_bwd(const float* dout, float* dy, int32_t N, int32_t HW, int32_t C, void* stream);

/* ---- Input plumbing -------------------------------------------------------------------------- */
/* "b ch r c s -> (b s) ch r c" (_xrNmrMcP.py:209-210): x [B,R,C,S] -> out [B*S,R,C] */
int koaf_slice_fold(const float* x, float* out, int32_t B, int32_t R, int32_t Cc, int32_t S,
                    void* stream);
/* F.interpolate(scale 0.5, align_corners=False, (bi|tri)linear) == 2x average pooling
 * (preproc/_pt.py:189-192).  x [B,R,C,S] -> out [B,R/2,C/2,S/fs], fs in {1,2}; S==1 for XR. */
int koaf_downscale2(const float* x, float* out, int32_t B, int32_t R, int32_t Cc, int32_t S,
                    int32_t fs, void* stream);

/* F.interpolate(x, scale_factor, mode = linear | bilinear | trilinear, align_corners=False, recompute_scale_factor=True) for any
 * scale factor (preproc/_pt.py:175-192): x [BC, in_size...] -> out [BC, out_size...], ndim = 1 .. 3 spatial dimensions,
 * out_size[d] = floor(in_size[d] * scale[d]) (computed by the caller as torch does). */
int koaf_resize(const float* x, float* out, int64_t BC, int32_t ndim, const int32_t* in_size, const int32_t* out_size,
                void* stream);

/* ---- Transformer pieces (_core_trf.py) ------------------------------------------------------- */
/* nn.Linear: y[M,N] = x[M,K] w[N,K]^T + b (+residual).  Small grids (few tokens) run split-K: ws = workspace
 * of koaf_linear_ws(M, N, K) floats (0 = not needed; NULL ws falls back to the unsplit kernel). */
int64_t koaf_linear_ws(int32_t M, int32_t N, int32_t K);
int koaf_linear_fwd(const float* x, const float* w, const float* b, const float* residual,
                    float* y, float* ws, int32_t M, int32_t N, int32_t K, void* stream);
/* dx[M,K] = dy[M,N] w[N,K]  (w read K-major, no re-pack) (+residual); ws: koaf_linear_ws(M, K, N) floats */
int koaf_linear_dgrad(const float* dy, const float* w, const float* residual, float* dx, float* ws,
                      int32_t M, int32_t N, int32_t K, void* stream);
/* dw[N,K] = dy^T x ; db[N] = column sums of dy (db nullable; ws: koaf_colsum_ws(M,N) floats) */
int koaf_linear_wgrad(const float* dy, const float* x, float* dw, float* db, float* ws, int32_t M,
                      int32_t N, int32_t K, void* stream);
/* nn.LayerNorm(D) rows: y, save mean/rstd (_core_trf.py:190-192,198,202) */
int koaf_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                       float* mean, float* rstd, int32_t rows, int32_t D, float eps, void* stream);
/* dx; dgamma/dbeta written (part: workspace of koaf_layernorm_bwd_ws floats) */
int64_t koaf_layernorm_bwd_ws(int32_t rows, int32_t D);
int koaf_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                       const float* rstd, float* dx, float* dgamma, float* dbeta, float* part,
                       int32_t rows, int32_t D, void* stream);
/* Attention core (_core_trf.py:170-180): qkv [B,n,3*h*d] with '(qkv h d)' split; scale applied to
 * QK^T; attn [B,h,n,n] emitted (it is returned by the reference, :182); out [B,n,h*d].  One launch
 * for n <= 512 and d % 4 == 0 (scores kept in LDS); larger n: two batched GEMMs + koaf_softmax_rows. */
int koaf_attention_fwd(const float* qkv, float* attn, float* out, int32_t B, int32_t n, int32_t h,
                       int32_t d, float scale, void* stream);
/* dqkv (every element written) from dout; ws: workspace of B*h*n*n floats (dS).  */
int koaf_attention_bwd(const float* dout, const float* qkv, const float* attn, float* dqkv,
                       float* ws, int32_t B, int32_t n, int32_t h, int32_t d, float scale,
                       void* stream);
int koaf_softmax_rows(float* x, int64_t rows, int32_t n, void* stream);
int koaf_softmax_bwd_rows(float* dp, const float* p, int64_t rows, int32_t n, float scale,
                          void* stream);
/* exact (erf) GELU, nn.GELU default (_core_trf.py:146) */
int koaf_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int koaf_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);
int koaf_relu_fwd(const float* x, float* y, int64_t n, void* stream);
int koaf_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
/* inverted dropout with a counter-based generator: y = x * keep(seed, i) / (1-p).  The same call
 * with x := dy is the backward.  (nn.Dropout / nn.Dropout2d on (N,C,1,1); streams differ from
 * torch's by construction -- SURVEY a10.)  */
/* epoch (nullable, both dropouts): device-resident step counter folded into the seed -- a captured (HIP-graph) step replays with
 * frozen arguments, the counter (koaf_counter_add) is what changes from step to step */
int koaf_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, const int64_t* epoch, void* stream);
int koaf_counter_add(int64_t* counter, int64_t delta, void* stream);   /* *counter += delta on the device */
/* nn.Dropout2d on an NHWC map x [N][HW][C] (_xrNmrMcP.py:62-72 with with_gap false): one draw per (image, channel),
 * generator index n*C + c -- identical to koaf_dropout on the pooled [N][C] output.  Backward = same call on dy. */
int koaf_dropout2d(const float* x, float* y, int32_t N, int32_t HW, int32_t C, float p, uint64_t seed,
                   const int64_t* epoch, void* stream);
/* out = a + b */
int koaf_add(const float* a, const float* b, float* out, int64_t n, void* stream);
/* column sums of x [rows][C] -> out [C] (bias gradients); part: koaf_colsum_ws floats or NULL */
int64_t koaf_colsum_ws(int32_t rows, int32_t C);
int koaf_colsum(const float* x, float* out, int32_t rows, int32_t C, float* part, void* stream);

/* ---- FocalLoss (_losses.py:89-108): loss = mean|sum( -(1-pt)^gamma * logpt ), logpt = -F.cross_entropy(x, t, weight, 'none') ----
 * logits [B,C,S] -- S = product of the spatial dims of a (b, ch, d0, d1, ...) input, 1 for (b, ch) --, target int64 [B,S],
 * class_weight [C] or NULL; writes the scalar loss and dlogits (= d loss / d logits, same layout).
 * ws: koaf_loss_ws(B, S) floats or NULL.  Up to 8192 elements (the models' (b, 2) logits) one block does everything; beyond that,
 * with a workspace, the same arithmetic runs on a grid of 4096-element blocks whose partial sums are added by one block in index
 * order (two fixed-order stages: the result does not depend on scheduling; without a workspace the one block walks everything). */
int64_t koaf_loss_ws(int32_t B, int64_t S);
int koaf_focal_loss(const float* logits, const int64_t* target, const float* class_weight, float* loss, float* dlogits,
                    int32_t B, int32_t C, int64_t S, float gamma, int32_t reduction_mean, float* ws, void* stream);
/* softmax-CE (CrossEntropyLoss wrapper = nn.CrossEntropyLoss(weight=class_weight), _losses.py:13-49): weighted mean
 * sum_i w[t_i] * (-log p_i[t_i]) / sum_i w[t_i] */
int koaf_ce_loss(const float* logits, const int64_t* target, const float* class_weight, float* loss, float* dlogits,
                 int32_t B, int32_t C, int64_t S, float* ws, void* stream);

/* ---- torch.optim.Adam (coupled L2) over a flat arena (_optimizers.py:47-52) ------------------ */
/* vmax (nullable): amsgrad -- the running maximum of the second moment, updated in place and used in the denominator */
int koaf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, double beta1,
                   double beta2, float eps, float weight_decay, int32_t step, int32_t adamw,
                   const float* hyper, float* vmax, void* stream);
/* Device-resident optimizer step state for captured (HIP-graph) train steps: ++*step; hyper[3] = {lr, lr / (1 - beta1^step),
 * sqrt(1 - beta2^step)} from the device scalars; hand `hyper` to koaf_adam_step (its host lr / step are then ignored). */
int koaf_adam_hyper(int32_t* step, const float* lr, double beta1, double beta2, float* hyper, void* stream);
int koaf_fill(float* p, float value, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KOAF_H */
