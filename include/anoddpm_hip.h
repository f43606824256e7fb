/*
 * anoddpm_hip.h -- C ABI of libanoddpm_hip.so, the MI355X (gfx950) implementation of the
 * AnoDDPM hot path.  Plain pointers and sizes only: no torch / C++ types cross this boundary.
 *
 * The upstream reference (Julian-Wyatt/AnoDDPM) is pure Python and has no FFI of its own; the
 * drop-in boundary its callers see is the Python module surface (GaussianDiffusion / UNet /
 * simplex).  This header is the layer directly below that surface: each entry point replaces
 * the reference code cited next to it.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every function returns ANODDPM_OK (0) or a negative ANODDPM_E* code; nothing throws,
 *     nothing allocates device memory, nothing synchronises (safe under hipGraph capture)
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream)
 *   - device pointers are marked [dev]; host pointers [host]
 *   - activations are NHWC fp32 ("pixels x channels"); an image is H*W pixels
 */
#ifndef ANODDPM_HIP_H
#define ANODDPM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANODDPM_OK 0
#define ANODDPM_EINVAL (-1)   /* bad argument (shape/alignment/range)          */
#define ANODDPM_ELAUNCH (-2)  /* HIP reported a launch / runtime error          */
#define ANODDPM_ENODEV (-3)   /* no HIP device visible                          */

int anoddpm_abi_version(void);          /* bumps whenever a struct below changes */
const char *anoddpm_last_error(void);   /* [host] text of the last failure on this thread */
int anoddpm_device_count(void);
int anoddpm_struct_size(int32_t which); /* sizeof of the n-th *_args struct, in declaration order */

/* ------------------------------------------------------------------ simplex ------------ */

/* simplex.py:166-192 (_init): seed -> permutation + gradient-index tables.  HOST function.
 * `seed` is the already int64-wrapped seed.  perm / pgi3: [host] int16[256] each. */
int anoddpm_simplex_perm_init(int64_t seed, int16_t *perm, int16_t *pgi3);

/* simplex.py:75-93 (rand_3d_fixed_T_octaves), :37-54 (rand_3d_octaves), :833-840 (_noise3a),
 * :321-830 (_noise3), :202-208 (_extrapolate3): multi-octave OpenSimplex-3D field.
 *   out[s][y][x] = sum_o persistence^o * noise3(x/f_o, y/f_o, z_s/f_o),  f_o = frequency/2^o
 * evaluated in fp64 in the reference's exact operation order (bit-exact), octave 0 first.
 * One launch fills `nslices` slices of H x W.
 *   zvals   [dev] int64[nslices] integer z index of each slice (timesteps, or 0..D-1);
 *           NULL means z_s = z0 + s
 *   tables  [dev] int16: table set n at tables + n*512 = {perm[256], pgi3[256]}
 *   table_sel [dev] int32* or NULL: *table_sel = index of the table set for this launch
 *           (lets a captured graph walk through pre-generated per-step tables)
 *   table_slice_stride: table set used by slice s = base + s*table_slice_stride (0 = shared)
 *   out_slice_stride: elements between consecutive slices in `out` (>= H*W)
 * _f64 stores doubles (Simplex_CLASS API); _f32 stores the fp64 result rounded to fp32
 * (what generate_simplex_noise's assignment into a float tensor does, GaussianDiffusion.py:125). */
typedef struct {
    void *out;                 /* [dev] double* or float* */
    const int64_t *zvals;      /* [dev] or NULL */
    const int16_t *tables;     /* [dev] */
    const int32_t *table_sel;  /* [dev] or NULL */
    int64_t z0;
    int64_t out_slice_stride;
    int32_t nslices, H, W;
    int32_t table_slice_stride;
    int32_t table_sel_scale;   /* table index = (*table_sel) * table_sel_scale + ... */
    int32_t octaves;
    double persistence;
    double frequency;
} anoddpm_simplex_args;

int anoddpm_simplex3_octaves_f64(const anoddpm_simplex_args *a, void *stream);
int anoddpm_simplex3_octaves_f32(const anoddpm_simplex_args *a, void *stream);

/* simplex.py:833-840 (_noise3a) / :31-35 (noise3, noise3array): one octave on arbitrary fp64
 * coordinate vectors, out[z][y][x] = noise3(X[x], Y[y], Z[z]); all pointers [dev]; tables = one
 * 512-entry set. */
int anoddpm_simplex3_grid_f64(double *out, const double *X, int32_t nx, const double *Y, int32_t ny,
                              const double *Z, int32_t nz, const int16_t *tables, void *stream);

/* 2-D OpenSimplex on a SQUARE n x n grid (upstream's _noise2a / rand_2d_octaves are only well defined for square
 * shapes: simplex.py:315-318 indexes noise[i * y.size + j], :69 adds a (W,H) array to a (H,W) field).
 *   octaves: out[i][j] = sum_o persistence^o * noise2(j / f_o, i / f_o), f_o = frequency / 2^o   (simplex.py:56-73)
 *   grid:    out[i][j] = noise2(X[j], Y[i])                                                       (simplex.py:311-318)
 * tables: the int16[512] table of anoddpm_simplex_perm_init (only perm[256] is used).  Bit-exact fp64. */
int anoddpm_simplex2_octaves_f64(double *out, int32_t n, const int16_t *tables, int32_t octaves,
                                 double persistence, double frequency, void *stream);
int anoddpm_simplex2_grid_f64(double *out, const double *X, const double *Y, int32_t n, const int16_t *tables,
                              void *stream);

/* ------------------------------------------------------------------ diffusion ---------- */

/* GaussianDiffusion.py:361-371 (sample_q) and :373-382 (sample_q_gradual):
 *   out = ca[t[b]] * x + cb[t[b]] * noise        (separate mul, mul, add: no FMA)
 * x, noise, out: [dev] fp32 [B][n]; t: [dev] int64[B]; ca, cb: [dev] fp32[T] (the reference's
 * fp64 tables rounded to fp32, which is what extract() does after its gather, :32-36). */
int anoddpm_q_sample(float *out, const float *x, const float *noise, const int64_t *t,
                     const float *ca, const float *cb, int32_t B, int64_t n, int32_t T,
                     void *stream);

/* GaussianDiffusion.py:269-318 (p_mean_variance + sample_p) with the model output given:
 *   pred_x0 = clamp(c_recip[t]*x_t - c_recipm1[t]*eps, -1, 1)
 *   mean    = c_coef1[t]*pred_x0 + c_coef2[t]*x_t
 *   x_prev  = mean + (t != 0) * sigma[t] * noise ,  sigma = exp(0.5*model_log_variance)
 * all fp32 with the reference's operation order (bit-exact vs. its CPU path).
 * x_prev may alias x_t.  pred_x0 and mean_out may be NULL.  noise may be NULL (treated as 0). */
typedef struct {
    float *x_prev;
    float *pred_x0;
    float *mean_out;
    const float *x_t;
    const float *eps;
    const float *noise;
    const int64_t *t;          /* [dev] int64[B] */
    const float *c_recip;      /* sqrt_recip_alphas_cumprod      fp32[T] */
    const float *c_recipm1;    /* sqrt_recipm1_alphas_cumprod    fp32[T] */
    const float *c_coef1;      /* posterior_mean_coef1           fp32[T] */
    const float *c_coef2;      /* posterior_mean_coef2           fp32[T] */
    const float *c_sigma;      /* exp(0.5*log(model_var))        fp32[T] */
    int64_t n;                 /* elements per sample */
    int32_t B, T;
} anoddpm_p_update_args;

int anoddpm_p_sample_update(const anoddpm_p_update_args *a, void *stream);

/* Advance a device-resident chain state after a reverse step: t[b] -= 1 for all b, *step += 1.
 * (The reference rebuilds t on the host every step, GaussianDiffusion.py:351-352.) */
int anoddpm_chain_advance(int64_t *t, int32_t B, int32_t *step, void *stream);

/* ------------------------------------------------------------------ UNet ops ----------- */

/* Operation codes for anoddpm_run_ops: one UNet forward is a flat list of these. */
enum {
    ANODDPM_OP_IGEMM = 1,        /* anoddpm_igemm_args       */
    ANODDPM_OP_GN_STATS = 2,     /* anoddpm_gn_args          */
    ANODDPM_OP_SOFTMAX = 3,      /* anoddpm_softmax_args     */
    ANODDPM_OP_RESAMPLE = 4,     /* anoddpm_resample_args    */
    ANODDPM_OP_LINEAR = 5,       /* anoddpm_linear_args      */
    ANODDPM_OP_POSEMB = 6,       /* anoddpm_posemb_args      */
    ANODDPM_OP_STEM = 7,         /* anoddpm_stem_args        */
    ANODDPM_OP_LAYOUT = 8,       /* anoddpm_layout_args      */
    ANODDPM_OP_CHAN_STATS = 9,   /* anoddpm_chan_stats_args  */
    ANODDPM_OP_GN_FINALIZE = 10, /* anoddpm_gn_finalize_args */
    ANODDPM_OP_HEAD = 11,        /* anoddpm_head_args        */
    /* 12 / 14 are the profiler's slots for the Winograd F(2x2) / F(4x4) launches of ANODDPM_OP_IGEMM, 13 for its cfg 5 launches */
    /* training step (backward twins and per-step weight packing) */
    ANODDPM_OP_WGRAD3 = 16,      /* anoddpm_wgrad_args        */
    ANODDPM_OP_WGRAD1 = 17,      /* anoddpm_wgrad1_args       */
    ANODDPM_OP_GN_BWD = 18,      /* anoddpm_gn_bwd_args       */
    ANODDPM_OP_PACK = 19,        /* anoddpm_pack_args         */
    ANODDPM_OP_SOFTMAX_BWD = 20, /* anoddpm_softmax_bwd_args  */
    ANODDPM_OP_TRANSPOSE = 21,   /* anoddpm_transpose_args    */
    ANODDPM_OP_LINEAR_BWD = 22,  /* anoddpm_linear_bwd_args   */
    ANODDPM_OP_STEM_BWD = 23,    /* anoddpm_stem_bwd_args     */
    ANODDPM_OP_HEAD_BWD = 24,    /* anoddpm_head_bwd_args     */
    ANODDPM_OP_COLSUM_FOLD = 25, /* anoddpm_colsum_fold_args  */
    ANODDPM_OP_ATTENTION = 26,   /* anoddpm_attention_args    */
    ANODDPM_OP_PACK_BATCH = 27,  /* anoddpm_pack_batch_args   */
    ANODDPM_OP_LINEAR_BWD_BATCH = 28, /* anoddpm_linear_bwd_batch_args */
    ANODDPM_OP_DROPOUT = 29,     /* anoddpm_dropout_args      */
    ANODDPM_OP_MAX = 32
};

/* Implicit-GEMM convolution / GEMM on the f32 MFMA pipe (v_mfma_f32_32x32x2_f32: exact fp32).
 *   out[z][p][n] = alpha * sum_{tap,k} A(z, p shifted by tap, k) * Bmat(tap, k, n)
 *                  + bias[n] + temb[b][n] + res[z][p][n]
 * Replaces: nn.Conv2d 3x3 / 1x1 (UNet.py:172,193,200,280,387), nn.Conv1d k=1 (UNet.py:115,117),
 * the two einsums of QKVAttention (UNet.py:148-152), and -- fused into the A load -- GroupNorm32
 * apply + SiLU (UNet.py:170-171,190-191,386,409-411), AvgPool2d / nearest x2 (UNet.py:70,89)
 * and torch.cat([h, skip]) (UNet.py:402: the two sources are read in place).
 * z = blockIdx.z decomposes as b = z / heads, head = z % heads. */
typedef struct {
    /* A operand: up to two NHWC sources concatenated along channels */
    const float *a0, *a1;           /* a1 NULL when single source */
    const float *gn_scale, *gn_shift; /* [B][K] per-sample per-channel affine or NULL */
    /* B operand */
    const float *bmat;              /* packed weights, or activations for b_mode 1/2 */
    /* epilogue */
    const float *bias;              /* [N] or NULL */
    const float *temb;              /* [B][temb_ld] or NULL */
    const float *res;               /* residual, same layout as out, or NULL */
    float *out;
    float *ws;                      /* split-K workspace [ksplit][Z][P][N] or NULL */
    int64_t a0_bs, a0_hs, a1_bs, a1_hs;   /* batch / head strides (floats) */
    int64_t b_bs, b_hs;
    int64_t o_bs, o_hs, r_bs, r_hs;
    int32_t c0, c1;                 /* channels taken from a0 / a1 (K = c0 + c1) */
    int32_t a0_ld, a1_ld;           /* floats between consecutive pixels */
    int32_t H, W;                   /* OUTPUT image dims; P = H*W                */
    int32_t ks;                     /* 1 or 3 (square, stride 1, pad ks/2)       */
    int32_t a_mode;                 /* 0 same-res, 1 source is half-res (nearest x2), 2 source is double-res (avg 2x2) */
    int32_t act;                    /* 1: SiLU on the A operand (after the affine) */
    int32_t b_mode;                 /* 0 packed [taps][K/4][N][4]; 1 rows [N][K] (ldb); 2 rows [K][N] (ldb) */
    int32_t ldb;
    int32_t N;
    int32_t temb_ld;
    int32_t out_ld, res_ld;
    int32_t B, heads;               /* Z = B*heads */
    int32_t ksplit;                 /* >= 1 */
    int32_t cfg;                    /* 0: 128x128 tile, 1: 64x64 tile, 2: Winograd F(2x2,3x3): ks 3, a_mode 0/1, H,W % 16 == 0,
                                       bmat = transformed weights [16][K/4][N][4] (G g G^T); 3: Winograd F(4x4,3x3): as 2 with
                                       N % 64 == 0, ksplit 1, bmat [36][K/4][N][4]; statistics: one row per 16x16 patch;
                                       4: streaming 1x1 for large maps (pointwise.hip): ks 1, a_mode 0, b_mode 0, heads 1, ksplit 1,
                                       no gn / act / stats, K % 128 == 0, K <= 512, c0 % 32 == 0, H*W % 32 == 0, N % 64 == 0;
                                       5: small maps without split-K (smallmap.hip), see anoddpm_smallmap_tile;
                                       6: Winograd F(2x2,3x3) on 16x16 / 32x32 maps without split-K (wino23s.hip), see anoddpm_wino23s_tile;
                                       bmat as cfg 2;
                                       7: OPT-IN side configuration, never chosen by default: cfg 3's layer with every fp32 operand split into
                                       three bf16 pieces and six products on v_mfma_f32_16x16x32_bf16 (winograd43b.hip): NOT the reference's
                                       arithmetic class.  N % 128 == 0, K % 32 == 0 per source, ksplit 1; bmat = anoddpm_pack_wino43_bf16x3 */
    float alpha;
    int32_t gn_ld;                  /* row length of gn_scale/gn_shift (= K) */
    float *stats;                   /* or NULL: per-channel partial sums of the OUTPUT, [B][tiles*2][N][2]
                                       {sum, sum of squares} per wave-row of each pixel tile (needs ksplit==1,
                                       heads==1); consumed by anoddpm_gn_finalize -- GroupNorm statistics
                                       without re-reading the tensor.  With ksplit > 1 the split-K reduction
                                       emits them instead, as [B][stats_rows][N][2] */
    int32_t stats_rows;             /* split-K only: number of pixel slabs (= rows) of the statistics */
    /* Split-K tail with the consumer's GroupNorm folded in (ksplit > 1, heads == 1, N % 128 == 0; `stats` must be NULL).  With
     * tail_csum set, the tail launch is partitioned by (GroupNorm group, image) instead of pixel slabs: a workgroup folds the K
     * slabs of its channels over ALL pixels of its image, so it owns complete per-channel sums -- it writes them to tail_csum and,
     * when tail_gamma is set, finishes GroupNorm(tail_groups, N + tail_c1) of torch.cat([out, other], 1) on the spot (UNet.py:409-411,
     * 402): scale / shift of its group's channels, the other source's channels taken from ITS folded sums.  One launch replaces the
     * statistics rows + anoddpm_gn_finalize of the plain tail.  (N + tail_c1) / tail_groups must be a multiple of 4. */
    double *tail_csum;              /* [B][N][2] {sum, sum of squares} over the image, fp64 */
    const double *tail_other;       /* [B][tail_c1][2] of the second source, or NULL */
    const float *tail_gamma, *tail_beta;   /* [N + tail_c1], or NULL: folded sums only */
    float *tail_scale, *tail_shift; /* [B][N + tail_c1] */
    float *tail_mean, *tail_rstd;   /* optional [B][tail_groups] (training) */
    int32_t tail_c1, tail_groups;
    float tail_eps;
    /* cfg 5 / 6 (smallmap.hip, wino23s.hip), and cfg 3 with fp64 sums (fold_fmt* == 1, see stats_csum): the GroupNorm of the A operand FINISHED IN THE KERNEL'S PROLOGUE from its producers' statistics
     * (UNet.py:409-411 over torch.cat([h, skip], 1)) -- the arguments of anoddpm_gn_finalize, consumed in place: no finalize
     * launch between producer and consumer.  With fold_gamma set, gn_scale / gn_shift are ignored.  fp64 fold in a fixed order
     * (rows ascending per channel, channels ascending per group), biased variance, fold_eps. */
    const float *fold_stats0, *fold_stats1;   /* as anoddpm_gn_finalize_args.stats0 / stats1 */
    const float *fold_gamma, *fold_beta;      /* [K], or NULL: no fold */
    int32_t fold_rows0, fold_rows1, fold_fmt0, fold_fmt1;
    int32_t fold_groups;
    float fold_eps;
    /* cfg 3 only: res_mode 1 = the residual is the block input at HALF resolution and is repeated 2x2 on the read (the
     * `x_upd = Upsample(x)` skip path of an up-sampling ResBlock, UNet.py:196-198, 89: F.interpolate(scale_factor=2, mode="nearest")
     * is never materialised); res: [B][(H/2)*(W/2)][res_ld], r_bs its batch stride.  0: res has the output's resolution. */
    int32_t res_mode;
    /* cfg 3 only, ksplit == 1, heads == 1, `stats` NULL (round 6): per-channel {sum, sum of squares} of the OUTPUT, added to
     * [B][N][2] fp64 with device-scope atomic adds, one pair per workgroup and channel (fp32 sums over the workgroup's 256 pixels,
     * as the `stats` rows hold them).  The buffer must be ZERO before the launch (anoddpm_posemb_args.zero clears a whole plan's
     * buffers at the start of a forward).  It is the fmt-1 statistics source of anoddpm_gn_finalize / fold_*: a cfg 3 consumer
     * finishes the GroupNorm in its prologue from it (fold_* with fold_fmt0 = fold_fmt1 = 1 -- the only formats cfg 3 folds),
     * so that no finalize launch sits between two F(4x4,3x3) layers.  fp64 sums of fp32 partials: the order of the atomic adds
     * changes the result by at most an ulp of the double. */
    double *stats_csum;
    /* cfg 3 on the 128-channel grid (anoddpm_f43_channel_sliced(...) == 1), ksplit == 1, heads == 1, plain operand (no gn / act),
     * no bias / temb / res (round 6, training): the launch is the DATA GRADIENT da of a 3x3 convolution whose operand was
     * a = SiLU(GroupNorm(x)) (diffusion_training.py:102 through UNet.py:170-172, 190-193), and the epilogue -- which holds da in
     * registers -- also performs the REDUCTION pass of anoddpm_gn_silu_backward: it reads x (the GroupNorm's input, one value per
     * output it stores; two concatenated sources as there), forms dy = da * silu'(gamma * xhat + beta), xhat = (x - mean) * rstd,
     * and writes gnb_partial[b][tile][c] = { sum dy, sum dy * xhat } over the workgroup's 16 x 16 pixels (fp32 sums of 256 values,
     * stored as fp64): exactly the `partial` rows of anoddpm_gn_bwd_args with nslab = (H / 16) * (W / 16), which is then called
     * with partial_ready = 1 and runs its fold + elementwise launches only.  One pass over x and da less per layer.
     * N (this launch's output channels) = gnb_c0 + the second source's channels; gnb_c0 % 16 == 0. */
    double *gnb_partial;            /* or NULL: plain launch */
    const float *gnb_x0, *gnb_x1;   /* the GroupNorm's input sources, NHWC at the OUTPUT resolution of this launch */
    const float *gnb_gamma, *gnb_beta;   /* [N] */
    const float *gnb_mean, *gnb_rstd;    /* [B][gnb_groups] */
    int64_t gnb_x0_bs, gnb_x1_bs;
    int32_t gnb_c0, gnb_x0_ld, gnb_x1_ld, gnb_groups;
} anoddpm_igemm_args;

int anoddpm_igemm(const anoddpm_igemm_args *a, void *stream);
/* 1 when a cfg 3 launch of this shape (ksplit 1) runs on the channel-sliced 128-channel kernel (winograd43r.hip) -- the one that
 * takes gnb_partial -- 0 when the launcher picks 64-channel workgroups (grids that would leave CUs idle, N % 128 != 0) */
int anoddpm_f43_channel_sliced(int32_t H, int32_t W, int32_t N, int32_t B);

/* cfg 5 of anoddpm_igemm: contractions on maps of <= 256 pixels WITHOUT split-K (smallmap.hip) -- the batch is folded into the
 * GEMM's M, a workgroup owns TM output rows x TN channels over all of K' = taps * K and its eight waves split K' (fold through
 * LDS, fixed order): one launch per layer instead of main + split-K tail (+ GroupNorm finalize, see fold_*).  ks 1 or 3, a_mode 0,
 * b_mode 0, heads 1, ksplit 1, H * W <= 256, W in {4, 8, 16}, K % 16 == 0, 16 <= K <= 1024, N % 32 == 0.  `stats` rows: one per
 * TM-pixel tile, [B][P / TM][N][2].  Returns the tile the launch will use as (TM / 16) * 16 + TN / 16, or 0 when the shape is
 * not taken (the caller then uses cfg 0 / 1 / 2). */
int anoddpm_smallmap_tile(int32_t ks, int32_t H, int32_t W, int32_t K, int32_t c0, int32_t N, int32_t B);

/* Weights of cfg 7: OIHW [N][K][3][3] fp32 -> U = G g G^T of F(4x4,3x3) (fp64 transform, rounded once to fp32) split into three bf16
 * planes, out = [3 pieces][36][K/8][N][8] bf16 (3 * 36 * N * K * 2 bytes).  K % 8 == 0. */
int anoddpm_pack_wino43_bf16x3(const float *w, void *out, int32_t N, int32_t K, void *stream);

/* cfg 6 of anoddpm_igemm: the 3x3 convolutions on 16x16 and 32x32 maps as Winograd F(2x2,3x3) WITHOUT split-K (wino23s.hip):
 * a workgroup owns 8x8 output pixels of one image x 32 or 64 channels over all of K, its eight waves split the sixteen transform
 * positions; one launch instead of main + split-K tail (+ GroupNorm finalize: fold_* as cfg 5).  a_mode 0 / 1, b_mode 0, heads 1,
 * ksplit 1, K % 32 == 0, 64 <= K <= 1024, N % 32 == 0; bmat = the F(2x2,3x3) weights of cfg 2.  `stats` rows: one per workgroup
 * tile, [B][(H / 8) * (W / 8)][N][2].  Returns the channel tiles per workgroup (2 or 4), or 0 when the shape is not taken. */
int anoddpm_wino23s_tile(int32_t H, int32_t W, int32_t K, int32_t c0, int32_t N, int32_t B, int32_t a_mode);

/* GroupNorm statistics -> per-sample per-channel affine (UNet.py:409-411 / nn.GroupNorm(32,C),
 * eps 1e-5, biased variance), over up to two concatenated NHWC sources:
 *   scale[b][c] = rstd[b,g(c)] * gamma[c];  shift[b][c] = beta[c] - mean[b,g(c)] * scale[b][c]
 * `partial` is scratch: double[B][nslab][64]. */
typedef struct {
    const float *a0, *a1;
    const float *gamma, *beta;      /* [C] */
    float *scale, *shift;           /* [B][C] */
    double *partial;
    int64_t a0_bs, a1_bs;
    int32_t c0, c1, a0_ld, a1_ld;
    int32_t P;                      /* pixels per image */
    int32_t B, groups, nslab;
    float eps;
} anoddpm_gn_args;

int anoddpm_gn_stats(const anoddpm_gn_args *a, void *stream);

/* Per-channel partial sums of an NHWC tensor in the format the igemm epilogue emits:
 * stats[b][slab][c] = {sum, sum of squares} over the slab's pixels (fp32 pairs). */
typedef struct {
    const float *a;
    float *stats;                   /* [B][nslab][C][2] */
    int64_t a_bs;
    int32_t C, a_ld, P, B, nslab;
} anoddpm_chan_stats_args;

int anoddpm_chan_stats(const anoddpm_chan_stats_args *a, void *stream);

/* GroupNorm(32, C) affine from per-channel partial sums of one or two concatenated sources
 * (UNet.py:409-411 over torch.cat([h, skip], 1), UNet.py:402): fp64 fold, biased variance, eps. */
typedef struct {
    const float *stats0, *stats1;   /* [B][rows][c][2]; stats1 NULL when single source */
    const float *gamma, *beta;      /* [C] */
    float *scale, *shift;           /* [B][C] */
    int32_t rows0, rows1, c0, c1;
    int32_t P, B, groups;
    float eps;
    float *mean_out, *rstd_out;     /* optional [B][groups]: saved for anoddpm_gn_silu_backward (training) */
    int32_t fmt0, fmt1;             /* 0: statsN = fp32 rows as above; 1: statsN = ONE row of fp64 pairs [B][c][2] (the tail_csum of
                                       anoddpm_igemm_args; rowsN is ignored) */
} anoddpm_gn_finalize_args;

int anoddpm_gn_finalize(const anoddpm_gn_finalize_args *a, void *stream);

/* Row softmax in place (UNet.py:151): x[r][0..L) <- softmax(x[r][:]); rows = B*heads*L. */
typedef struct {
    float *x;
    int64_t rows;
    int32_t L;
} anoddpm_softmax_args;

int anoddpm_softmax_rows(const anoddpm_softmax_args *a, void *stream);

/* Fused QKVAttention core (UNet.py:137-153; QKVAttentionLegacy channel order): per (image, head)
 *     weight = softmax(scale * q^T k),   a = weight v
 * in one launch: qkv [B][L][3*heads*ch] holds, for head h, q at channel h*3*ch, k at +ch, v at +2*ch (the output of the to_qkv
 * convolution, UNet.py:115,122); out [B][L][heads*ch]; probs (or NULL) receives the softmax [B*heads][L][L] (the training
 * backward reads it).  scale = 1/sqrt(ch) (the reference scales q and k by ch^-1/4 each, UNet.py:147-150).
 * Needs L % 16 == 0, 16 <= L <= 1024 (the score rows of a 16-query block live in LDS), ch a power of two in [16, 512].
 * Other shapes: three anoddpm_igemm / anoddpm_softmax_rows launches (b_mode 1 / 2). */
typedef struct {
    const float *qkv;
    float *out;
    float *probs;
    int32_t B, L, heads, ch;
    float scale;
} anoddpm_attention_args;

int anoddpm_attention(const anoddpm_attention_args *a, void *stream);

/* 2x resampling of an NHWC tensor (the x_upd path of ResBlock, UNet.py:177-181,207):
 * mode 1: nearest x2 up (in H x W -> out 2H x 2W); mode 2: 2x2 average pool (in -> H/2 x W/2); mode 3: the even pixels
 * (2i, 2j) -> H/2 x W/2, i.e. a stride-2 convolution's outputs picked from the stride-1 result (Downsample, UNet.py:60-75);
 * mode 4: the adjoint of mode 3 -- in H x W -> out 2H x 2W with out[2i][2j] = in[i][j] and zeros elsewhere (the gradient of the
 * stride-2 convolution's output placed on the stride-1 grid).
 * scale (0 is read as 1) multiplies the result and accumulate != 0 adds it to `out`: the backward of one mode is the
 * other one scaled -- d(avg pool) = nearest-up * 0.25, d(nearest-up) = avg pool * 4 (the sum of the four children). */
typedef struct {
    const float *in;
    float *out;
    int32_t B, H, W, C;             /* INPUT dims */
    int32_t mode;
    float scale;
    int32_t accumulate;
    /* optional second output of mode 2: out_act = 2x2 average of silu(gn_scale[b][c] * in + gn_shift[b][c]) -- the operand the
     * first 3x3 convolution of a down block consumes (UNet.py:170-171, 70, 206), produced in the same pass over `in` so that
     * the convolution itself can run on the Winograd kernels */
    const float *gn_scale, *gn_shift;   /* [B][C] or NULL */
    float *out_act;                     /* [B][H/2][W/2][C] or NULL */
} anoddpm_resample_args;

int anoddpm_resample2x(const anoddpm_resample_args *a, void *stream);

/* Small-batch linear layer (UNet.py:273-275 time MLP; :185-188 per-block embedding projections,
 * all blocks batched into one launch by concatenating their weight rows):
 *   out[b][n] = act_out( sum_k act_in(in[b][k]) * w[n][k] + bias[n] ); 16 batch rows per launch (larger B: one launch per 16 rows). */
typedef struct {
    const float *in;                /* [B][K] */
    const float *w;                 /* [N][K] (nn.Linear layout) */
    const float *bias;              /* [N] or NULL */
    float *out;                     /* [B][N] */
    int32_t B, K, N;
    int32_t act_in, act_out;        /* 1 = SiLU */
} anoddpm_linear_args;

int anoddpm_linear_small(const anoddpm_linear_args *a, void *stream);

/* Sinusoidal timestep features (UNet.py:50-57): out[b][i] = sin(t_b*f_i), out[b][half+i] = cos(t_b*f_i),
 * f_i = exp(-i * ln(1e4)/half) in fp32 (table supplied by the caller so that it is bit-identical
 * to the host-computed one; t_b*f_i is one fp32 multiply as in torch.outer). */
typedef struct {
    const int64_t *t;               /* [dev] int64[B] */
    const float *freqs;             /* [dev] fp32[dim/2]: f_i, precomputed once on the host */
    float *out;                     /* [B][dim] */
    int32_t B, dim;
    float scale;
    /* optional (round 6): `zero_doubles` doubles at `zero` are cleared by the same launch -- the statistics accumulators
     * (anoddpm_igemm_args.stats_csum) of a whole forward plan, whose first op this is */
    double *zero;
    int64_t zero_doubles;
} anoddpm_posemb_args;

int anoddpm_posemb(const anoddpm_posemb_args *a, void *stream);

/* Stem convolution (UNet.py:280): 3x3, pad 1, Cin <= 4 NCHW input -> Cout NHWC output.
 * w: [9][Cin][Cout] fp32. */
typedef struct {
    const float *x;                 /* [B][Cin][H][W] (NCHW, as the caller hands it over) */
    const float *w;
    const float *bias;
    float *out;                     /* [B][H][W][Cout] */
    int32_t B, H, W, Cin, Cout;
    /* optional (Cin <= 2, W % 8 == 0, x 16-byte aligned): GroupNorm partial sums of the output, [B][stats_rows][Cout][2] floats
     * {sum, sum of squares}, one row per workgroup range -- the row format of anoddpm_chan_stats / the contraction epilogues.
     * stats_rows must be anoddpm_stem_stats_rows() of the shape. */
    float *stats;
    int32_t stats_rows;
} anoddpm_stem_args;

/* rows per image the fused stem statistics use for this shape, 0 when the shape has no fused form */
int anoddpm_stem_stats_rows(int H, int W, int Cin, int Cout);

int anoddpm_conv_stem(const anoddpm_stem_args *a, void *stream);

/* Head convolution (UNet.py:384-388): silu(GroupNorm affine(x)) -> 3x3 pad 1 -> Cout <= 4 channels.
 * x: NHWC [B][H][W][C]; w: [9][C][Cout]; out: NCHW [B][Cout][H][W] (the caller's layout). HBM-bound. */
typedef struct {
    const float *x;
    const float *w;
    const float *bias;              /* [Cout] or NULL */
    const float *gn_scale, *gn_shift; /* [B][C] */
    float *out;
    int32_t B, H, W, C, Cout;
} anoddpm_head_args;

int anoddpm_conv_head(const anoddpm_head_args *a, void *stream);

/* Layout change at the API edge: NHWC [B][P][C] -> NCHW [B][C][P] (C small, UNet output). */
typedef struct {
    const float *in;
    float *out;
    int32_t B, P, C;
    int32_t in_ld;
} anoddpm_layout_args;

int anoddpm_nhwc_to_nchw(const anoddpm_layout_args *a, void *stream);

/* Flat op list: the native executor behind UNetModel.forward (UNet.py:390-406). */
typedef struct {
    int32_t code;                   /* ANODDPM_OP_* */
    int32_t flags;
    const void *args;               /* [host] pointer to the matching *_args struct */
} anoddpm_op;

int anoddpm_run_ops(const anoddpm_op *ops, int32_t n, void *stream);

/* Per-class kernel timing with HIP events on the launch stream (bench.py roofline leg).
 * enable=1 starts recording one event pair per launched op; collect() synchronises the events
 * and returns accumulated milliseconds and launch counts per op code (arrays of ANODDPM_OP_MAX = 32); ANODDPM_OP_IGEMM
 * launches that run the Winograd kernels are booked under index 12 (cfg 2) / 14 (cfg 3) instead of 1, and the Winograd-domain
 * launches of ANODDPM_OP_WGRAD3 (algo 1) under index 15 instead of 16. */
int anoddpm_prof_enable(int32_t enable);
int anoddpm_prof_active(void);             /* 1 while event recording is on (graph capture must be avoided) */
int anoddpm_prof_collect(double *ms_per_code, int64_t *launches_per_code);
/* the ops recorded up to the last anoddpm_prof_collect, in launch order: profiler slot + milliseconds each; returns their count */
int anoddpm_prof_list(int32_t *codes, float *ms, int32_t cap);

/* ------------------------------------------------------------------ training ----------- */

/* Fused AdamW + EMA over a flat fp32 parameter buffer (diffusion_training.py:75,105,107 and
 * UNet.py:423-427): decoupled weight decay, bias-corrected moments, then
 * ema = decay*ema + (1-decay)*p.  grad_scale multiplies g first (global-norm clip factor). */
typedef struct {
    float *p, *m, *v, *ema;
    const float *g;
    const float *grad_scale;        /* [dev] fp32 scalar or NULL */
    int64_t n;
    float lr, beta1, beta2, eps, weight_decay, ema_decay;
    int32_t step;                   /* 1-based */
} anoddpm_adamw_args;

int anoddpm_adamw_ema(const anoddpm_adamw_args *a, void *stream);

/* Global gradient norm and clip factor of a flat fp32 buffer (clip_grad_norm_(params, max_norm), diffusion_training.py:104):
 *   out[0] = sum of squares, out[1] = its square root, out[2] = min(max_norm / (out[1] + 1e-6), 1)   (1 when max_norm <= 0)
 * out: [dev] fp32[3]; workspace: [dev] >= 2048 doubles.  Two-stage fp64 reduction in a fixed order, no atomics: replicas that
 * hold identical gradients compute identical clip factors. */
int anoddpm_sumsq(const float *g, int64_t n, float *out, double *workspace, float max_norm, void *stream);

/* ------------------------------------------------------------------ anomaly map + segmentation counts
 * One pass over an image and its `navg` reconstructions (the (t_distance, avg) chains of detection_A/B):
 *   mean over the chains                                  GaussianDiffusion.py:517, 572
 *   sqerr = (mean - real)^2                                detection.py:229; evaluation.py:31
 *   mse_img = sqerr*2 - 1 ; thr_img = (mse_img > 0)*2 - 1  GaussianDiffusion.py:518-520, 581-583; evaluation.py:13-15
 *   pred = (sqerr > threshold)                             detection.py:232 (threshold 0.5)
 * and the sums behind dice_coeff / IoU / precision / recall / FPR / PSNR (evaluation.py:26-76) as
 * counts[b][ANODDPM_ANOMALY_NCOUNTS] (fp64):
 *   0 sum(pred)  1 sum(mask)  2 sum(pred*mask)  3 #(mask==1&pred==1)  4 #(mask==1&pred==0)  5 #(mask==0&pred==1)
 *   6 #(mask==0&pred==0)  7 #(mask!=0 & pred!=0)  8 #(mask!=0 | pred!=0)  9 sum(sqerr)  10 max(real)  11 reserved
 * recon: [navg] slices of n floats per image, slice stride recon_as, image stride recon_bs (elements).
 * Any of mean / sqerr / mse_img / thr_img / pred may be NULL.  mask may be NULL (treated as all zero).
 * workspace: ANODDPM_ANOMALY_BLOCKS * B * NCOUNTS doubles [dev].  Deterministic (fixed-order fold). */
#define ANODDPM_ANOMALY_NCOUNTS 12
#define ANODDPM_ANOMALY_BLOCKS 64
typedef struct anoddpm_anomaly_args {
    const float *recon;
    const float *real;
    const float *mask;
    float *mean, *sqerr, *mse_img, *thr_img, *pred;
    double *counts;                 /* [B][ANODDPM_ANOMALY_NCOUNTS] */
    double *workspace;
    int64_t workspace_doubles;
    int64_t n;                      /* pixels (x channels) per image */
    int64_t recon_as, recon_bs;
    int32_t navg, B;
    float threshold;
} anoddpm_anomaly_args;

int anoddpm_anomaly_map(const anoddpm_anomaly_args *a, void *stream);

/* Variational-bound terms of one reverse step (GaussianDiffusion.py:384-397 calc_vlb_xt, and the two MSE curves of
 * calc_total_vlb :445-478): per sample b
 *   out[0*B + b] = mean_flat( t==0 ? -discretised_gaussian_log_likelihood(x_0; mean, 0.5*logvar)
 *                                  : normal_kl(true_mean, post_logvar, mean, model_logvar) ) / ln 2
 *   out[1*B + b] = mean_flat((pred_x_0 - x_0)^2)
 *   out[2*B + b] = mean_flat((predict_eps_from_x_0(x_t, t, pred_x_0) - noise)^2)        (0 when noise == NULL)
 * with pred_x_0 = clamp(c_recip[t]*x_t - c_recipm1[t]*eps, -1, 1) and mean = c_coef1[t]*pred_x_0 + c_coef2[t]*x_t.
 * Tables are fp32 copies of the reference's fp64 tables (sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod,
 * posterior_mean_coef1/2, posterior_log_variance_clipped, log(append(posterior_variance[1], betas[1:]))).
 * workspace: >= 64 * B * 3 doubles [dev].  Deterministic. */
typedef struct anoddpm_vlb_args {
    const float *x0, *xt, *eps, *noise;
    const int64_t *t;
    const float *c_recip, *c_recipm1, *c_coef1, *c_coef2, *c_post_logvar, *c_model_logvar;
    float *pred_x0;                 /* optional [B][n] */
    float *out;                     /* [3][B] */
    double *workspace;
    int64_t workspace_doubles;
    int64_t n;
    int32_t B, T;
} anoddpm_vlb_args;

int anoddpm_vlb_terms(const anoddpm_vlb_args *a, void *stream);

/* Training loss of one batch and its gradient (GaussianDiffusion.py:399-417 calc_loss, :419-434 p_loss; the hybrid loss's
 * VLB term is calc_vlb_xt :384-397):
 *   kind 0 (l1):  per_sample[b] = mean_flat(|eps - noise|)
 *   kind 1 (l2):  per_sample[b] = mean_flat((eps - noise)^2)
 *   kind 2 (hybrid): vlb[b] as anoddpm_vlb_terms' out[0]; per_sample[b] = vlb[b] + mean_flat((eps - noise)^2)
 *   total[0] = mean_b(per_sample[b] * weights[b])                          (weights NULL = 1: loss_weight "none")
 * anoddpm_loss_backward writes d_eps[b][i] = d( sum_b g_per[b] per_sample[b] + sum_b g_vlb[b] vlb[b] + g_total total ) / d eps[b][i]
 * (each upstream gradient may be NULL = 0) -- the tensor loss.backward() hands to the UNet's backward.
 * Element math fp32, per-sample sums fp64 folded in a fixed order (deterministic).  workspace: >= 64 * B * 2 doubles [dev]. */
typedef struct anoddpm_loss_args {
    const float *eps, *noise;       /* [B][n] */
    const float *x0, *xt;           /* [B][n], hybrid only */
    const int64_t *t;               /* [B], hybrid only */
    const float *weights;           /* [B] or NULL */
    const float *c_recip, *c_recipm1, *c_coef1, *c_coef2, *c_post_logvar, *c_model_logvar;   /* fp32[T], hybrid only */
    float *per_sample;              /* [B] */
    float *vlb;                     /* [B] or NULL */
    float *total;                   /* [1] or NULL */
    double *workspace;
    int64_t workspace_doubles;
    const float *g_per, *g_vlb, *g_total;   /* backward: upstream gradients [B], [B], [1]; each may be NULL */
    float *d_eps;                   /* backward: [B][n] */
    int64_t n;
    int32_t B, T, kind;
} anoddpm_loss_args;

int anoddpm_loss_forward(const anoddpm_loss_args *a, void *stream);
int anoddpm_loss_backward(const anoddpm_loss_args *a, void *stream);

/* nn.Dropout(p) of ResBlock.out_layers (UNet.py:192: GroupNorm32 -> SiLU -> Dropout -> conv) in training mode.  The mask is a
 * counter-based hash of (seed, element index): the same (seed, index) gives the same mask in the forward and the backward launch,
 * no mask tensor exists.  keep probability 1 - p, kept values scaled by 1 / (1 - p) (torch semantics; the random STREAM is this
 * library's own -- no implementation reproduces torch's philox offsets, parity tests inject the mask they read back).
 *   mode 0 (forward):  out[b][i] = keep ? silu(x[b][i] * gn_scale[b][c] + gn_shift[b][c]) / (1 - p) : 0      (c = i % C)
 *   mode 1 (backward): out[b][i] = keep ? x[b][i] / (1 - p) : 0                                             (x = d(out), may alias out)
 * x, out: NHWC [B][n]; n = pixels * C. */
typedef struct anoddpm_dropout_args {
    const float *x;
    float *out;
    const float *gn_scale, *gn_shift;   /* [B][C], mode 0 */
    int64_t n;                      /* elements per image */
    uint64_t seed;
    int32_t B, C, mode;
    float p;
} anoddpm_dropout_args;

int anoddpm_dropout(const anoddpm_dropout_args *a, void *stream);

/* ------------------------------------------------------------------ backward twins (training, diffusion_training.py:102)
 * Weight gradient of a 3x3 / stride 1 / pad 1 convolution whose input the forward consumed through the fused operand
 * load of anoddpm_igemm (GroupNorm-apply + SiLU, nearest x2, two-source concat):
 *   dw[co][ci][ky][kx] (OIHW) = sum_{b,y,x} dy[b][y][x][co] * A[b][y+ky-1][x+kx-1][ci]
 * The data gradient needs no entry point of its own: it is anoddpm_igemm on dy with the spatially flipped,
 * channel-transposed weights.  ws: [B * (W/TW) * ceil(H/band)][9][K][N] floats (TW = the largest of 32, 16, 8, 4, 2 dividing W);
 * the partial tiles are folded in a fixed order (deterministic).  accumulate != 0 adds into dw. */
typedef struct anoddpm_wgrad_args {
    const float *a0, *a1;           /* the conv's input sources (NHWC), a1 NULL when single source */
    const float *gn_scale, *gn_shift; /* [B][gn_ld] per-sample per-channel affine or NULL */
    const float *dy;                /* [B][H*W][dy_ld] gradient w.r.t. the conv output */
    float *dw;                      /* [N][K][3][3] */
    float *ws;
    int64_t ws_floats;
    int64_t a0_bs, a1_bs, dy_bs;    /* batch strides (floats) */
    int32_t c0, c1, a0_ld, a1_ld, dy_ld;
    int32_t H, W;                   /* OUTPUT image dims */
    int32_t N, B;
    int32_t a_mode;                 /* 0 same-res, 1 source is half-res (nearest x2), 2 source is double-res (2x2 average) */
    int32_t act;                    /* 1: SiLU after the affine */
    int32_t gn_ld;
    int32_t band;                   /* image rows per work item (split-K granularity) */
    int32_t accumulate;
    float *colsum;                  /* optional [items][N]: per work item sum over its pixels of dy (items of one image are
                                       consecutive: item = (b * nband + band) * nseg + segment) -> bias / embedding gradients */
    int32_t algo;                   /* 0: direct (nine-tap MFMA tiles, any shape above); 1: Winograd F(4x4,3x3) domain (wgrad43.hip),
                                       the adjoint of the cfg-3 forward kernel: dU = sum_tiles V (.) Z, dg = G^T dU G -- needs
                                       gn + act == 1, a_mode 0 / 1, H % 8 == 0, W % 16 == 0, K % 32 == 0, N % 64 == 0,
                                       c0 % 16 == 0, B <= 15; ws: anoddpm_wgrad43_groups(...) * 9 * K * N floats; colsum:
                                       [B][anoddpm_wgrad43_colsum_items(...)][N], one row per workgroup set and tile row (the
                                       kernel sums over its patches of an image); `band` is ignored */
    float *dimg, *dbias;            /* algo 1 only, optional (need colsum): dimg[B][N] = per-image sums of dy (embedding gradient),
                                       dbias[N] += their sum over the images -- anoddpm_colsum_fold done by N / 32 extra
                                       workgroups of the fold launch instead of a launch of its own (round 6) */
} anoddpm_wgrad_args;

int anoddpm_conv3x3_wgrad(const anoddpm_wgrad_args *a, void *stream);
int anoddpm_wgrad43_groups(int32_t K, int32_t N, int32_t B, int32_t H, int32_t W);   /* workgroup sets (workspace slabs) of algo 1 */
int anoddpm_wgrad43_colsum_items(int32_t K, int32_t N, int32_t B, int32_t H, int32_t W);   /* column-sum rows per image of algo 1 */

/* Device-side weight packing for the 3x3 kernels (training re-packs after every optimizer step).  w: OIHW [N][K][3][3].
 * mode 0: direct layout [9][I/4][O][4]; mode 1: Winograd F(2x2,3x3) U = G g G^T as [16][I/4][O][4]; mode 2: Winograd
 * F(4x4,3x3) as [36][I/4][O][4].  bwd != 0 packs the
 * data-gradient weights W'[o=k][i=n][a][b] = w[n][k][2-a][2-b] (I = N, O = K), else I = K, O = N. */
int anoddpm_pack_conv3x3(const float *w, float *out, int32_t N, int32_t K, int32_t mode, int32_t bwd, void *stream);

/* Backward of a = SiLU(GroupNorm32(x)) (UNet.py:170-171,190-191,409-411; act == 0: GroupNorm alone, UNet.py:113) as the
 * fused operand load of anoddpm_igemm consumed it, over up to two concatenated NHWC sources:
 *   y = gamma*xhat + beta, xhat = (x - mean)*rstd;   dy = da * silu'(y);
 *   dgamma[c] += sum_{b,p} dy*xhat;   dbeta[c] += sum_{b,p} dy;
 *   dx = rstd * (gamma*dy - mean_g(gamma*dy) - xhat * mean_g(gamma*dy*xhat))          (means over the group, per image)
 * `da` is the gradient w.r.t. the tensor the conv read: with a_mode 1 (forward nearest-x2 of a) it is at twice the
 * source resolution and the four children are summed; with a_mode 2 (forward 2x2 average) it is at half resolution and
 * each source pixel takes a quarter of its parent.  dx0 / dx1 receive the gradient w.r.t. the sources (acc_dx != 0:
 * added to what is there -- gradient fan-in of skip connections and residuals).  Three launches: per-channel partial
 * sums (deterministic two-stage fp64 fold), then the elementwise pass.
 * partial: double[B][nslab][C][2]; coef: float[B][C][4] scratch. */
typedef struct anoddpm_gn_bwd_args {
    const float *x0, *x1;           /* forward sources (x1 NULL when single) */
    const float *da;                /* [B][Pa][C] with row length da_ld */
    const float *gamma, *beta;      /* [C] */
    const float *mean, *rstd;       /* [B][groups] from anoddpm_gn_finalize */
    float *dx0, *dx1;
    float *dgamma, *dbeta;          /* [C], accumulated into */
    double *partial;
    float *coef;
    int64_t x0_bs, x1_bs, da_bs, dx0_bs, dx1_bs;
    int32_t c0, c1, x0_ld, x1_ld, da_ld, dx0_ld, dx1_ld;
    int32_t Hs, Ws;                 /* SOURCE image dims (P = Hs*Ws) */
    int32_t B, groups, nslab;
    int32_t act, a_mode;
    int32_t acc_dx;                 /* bit 0: add into dx0, bit 1: add into dx1 (else overwrite) */
    const float *dres;              /* optional [B][P][C] (row length dres_ld, batch stride dres_bs): added to the source
                                       gradient -- the identity residual of a block (UNet.py:216 / :125) */
    int64_t dres_bs;
    int32_t dres_ld;
    int32_t partial_ready;          /* 1: `partial` was written by the producer of da (anoddpm_igemm_args.gnb_partial): no reduction launch */
} anoddpm_gn_bwd_args;

int anoddpm_gn_silu_backward(const anoddpm_gn_bwd_args *a, void *stream);

/* Weight gradient of a pointwise (1x1 / Conv1d k=1) convolution (UNet.py:115,117,200) whose input was consumed through
 * the fused operand load (optional GroupNorm-apply, optional SiLU, two-source concat):
 *   dw[n][k] (+)= sum_{b,p} dy[b][p][n] * A[b][p][k]            dbias[n] += sum_{b,p} dy[b][p][n]
 * Contraction over pixels on the fp32 matrix pipe: a workgroup owns a 128 x 128 (k, n) tile of one work item
 * (image, `span` consecutive pixels); partial tiles go to ws[item][K][N] and are folded in a fixed order.
 * ws: >= (B * ceil(P/span)) * (K*N + N) floats. */
typedef struct anoddpm_wgrad1_args {
    const float *a0, *a1;
    const float *gn_scale, *gn_shift; /* [B][gn_ld] or NULL */
    const float *dy;                /* [B][P][dy_ld] */
    float *dw;                      /* [N][K] */
    float *dbias;                   /* [N], accumulated into; or NULL */
    float *ws;
    int64_t ws_floats;
    int64_t a0_bs, a1_bs, dy_bs;
    int32_t c0, c1, a0_ld, a1_ld, dy_ld;
    int32_t P, N, B;
    int32_t act, gn_ld;
    int32_t span;                   /* pixels per work item, multiple of 32 */
    int32_t accumulate;             /* != 0: add into dw */
} anoddpm_wgrad1_args;

int anoddpm_wgrad_pointwise(const anoddpm_wgrad1_args *a, void *stream);

/* Device-side weight packing (training re-packs after every optimizer step).
 *   kind 0: 3x3 direct   (anoddpm_pack_conv3x3 mode 0)       w OIHW [N][K][3][3]
 *   kind 1: 3x3 Winograd (anoddpm_pack_conv3x3 mode 1)
 *   kind 2: pointwise    w [N][K] -> [K/4][N][4]; bwd != 0: the data-gradient matrix W'[i=n][o] = w[n][k0 + o],
 *           o < kc, packed [N/4][kc][4] (a column range: the two sources of a concatenated input get separate matrices)
 *   kind 3: small conv   w OIHW [N][K][3][3] -> [9][K][N] (stem / head kernels)
 *   kind 4: plain copy of N*K floats
 *   kind 5: 3x3 Winograd F(4x4,3x3) (anoddpm_pack_conv3x3 mode 2): [36][I/4][O][4] */
typedef struct anoddpm_pack_args {
    const float *w;
    float *out;
    int32_t N, K, kind, bwd, k0, kc;
} anoddpm_pack_args;

int anoddpm_pack_weights(const anoddpm_pack_args *a, void *stream);

/* All pack jobs of a model in one launch: `jobs` is a DEVICE array of njobs anoddpm_pack_args (each validated on the host the way
 * anoddpm_pack_weights does), `block0` a DEVICE array of njobs + 1 prefix sums of anoddpm_pack_job_blocks(job) (256-thread blocks),
 * nblocks = block0[njobs]. */
typedef struct anoddpm_pack_batch_args {
    const anoddpm_pack_args *jobs;
    const int32_t *block0;
    int32_t njobs, nblocks;
} anoddpm_pack_batch_args;

int anoddpm_pack_batch(const anoddpm_pack_batch_args *b, void *stream);
int64_t anoddpm_pack_job_blocks(const anoddpm_pack_args *a);

/* Backward of the row softmax (UNet.py:151), in place on the incoming gradient:
 *   ds[r][j] = p[r][j] * (dp[r][j] - sum_j dp[r][j] p[r][j]) */
typedef struct anoddpm_softmax_bwd_args {
    const float *p;                 /* softmax output [rows][L] */
    float *dp;                      /* in: dL/dp, out: dL/ds */
    int64_t rows;
    int32_t L;
} anoddpm_softmax_bwd_args;

int anoddpm_softmax_rows_backward(const anoddpm_softmax_bwd_args *a, void *stream);

/* out[z][j][i] = in[z][i][j] for Z square L x L matrices (attention weights / their gradients, so that every
 * backward contraction of QKVAttention reads its A operand row-major). */
typedef struct anoddpm_transpose_args {
    const float *in;
    float *out;
    int32_t Z, L;
} anoddpm_transpose_args;

int anoddpm_transpose_square(const anoddpm_transpose_args *a, void *stream);

/* Backward of anoddpm_linear_small with act_out == 0: y = act_in(x) W^T + b.
 *   dw[n][k] (+)= sum_b dy[b][n] * act_in(x[b][k]);  db[n] (+)= sum_b dy[b][n];
 *   dx[b][k] (+)= act_in'(x[b][k]) * sum_n dy[b][n] * w[n][k]            (dx may be NULL) */
typedef struct anoddpm_linear_bwd_args {
    const float *x;                 /* [B][K] the forward INPUT (before act_in) */
    const float *w;                 /* [N][K] */
    const float *dy;                /* [B][N] */
    float *dw, *db, *dx;
    int32_t B, K, N, act_in;
    int32_t acc_w, acc_x;           /* != 0: add into dw+db / dx */
} anoddpm_linear_bwd_args;

int anoddpm_linear_small_backward(const anoddpm_linear_bwd_args *a, void *stream);

/* The same for njobs linear layers that share the input x (the per-block embedding projections, UNet.py:185-188, 213: one launch in
 * the forward): `jobs` is a DEVICE array of anoddpm_linear_bwd_args of which w, dy, dw, db and N are read; x, B, K, act_in, acc_w
 * come from this header.  dx (or NULL) receives act_in'(x) * sum over the jobs, folded in job order through ws [njobs][B][K]. */
typedef struct anoddpm_linear_bwd_batch_args {
    const anoddpm_linear_bwd_args *jobs;
    const float *x;
    float *dx;
    float *ws;
    int32_t njobs, max_n;           /* max_n: the largest N among the jobs */
    int32_t B, K, act_in, acc_w, acc_x;
} anoddpm_linear_bwd_batch_args;

int anoddpm_linear_small_backward_batch(const anoddpm_linear_bwd_batch_args *h, void *stream);

/* Backward of anoddpm_conv_stem (UNet.py:280): dw (OIHW [Cout][Cin][3][3]) += x (*) dy, db[co] += sum dy, and optionally
 * dx (NCHW) = dy (*) flipped w.  ws: >= nblk * (Cin * 9 + 1) * Cout floats with nblk = B * ceil(H*W / 1024). */
typedef struct anoddpm_stem_bwd_args {
    const float *x;                 /* [B][Cin][H][W] */
    const float *w;                 /* OIHW (the parameter itself) */
    const float *dy;                /* [B][H*W][Cout] */
    float *dw, *db;                 /* accumulated into */
    float *dx;                      /* [B][Cin][H][W] or NULL (overwritten) */
    float *ws;
    int64_t ws_floats;
    int32_t B, H, W, Cin, Cout;
} anoddpm_stem_bwd_args;

int anoddpm_conv_stem_backward(const anoddpm_stem_bwd_args *a, void *stream);

/* Backward of anoddpm_conv_head (UNet.py:384-388) up to the activated tensor a = silu(scale*x + shift):
 *   da[b][p][c] = sum_{o,tap} w[o][c][tap] * dy[b][o][p - off(tap)]           (NHWC, overwritten)
 *   dw[o][c][tap] += sum_{b,p} a[b][p + off(tap)][c] * dy[b][o][p];  db[o] += sum dy
 * ws: >= nblk * (9 * C + 1) * Cout floats with nblk = B * ceil(H*W / 512). */
typedef struct anoddpm_head_bwd_args {
    const float *x;                 /* [B][H*W][C] */
    const float *gn_scale, *gn_shift; /* [B][C] */
    const float *w;                 /* OIHW [Cout][C][3][3] (the parameter itself) */
    const float *dy;                /* [B][Cout][H][W] */
    float *da;                      /* [B][H*W][C] */
    float *dw, *db;                 /* accumulated into */
    float *ws;
    int64_t ws_floats;
    int32_t B, H, W, C, Cout;
} anoddpm_head_bwd_args;

int anoddpm_conv_head_backward(const anoddpm_head_bwd_args *a, void *stream);

/* Fold of the per-work-item column sums anoddpm_conv3x3_wgrad publishes ([B][ipb][N]: items of an image are
 * consecutive): dimg[b][n] = sum_items (written: the embedding-projection gradient, UNet.py:213) and
 * dbias[n] += sum_b dimg[b][n] (optional). */
typedef struct anoddpm_colsum_fold_args {
    const float *colsum;
    float *dimg;                    /* [B][N], written (scratch when only the bias gradient is wanted) */
    float *dbias;                   /* [N] or NULL, accumulated into */
    int32_t B, ipb, N;
} anoddpm_colsum_fold_args;

int anoddpm_colsum_fold(const anoddpm_colsum_fold_args *a, void *stream);

/* ------------------------------------------------------------------ MRI slice loader (dataset.py:575-643) ----------
 * Volume normalisation of MRIDataset.__getitem__ (dataset.py:585-592): clip to [mean - std, mean + 2 std] (population
 * std, fp64 two-pass), divide by the range, store fp32.  workspace: >= 2*256 + 4 doubles [dev]. */
int anoddpm_volume_normalise(const double *vol, int64_t n, float *out, double *workspace, void *stream);

/* Slice cut (dataset.py:621: image[:, s:s+1, :].reshape(X, Z)), optional RandomAffine resampling (PIL AFFINE / NEAREST:
 * affine = six 16.16 fixed-point coefficients per item, source = (a2 + a0 x + a1 y) >> 16, (a5 + a3 x + a4 y) >> 16, fill 0)
 * and torchvision CenterCrop(crop) with its zero padding: out[b][oy][ox] = img[oy + crop_top][ox - pad_left]. */
typedef struct anoddpm_mri_slice_args {
    const float *const *vols;       /* [dev] B pointers to resident fp32 volumes [X][Y_b][Z] */
    const int32_t *ydim;            /* [dev] Y_b */
    const int32_t *slice_idx;       /* [dev] */
    const long long *affine;        /* [dev] [B][6] or NULL (no augmentation) */
    float *out;                     /* [B][crop][crop] */
    int32_t B, X, Z, crop, pad_left, crop_top;
} anoddpm_mri_slice_args;

int anoddpm_mri_slice_prepare(const anoddpm_mri_slice_args *a, void *stream);

/* PIL Image.resize(BILINEAR) on fp32 images (what torchvision Resize does to a mode-"F" image), then optionally
 * Normalize: out = (v - mean) / std.  Coefficient tables are PIL's precompute_coeffs (host, fp64): for output index o the
 * taps in[kmin[o] .. kmin[o] + kn[o]) with weights k[o][0 .. kn[o]), row length kmax.  tmp: [B][in_h][out_w] floats. */
typedef struct anoddpm_resize_args {
    const float *in;                /* [B][in_h][in_w] */
    float *tmp, *out;               /* out: [B][out_h][out_w] */
    const double *kx;               /* [out_w][kmax_x] */
    const int32_t *kx_min, *kx_n;
    const double *ky;               /* [out_h][kmax_y] */
    const int32_t *ky_min, *ky_n;
    int32_t B, in_h, in_w, out_h, out_w, kmax_x, kmax_y;
    float mean, std;
    int32_t normalize;
} anoddpm_resize_args;

int anoddpm_resize_bilinear_pil(const anoddpm_resize_args *a, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ANODDPM_HIP_H */
