/* C ABI of libdm_hip.so: hand-written gfx950 (MI355X) kernels for the latent-diffusion
 * denoising hot path (U-Net noise predictor inside the DDPM sampling loop).
 *
 * The reference (sumanthyahoo9/diffusion_model) is pure Python/PyTorch and has no FFI: its
 * boundary for this path is the call `model(sample, context, time_emb)` made by
 * DDPMScheduler.sample (src/schedulers/ddpm_scheduler.py:203) plus the torch.nn operators the
 * model is built from.  Each entry point below therefore cites the reference operator call
 * site(s) it replaces.  All functions:
 *   - take raw device pointers, sizes and a hipStream_t; the compute entry points never allocate device memory,
 *     free or synchronise (so a caller may capture them into a hipGraph) and take their workspaces from the
 *     caller; the exceptions are the plumbing calls dm_graph_end / dm_event_create (which create runtime objects)
 *     and dm_event_elapsed_ms (which waits for the event);
 *   - return 0 (DM_OK) or a positive DM_ERR_* code (1 argument/shape, 2 HIP launch error, 3 alignment,
 *     4 attention head dimension without a kernel, 5 too large for a fixed scratch); a HIP runtime error is also
 *     named on stderr;
 *   - are re-entrant per stream.  Process-wide state is limited to: a per-kernel, per-device "dynamic LDS limit
 *     raised" flag; constant 16-byte zero / ones device buffers; the 1 MiB device scratch of dm_groupnorm_noalloc.
 * Activation matrices are row-major [rows][ld] with channels contiguous (row = (batch, y, x)
 * or (batch, token)); `dtype` selects fp32 storage/exact-f32 MFMA or bf16 storage/f32 accumulate.
 */
#ifndef DM_HIP_H
#define DM_HIP_H

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM_ABI_VERSION 13

#define DM_F32 0
#define DM_BF16 1

#define DM_EPI_PLAIN 0    /* C = acc (+bias) (+rowbias[batch]) (+R) */
#define DM_EPI_GEGLU 1    /* C[m][n] = (acc_v + b_v) * gelu_erf(acc_g + b_g), N counts both halves */
#define DM_EPI_NCHW_F32 2 /* plain epilogue, stored as fp32 (B, N, H*W) */
#define DM_EPI_QUICKGELU 3 /* plain epilogue, then x * sigmoid(1.702 x) (src/models/clip.py:114) */

#define DM_TILE_AUTO 0
#define DM_TILE_128x128 1
#define DM_TILE_64x128 2
#define DM_TILE_64x64 3
#define DM_TILE_256x128 4
#define DM_TILE_32x64 5
/* direct-to-LDS (global_load_lds) main loop, BMxBN block tile, Sn = LDS ring depth */
#define DM_TILE_D128x128_S2 11
#define DM_TILE_D128x128_S3 12
#define DM_TILE_D256x128_S3 13
#define DM_TILE_D64x128_S3 14
#define DM_TILE_D64x128_S4 15
#define DM_TILE_D64x64_S4 16
#define DM_TILE_D32x64_S4 17
#define DM_TILE_D256x64_S3 18
#define DM_TILE_D128x64_S4 19
#define DM_TILE_D256x64_S2 20
#define DM_TILE_D128x64_S2 21
#define DM_TILE_D64x64_S2 22
#define DM_TILE_D256x256_S2 23 /* bf16 only: 8 waves of 128x64, two per SIMD taking turns between fragment reads and
                                * MFMAs; half the operand bytes per MFMA of the 128x128 tile.  Whole k-tiles only
                                * (dense: kpad == k; conv: cin a multiple of 64), convolutions stride 1 / same size */
/* E: two LDS buffers, a k-tile's fragments are read into registers before its buffer is refilled with the
 * tile two steps ahead (two tiles in flight at the LDS footprint, hence occupancy, of a 2-stage ring) */
#define DM_TILE_E128x128 24
#define DM_TILE_E128x64 25
#define DM_TILE_E64x64 26
#define DM_TILE_E64x128 27
/* U: two k-tiles per barrier.  Four LDS stages as two halves of two k-tiles; the block waits for the pair it is about to
 * multiply, passes one barrier, requests the next pair into the other half and multiplies both tiles */
#define DM_TILE_U64x64 28
/* H: 128 x 128, bf16 stride-1 same-size 3x3 convolutions on images at most 32 pixels wide.  The activation of a 64-channel
 * chunk is held in LDS once - the tile's pixels and a one-pixel border, as two half-chunks of 32 channels - and the nine taps
 * read it at nine offsets, so a k-tile requests weights only (a third fewer requests than the ring loop, where every tap
 * fetches its own shifted copy).  Goes with conv_chunk_major = 2 (dm_conv_halo_supported says which geometries qualify). */
#define DM_TILE_H128x128 29
/* P: row-panel form for short contractions (bf16, dense, K = 320 or 640, N a multiple of 64, plain or GEGLU
 * epilogue): a block owns 64 / 128 rows, keeps them in registers for the whole launch and walks ALL column
 * tiles of 64 (weights streamed through an LDS ring), so the activations are fetched once and the
 * row sums of a following LayerNorm need no atomics.  For these tiles `splitk` (>= 1) is the number of column slices
 * the column tiles are dealt to (grid.y), not a split of k; no workspace.  P128 exists for K = 320 only. */
#define DM_TILE_P64 31
#define DM_TILE_P128 32

int dm_abi_version(void);

/* GEMM / implicit-GEMM conv3x3 with fused epilogue.
 * Replaces: nn.Linear (src/models/attention.py:19-20,51-54; src/models/blocks.py:109-110),
 * nn.Conv2d 1x1 (blocks.py:42,99,112), nn.Conv2d 3x3 pad 1 stride 1/2 (blocks.py:33,37;
 * src/models/unet.py:29,92,97,104,111; src/models/diffusion.py:32), F.interpolate(nearest,x2)
 * folded into the following conv (unet.py:33-34), the broadcast time add (blocks.py:67), the
 * residual adds (blocks.py:72,139,145,153,159) and GEGLU (blocks.py:150-151).
 * `w` is a packed weight matrix made by dm_pack_weight_order() + a gather on the caller side. */
typedef struct DmGemmDesc {
    const void* a;        /* activations [m][lda] (conv: NHWC image rows (b, y, x)) */
    long lda;
    const void* w;        /* packed weights, npad x kpad elements, rows in dm_pack_weight_order order, stored
                           * blocked as [npad/64][kpad/bk][64 rows][bk] with bk = 128 bytes of k (32 fp32 /
                           * 64 bf16): one k-tile of a 64-row group is a contiguous 8 KiB piece */
    void* c;              /* output [m][ldc] (or fp32 NCHW); c, r 16-byte aligned, ldc, ldr multiples of 16 bytes */
    long ldc;
    const float* bias;    /* [n] fp32 or NULL */
    const float* rowbias; /* [batch][ld_rowbias] fp32 or NULL; batch = row / rows_per_batch */
    long ld_rowbias;
    int rows_per_batch;
    const void* r;        /* residual [m][ldr], same dtype as c, or NULL */
    long ldr;
    int m, n, k;          /* k = 9*cin for conv */
    int npad, kpad;       /* packed weight extents */
    int dtype;
    int conv;             /* 0 dense, 1 conv3x3 pad 1 */
    int hin, win, hout, wout, cin, stride, up;
    int epi;
    int tile;             /* DM_TILE_* (DM_TILE_AUTO: library chooses tile and split) */
    int splitk;           /* used when tile != AUTO.  > 1: k is cut into `splitk` slices, one block each, summed by a
                           * second launch (plain epilogue only).  < 0: work-list form for the direct-to-LDS tiles:
                           * -splitk blocks share the (output tile, k-tile) list in equal runs; a tile cut between
                           * blocks is summed, in block order, by the last of them to arrive (any epilogue). */
    void* workspace;      /* split-K slabs, >= dm_gemm_workspace_bytes(); work-list form: 256-byte aligned,
                           * >= dm_gemm_streamk_workspace_bytes(), whose leading arrival counters must read zero
                           * at launch (the kernel leaves them zero) - do not share it with split-K slabs */
    long workspace_bytes;
    double* stats;        /* NULL, or per-GROUP sums of the STORED output for the GroupNorm that reads it next: output
                           * column n is channel stats_ch0 + n of the normalised tensor (the output may be a channel
                           * slice of it) and belongs to group g = (stats_ch0 + n) / stats_cpg;
                           * stats[(b*2 + 0)*ld_stats + g] += sum over the rows of sample b of c[m][n],
                           * stats[(b*2 + 1)*ld_stats + g] += the same of c[m][n]^2 (fp64 atomic adds, two per group
                           * and wave; the caller zeroes the buffer).  Needs rows_per_batch, the plain epilogue, no
                           * split-K reduction launch, and dm_gemm_stats_supported(tile, rows_per_batch, splitk) != 0. */
    long ld_stats;        /* >= number of groups of the normalised tensor */
    int stats_cpg;        /* channels per group */
    int stats_ch0;
    double* stats2;       /* NULL, or a second partition of the same columns (an encoder output that is normalised on
                           * its own by the next stage AND as the right-hand slice of a decoder's concatenated input):
                           * same meaning with its own ld / cpg / ch0; needs stats */
    long ld_stats2;
    int stats2_cpg;
    int stats2_ch0;
    /* LayerNorm folded into the GEMMs on either side of it (blocks.py:137-138,143-144,149-150):
     * ln_out: the producer adds, per output row m, sum and sum of squares of the stored row to
     *         ln_out[2m], ln_out[2m+1] (fp64 atomics; caller zeroes; plain epilogue; no split-K launch);
     * ln_in : the consumer takes the same array for its A rows and computes
     *         rstd_m * (a_m W'^T - mean_m * colsum(W')) + bias, where the caller has packed W' = W diag(gamma),
     *         passes ln_colsum[p] = sum_k W'_packed[p][k] (packed row order, npad entries, fp32) and has put
     *         beta W^T (+ b) into `bias`.  ln_eps is the LayerNorm's epsilon; the width is k. */
    const double* ln_in;
    const float* ln_colsum;
    float ln_eps;
    double* ln_out;
    /* Second operand appended to the k axis (NULL = none); direct-to-LDS tiles only; cin2 and k - cin2 (conv: cin)
     * multiples of the k-tile (32 fp32 / 64 bf16).
     * conv: a pointwise (1x1) term over a second tensor, k = 9*cin + cin2:
     *   out[m] += sum_c a2[m][c] * w[n][9*cin + c]  (stride-1, same-size convolutions).  Folds ResidualBlock's
     *   residual_layer (blocks.py:42,70-72) into its second convolution.
     * dense: columns k - cin2 .. k - 1 of the A row come from a2[m][0 .. cin2) instead of a (k == kpad).  Lets
     *   AttentionBlock's linear_geglu_2 -> += residue -> conv_output chain (blocks.py:152-159) run as one GEMM
     *   over [gelu product | residue] with pre-multiplied weights. */
    const void* a2;
    long lda2;
    int cin2;
    /* conv only: 0 = zero padding 1 on every side (nn.Conv2d(padding=1)); 1 = padding only below and right, none
     * above / left, i.e. F.pad(x, (0, 1, 0, 1)) followed by a padding-0 convolution - the stride-2 convolutions of
     * the VAE encoder (src/models/encoder.py:179-185).  Not with `up`, not with `a2`. */
    int pad_hi_only;
    /* GroupNorm (+ SiLU) of the A operand applied inside this launch (NULL = none; fp32 only): the tensor `a` is the
     * INPUT of nn.GroupNorm(gn_groups, C) (blocks.py:58-59,68-69,128; diffusion.py:44-45) and the product is taken
     * over silu(groupnorm(a)) (conv) / groupnorm(a) (dense) without that tensor ever being written.  gn_sums are the
     * per-group fp64 (sum, sumsq) of `a`, [batch][2][ld_gn], as accumulated through `stats` by the launch that wrote
     * it.  Every block turns them into per-channel scale / shift in LDS and applies them to its A fragments in
     * registers; padding taps stay zero, the appended operand `a2` is not normalised.  Needs a direct-to-LDS tile
     * (DM_TILE_D*), C = cin (conv) or k - cin2 (dense) a multiple of 32 and of gn_groups and at most 1280,
     * rows_per_batch a multiple of the tile's rows, stride 1 / same size / no upsampling, gn_silu = 1 for conv and
     * 0 for dense; dm_gemm_gn_supported() tells. */
    const double* gn_sums;
    long ld_gn;
    const float* gn_gamma;
    const float* gn_beta;
    float gn_eps;
    int gn_silu;
    int gn_groups;
    /* Opt-in arithmetic modes for fp32 problems (0 = exact-fp32 MFMA, the default); no counterpart in the reference,
     * which computes in fp32.  Direct-to-LDS tiles, not together with gn_sums.
     * 2: `w` is a two-term split-bf16 image - per packed row and k-tile of 32 values, 32 bf16 hi terms followed by 32
     *    bf16 lo terms (w ~ hi + lo, lo = bf16(w - hi)); same blocking and bytes as the fp32 image - and the launch
     *    forms a_hi w_hi + a_lo w_hi + a_hi w_lo on the bf16 MFMA with the fp32 activations split in registers (fp32
     *    accumulation; error of a product ~2^-16 instead of 2^-24).  Inside a k-tile the terms are stored in the k
     *    order 0-3, 16-19, 4-7, 20-23, 8-11, 24-27, 12-15, 28-31 (the eight values one lane group multiplies).
     * 3: three terms per operand (hi, mid, lo: 24 bits) and the six products down to 2^-16 of the leading one; what is
     *    dropped is 2^-24 of a product.  The image has two 128-byte rows per packed row and k-tile, stored
     *    [npad/64][kpad/32][2 planes][64 rows][128 B]: plane 0 = 32 hi | 32 mid terms, plane 1 = 32 lo terms | zeros. */
    int w_split;
    /* conv3x3 only.  0: the packed k axis runs tap-major (k = tap * cin + c: all channels of tap 0, then tap 1, ...).
     * 1: chunk-major - the channels are cut into k-tiles (64 bf16 values) and the nine taps of one chunk
     *    follow each other (k-tile index = chunk * 9 + tap), any pointwise operand still last.  The nine requests for a
     *    chunk then touch the same few image rows back to back and all but the first are served by the XCD's L2; in
     *    tap-major order a tap returns to a pixel only after cin / k-tile steps, long after the line has left L2.
     *    bf16 only (the exact-fp32 loop moves an eighth of the operand bytes per MFMA cycle and only pays the extra
     *    address arithmetic).  Needs stride 1, equal input and output size, no fused upsampling, padding 1, cin a whole
     *    number of k-tiles and a direct-to-LDS tile (not DM_TILE_128x128 .. DM_TILE_32x64). */
    /* 2: the order of DM_TILE_H128x128 (and only of it): per 64-channel chunk the 18 units (half h of 32 channels, tap t)
     *    in the order h-major, t-minor, 32 values each, i.e. k = chunk * 576 + (h * 9 + t) * 32 + e; pointwise operand last. */
    int conv_chunk_major;
} DmGemmDesc;

int dm_gemm(const DmGemmDesc* d, hipStream_t stream);
/* Whether DM_TILE_H128x128 takes a stride-1 same-size 3x3 convolution over m = batch * h * w output pixels of h x w images
 * with cin channels and row stride lda (elements).  No counterpart in the reference: a property of this library's tile. */
int dm_conv_halo_supported(int m, int h, int w, int cin, long lda);
void dm_gemm_plan(int dtype, int m, int n, int k, int epi, int* tile, int* splitk);
long dm_gemm_workspace_bytes(int m, int n, int splitk);
/* -1: the tile has no work-list form, or the problem has more output tiles (2M) than the counter area holds.  The
 * first 8 MiB of the workspace are the arrival counters (one per output tile; must read zero at launch, left zero),
 * the accumulator slabs follow; the split is the same for every launch, so launches of any shapes may share one
 * workspace. */
long dm_gemm_streamk_workspace_bytes(int tile, int m, int n, int blocks);
int dm_gemm_stats_supported(int tile, int rows_per_batch, int splitk);
/* 1 when the descriptor's problem can run on the row-panel tile `tile` (DM_TILE_P*) with `nsplit` column slices. */
int dm_gemm_panel_supported(const DmGemmDesc* d, int tile, int nsplit);
/* Feed-forward half of an AttentionBlock as one launch (bf16, C = 320):
 *   out = conv_output(linear_geglu_2(v * gelu(g)) + t) + x,  [v | g] = linear_geglu_1(layernorm_3(t))
 * replaces /root/reference/src/models/blocks.py:137-153 (layernorm_3, linear_geglu_1, the GELU gate, linear_geglu_2, the
 * residue add, conv_output and the long residue).  The 4C-wide product stays on the CU.  Operands are the ones the two
 * dm_gemm calls of the unfused form take: w1 = linear_geglu_1 packed with the GEGLU row interleave and the LayerNorm folded
 * in (bias1, ln_colsum; ln_in = per-row (sum, sumsq) of t), w2 = the matrix [W_c W_2 | W_c] over the row [product | t]
 * packed as a dm_gemm weight image of npad2 x 5C (bias2 = W_c b_2 + b_c), r = the block input x (may be NULL).
 * m must be a multiple of 128; with statistics (stats / stats2 as in DmGemmDesc) rows_per_batch must be a multiple of 128. */
typedef struct DmFfnDesc {
    const void* a; long lda;
    const void* w1; const float* bias1;
    const double* ln_in; const float* ln_colsum; float ln_eps;
    const void* w2; const float* bias2; int npad2;
    const void* r; long ldr;
    void* c; long ldc;
    int m, c_model, dtype;
    int rows_per_batch;
    double* stats; long ld_stats; int stats_cpg, stats_ch0;
    double* stats2; long ld_stats2; int stats2_cpg, stats2_ch0;
} DmFfnDesc;
int dm_ffn_geglu(const DmFfnDesc* d, hipStream_t stream);
/* 1 when dm_ffn_geglu takes a problem of this shape */
int dm_ffn_geglu_supported(int dtype, int m, int c_model, int rows_per_batch, int want_stats);
/* 1 when DmGemmDesc.gn_sums may be used with this tile: C = channels that are normalised (see above). */
int dm_gemm_gn_supported(int dtype, int tile, int conv, int c, int groups, int rows_per_batch, int splitk);
/* order[p] = source weight row of packed row p (or -1 for a zero row), for p in [0, npad). */
int dm_pack_weight_order(int n, int npad, int geglu, int* order_host);
/* Device-side packing of an ACTIVATION matrix into the `w` operand layout, so that a product of two activation
 * matrices runs on dm_gemm: dst (npad x kpad, blocked and row-permuted as DmGemmDesc.w) holds the n x k matrix
 * X[i][j] = src[i*ld + j] (transposed == 0) or src[j*ld + i] (transposed != 0), zero padded.  Used for the
 * single-head attention of the VAE (scores = Q K^T with K packed; output = P V with V packed transposed), whose head
 * width of 512 does not fit the flash kernel's LDS tiles (mirror of src/models/encoder.py:56-60). */
int dm_pack_rows(const void* src, long ld, int n, int k, int transposed, void* dst, int npad, int kpad, int dtype,
                 hipStream_t stream);

/* GroupNorm (+ optional SiLU) over NHWC rows.  Replaces nn.GroupNorm(32, C) + F.silu
 * (blocks.py:58-59,68-69,128; diffusion.py:44-45).  Two launches: per-slab (sum, sumsq)
 * partials, then normalise.  partial: fp32 [batch][nsplit][groups][2].
 * dm_groupnorm_apply_sums: one launch; the statistics come from the per-group fp64 sums that
 * dm_gemm accumulated while it wrote x (DmGemmDesc.stats: [batch][2][ld_sums], ld_sums >= groups). */
int dm_groupnorm_stats(const void* x, long ldx, int batch, int hw, int c, int groups, int dtype,
                       float* partial, int nsplit, hipStream_t stream);
int dm_groupnorm_apply(const void* x, long ldx, void* y, long ldy, int batch, int hw, int c, int groups,
                       int dtype, const float* partial, int nsplit, const float* gamma, const float* beta,
                       float eps, int silu, hipStream_t stream);
int dm_groupnorm_apply_sums(const void* x, long ldx, void* y, long ldy, int batch, int hw, int c, int groups,
                            int dtype, const double* sums, long ld_sums, const float* gamma, const float* beta,
                            float eps, int silu, hipStream_t stream);

/* Slab count the library itself uses for the two-launch GroupNorm (>= 16 rows per slab, <= 128 slabs). */
int dm_groupnorm_default_nsplit(int batch, int hw);

/* Resolves, for the CURRENT device, the address of the library's fixed device scratch (see
 * dm_groupnorm_noalloc).  Optional: the entry point that needs it does the same on its first call on a device;
 * call dm_init() once per device beforehand when that first call would fall inside a stream capture. */
int dm_init(void);

/* GroupNorm (+ SiLU) with NO allocation and no workspace argument.  No counterpart in the reference: it exists
 * for callers that cannot call hipMalloc (code under hipGraph capture, tight serving loops, harnesses that
 * forbid allocation).  Same result, bit for bit, as dm_groupnorm_stats + dm_groupnorm_apply with
 * nsplit = dm_groupnorm_default_nsplit(batch, hw); the slab partials live in a 1 MiB file-scope device array that
 * every device holds for the life of the process.  Returns DM_ERR_TOO_LARGE (5) when
 * batch * nsplit * groups * 2 floats exceed that array (2^18 floats; e.g. batch <= 512 at 32 groups).
 * All calls on one device share the array: calls that overlap in time on one device (different streams) race. */
int dm_groupnorm_noalloc(const void* x, long ldx, void* y, long ldy, int batch, int hw, int c, int groups,
                         int dtype, const float* gamma, const float* beta, float eps, int silu, hipStream_t stream);

/* In place x[r][:] = softmax(scale * x[r][:]) over `cols` columns (a multiple of 16 bytes) of `rows` rows.  The
 * softmax of the materialised-score attention above (F.softmax(w / sqrt(d)), src/models/attention.py:38-39). */
int dm_softmax_rows(void* x, long ld, int rows, int cols, float scale, int dtype, hipStream_t stream);

/* LayerNorm over the channel dimension of [rows][c].  Replaces nn.LayerNorm (blocks.py:137,143,149). */
int dm_layernorm(const void* x, long ldx, void* y, long ldy, int rows, int c, int dtype,
                 const float* gamma, const float* beta, float eps, hipStream_t stream);

/* softmax(q k^T / sqrt(d)) v for all heads, never materialising the score matrix.
 * Replaces attention.py:34-43 (self; causal != 0 adds the upper-triangular mask of attention.py:35-37,
 * used by the CLIP text encoder, and needs sq == sk) and attention.py:71-77 (cross).
 * q rows = (b, sq) at q + (b*sq + s)*ldq + h*d; k and v likewise with sk rows per sample (their
 * own leading dimensions).  V is read row-major as the projection wrote it; no transposed copy. */
int dm_attention(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                 int batch, int heads, int sq, int sk, int d, int causal, int dtype, hipStream_t stream);
/* dm_attention for fp32 tensors in the opt-in split-bf16 product mode (see DmGemmDesc.w_split): K, Q, V and the
 * probabilities are split into two bf16 terms in registers and both products run on the bf16 MFMA (three products per
 * fp32 product, fp32 accumulation, softmax in fp32).  Not the reference's arithmetic; same arguments and layouts as
 * dm_attention with dtype DM_F32.  Small wide-head problems that dm_attention runs with its head-dimension split keep
 * that exact-fp32 form. */
int dm_attention_split_f32(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                           int batch, int heads, int sq, int sk, int d, int causal, hipStream_t stream);

/* y[b][n] = sum_k f(x[b][k]) * w[n][k] + bias[n], fp32, f = SiLU when silu_in.  Replaces the
 * batch-sized Linears of the time path (src/models/embeddings.py:42-44; blocks.py:63-64). */
int dm_linear_rows(const float* x, long ldx, const float* w, const float* bias, float* y, long ldy, int batch,
                   int n, int k, int silu_in, hipStream_t stream);

/* Sinusoidal timestep features, [cos | sin], freqs = exp(-ln(max_period) i / half).
 * Replaces DDPMScheduler._get_timestep_embedding (ddpm_scheduler.py:212-244). */
int dm_timestep_embedding(const long long* timesteps, int batch, int dim, float max_period, float* out,
                          hipStream_t stream);

/* Reverse-process update, fp32, elementwise (ddpm_scheduler.py:99-138):
 *   x0 = (x - coef[0]*eps)/coef[1]; clamp to [-1,1] if clip; out = coef[2]*x0 + coef[3]*x (+ coef[4]*noise)
 * coef points at DEVICE memory: 5 floats per step; the row used is coef + 5 * (*step_index) when
 * step_index != NULL (device int, lets one captured graph serve every step), else coef itself.
 * noise may be NULL (t == 0); with step_index it is advanced by (*step_index) * n elements. */
int dm_ddpm_step(const float* x, const float* eps, const float* noise, float* out, long n, const float* coef,
                 const int* step_index, int clip, hipStream_t stream);
/* DDIM update (src/schedulers/ddim_scheduler.py:132-164): identical, except that the second term is
 * coef[3]*eps (the "direction pointing to x_t") instead of coef[3]*x. */
int dm_ddim_step(const float* x, const float* eps, const float* noise, float* out, long n, const float* coef,
                 const int* step_index, int clip, hipStream_t stream);
/* out = a[b]*x + c[b]*y with per-sample scalars gathered from device tables by timestep:
 * a = ta[t[b]], c = sign*tc[t[b]].  add_noise / get_velocity (base_scheduler.py:145-214). */
int dm_scale_add(const float* x, const float* y, float* out, int batch, long per_sample, const float* ta,
                 const float* tc, const long long* t, float sign_c, hipStream_t stream);
/* out = alpha*x + beta*y (+ gamma*z), fp32 scalars from the host (K-sampler steps, CFG combine). */
int dm_axpbypcz(const float* x, float alpha, const float* y, float beta, const float* z, float gamma, float* out,
                long n, hipStream_t stream);

/* K-sampler loop body with the per-step scalars in DEVICE tables indexed by *step_index (one captured graph serves
 * every step).  dm_ksampler_input: out[c][i] = scale_table[s] * x[i] for c < copies - the model input
 * latents * sampler.get_input_scale(), repeated for classifier-free guidance (scripts/inference.py:300-304).
 * dm_ksampler_step: guided = cfg_scale * (cond - uncond) + uncond with eps = [cond | uncond] when do_cfg
 * (inference.py:313-315), else guided = eps; out = x + coef[2s] * guided (+ coef[2s+1] * noise[s*n ..] when that
 * coefficient is non-zero): KEulerSampler.step (src/sampling/k_euler.py:78-111) and KEulerAncestralSampler.step
 * (src/sampling/k_euler_ancestral.py:80-127).  Separately rounded fp32 operations in the eager path's order. */
int dm_ksampler_input(const float* x, float* out, long n, int copies, const float* scale_table, const int* step_index,
                      hipStream_t stream);
int dm_ksampler_step(const float* x, const float* eps, const float* noise, float* out, long n, const float* coef,
                     const int* step_index, float cfg_scale, int do_cfg, hipStream_t stream);
/* KLMSSampler.step (src/sampling/k_lms.py:113-137) for the captured loop: the (guided, as in dm_ksampler_step) model
 * output of step s = *step_index is stored into history[(s % ring) * n ...] and
 * out = coef[s][0] * o_s + coef[s][1] * o_(s-1) + ... over the order[s] newest outputs (coef: [steps][max_order] fp32,
 * order: [steps] int32, ring >= max_order history slots of n floats).  The incoming latents do not enter the result,
 * as in the reference. */
int dm_klms_step(const float* eps, float* history, float* out, long n, const float* coef, const int* order,
                 int max_order, int ring, const int* step_index, float cfg_scale, int do_cfg, hipStream_t stream);

/* out[(b, s)][:] = table[tokens[b][s]][:] + pos[s][:] in the engine dtype.  Replaces CLIPEmbedding.forward
 * (src/models/clip.py:52-57): nn.Embedding lookup + learned positional add.  tokens: int64 [rows],
 * rows = batch*seq; table fp32 [vocab][dim]; pos fp32 [seq][dim]. */
int dm_embed_tokens(const long long* tokens, const float* table, const float* pos, void* out, long ld, int rows,
                    int seq, int dim, int vocab, int dtype, hipStream_t stream);

/* y = ((x - old_min) / old_span) * new_span + new_min, optionally clamped to [lo, hi]; every step separately
 * rounded in the tensor's dtype (fp32 or bf16).  Replaces rescale (scripts/utils.py:15-53); the host passes
 * old_span = old_max - old_min and new_span = new_max - new_min as the reference forms them. */
int dm_rescale(const void* x, void* y, long n, int dtype, float old_min, float old_span, float new_span,
               float new_min, int clamp, float lo, float hi, hipStream_t stream);
/* (B, C, H, W) images -> rescaled as above, clamped to [lo, hi] (within [0, 255]), truncated to uint8 and
 * stored (B, H, W, C).  Replaces the post-processing chain of scripts/inference.py:337-343
 * (rescale(..., clamp=True) -> move_channel(to="last") -> .to(torch.uint8)) in one pass.  hw = H*W. */
int dm_images_to_u8(const void* x, int dtype, int batch, int c, long hw, float old_min, float old_span,
                    float new_span, float new_min, float lo, float hi, unsigned char* out, hipStream_t stream);

/* (B, H, W, C) uint8 pixels -> (B, C, H, W) fp32 / bf16, rescaled as dm_rescale does (no clamp).  Replaces the
 * tensor conversion + rescale(image, (0, 255), (-1, 1)) + move_channel(to="first") of prepare_image_tensor
 * (scripts/utils.py:203-212).  A uint8 value converts exactly; in bf16 mode it is first rounded to bf16, as
 * torch.tensor(array, dtype=torch.bfloat16) does. */
int dm_u8_to_images(const unsigned char* x, int batch, int c, long hw, float old_min, float old_span, float new_span,
                    float new_min, void* out, int dtype, hipStream_t stream);

/* VAE encoder tail (src/models/encoder.py:191-212): moments fp32 (B, 8, hw) = [mean | log-variance] ->
 * out (B, 4, hw) = (mean + sqrt(exp(clamp(logvar, -30, 20))) * noise) * scaling. */
int dm_vae_sample(const float* moments, const float* noise, float* out, int batch, long hw, float scaling,
                  hipStream_t stream);

/* Layout / dtype plumbing around the model boundary (fp32 NCHW outside, [rows][C] inside). */
int dm_nchw_to_rows(const float* src, int batch, int c, int hw, void* dst, long ld, int cpad, int dtype,
                    hipStream_t stream);
int dm_rows_to_nchw(const void* src, long ld, int batch, int c, int hw, float* dst, int dtype, hipStream_t stream);
int dm_convert(const void* src, int src_dtype, void* dst, int dst_dtype, long n, hipStream_t stream);
/* dst[b][:] = table[*index][:] for b < batch (fp32); *index += step afterwards when advance != 0 is
 * requested through dm_advance_counter. */
int dm_gather_row(const float* table, long ld, const int* index, float* dst, long ld_dst, int batch, int n,
                  hipStream_t stream);
int dm_advance_counter(int* counter, int delta, hipStream_t stream);
/* Zero `bytes` bytes (a memset node when captured): statistics buffers, arrival counters. */
int dm_zero(void* ptr, long bytes, hipStream_t stream);

/* hipGraph capture of a sequence of the launches above (the sampler-loop body). */
int dm_graph_begin(hipStream_t stream);
int dm_graph_end(hipStream_t stream, void** graph_exec_out);
int dm_graph_launch(void* graph_exec, hipStream_t stream);
int dm_graph_destroy(void* graph_exec);

/* Event timing on the caller's stream (used by the benchmark to time kernel classes). */
int dm_event_create(void** ev);
int dm_event_record(void* ev, hipStream_t stream);
int dm_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int dm_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* DM_HIP_H */
