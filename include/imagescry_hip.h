/*
 * imagescry_hip.h -- C ABI of the MI355X (gfx950) embed-and-search hot path.
 *
 * This is the drop-in boundary.  The reference
 * (libertininick/imagescry) is pure Python and has no FFI; each entry point below
 * names the reference Python function whose arithmetic it replaces.  The Python
 * surface in `imagescry_amd/` (same class / function names as the reference) binds
 * these symbols through ctypes -- see INTEGRATION.md for the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless a parameter says
 *     "host"; nothing is allocated or freed inside the library;
 *   - `stream` is a `hipStream_t` passed as `void*` (NULL = the default stream); all
 *     work is enqueued on it and the call returns without synchronising, so the
 *     functions may be captured into a hipGraph;
 *   - scratch memory is a caller-provided workspace whose size is queried first;
 *   - return value: 0 = ISC_OK, negative = error (see `isc_strerror`); never throws;
 *   - layouts are row-major / NCHW or NHWC as stated per function, dense unless a
 *     leading dimension is given (in ELEMENTS).
 */
#ifndef IMAGESCRY_HIP_H
#define IMAGESCRY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISC_ABI_VERSION 4

/* element types */
#define ISC_U8 0
#define ISC_F16 1
#define ISC_F32 2

/* status codes */
#define ISC_OK 0
#define ISC_ERR_INVALID_ARG (-1)  /* NULL pointer, non-positive size, bad enum        */
#define ISC_ERR_UNSUPPORTED (-2)  /* valid request this build has no kernel for        */
#define ISC_ERR_WORKSPACE (-3)    /* workspace missing or smaller than the queried size */
#define ISC_ERR_LAUNCH (-4)       /* hipGetLastError() != hipSuccess after a launch    */
#define ISC_ERR_NO_DEVICE (-5)    /* no HIP device / wrong architecture                */
#define ISC_ERR_ALIGNMENT (-6)    /* pointer or leading dimension not aligned as required */

/* activation selector for the encoder blocks */
#define ISC_ACT_NONE 0
#define ISC_ACT_RELU 1
#define ISC_ACT_GELU 2 /* exact erf form */
#define ISC_ACT_SILU 3    /* x * sigmoid(x) */
#define ISC_ACT_SIGMOID 4
#define ISC_ACT_RESIDUAL_AFTER 0x100 /* OR into `act`: out = act(conv + bias) + residual (default: residual inside act) */

int isc_abi_version(void);
/* how the library was built: bit 0 (ISC_BUILD_ABLATION) = compiled with -DISC_ABLATION, i.e. it contains the timing
 * variants of the kernels that return WRONG results by design and reads ISC_DEBUG_MODE-style environment variables.  The
 * production library returns 0; a binding must refuse a library with bit 0 set unless ablation runs were asked for. */
#define ISC_BUILD_ABLATION 1
int isc_build_flags(void);
const char* isc_strerror(int status);
/* device properties the host side sizes launches with; any out pointer may be NULL (host pointers) */
int isc_device_info(int* num_cus, int* lds_bytes_per_cu, char* arch_name, int arch_name_len);

/* Per-kernel device timing for the roofline line of bench.py.  While enabled, every launch of the kernels
 * below is bracketed by hipEvents recorded on the caller's stream; `isc_timing_read` synchronises those events,
 * returns the summed device time and the number of KERNEL launches since the last read (the unit a rocprofv3 kernel
 * trace counts in: a convolution that runs as whole rounds + a half-tile remainder counts two), and clears them (host
 * pointers). */
#define ISC_KERNEL_DOTS_FILTER 0 /* k_dots_filter: the MFMA score + threshold-filter pass of isc_cosine_topk */
#define ISC_KERNEL_CONV 1        /* k_conv_f32 / k_conv_halo_f32: the implicit-GEMM convolutions of isc_conv2d_nhwc */
#define ISC_KERNEL_GEMM_F16 2    /* k_gemm_f16: the fp16 GEMM of isc_gemm_f16 (transformer encoder) */
#define ISC_KERNEL_COUNT 3
int isc_timing_enable(int enable);
int isc_timing_read(int kernel_id, double* total_ms, int* launches);

/* ---------------------------------------------------------------------------------------------
 * Preprocess
 * ------------------------------------------------------------------------------------------- */

/* Batch-wide per-channel mean and UNBIASED standard deviation of an NCHW image batch.
 * Replaces `image_tensor.mean(dim=(0,2,3))` / `.std(dim=(0,2,3))` in
 * reference src/imagescry/image/transforms.py:62-65.
 *   x            [B,C,H,W] of `dtype` (ISC_U8 or ISC_F32), contiguous
 *   mean, stdev  float [C] outputs
 * u8 input is accumulated exactly (integer sums of x and x*x), f32 input in float64. */
int isc_channel_stats_workspace_bytes(int dtype, int B, int C, int H, int W, size_t* bytes);
int isc_channel_stats(const void* x, int dtype, int B, int C, int H, int W, float* mean, float* stdev,
                      void* workspace, size_t workspace_bytes, void* stream);

/* y = clip((float(x) - mean[c]) / (stdev[c] + eps), lo, hi).  Pass -INFINITY / +INFINITY to disable a bound.
 * Replaces reference src/imagescry/image/transforms.py:58-72.
 *   mean/stdev are float arrays of `stat_batch`*C values with stat_batch in {1, B}
 *   (the reference's `#B C 1 1` broadcast rule, transforms.py:19-20). */
int isc_normalize_clip(const void* x, int dtype, int B, int C, int H, int W, const float* mean, const float* stdev,
                       int stat_batch, float eps, float lo, float hi, float* y, void* stream);

/* isc_normalize_clip written channels-last for the convolution stems: y float [B,H,W,4] with
 * y[b][h][w][c] = the value isc_normalize_clip puts at [b][c][h][w] (bit for bit) for c < C and 0 for C <= c < 4;
 * C <= 4.  What `Embedder.predict_step` feeds its encoder when preprocess and forward run back to back
 * (src/imagescry/models/embedding.py:70-72): one pass instead of normalise (NCHW) + isc_nchw_to_nhwc. */
int isc_normalize_clip_nhwc4(const void* x, int dtype, int B, int C, int H, int W, const float* mean, const float* stdev,
                             int stat_batch, float eps, float lo, float hi, float* y, void* stream);

/* Bilinear resize, align_corners=False, no antialias; the input is cast to float first.
 * Replaces `interpolate(image.float(), ..., mode="bilinear", align_corners=False)` in
 * reference src/imagescry/image/transforms.py:103-121.  Source coordinate
 * (dst + 0.5) * (in / out) - 0.5 clamped at 0, as torch's upsample_bilinear2d.
 *   x [planes,H1,W1] of `dtype` (ISC_U8 or ISC_F32); y float [planes,H2,W2]; planes = B*C. */
int isc_resize_bilinear(const void* x, int dtype, int planes, int H1, int W1, int H2, int W2, float* y, void* stream);

/* y[b,:,s] = x[b,:,s] / max(||x[b,:,s]||_2, eps) for x float [B,E,S] (S = H*W; channel dimension normalised).
 * Replaces `nn.functional.normalize(x, p=2, dim=1)` in reference src/imagescry/models/embedding.py:74. */
int isc_l2norm_channels(const float* x, int B, int E, int S, float eps, float* y, void* stream);

/* Packed bank layout -- how an embedding bank sits in HBM for the search kernels.
 *   rows are grouped in tiles of 256; the embedding axis is cut into K steps of 128 bytes (64 halves / 32 floats,
 *   zero padded); storage order is [tile][K step][row in tile][128 B], so the block one K step of one tile needs is
 *   32 KiB of contiguous memory and a workgroup's chunk of tiles is one linear stream.
 *   ROW ORDER: packed position p holds ORIGINAL row (mul * p) mod N, with mul ~ N / golden ratio coprime to N
 *   (isc_bank_permutation): every prefix of the packed bank is an even sample of the original rows, so a bank whose
 *   rows arrive sorted or clustered by similarity -- the reference's store returns all cells of one image adjacent,
 *   src/imagescry/storage/operations.py:135-144 -- looks exchangeable to the search filter.  Row indices at this API
 *   are always ORIGINAL rows; the permutation is internal to pack / unpack / search.
 * The byte size is a multiple of one tile (rows padded to a multiple of 256); the caller zero-fills the padding rows
 * (isc_bank_pack only writes real rows). */
int isc_bank_packed_bytes(int dtype, int64_t N, int D, size_t* bytes);

/* host: the row permutation of an N-row bank (orig = mul * p mod N, p = mul_inv * orig mod N); N < 2^31 */
int isc_bank_permutation(int64_t N, int64_t* mul, int64_t* mul_inv);

/* Write rows [first_row, first_row + n_rows) of a bank of `n_total` rows into its packed image, optionally
 * L2-normalising each row with the `F.normalize(x, p=2, dim=1)` formula x / max(||x||, eps)
 * (reference src/imagescry/models/embedding.py:74) and casting to the bank dtype.
 * Used once when an embedding bank is built from `EmbeddingBatch.get_flat_vectors()` rows
 * (reference src/imagescry/data.py:112-118).
 *   norm_bound   optional device float, updated with atomic max: an upper bound of the Euclidean norms of the rows AS
 *                STORED.  Zero it before the first call; isc_cosine_topk's rounding-error guard takes it. */
int isc_bank_pack(const void* rows, int in_dtype, int64_t n_rows, int D, int64_t ldx, int64_t first_row,
                  int64_t n_total, int normalize, float eps, void* packed, int dtype, float* norm_bound, void* stream);

/* Inverse of isc_bank_pack for rows [first_row, first_row + n_rows): packed -> row-major [n_rows, D] of `dtype`. */
int isc_bank_unpack(const void* packed, int dtype, int D, int64_t n_total, int64_t first_row, int64_t n_rows,
                    void* rows, int64_t ldy, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Encoder blocks (float32, NHWC activations, KRSC weights)
 * ------------------------------------------------------------------------------------------- */

/* NCHW float -> NHWC float repack ([B,C,H,W] -> [B,H,W,Cpad], channels C..Cpad-1 zero). */
int isc_nchw_to_nhwc(const float* x, int B, int C, int H, int W, int Cpad, float* y, void* stream);

/* out = act(conv2d(x, w) + bias [+ residual]) as an implicit GEMM on the f32 matrix cores.
 * The build's stand-in for the torchvision backbone the reference calls at
 * src/imagescry/models/embedding.py:167-177 (BatchNorm folded into w / bias by the host).
 *   x        float [B,H,W,Cin]         NHWC, Cin % 4 == 0.  Cin % 32 != 0 selects "packed-K mode" (the RGB stem as
 *                                      RGB + one zero channel, 24- / 48-channel stages): no residual restriction, but
 *                                      no `gate` (isc_conv2d_nhwc_gated) and no centring (isc_linear_centered)
 *   w        float [Cout,R,S,Cin]      KRSC; in packed-K mode every row [R*S*Cin] is zero-padded to a multiple of 32
 *                                      floats, i.e. [Cout, ceil(R*S*Cin/32)*32]
 *   bias     float [Cout] or NULL
 *   residual float [B,Ho,Wo,Cout] or NULL (added before the activation)
 *   out      float [B,Ho,Wo,Cout],  Ho = (H + 2*pad - R)/stride + 1 (same for Wo)
 * A linear layer is the case H=W=R=S=1. */
int isc_conv2d_nhwc(const float* x, int B, int H, int W, int Cin, const float* w, int Cout, int R, int S, int stride,
                    int pad, const float* bias, const float* residual, int act, float* out, void* stream);

/* A 1 x 1 convolution over TWO inputs, K-concatenated:
 *     out[b,h,w,:] = act( w[:, :Cin] . x[b,h,w,:] + w[:, Cin:] . x2[b, h*stride2, w*stride2, :] + bias [+ residual] )
 * A ResNet bottleneck's projection shortcut folded into its last convolution (torchvision Bottleneck: `out = relu(
 * bn3(conv3(h)) + downsample(x))`; the build's stand-in for the backbone call at src/imagescry/models/embedding.py:167-177):
 * the shortcut's [B,H,W,Cout] map is never written and read back as a residual.
 *   x   float [B,H,W,Cin], Cin % 32 == 0;  x2 float [B,H2,W2,Cin2], Cin2 % 32 == 0, (H2-1)/stride2+1 == H (same for W)
 *   w   float [Cout, Cin + Cin2] (the two folded weight matrices side by side); bias = the sum of the two biases */
int isc_conv2d_nhwc_dual(const float* x, int B, int H, int W, int Cin, const float* x2, int H2, int W2, int Cin2,
                         int stride2, const float* w, int Cout, const float* bias, const float* residual, int act,
                         float* out, void* stream);

/* Same as isc_conv2d_nhwc with the input first multiplied by a per-(image, input channel) gate float [B, Cin]:
 * the squeeze-excitation scale of an MBConv block fused into its 1x1 projection convolution. */
int isc_conv2d_nhwc_gated(const float* x, int B, int H, int W, int Cin, const float* gate, const float* w, int Cout, int R,
                          int S, int stride, int pad, const float* bias, const float* residual, int act, float* out,
                          void* stream);

/* Depthwise R x R convolution (groups == channels), NHWC float, weights float [R,R,C], bias float [C] or NULL,
 * C % 4 == 0: y = act(dwconv(x, w) + bias).  The depthwise stage of torchvision's MBConv block, which the reference
 * runs inside `efficientnet_v2_*.features` (src/imagescry/models/embedding.py:133-147). */
int isc_dwconv2d_nhwc(const float* x, int B, int H, int W, int C, const float* w, int R, int stride, int pad,
                      const float* bias, int act, float* y, void* stream);

/* The depthwise stage of an MBConv block with its squeeze-excitation neighbours folded in (torchvision `MBConv`:
 * depthwise conv, `SqueezeExcitation.avgpool`, scale; reached through src/imagescry/models/embedding.py:133-147):
 *   pooled[B, C] = mean over (Ho, Wo) of act(dwconv(x, w) + bias)             when `pooled` is not NULL
 *   y            = act(dwconv(x, w) + bias) [* gate[b, c] when `gate` != NULL] when `y` is not NULL
 * For 3 x 3 / stride 1 / pad 1 the depthwise kernel produces both itself (`pooled` needs W <= 14, else a second pass over
 * y by isc_global_avgpool_nhwc); y == NULL (the pooling alone) and `gate` exist for that shape only, other shapes
 * return ISC_ERR_UNSUPPORTED for them.  A block runs it twice -- pooled, isc_se_gate, then the gated y -- which makes the
 * block's projection a plain isc_conv2d_nhwc; isc_conv2d_nhwc_gated is the route for the other shapes. */
int isc_dwconv2d_nhwc_pool(const float* x, int B, int H, int W, int C, const float* w, int R, int stride, int pad,
                           const float* bias, int act, const float* gate, float* y, float* pooled, void* stream);

/* Squeeze-excitation gate: gate[B, C] = sigmoid(w2 . silu(w1 . pooled + b1) + b2) -- torchvision's
 * `SqueezeExcitation` (fc1, SiLU, fc2, Sigmoid on the pooled map) inside the MBConv blocks the reference runs through
 * src/imagescry/models/embedding.py:133-147.  pooled float [B, C]; w1 float [S, ld1] (row stride ld1 >= C floats),
 * b1 float [S] or NULL; w2 float [C, ld2] (ld2 >= S), b2 float [C] or NULL; C, S, ld1, ld2 multiples of 4,
 * C + S <= 16384, S <= 256. */
int isc_se_gate(const float* pooled, int B, int C, const float* w1, int ld1, const float* b1, int S, const float* w2,
                int ld2, const float* b2, float* gate, void* stream);

/* out[n, K] = (x[n, F] - mean[F]) . w[K, F]^T + bias[K]   (mean and bias may be NULL; F % 32 == 0, K % 4 == 0).
 * The centring happens before the product, as in `torch.matmul(x - feature_means, component_vectors)` of
 * reference src/imagescry/models/decomposition.py:91 (`PCA.forward`); w is `component_vectors` transposed. */
int isc_linear_centered(const float* x, int64_t n, int F, const float* mean, const float* w, int K, const float* bias,
                        float* out, void* stream);

/* im2col for the stem convolution (small Cin): x NCHW float [B,C,H,W] -> patches float [B*Ho*Wo, Kpad] with the K axis
 * ordered (r, s, c) and zero-padded from R*S*C to Kpad. */
int isc_im2col_nchw(const float* x, int B, int C, int H, int W, int R, int S, int stride, int pad, int Kpad, float* y,
                    void* stream);

/* max pooling, NHWC float, window R x R, -inf padding (torch.nn.functional.max_pool2d semantics). */
int isc_maxpool_nhwc(const float* x, int B, int H, int W, int C, int R, int stride, int pad, float* y, void* stream);

/* global average pooling, NHWC float [B,H,W,C] -> [B,C]. */
int isc_global_avgpool_nhwc(const float* x, int B, int H, int W, int C, float* y, void* stream);

/* The tail of a pooled encoder in ONE launch: out[b] = linear(mean over (H, W) of x[b]) [/ max(||.||_2, eps) when
 * `normalize`]: global average pool, projection and the `F.normalize(x, p=2, dim=1)` of the reference's predict_step
 * (src/imagescry/models/embedding.py:70-76) for an embedder whose output map is [B, E, 1, 1].  x float NHWC
 * [B, H, W, C], w float [E, C] (16-byte aligned), bias float [E] or NULL, out float [B, E]; C % 4 == 0, C + E <= 8192.
 * The pool and the normalisation reproduce isc_global_avgpool_nhwc and isc_l2norm_channels bit for bit. */
int isc_pool_linear_l2norm(const float* x, int B, int H, int W, int C, const float* w, const float* bias, int E,
                           int normalize, float eps, float* out, void* stream);

/* ---- transformer encoder blocks (ViT-B/16, BASELINE.json configs[4]); fp16 operands, float32 accumulation -------
 * The reference's encoder is any `EmbeddingModule.forward` (models/embedding.py:91-104); these are the blocks a
 * ViT forward is composed of (torch.nn.Linear / LayerNorm / scaled_dot_product_attention in a torch build). */

/* PACKED fp16 matrix layout (flags below): the embedding bank's layout (isc_bank_pack) applied to GEMM operands --
 * rows in tiles of 256, columns in K steps of 64 halves, stored [tile][K step][row][64 halves]; a matrix of R rows
 * and C columns (C % 64 == 0) occupies ceil(R / 256) * 256 * C halves.  element (r, c) sits at
 * (((r / 256) * (C / 64) + c / 64) * 256 + r % 256) * 64 + c % 64.  One K step of one tile is 32 KiB of contiguous
 * memory (one LDS-DMA instruction = one contiguous KiB), and a 64-wide attention head of a token is one 128-byte segment. */
#define ISC_GEMM_A_PACKED 1   /* `a` is packed */
#define ISC_GEMM_W_PACKED 2   /* `w` is packed */
#define ISC_GEMM_OUT_PACKED 4 /* `out` (fp16 only, N % 64 == 0) is written packed */
#define ISC_GEMM_TILE_256 8   /* use the 256 x 256-tile kernel (one wave per SIMD, LDS-DMA rings); needs K >= 192,
                                 act == ISC_ACT_NONE and operands below 4 GiB, else ISC_ERR_UNSUPPORTED */
#define ISC_GEMM_TILE_128 16  /* use the 128 x 128-tile kernel even where the streaming kernel applies (packed a and w,
                                 N % 256 == 0: 256 x 256 tiles on the search kernel's LDS-DMA ring loop, the default) */

/* out[M,N] = act(a[M,K] . w[N,K]^T + bias[N]) + residual[M,N].   a, w fp16, row-major (w in torch.nn.Linear layout)
 * or packed per `flags`; bias, residual float32 row-major (either may be NULL); act ISC_ACT_NONE or ISC_ACT_GELU
 * (0.5 x (1 + erf(x / sqrt 2)), erf by the Abramowitz-Stegun 7.1.26 polynomial, |error| <= 1.5e-7);
 * out fp16 or float32 (`out_dtype`).  K % 64 == 0, N % 4 == 0, all pointers 16-byte aligned. */
int isc_gemm_f16(const void* a, int64_t M, int K, const void* w, int N, const float* bias, const float* residual,
                 int act, void* out, int out_dtype, int flags, void* stream);

/* LayerNorm over the last axis (biased variance, float32 statistics): x float32 [rows, D] with row stride ldx,
 * y fp16 or float32 (`y_dtype`) with row stride ldy (strides in elements, multiples of 4), or -- y_packed != 0, fp16,
 * D % 64 == 0 -- in the packed layout.  D % 4 == 0, D <= 2048. */
int isc_layernorm(const float* x, int64_t rows, int D, int64_t ldx, const float* gamma, const float* beta, float eps,
                  void* y, int y_dtype, int64_t ldy, int y_packed, void* stream);

/* softmax(q k^T / sqrt(head_dim)) v per (image, head).  qkv fp16 [B * T, 3 * heads * head_dim] laid out as the output
 * of one fused Linear whose weight rows are [query; key; value], each head-major; out fp16 [B * T, heads * head_dim];
 * both row-major (packed == 0) or both packed.  head_dim == 64, T <= 224. */
int isc_attention_f16(const void* qkv, int B, int T, int heads, int head_dim, void* out, int packed, void* stream);

/* non-overlapping patches of an NCHW float32 image batch as fp16 GEMM rows (row-major or packed):
 * patches[(b, ph, pw)][c * P * P + r * P + s] = x[b][c][ph * P + r][pw * P + s]   (torch Conv2d(kernel=stride=P) weight
 * order).  P % 8 == 0, H % P == 0, W % P == 0. */
int isc_patchify_f16(const float* x, int B, int C, int H, int W, int patch, void* patches, int packed, void* stream);

/* tokens[b][0] = cls + pos[0]; tokens[b][t] = patch_embed[b * (T - 1) + t - 1] + pos[t]; all float32, D % 4 == 0. */
int isc_vit_assemble(const float* patch_embed, const float* cls_token, const float* pos_embed, int B, int T, int D,
                     float* tokens, void* stream);

/* ---------------------------------------------------------------------------------------------
 * PCA fit (reference src/imagescry/models/decomposition.py:118-131: mean, centring, SVD of the [N, F] rows on
 * the host).  The N-sized work runs here -- float64 feature sums, centred + transposed chunks, their F x F Gram matrices
 * on the f32 matrix cores; the F x F eigenproblem is the host's (imagescry_amd/decomposition.py).
 * ------------------------------------------------------------------------------------------- */

/* sums[f] = sum_i x[i][f] in float64, bit-reproducible (fixed two-stage order).  x float [n, F] with row stride ldx.
 * Replaces `x.mean(dim=0, keepdim=True)` (decomposition.py:120). */
int isc_feature_sums_workspace_bytes(int64_t n, int F, size_t* bytes);
int isc_feature_sums(const float* x, int64_t n, int F, int64_t ldx, double* sums, void* workspace, size_t workspace_bytes,
                     void* stream);

/* xt[f][i] = x[i][f] - mean[f] (i < n), 0 for n <= i < ldn and for F <= f < Fpad; xt float [Fpad, ldn].
 * Replaces `x - self.feature_means` (decomposition.py:123) and hands the Gram kernel its K-contiguous operand. */
int isc_center_transpose(const float* x, int64_t n, int F, int64_t ldx, const float* mean, float* xt, int Fpad,
                         int64_t ldn, void* stream);

/* gram[f1][f2] = sum_i xt[f1][i] * xt[f2][i] on the f32 matrix cores (k_conv_f32 with xt as both operands); xt float
 * [F, n], n % 32 == 0, F % 4 == 0, F * n < 2^31; gram float [F, F].  The right singular vectors / singular values the
 * reference takes from `torch.linalg.svd(x_centered)` (decomposition.py:126) are the eigenpairs of this matrix. */
int isc_gram_rows(const float* xt, int F, int64_t n, float* gram, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Search
 * ------------------------------------------------------------------------------------------- */

/* Brute-force cosine top-k of `Q` queries against `N` bank rows.
 * No reference symbol exists; semantics are the oracle's
 * (oracle/search_oracle.py): score = float32(dot_f64(q,b) / max(||q||_2,1e-12)), bank rows used as
 * stored, result ordered by (score descending, row index ascending; NaN scores last).
 * The result is FINAL when the stream has run the call: the matrix-core pass is only a filter, the candidates are
 * re-scored in float64, a rounding-error guard proves per query that the filter lost nothing, and the queries it cannot
 * prove (near-duplicate rows around the k-th neighbour, overflowed candidate buffers) are searched again on the device:
 * one more matrix-core pass over the bank with a fixed threshold just below the k-th exact score found so far, EVERY
 * survivor re-scored in float64 -- and, for what even that cannot hold (thousands of rows tied at the k-th score, NaN
 * scores, queries of denormal or overflowing scale), the exhaustive float64 sweep.  No host round trip.
 *   bank         N rows of D values of `dtype` (ISC_F16 or ISC_F32) in the PACKED layout above (isc_bank_pack),
 *                16-byte aligned
 *   queries      row-major [Q, D] of `q_dtype` (ISC_F16 or ISC_F32, independent of the bank's), leading dimension ldq
 *                (elements of q_dtype).  A query is ROUNDED TO THE BANK DTYPE first (float32 -> fp16 round to nearest
 *                even, exactly `Tensor.to(float16)`; fp16 -> float32 is exact) while it is packed, so the reference-shaped
 *                call `bank.search(predict_step(batch).get_flat_vectors())` (float32 vectors, reference
 *                src/imagescry/data.py:112-118) against an fp16 bank needs no cast kernel
 *   k            1 <= k <= min(N, ISC_TOPK_MAX_K)
 *   index_base   added to every returned row index (global index of this shard's row 0)
 *   norm_bound   device float: upper bound of the stored rows' norms (isc_bank_pack); NULL = rows are unit length
 *   out_scores   float   [Q, k]
 *   out_indices  int64_t [Q, k]
 *   status       int32_t [4] device words, diagnostics only:
 *                  [0] = candidate buffers that overflowed, [1] = queries the first pass could not prove (searched
 *                  again), [2] = float bits of max |filter score - exact dot| / guard bound over the re-scored
 *                  candidates (must stay < 1), [3] = queries answered by the exhaustive float64 sweep (a subset of [1])
 * Calls with Q > 1024 run as passes of 1024 queries over the same workspace.  D <= ISC_SEARCH_MAX_D.
 */
#define ISC_TOPK_MAX_K 120
#define ISC_SEARCH_MAX_D 8192
#define ISC_SEARCH_MAX_Q (1 << 24)
#define ISC_SEARCH_PASS_QUERIES 1024 /* queries per pass; the workspace size depends on min(Q, this) rounded up to a query tile */
int isc_cosine_topk_workspace_bytes(int dtype, int64_t N, int D, int Q, int k, size_t* bytes);
int isc_cosine_topk(const void* bank, int dtype, int64_t N, int D, const void* queries, int q_dtype, int Q, int64_t ldq,
                    int k, int64_t index_base, const float* norm_bound, float* out_scores, int64_t* out_indices,
                    int32_t* status, void* workspace, size_t workspace_bytes, void* stream);

/* Same contract and the same limits, data-independent cost: every score of every query is evaluated in float64
 * (vector FMA, no matrix cores, about one bank stream per four queries).  The kernel isc_cosine_topk falls back to
 * per query; exported as the reference implementation of the search on the device. */
int isc_cosine_topk_exhaustive_workspace_bytes(int dtype, int64_t N, int D, int Q, int k, size_t* bytes);
int isc_cosine_topk_exhaustive(const void* bank, int dtype, int64_t N, int D, const void* queries, int q_dtype, int Q,
                               int64_t ldq, int k, int64_t index_base, float* out_scores, int64_t* out_indices,
                               void* workspace, size_t workspace_bytes, void* stream);

/* Merge G partial results (e.g. one per bank shard after the all-gather) into the final top-k by
 * (score descending, index ascending): scores float [G,Q,kin], indices int64 [G,Q,kin] -> [Q,kout], kout <= G*kin <= 4096.
 * `stride_g_*` = distance in ELEMENTS between the [Q,kin] blocks of consecutive shards (0 = dense); this lets the merge
 * read the all-gathered exchange buffers in place. */
int isc_topk_merge(const float* scores, const int64_t* indices, int G, int Q, int kin, int kout, int64_t stride_g_scores,
                   int64_t stride_g_indices, float* out_scores, int64_t* out_indices, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IMAGESCRY_HIP_H */
