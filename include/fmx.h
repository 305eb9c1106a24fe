/* fmx.h -- C ABI of libfmx_gfx950.so: the MI355X (gfx950 / CDNA4) kernels behind Forge's txt2img
 * denoising hot path.  Plain pointers and sizes only; no torch types.  Every pointer is a DEVICE
 * pointer unless a comment says "host"; `stream` is a hipStream_t passed as void*.  All entry points
 * are asynchronous on `stream`, allocate nothing, keep no global state besides the last-error
 * string, and return 0 on success or a non-zero code (hipError_t value, or FMX_E_* below);
 * `fmx_last_error()` describes the last failure on the calling thread.
 *
 * Reference interfaces replaced (the reference has no FFI; these are the Python call sites whose
 * arithmetic PyTorch/ATen executes today -- see INTEGRATION.md for the binding on the Forge side):
 *   fmx_gemm_conv_f16      F.linear  backend/operations.py:153,156 ; Conv2d._conv_forward :173,176 ;
 *                          nearest Upsample+conv backend/nn/unet.py:340-355, backend/nn/vae.py:35-57 ;
 *                          torch.cat([h, hsp]) backend/nn/unet.py:741 (two-source A operand) ;
 *                          GEGLU backend/nn/unet.py:104-111 ; ResBlock emb add / skip add :469-478
 *   fmx_attention_f16      attention_function  backend/attention.py:324-339 (and :37-93, incl. its additive / bool `mask`)
 *   fmx_strided_copy4      the reshape / permute / .to(dtype) around it: backend/attention.py:51-60,90-93,330-338,415-418
 *   fmx_attention_single_head512_f16  attention_function_single_head_spatial backend/attention.py:412-422 at the VAE's 512 channels
 *   fmx_softmax_rows_f16   sim.softmax(dim=-1) backend/attention.py:85 (materialised-score variant: single heads wider than 160 other than 512)
 *   fmx_groupnorm_*        F.group_norm backend/operations.py:308 (+ SiLU backend/nn/unet.py:394-398)
 *   fmx_layernorm_f16      F.layer_norm backend/operations.py:327
 *   fmx_rmsnorm_f16        T5LayerNorm backend/nn/t5.py:15-25 (the T5-XXL text encoder of Flux, backend/diffusion_engine/flux.py:87-88)
 *   fmx_timestep_embedding timestep_embedding backend/nn/unet.py:55-67 (and backend/nn/flux.py:52-73 with t*1000)
 *   fmx_layernorm_mod_f16  LayerNorm(no affine) + adaLN modulate backend/nn/flux.py:209-210,228-229,255,260,286,326
 *   fmx_flux_qk_norm_rope_f16  QKNorm (RMSNorm) + apply_rope + [B,H,L,D] permute backend/nn/flux.py:43-49,115-139,217-247
 *   fmx_silu_f16           nn.SiLU in time_embed / emb_layers backend/nn/unet.py:519-523,415-418
 *   fmx_unet_pack_input    KModel.apply_model input half  backend/modules/k_model.py:25-36
 *   fmx_cfg_combine        calculate_denoised backend/modules/k_prediction.py:81-92 + CFG combine
 *                          backend/sampling/sampling_function.py:276-288,312
 *   fmx_sampler_*          k_diffusion/sampling.py:120-137 (Euler), :141-159 (Euler a), :649-671 (DPM++2M)
 *   fmx_vae_pack_latent / fmx_vae_unpack_image
 *                          process_out backend/nn/vae.py:315 ; clamp((y+1)/2) backend/patcher/vae.py:142
 *   fmx_philox_randn       modules/rng_philox.py:32-102 ("NV" noise source)
 *   fmx_sampler_lincomb / fmx_sampler_error_norm
 *                          every other update of k_diffusion/sampling.py, modules/sd_samplers_timesteps_impl.py, uni_pc.py (a linear
 *                          combination of latent-sized tensors with host-computed coefficients); DPM adaptive's error norm :507-560
 *   fmx_layernorm_padded_f16  F.layer_norm writing at a padded per-image row stride (ragged token counts, backend/nn/unet.py:183-279)
 *   fmx_resize_separable_f32  F.interpolate of the hires pass modules/processing.py:1430-1480 (all latent upscale modes)
 *   fmx_blend_masked       inpaint latent blending modules/sd_samplers_cfg_denoiser.py:178-213 ; regional cond averaging
 *                          backend/sampling/sampling_function.py:276-288
 *   fmx_add_control_nchw / fmx_add_scaled_f16
 *                          ControlNet / T2I-Adapter residual injection backend/nn/unet.py:44-52
 *   fmx_avgpool2x2_nhwc_f16 / fmx_act_f16
 *                          T2I-Adapter Downsample + ReLU backend/nn/cnets/t2i_adapter.py:42-62,76-101 ; CLIP quick-GELU / GELU
 *   fmx_embed_tokens       CLIP token + position embedding (transformers CLIPTextEmbeddings, called from backend/nn/clip.py)
 *   fmx_vae_sample_posterior  DiagonalGaussianDistribution.sample + process_in backend/nn/vae.py:16-29,312-313
 *   fmx_*_bf16             the bfloat16 build of the Flux path's kernels (last section)
 */
#ifndef FMX_H
#define FMX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMX_ABI_VERSION 11

#define FMX_OK 0
#define FMX_E_BADARG 10001   /* shape / alignment / null-pointer contract violated */
#define FMX_E_UNSUPPORTED 10002

/* activation / epilogue selector for fmx_gemm_conv_f16 */
#define FMX_ACT_NONE 0
#define FMX_ACT_GEGLU 1 /* weight rows interleaved [16 value | 16 gate] (see fmx_geglu_interleave_rows) */
#define FMX_ACT_GELU_TANH 2 /* nn.GELU(approximate="tanh") of the Flux MLPs (backend/nn/flux.py:193,280) */

int fmx_abi_version(void);
const char* fmx_last_error(void);
/* identity of the loaded binary: "src=<16 hex digits: sha256 over csrc sources + this header at build time> abi=<n> arch=gfx950" -- what a
 * measurement (bench.py's `roofline.traffic`, a committed PMC summary) is keyed on, instead of whatever sources lie next to the .so */
const char* fmx_build_info(void);
/* development A/B knobs (environment variables FMX_GEMM_*, FMX_ATTN_*, FMX_GN_*, ... that select another kernel for the same call) are read ONLY in a
 * process that carries FMX_ALLOW_KNOBS=1; this lists "NAME=value,..." of the knobs that took effect (ignored = 0) or were set but ignored (ignored = 1) */
int fmx_active_knobs(char* buf, int buf_len, int ignored);
/* host out-params: number of CUs, wave size, gcn arch name (e.g. "gfx950:sramecc+:xnack-") */
int fmx_device_info(int* cu_count, int* wave_size, char* arch, int arch_len);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear on MFMA:   OUT[M, nout] = epilogue( A (*) W^T )
 *   A operand : NHWC fp16 activations, optionally the channel-concatenation of two tensors
 *               (a0 with c0 channels, a1 with c1 channels; c1 = 0 for a single source).
 *               a?_stride = element stride between consecutive pixels (0 -> dense = c?).
 *               With kh = 1 this is a plain matrix [M = n*h*w][c0+c1] (Linear / 1x1 conv).
 *               If up_h > 0 the input is first nearest-resized from (h, w) to (up_h, up_w)
 *               (src = floor(dst * in / out)) -- the fused Upsample -- and the conv runs on that.
 *   W operand : fp16 [nout][kh*kh*(c0+c1)] row-major (ldw elements per row, 0 -> dense),
 *               K ordered (ky, kx, channel).
 *   epilogue  : acc*alpha (+ bias[col]) (+ rowvec[row / (oh*ow)][col]) -> act -> (* gate[row / (oh*ow)][col])
 *               -> (+ residual[row][col])    (gate: the adaLN gates of Flux, backend/nn/flux.py:254-262,301)
 *               stored as fp16 (or fp32 when out_f32 != 0) at out[row*ld_out + col].
 *               act = FMX_ACT_GEGLU: out has nout/2 columns, value*gelu_erf(gate).
 * Requirements: (c0+c1) % 64 == 0, c0 % 64 == 0, all tensors 16-byte aligned, strides % 8 == 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fmx_gemm_args {
  const void* a0;
  const void* a1;
  int32_t c0, c1;
  int32_t a0_stride, a1_stride;
  int32_t n, h, w;
  int32_t oh, ow;
  int32_t kh, stride, pad;
  int32_t up_h, up_w;
  const void* wgt;
  int32_t ldw;
  int32_t nout;
  const void* bias;
  const void* rowvec;
  int32_t ld_rowvec;
  const void* residual;
  int32_t ld_res;
  float alpha;
  int32_t act;
  void* out;
  int32_t ld_out;
  int32_t out_f32;
  const void* zero_page; /* >= 256 bytes of zeros (device), used for padding taps / tails */
  const void* gate;      /* optional fp16 [n][ld_gate] per-image column scale applied after act (fp16-output fast epilogue only) */
  int32_t ld_gate;
  /* Optional split-K workspace (ABI 4): device memory the library may use when the problem has fewer output tiles than the chip has
   * workgroup slots (small batches).  The first 64 KiB are arrival counters: ZERO them once after allocation, the library leaves them zero
   * after every call.  One workspace serves one stream at a time.  NULL / 0 = never split.  64 MiB covers every shape that profits. */
  void* workspace;
  int64_t workspace_bytes;
  /* LayerNorm folded into its consumer (ABI 5; backend/nn/unet.py:262-279, norm2 / norm3 of a BasicTransformerBlock).  With ln_partial set,
   * this GEMM computes  LN(x) W^T + b  from the UN-normalised x:  wgt = W * gamma (per input channel), ln_colsum[n] = sum_k wgt[n][k] in fp32
   * (of the fp16 values), bias = W beta + b;  ln_partial = the row statistics fmx_gemm_linear_rowstats_f16 left for x, ln_parts entries per
   * row (1..8: one per 160 output columns of the producer).  Linear only (kh 1), act NONE or GEGLU, no residual / rowvec / gate, fp16 output; the
   * 256x320 tile (or, under the development knob FMX_GEMM_4W, the 256x160 two-workgroups-per-CU tile of csrc/fmx_gemm4w.hip: same arrays). */
  const void* ln_partial;
  int32_t ln_parts;
  const void* ln_colsum;
  float ln_eps;
  /* The same fold for the OPERAND-SWAPPED GEMM (ABI 7; norm1 -> attn1.to_v, unet.py:262-266, whose projection runs as V^T = Wv x^T so that
   * attention reads V^T without a transpose): here the LayerNorm rows are the COLUMNS of the output.  a0 = W * gamma [M = channels][K], wgt =
   * the UN-normalised x [nout = tokens][K];  out[m][n] = acc * rstd[n] - mean[n] rstd[n] colsum[m] + (W beta)[m]:
   *   ln_col_ab [nout][2] fp32 = {rstd, -mean * rstd} per token (fmx_layernorm_rowstats_finalize from the producer's row statistics),
   *   ln_row_cb [M][2]    fp32 = {sum_k a0[m][k] (of the fp16 values), (W beta)[m]} per channel.
   * Linear only, act NONE, no bias / residual / rowvec / gate, fp16 output, M % 320 == 0; always the 320x256 tile. */
  const void* ln_col_ab;
  const void* ln_row_cb;
  /* optional output of an ln_partial (consumer) GEMM: the {rstd, -mean * rstd} pairs it derived for its input rows, [M][2] fp32 -- what
   * fmx_layernorm_rowstats_finalize computes, for the operand-swapped GEMM that follows on the same rows (q|k projection -> V^T projection) */
  void* ln_ab_out;
  /* Cross-attention as the EPILOGUE of the query projection (ABI 10; backend/nn/unet.py:145-155 CrossAttention.forward + :254-267, the attn2 of a
   * BasicTransformerBlock: q = to_q(norm2(x)), out = softmax(q k^T / sqrt(d)) v against the text context's cached K / V^T).  With xa_k set, an
   * ln_partial (LayerNorm-consumer) GEMM does not store Q: a 256 x 320 tile of Q holds five whole 64-wide heads of 256 queries of ONE image; after
   * the K loop the tile's K / V^T rows (xa_nk <= 80 keys x 320 columns) are staged in the idle LDS, S = K Q^T, a one-pass softmax and O = V^T P^T
   * run per head out of the accumulators (Q rounded to fp16 and pre-scaled exactly as fmx_attention_f16 does), and `out` receives O [M][nout].
   *   xa_k  [images * xa_k_bs rows][xa_k_rs] fp16: K of the context, row = key, head h at columns h*64 .. (rows >= xa_nk are never read beyond key 79)
   *   xa_vt [nout rows][xa_vt_ds] fp16: V^T, row = head h * 64 + d, image i's keys at columns i * xa_vt_bs ..
   * Linear only, act NONE, bias optional, no residual / rowvec / gate, fp16 output; d_head 64; nout % 320 == 0; xa_rows (queries per image) % 256
   * == 0 and M % xa_rows == 0; always the 256 x 320 tile.  fp16 build only. */
  const void* xa_k;
  const void* xa_vt;
  int32_t xa_k_rs;   /* elements between keys in xa_k (= heads * 64) */
  int32_t xa_k_bs;   /* key ROWS per image in xa_k (the padded token count) */
  int32_t xa_vt_ds;  /* elements between rows of xa_vt (= images * padded token count) */
  int32_t xa_vt_bs;  /* elements (keys) between images in a row of xa_vt */
  int32_t xa_nk;     /* keys per image, 1..80 */
  int32_t xa_rows;   /* queries per image */
  float xa_scale;    /* softmax scale (d_head^-0.5) */
  int64_t xa_k_bytes, xa_vt_bytes;   /* extents of the two tensors (buffer descriptors) */
} fmx_gemm_args;

int fmx_gemm_conv_f16(const fmx_gemm_args* args /* host */, void* stream);

/* The same GEMM (linear with bias + residual: the projections that close attention / feed-forward, unet.py:268,272,277), additionally
 * leaving per-row partial sums of its fp16 OUTPUT for a LayerNorm that follows: row_partial[m][parts][{sum, sum of squares}] fp32, where
 * parts = 2 * ceil(nout / 320) <= parts_cap is returned in *parts_out.  *parts_out = 0 means the dispatcher chose another tile shape for
 * this problem (small M): the GEMM ran as usual, nothing was written, and the caller applies its LayerNorm with fmx_layernorm_f16.
 * The residual is optional since ABI 7 (proj_in of a SpatialTransformer, unet.py:311-316, feeds norm1 of its first block). */
int fmx_gemm_linear_rowstats_f16(const fmx_gemm_args* args /* host */, float* row_partial, int32_t parts_cap, int32_t* parts_out /* host */,
                                 void* stream);

/* Row statistics -> per-row LayerNorm terms:  ab[m] = {rstd, -mean * rstd}  with mean / rstd over the `c` elements of row m, from the
 * `parts` partial {sum, sum of squares} pairs fmx_gemm_linear_rowstats_f16 left (fixed summation order).  Feeds ln_col_ab above. */
int fmx_layernorm_rowstats_finalize(const float* row_partial, int32_t parts, int64_t rows, int32_t c, float eps, float* ab, void* stream);

/* Reorders the rows of a GEGLU projection weight [2*inner][k] (and bias [2*inner]) on the DEVICE into
 * the [16 value rows | 16 gate rows] interleave FMX_ACT_GEGLU expects.  inner % 16 == 0. */
int fmx_geglu_interleave_rows(const void* w_in, const void* b_in, void* w_out, void* b_out,
                              int32_t inner, int32_t k, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-head attention  O = softmax(Q K^T * scale) V   (flash-style, never materialises N x N)
 *   q  : fp16, element (b, i, h, d) at q[b*q_bs + i*q_rs + h*dpad + d]
 *   k  : fp16, element (b, j, h, d) at k[b*k_bs + j*k_rs + h*dpad + d]
 *   vt : fp16 V TRANSPOSED, element (b, h, d, j) at vt[b*vt_bs + h*vt_hs + d*vt_ds + j]
 *   o  : fp16, element (b, i, h, d) at o[b*o_bs + i*o_rs + h*dpad + d]
 *   dpad in {48, 64, 80, 128, 160}: head dim padded to a multiple of 16 with ZERO columns (rows in vt);
 *   nk_pad = number of key columns present per (b,h,d) row of vt / rows of k (multiple of 64),
 *   nk = number of valid keys (<= nk_pad, keys >= nk are masked out).  nq arbitrary.
 *   All base pointers 16-byte aligned; strides % 8 == 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct fmx_attn_args {
  const void* q;
  const void* k;
  const void* vt;
  void* o;
  int64_t q_bs, q_rs;
  int64_t k_bs, k_rs;
  int64_t vt_bs, vt_hs, vt_ds;
  int64_t o_bs, o_rs;
  int32_t batch, heads, nq, nk, nk_pad, dpad;
  float scale;
  int32_t causal; /* 1: key j attends only for j <= query i (CLIP text encoder, transformers causal mask); occupies former padding */
  const void* zero_page;
  /* optional additive mask (attention_function's `mask`, backend/attention.py:74-88 / SDPA attn_mask): fp16, added to the score before the
   * softmax, element (b, h, i, j) at mask[b*mask_bs + h*mask_hs + i*mask_qs + j]; a stride of 0 broadcasts that dimension; every addressed row
   * holds nk_pad keys; 16-byte aligned, strides % 8 == 0.  -inf entries mask a key out (a bool mask is converted by fmx_strided_copy4). */
  const void* mask;
  int64_t mask_bs, mask_hs, mask_qs;
} fmx_attn_args;

int fmx_attention_f16(const fmx_attn_args* args /* host */, void* stream);

/* Fused attention of ONE head of width 512 (the VAE mid-block attention: backend/nn/vae.py:118-137 -> attention.py:412-422), N up to 16 384
 * tokens, without materialising the scores: O = softmax(Q K^T * scale) V, the head dimension split across the waves of a workgroup.
 *   q, k : fp16 token-major, token i of image b at q[b*q_bs + i*q_rs + c], c < 512 (q and k may be the two halves of one [B*N][1024] buffer)
 *   vt   : fp16 V TRANSPOSED, (b, c, j) at vt[b*vt_bs + c*vt_ds + j]; every row holds nk_pad keys (multiple of 32), keys >= nk must be finite (zeros)
 *   o    : fp16 token-major like q.   16-byte aligned bases, strides % 8 == 0, an image's K / V^T spans < 2 GB. */
int fmx_attention_single_head512_f16(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs, const void* vt,
                                     int64_t vt_bs, int64_t vt_ds, void* o, int64_t o_bs, int64_t o_rs, int32_t batch, int32_t nq, int32_t nk,
                                     int32_t nk_pad, float scale, void* stream);

/* In-place row softmax over fp16 scores: x[r*ld + j], j < ncols, r < nrows (fp32 math). */
int fmx_softmax_rows_f16(void* x, int64_t nrows, int32_t ncols, int64_t ld, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GroupNorm over NHWC fp16 (optionally the channel-concat of two tensors), fp32 statistics, 1 read + 1 write.
 *   statistics : partial[n][chunk][c][2] fp32 = {sum x, sum x^2} of channel c over the chunk's pixels of image n.  Produced EITHER by the
 *                convolution / linear that wrote the tensor -- fmx_gemm_conv_stats_f16 below emits one chunk per 256-row output tile from
 *                its epilogue, so the tensor is never re-read for statistics -- OR by fmx_groupnorm_stats_f16 (any nchunks <= 1024; one
 *                source tensor [n][hw][c] with pixel stride ld).  A partial buffer is read-only for apply: a tensor that feeds two
 *                GroupNorms (a UNet skip connection) keeps its statistics.
 *   apply      : folds the chunks of both sources into per-channel {scale, shift} (scale_shift: fp32 workspace [n][c0+c1][2], fixed
 *                summation order -> deterministic), then y = x * scale[c] + shift[c] = (x - mean_g) * rstd_g * gamma[c] + beta[c],
 *                optional SiLU, fp16 out [n][hw][c0+c1] (the concatenation is materialised only here).
 * (c0+c1) % groups == 0, c0 % 8 == 0, c1 % 8 == 0; ld0 / ld1 = pixel strides of the sources (elements, % 8 == 0).
 * ---------------------------------------------------------------------------------------------- */
int fmx_groupnorm_stats_f16(const void* x, int32_t c, int64_t ld, int32_t n, int32_t hw, float* partial, int32_t nchunks, void* stream);
int fmx_groupnorm_apply_f16(const void* x0, const void* x1, int32_t c0, int32_t c1, int64_t ld0, int64_t ld1, int32_t n, int32_t hw,
                            const float* partial0, int32_t nchunks0, const float* partial1, int32_t nchunks1, int32_t groups, float eps,
                            const void* gamma, const void* beta, int32_t silu, float* scale_shift, void* y, void* stream);

/* fmx_gemm_conv_f16 that ALSO leaves the GroupNorm statistics of its fp16 output [M = n*oh*ow][nout] (ld_out == nout, no GEGLU) in
 * partial[n][chunks][nout][2]: from the epilogue of the 256-row tiles when an image is a whole number of them (oh*ow % 256 == 0) and the
 * tile choice is a 256-row kernel, else by a statistics pass launched behind the GEMM -- the caller cannot tell the difference except by
 * *chunks_out (host): the chunk count it must hand to fmx_groupnorm_apply_f16.  partial must hold n * max_chunks * nout * 2 floats,
 * max_chunks >= max(oh*ow / 256, fallback_chunks); fallback_chunks (1..1024) is the chunk count of the separate pass. */
int fmx_gemm_conv_stats_f16(const fmx_gemm_args* args /* host */, float* partial, int32_t max_chunks, int32_t fallback_chunks,
                            int32_t* chunks_out /* host */, void* stream);

/* LayerNorm over the last dim of fp16 [rows][c] (c % 8 == 0, c <= 4096), fp32 two-pass statistics. */
int fmx_layernorm_f16(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int32_t c,
                      float eps, void* stream);

/* RMS LayerNorm (ABI 9; T5LayerNorm, backend/nn/t5.py:15-25): y = x * rsqrt(mean(x^2) + eps) * weight over the last dim of fp16 [rows][c]
 * (c % 8 == 0, c <= 4096), fp32 sum of squares. */
int fmx_rmsnorm_f16(const void* x, const void* weight, void* y, int64_t rows, int32_t c, float eps, void* stream);

/* Same, with the output rows of every image re-spaced: input row i of image b (b = r / rows_per_image) goes to output row
 * b * out_rows_per_image + i.  Used when the token count per image is not a multiple of the attention key tile (64): the rows in between
 * stay whatever the caller left there (zeros), so ONE batched Q|K / V^T projection serves all images. */
int fmx_layernorm_padded_f16(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int32_t c, float eps,
                             int64_t rows_per_image, int64_t out_rows_per_image, void* stream);

/* Flux adaLN: y = (1 + scale[b]) * LayerNorm_noaffine(x) + shift[b]; x, y fp16 [rows][c], b = row / rows_per_batch;
 * scale / shift fp16 vectors of batch element b at scale + b*ld_mod, shift + b*ld_mod (views into the Modulation output). */
int fmx_layernorm_mod_f16(const void* x, const void* scale, const void* shift, int64_t ld_mod, int64_t rows_per_batch, void* y,
                          int64_t rows, int32_t c, float eps, void* stream);

/* Flux attention front end for one stream (txt or img) of `tokens` rows per batch element:
 *   qkv [batch*tokens][ld_qkv] fp16 = [q | k | v], each [heads][128]  ->  per-head RMSNorm(q)*q_scale, RMSNorm(k)*k_scale (eps),
 *   rotary embedding with pe[(row_off + t)][64][2] = (cos, sin) fp32, written to q_out / k_out [batch][l_pad][heads*128] at rows
 *   row_off + t, and v transposed to vt_out [heads*128][batch*l_pad] at columns b*l_pad + row_off + t. */
int fmx_flux_qk_norm_rope_f16(const void* qkv, int64_t ld_qkv, const void* q_scale, const void* k_scale, const float* pe, void* q_out,
                              void* k_out, void* vt_out, int32_t batch, int32_t tokens, int32_t heads, int32_t head_dim,
                              int32_t row_off, int32_t l_pad, float eps, void* stream);

/* emb[b][0:half] = cos(t[b]*f_k), emb[b][half:] = sin(t[b]*f_k), f_k = exp(-ln(max_period)*k/half); fp32 math,
 * fp16 out [b][dim] (dim even). */
int fmx_timestep_embedding(const float* t, void* emb, int32_t b, int32_t dim, float max_period, void* stream);

int fmx_silu_f16(const void* x, void* y, int64_t n, void* stream);
/* h[p][c] += ctrl[b][c][p'] : ControlNet residual injection (backend/nn/unet.py:44-52 `h += ctrl`).  h: fp16 NHWC [B*npix][C]
 * (the executor's activation layout), ctrl: fp32 NCHW [B][C][npix] as the ControlNet produces it. */
int fmx_add_control_nchw(void* h, const float* ctrl, int32_t b, int32_t c, int64_t npix, void* stream);
/* h[i] += alpha * c[i], h fp16, c fp16 (c_is_f32 == 0) or fp32, both contiguous in the SAME layout: ControlNet residuals that are already
 * channels-last (the native ControlNet's outputs; `h += ctrl` of backend/nn/unet.py:44-52 with the `x *= strength` of
 * patcher/controlnet.py:236 folded into alpha). */
int fmx_add_scaled_f16(void* h, const void* c, int32_t c_is_f32, float alpha, int64_t n, void* stream);
/* 2x2 / stride-2 average pooling on fp16 NHWC [n][h][w][c] (h, w even): the conv-less Downsample of backend/nn/cnets/t2i_adapter.py:42-62 */
int fmx_avgpool2x2_nhwc_f16(const void* x, void* y, int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
int fmx_cast_f32_to_f16(const float* x, void* y, int64_t n, void* stream);

/* Strided 4-D copy with element conversion -- the layout adapter of the attention- and op-level drop-ins (backend/attention.py:324-339 takes
 * [B, N, heads*d] or [B, heads, N, d] in any float type; the ForgeOperations modules take NCHW): dst[i0][i1][i2][i3] = convert(src[...]) for
 * i < dims, element strides per tensor (host arrays of 4).  kinds: 0 fp16, 1 fp32, 2 bf16; source kind 3 = bool "attend" mask (1 byte per
 * element) -> 0 / -inf.  Padding of the destination is the caller's (zero it first). */
int fmx_strided_copy4(const void* src, int32_t src_kind, const int64_t* src_strides /* host[4] */, void* dst, int32_t dst_kind,
                      const int64_t* dst_strides /* host[4] */, const int32_t* dims /* host[4] */, void* stream);
/* y = act(x), fp16, kind 0 = quick_gelu x*sigmoid(1.702x) (CLIP-L), 1 = exact erf GELU (CLIP-G), 2 = ReLU (T2I-Adapter ResnetBlock) */
int fmx_act_f16(const void* x, void* y, int64_t n, int32_t kind, void* stream);
/* CLIP text embeddings (transformers CLIPTextEmbeddings): out[b*T + t][:] = tok_emb[ids[b*T + t]][:] + pos_emb[t][:], fp16, c % 8 == 0 */
int fmx_embed_tokens(const int32_t* ids, const void* tok_emb, const void* pos_emb, void* out, int32_t batch, int32_t tokens, int32_t c,
                     int32_t vocab, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Sampler-side fused elementwise (fp32 latents NCHW [b][c][h][w], as the reference keeps them)
 * ---------------------------------------------------------------------------------------------- */
/* UNet input pack: xc = x / sqrt(sigma^2 + sigma_data^2) (k_prediction.py:74-79) for `reps` stacked copies
 * of the batch (reps = 2 under CFG: [uncond ; cond]), written as the 3x3 im2col matrix of the first conv:
 * out fp16 [reps*b*h*w][64], column (ky*3+kx)*c + ch for c*9 <= 64, zero elsewhere (zero padding taps). */
int fmx_unet_pack_input(const float* x, const float* sigma, float sigma_data, int32_t b, int32_t c, int32_t h,
                        int32_t w, int32_t reps, void* out, void* stream);

/* eps: fp16 NHWC [reps*b][h][w][ld_eps>=c], the model output.  Per half, calculate_denoised (k_prediction.py:81-92):
 *   prediction_type 0 (epsilon; also Flux 'const'): x - out*sigma
 *   prediction_type 1 (v_prediction): x*sd^2/(sigma^2+sd^2) - out*sigma*sd/sqrt(sigma^2+sd^2)      (sd = sigma_data)
 *   prediction_type 2 (edm):          x*sd^2/(sigma^2+sd^2) + out*sigma*sd/sqrt(sigma^2+sd^2)
 * reps == 2: eps holds [uncond ; cond]; out = uncond + (cond - uncond)*cond_scale; reps == 1: out = cond.
 * Optional outputs cond_pred / uncond_pred (fp32 NCHW, may be null). */
int fmx_cfg_combine(const void* eps, int32_t ld_eps, const float* x, const float* sigma, int32_t b, int32_t c,
                    int32_t h, int32_t w, int32_t reps, float cond_scale, float* denoised, float* cond_pred,
                    float* uncond_pred, int32_t prediction_type, float sigma_data, void* stream);

/* x_out = x + (x - denoised)/sigma * (sigma_next - sigma)   (Euler; also the deterministic part of Euler a
 * with sigma_next := sigma_down); if noise != null: x_out += noise * noise_scale. */
int fmx_sampler_euler_step(const float* x, const float* denoised, float sigma, float sigma_next, const float* noise,
                           float noise_scale, float* x_out, int64_t n, void* stream);
/* x_out = a*x + bcoef*denoised + ccoef*old_denoised (old may be null when ccoef == 0)  (DPM++ 2M update) */
int fmx_sampler_lincomb3(const float* x, const float* denoised, const float* old_denoised, float a, float bcoef,
                         float ccoef, float* x_out, int64_t n, void* stream);
/* x_out = sum_{k < n_terms} coefs[k] * srcs[k]  (1 <= n_terms <= 8; `srcs` / `coefs` are HOST arrays of device pointers / scalars;
 * x_out may alias a source).  The update of every multi-stage / multistep k-diffusion sampler with host-side coefficients:
 * Heun, DPM2(a), DPM++ 2S a, LMS, HeunPP2, IPNDM(_V), DEIS, Restart (k_diffusion/sampling.py:189-341, 573-603, 771-981,
 * modules/sd_samplers_extra.py:7-74). */
int fmx_sampler_lincomb(const float* const* srcs, const float* coefs, int32_t n_terms, float* x_out, int64_t n, void* stream);
/* out[0] = || (x_low - x_high) / max(atol, rtol * max(|x_low|, |x_prev|)) ||_2 / sqrt(n)   -- the local error estimate of the adaptive
 * DPM-Solver (k_diffusion/sampling.py:531-532).  workspace: >= 256 floats of device memory.  Deterministic summation order. */
int fmx_sampler_error_norm(const float* x_low, const float* x_high, const float* x_prev, float atol, float rtol, float* workspace,
                           float* out, int64_t n, void* stream);
/* Separable linear resize of fp32 planes [planes][h][w] -> [planes][oh][ow]:
 *   out[p][oy][ox] = sum_{a<ky} sum_{b<kx} yweights[oy][a] * xweights[ox][b] * in[p][ystart[oy] + a][xstart[ox] + b]
 * (tables in device memory; ystart[oy] + ky <= h and xstart[ox] + kx <= w).  The hires-fix latent upscale, i.e. what
 * torch.nn.functional.interpolate does at modules/processing.py:1459 for bilinear / bicubic / nearest(-exact), antialiased or not. */
int fmx_resize_separable_f32(const float* in, float* out, const int32_t* ystart, const float* yweights, const int32_t* xstart,
                             const float* xweights, int32_t planes, int32_t h, int32_t w, int32_t oh, int32_t ow, int32_t ky, int32_t kx,
                             void* stream);
int fmx_scale_f32(const float* x, float s, float* y, int64_t n, void* stream);
/* out = a * a_mask + b * b_mask, fp32, elementwise over n (inpaint latent blending: modules/sd_samplers_cfg_denoiser.py:181,205
 * `x * nmask + noisy_init * mask`, `denoised * nmask + init_latent * mask`; processing.py:1866).  out may alias a or b. */
int fmx_blend_masked(const float* a, const float* a_mask, const float* b, const float* b_mask, float* out, int64_t n, void* stream);

/* VAE: latent fp32 NCHW [b][c][h][w] -> (z/scaling_factor + shift) as fp16 NHWC [b*h*w][ld] (ld >= c, rest 0) */
int fmx_vae_pack_latent(const float* z, float scaling_factor, float shift, int32_t b, int32_t c, int32_t h, int32_t w,
                        void* out, int32_t ld, void* stream);
/* generic 3x3 im2col for tiny channel counts: x fp16 NHWC [n][h][w][ldx] (first c channels) -> out fp16
 * [n*h*w][64] with column (ky*3+kx)*c + ch, c*9 <= 64 */
int fmx_im2col3x3_smallc(const void* x, int32_t ldx, int32_t n, int32_t c, int32_t h, int32_t w, void* out, void* stream);
/* 3x3 convolution (stride 1, zero padding 1) with at most 4 output channels: the VAE decoder's conv_out (/root/reference/backend/nn/vae.py:248-271, the
 * last layer: 128 -> 3 channels at the full image size) and the UNet's `out` convolution (backend/nn/unet.py:760-764: 320 -> 4).  x NHWC [n][h][w][c] 16-bit,
 * c a multiple of 32 (walked in channel chunks of 128 / 64 / 32); wgt [nout][ky][kx][c] (the GEMM entry's
 * weight layout); bias [nout] or null; out [n*h*w][ld_out], columns >= nout of an ld_out = 4 output are written as zeros.  A direct kernel (the input
 * patch of a 4 x 32 pixel tile staged once in LDS) instead of the implicit GEMM's nine-fold im2col gather; HBM-bound (ABI 8). */
int fmx_conv3x3_narrow_f16(const void* x, int32_t n, int32_t h, int32_t w, int32_t c, const void* wgt, const void* bias, int32_t nout, void* out,
                           int32_t ld_out, void* stream);
/* GroupNorm + SiLU + 3x3 convolution (stride 1, padding 1) in one kernel, 128 output channels: the `norm -> swish -> conv` halves of the VAE's
 * ResnetBlock.forward (backend/nn/vae.py:98-114: norm1 / conv1, norm2 / conv2 + `x + h`) at the decoder's full-resolution level (ABI 11).
 *   out[m][o] = bias[o] + residual[m][o] + sum_{ky,kx,c} wgt[o][ky][kx][c] * silu(x[pixel(m) + (ky-1, kx-1)][c] * scale[c] + shift[c])     (zero outside the image)
 * with {scale, shift} = GroupNorm(groups, eps, gamma, beta) of x from its chunk statistics x_partial [n][x_nchunks][cin][2] ({sum, sum of squares} as the
 * producing GEMM's epilogue / fmx_groupnorm_stats left them); the table is written to scale_shift [n][cin][2] (workspace, fp32).  x NHWC [n][h][w][cin],
 * cin a multiple of 64; wgt in the GEMM entry's layout [128][ky][kx][cin]; bias [128] or null; residual [n*h*w][ld_res] or null; one rounding of the
 * sum.  stats (optional): [n][stats_cap][128][2] receives the GroupNorm statistics of the OUTPUT as stored, one record per 8 x 32-pixel tile
 * (*stats_nchunks = records per image, 0 without stats) -- the operand of the next fmx_groupnorm_apply / fmx_conv3x3_gn_silu.
 * The normalised tensor the unfused pair (fmx_groupnorm_apply + fmx_gemm_conv) stores between its launches exists only as a tile's LDS patch here,
 * holding the same values (same arithmetic, same rounding). */
typedef struct fmx_conv_gn_args {
  const void* x;
  int32_t n, h, w, cin;
  const float* x_partial;
  int32_t x_nchunks, groups;
  float eps;
  const void* gamma;
  const void* beta;
  float* scale_shift;
  const void* wgt;
  int32_t cout;
  const void* bias;
  const void* residual;
  int64_t ld_res;
  void* out;
  int64_t ld_out;
  float* stats;
  int32_t stats_cap;
} fmx_conv_gn_args;
int fmx_conv3x3_gn_silu_f16(const fmx_conv_gn_args* args /* host */, int32_t* stats_nchunks /* host, may be null */, void* stream);
/* Upsample: 3x3 convolution (padding 1) of the x2 NEAREST-upsampled input -- backend/nn/unet.py:328-355 (Upsample.forward: F.interpolate(scale 2,
 * "nearest") then conv) and backend/nn/vae.py:35-57 -- WITHOUT the nine-fold work on the upsampled grid (ABI 11).  Of the 3 x 3 taps of an output pixel
 * only 2 x 2 DISTINCT input pixels are touched: output rows 2iy read input rows {iy-1, iy} with weights {w[0], w[1]+w[2]}, rows 2iy+1 read {iy, iy+1}
 * with {w[0]+w[1], w[2]}; columns alike.  So the layer is FOUR 2 x 2 convolutions on the input grid, one per output parity (py, px), each writing every
 * second pixel of every second output row: 4 / 9 of the multiply-adds, no upsampled tensor, no gather.
 *   x    : fp16 NHWC [n][h][w][c], c a multiple of 64, w a multiple of 32
 *   wgt4 : fp16 [4][nout][2][2][c] -- phase 2*py + px, taps (dy, dx) row-major -- the tap SUMS of the layer's [nout][3][3][c] weight, formed in fp32
 *          and rounded once to the element type by the caller (forge_amd.hipops.fold_up2x_weights): one more rounding of the same size as the
 *          weight's own, which the rounding oracles of tests/ model at exactly this site
 *   out  : fp16 NHWC [n][2h][2w][nout] dense; bias [nout] or null
 *   stats: optional GroupNorm statistics of the output, [n][stats_cap][nout][2], 4 * h*w/256 records per image (h*w a multiple of 256)
 * Runs on the 256-row implicit-GEMM kernels (their bit-mask address form, K = 4 c) with scattered output rows. */
int fmx_conv3x3_up2x_f16(const void* x, int32_t n, int32_t h, int32_t w, int32_t c, const void* wgt4, const void* bias, int32_t nout, void* out,
                         float* stats, int32_t stats_cap, int32_t* stats_nchunks /* host, may be null */, const void* zero_page, void* stream);
/* fmx_conv3x3_narrow_f16 on silu(group_norm(x)) without storing it (ABI 11): the `norm_out -> swish -> conv_out` tail of the VAE decoder (backend/nn/vae.py:266-271)
 * and of the UNet (`out`, backend/nn/unet.py:760-764).  x is the UN-normalised tensor with its chunk statistics x_partial [n][x_nchunks][c][2]; the (image, channel)
 * {scale, shift} table is written to scale_shift [n][c][2] (fp32 workspace) and applied while the tile's input patch is staged in LDS, with fmx_groupnorm_apply's
 * arithmetic and rounding (the staged values are the tensor the two-launch form stores; zeros outside the image).  Other arguments as fmx_conv3x3_narrow_f16. */
int fmx_conv3x3_narrow_gn_silu_f16(const void* x, int32_t n, int32_t h, int32_t w, int32_t c, const float* x_partial, int32_t x_nchunks, int32_t groups, float eps,
                                   const void* gamma, const void* beta, float* scale_shift, const void* wgt, const void* bias, int32_t nout, void* out,
                                   int32_t ld_out, void* stream);
/* VAE output: y fp16 NHWC [b*h*w][ld] (first c channels) -> clamp((y+1)/2, 0, 1) fp32 NHWC [b][h][w][c] */
int fmx_vae_unpack_image(const void* y, int32_t ld, int64_t npix, int32_t c, float* out, void* stream);

/* *count (device int32) = how many of the n fp16 values at x (16-byte aligned) are inf or NaN.  The VAE's overflow guard: an fp16 decode whose
 * output is not finite is repeated in bfloat16 (trained SDXL VAE weights leave fp16's range; the reference avoids fp16 there altogether,
 * backend/memory_management.py:190-205). */
int fmx_count_nonfinite_f16(const void* x, int64_t n, int32_t* count, void* stream);

/* Posterior sample of the VAE encoder (backend/nn/vae.py:16-29 DiagonalGaussianDistribution.sample, :312-313 process_in):
 *   out[b][c][p] = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise[b][c][p] - shift) * scale
 * moments: fp16 [B*npix][ld], channels 0..lc-1 = mean, lc..2lc-1 = logvar (the quant_conv output, NHWC);
 * noise / out: fp32 NCHW [B][lc][npix].  scale = 1, shift = 0 gives the raw sample. */
int fmx_vae_sample_posterior(const void* moments, int32_t ld, const float* noise, int32_t b, int32_t lc, int64_t npix, float scale, float shift,
                             float* out, void* stream);

/* Philox4x32-10 + Box-Muller ("NV" noise source, modules/rng_philox.py:32-102): out[i], i < n, for
 * counter (offset, 0, i, 0) and key = seed.  If raw_u32 != null also stores the 4 raw words per i. */
int fmx_philox_randn(uint64_t seed, uint32_t offset, float* out, uint32_t* raw_u32, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * bfloat16 build of the Flux path.  The reference computes Flux in bf16 (backend/loader.py picks the storage dtype of the checkpoint,
 * bf16 for Flux.1; fp16 overflows in the late single-stream blocks of trained weights).  These entry points are the SAME kernels as
 * their _f16 counterparts compiled with bfloat16 elements (storage, MFMA operands; fp32 accumulation, softmax and epilogue math as
 * before): identical argument contracts, every "fp16" in the descriptions above reads "bf16".  fmx_layernorm_padded / fmx_softmax_rows come
 * along with their files; the Flux executor uses the other seven.
 * ---------------------------------------------------------------------------------------------- */
int fmx_gemm_conv_bf16(const fmx_gemm_args* args /* host */, void* stream);
int fmx_attention_bf16(const fmx_attn_args* args /* host */, void* stream);
int fmx_softmax_rows_bf16(void* x, int64_t nrows, int32_t ncols, int64_t ld, void* stream);
int fmx_layernorm_bf16(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int32_t c, float eps, void* stream);
int fmx_rmsnorm_bf16(const void* x, const void* weight, void* y, int64_t rows, int32_t c, float eps, void* stream);
int fmx_layernorm_padded_bf16(const void* x, const void* gamma, const void* beta, void* y, int64_t rows, int32_t c, float eps,
                              int64_t rows_per_image, int64_t out_rows_per_image, void* stream);
int fmx_layernorm_mod_bf16(const void* x, const void* scale, const void* shift, int64_t ld_mod, int64_t rows_per_batch, void* y,
                           int64_t rows, int32_t c, float eps, void* stream);
int fmx_flux_qk_norm_rope_bf16(const void* qkv, int64_t ld_qkv, const void* q_scale, const void* k_scale, const float* pe, void* q_out,
                               void* k_out, void* vt_out, int32_t batch, int32_t tokens, int32_t heads, int32_t head_dim,
                               int32_t row_off, int32_t l_pad, float eps, void* stream);
int fmx_timestep_embedding_bf16(const float* t, void* emb, int32_t b, int32_t dim, float max_period, void* stream);
int fmx_silu_bf16(const void* x, void* y, int64_t n, void* stream);

/* bfloat16 build of the VAE (ABI 6).  The reference decodes in bf16 wherever the part supports it and in fp32 otherwise
 * (backend/memory_management.py:190-205 VAE_DTYPES, :840-855 vae_dtype(); --vae-in-fp16 / --vae-in-bf16 select by hand): trained SDXL VAE
 * weights overflow fp16 in the decoder's upper levels.  Same kernels, same contracts as the _f16 entry points of the same name ("fp16" reads
 * "bf16": activations, weights, norm parameters; statistics, accumulation and softmax stay fp32); the 3x3 im2col of tiny channel counts
 * (fmx_im2col3x3_smallc) moves 16-bit words and serves both types. */
int fmx_gemm_conv_stats_bf16(const fmx_gemm_args* args /* host */, float* partial, int32_t max_chunks, int32_t fallback_chunks,
                             int32_t* chunks_out /* host */, void* stream);
int fmx_groupnorm_stats_bf16(const void* x, int32_t c, int64_t ld, int32_t n, int32_t hw, float* partial, int32_t nchunks, void* stream);
int fmx_groupnorm_apply_bf16(const void* x0, const void* x1, int32_t c0, int32_t c1, int64_t ld0, int64_t ld1, int32_t n, int32_t hw,
                             const float* partial0, int32_t nchunks0, const float* partial1, int32_t nchunks1, int32_t groups, float eps,
                             const void* gamma, const void* beta, int32_t silu, float* scale_shift, void* y, void* stream);
int fmx_attention_single_head512_bf16(const void* q, int64_t q_bs, int64_t q_rs, const void* k, int64_t k_bs, int64_t k_rs, const void* vt,
                                      int64_t vt_bs, int64_t vt_ds, void* o, int64_t o_bs, int64_t o_rs, int32_t batch, int32_t nq, int32_t nk,
                                      int32_t nk_pad, float scale, void* stream);
int fmx_vae_pack_latent_bf16(const float* z, float scaling_factor, float shift, int32_t b, int32_t c, int32_t h, int32_t w,
                             void* out, int32_t ld, void* stream);
int fmx_vae_unpack_image_bf16(const void* y, int32_t ld, int64_t npix, int32_t c, float* out, void* stream);
int fmx_conv3x3_narrow_bf16(const void* x, int32_t n, int32_t h, int32_t w, int32_t c, const void* wgt, const void* bias, int32_t nout, void* out,
                            int32_t ld_out, void* stream);
int fmx_conv3x3_narrow_gn_silu_bf16(const void* x, int32_t n, int32_t h, int32_t w, int32_t c, const float* x_partial, int32_t x_nchunks, int32_t groups, float eps,
                                    const void* gamma, const void* beta, float* scale_shift, const void* wgt, const void* bias, int32_t nout, void* out,
                                    int32_t ld_out, void* stream);
int fmx_conv3x3_up2x_bf16(const void* x, int32_t n, int32_t h, int32_t w, int32_t c, const void* wgt4, const void* bias, int32_t nout, void* out,
                          float* stats, int32_t stats_cap, int32_t* stats_nchunks, const void* zero_page, void* stream);
int fmx_conv3x3_gn_silu_bf16(const fmx_conv_gn_args* args /* host */, int32_t* stats_nchunks /* host, may be null */, void* stream);
int fmx_vae_sample_posterior_bf16(const void* moments, int32_t ld, const float* noise, int32_t b, int32_t lc, int64_t npix, float scale,
                                  float shift, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * HIP-graph helpers: capture everything launched on `stream` between begin/end into an executable graph.
 * ---------------------------------------------------------------------------------------------- */
int fmx_graph_begin(void* stream);
int fmx_graph_end(void* stream, void** graph_exec_out /* host */);
int fmx_graph_launch(void* graph_exec, void* stream);
int fmx_graph_destroy(void* graph_exec);

/* HIP event timing on an arbitrary stream (torch.cuda.Event only sees torch's current stream). */
int fmx_event_create(void** ev_out /* host */);
int fmx_event_record(void* ev, void* stream);
int fmx_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out /* host */); /* synchronises ev_stop */
int fmx_event_destroy(void* ev);

#ifdef __cplusplus
}
#endif
#endif /* FMX_H */
