/* dvla.h -- C ABI of libdvla_hip.so: the hand-written CDNA4 (gfx950) kernels under DreamVLA's
 * transformer hot path.
 *
 * The reference (Zhangwenyao1/DreamVLA) has no native code and no FFI: every "kernel" on this path is
 * an ATen / F.scaled_dot_product_attention call issued from eager Python.  Each
 * entry point below therefore cites the reference *call sites* whose ATen op(s) it replaces; the
 * Python-side binding (ctypes) is dreamvla_amd/_lib.py and the drop-in nn.Module surface is
 * models/dreamvla_model.py.
 *
 * Conventions: plain pointers + sizes, no torch types.  All tensors are device pointers.  bf16 is the
 * raw 16-bit pattern.  Every function enqueues work on `stream` (a hipStream_t passed as void*),
 * never synchronises, is re-entrant per stream, and returns 0 on success or a negative DVLA_ERR_* code (no
 * exceptions cross the ABI).  Workspaces are the caller's (torch's caching allocator) with ONE exception: the
 * stream-K schedules of dvla_gemm_bf16 keep a 64-MiB slab area + flags per (device, stream), hipMalloc'ed at the
 * first launch that uses it (never during stream capture) and kept for the life of the process -- see
 * dvla_set_gemm_variant / dvla_set_gemm_schedule.  dtype codes: 0 = bf16, 1 = fp32.
 */
#ifndef DVLA_H_
#define DVLA_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DVLA_DT_BF16 0
#define DVLA_DT_F32 1

/* activation codes (applied in fp32 on the accumulator) */
#define DVLA_ACT_NONE 0
#define DVLA_ACT_GELU_ERF 1   /* timm Mlp nn.GELU():            models/vit_mae.py:73, dreamvla_model.py:349 */
#define DVLA_ACT_GELU_TANH 2  /* HF ACT2FN["gelu_new"]:         models/gpt2.py:292-301; DiT approx_gelu models.py:133 */
#define DVLA_ACT_RELU 3       /* action MLP head / depth pred:  dreamvla_model.py:458-471,842 */
#define DVLA_ACT_SILU 4       /* DiT TimestepEmbedder:          action_model/models.py:34 */
#define DVLA_ACT_QUICK_GELU 5 /* CLIP text tower QuickGELU (openai/CLIP model.py) */
#define DVLA_ACT_TANH 6
#define DVLA_ACT_SIGMOID 7

/* ABI revision of this header; dreamvla_amd/_lib.py refuses a library that reports another one (a stale prebuilt .so then fails
 * with a clear message instead of a missing-symbol error).  3 = round 3 (dvla_last_gemm_variant, variant 10, ...); 4 = k-sums on the
 * weight-gradient GEMM (ksum_* fields of dvla_gemm_params); 5 = dvla_ddim_cfg_step, dvla_act_bwd_colsum, a_layernorm,
 * GEMM configuration 11; 6 = dvla_dit_sample (the evaluation sampler as one persistent kernel); 7 = round 5: the sampler's
 * status word no longer carries over to the next launch (workspace words 33 / 34), dvla_dit_sample_inject_timeouts. */
#define DVLA_ABI_VERSION 8
int dvla_abi_version(void);

/* ---------------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = epilogue( A[M,K] * B[N,K]^T )     bf16 operands, fp32 MFMA accumulate
 *   a_trans = 0: A(m,k) at A[m*lda + k]      a_trans = 1: A(m,k) at A[k*lda + m]
 *   b_trans = 0: B(n,k) at B[n*ldb + k]  (nn.Linear weight (out,in))
 *   b_trans = 1: B(n,k) at B[k*ldb + n]  (HF Conv1D weight (in,out), models/gpt2.py:53-54,291-292)
 * epilogue, in order: v = acc; v += bias[n]; preact[m,n] = v; v = act(v); v *= act'(dact_aux[m,n]) (dact);
 *   v = dropout(v) (p, seed; element (m,n)); v += residual[m,n]; C = v (or C += v when accumulate, fp32 C).
 * Replaces: nn.Linear / Conv1D / timm PatchEmbed-as-GEMM + bias + GELU + dropout + residual add at
 *   models/vit_mae.py:188-203 (via timm Block), models/gpt2.py:160,172-173,296-301,329-339,
 *   models/perceiver_resampler.py:11-18,49-51,61, models/dreamvla_model.py:652-664,718-724,800-809,
 *   models/action_model/models.py:34-41,128-141,158-160, and their autograd backward GEMMs.
 * split_k > 1: K is cut in split_k slices written to `workspace` (split_k*M*N fp32) and reduced by a
 *   second kernel; only valid with the plain epilogue (used for weight gradients).
 */
typedef struct dvla_gemm_params {
  const void* A; int64_t lda; int32_t a_trans;
  const void* B; int64_t ldb; int32_t b_trans;
  void* C; int64_t ldc; int32_t c_dtype;
  int64_t M, N, K;
  const void* bias; int32_t bias_dtype;
  int32_t act;
  void* preact; int64_t ld_preact;
  const void* dact_aux; int64_t ld_dact; int32_t dact;
  float dropout_p; uint32_t seed_lo; uint32_t seed_hi;
  const void* residual; int64_t ld_res; int64_t res_rows; /* >0: residual row = m % res_rows (broadcast tables) */
  int32_t accumulate;
  int32_t split_k; void* workspace;
  /* (ABI 4) k-sums of one operand, computed by the GEMM that streams it anyway -- the bias gradient db = sum_m dz[m, :] of a
   * weight-gradient GEMM dW = dz^T x (utils/train_utils.py:599-608 leaves it to autograd's sum kernel):
   *   ksum_operand 0: none; 1: ksum[i] = sum_k A(i, k), i < M; 2: ksum[j] = sum_k B(j, k), j < N   (fp32 accumulation of the
   *   bf16 operand values, result in ksum_dtype = DVLA_DT_F32 / DVLA_DT_BF16).
   *   ksum_workspace: dvla_gemm_ksum_partial_rows(split_k) x (M or N) floats.
   * The ring kernels sum the fragments they feed to the matrix pipe (one v_dot2c_f32_bf16 per dword, in the MFMA
   * shadow); since ABI 7 the phase kernel sums too (both operands r-contiguous: the weight-gradient layout), in the slack of its
   * fragment-read segments, the sum of a tile row spread over up to four of its tiles -- dvla_gemm_ksum_partial_rows() accounts
   * for their partial rows.  A configuration without that code runs the column-sum kernel on the operand instead, which needs
   * the operand stored k-major (a_trans / b_trans = 1); otherwise DVLA_ERR_UNSUPPORTED. */
  void* ksum; int32_t ksum_dtype; int32_t ksum_operand; float* ksum_workspace;
  /* (ABI 5) a_layernorm != 0: the rows of A are layer-normalised on their way into the product -- A'(m, :) = (A(m, :) - mean_m) *
   * rsqrt(var_m + a_ln_eps), no affine parameters, rounded to bf16 like the output of dvla_layernorm_fwd -- i.e. the GEMM computes
   * Linear(LayerNorm(x)) from x.  The DiT blocks' parameter-free LayerNorms in front of qkv / fc1 / the output layer
   * (models/action_model/models.py:129-141,158-160) at evaluation: 250 launches of a control step less.  Few-rows kernel only
   * (configuration 11: M <= 512, k-contiguous operands, 512 <= K <= 1536); DVLA_ERR_UNSUPPORTED otherwise. */
  int32_t a_layernorm; float a_ln_eps;
} dvla_gemm_params;
int64_t dvla_gemm_ksum_partial_rows(int32_t split_k);
int dvla_gemm_bf16(const dvla_gemm_params* p, void* stream);
/* tuning hook: 0 = automatic kernel choice by the built-in cost model (default; env DVLA_GEMM_VARIANT overrides at load);
 * 11 = the few-rows kernel (M <= 512, both operands k-contiguous, K % 16 == 0: one 32 x 32 tile of C per workgroup, K split over
 * its four or eight waves, fragments straight from global memory -- the evaluation-time shapes; what 0 picks for such problems);
 * 2 = register-staged 128x128 kernel; 4 / 6 / 7 = LDS-DMA ring kernels 256x256 / 128x128 / 256x128 (K-tile 64); 8 = phase
 * kernel (256x256, K-tile 64, two wave groups in ping-pong); 9 = the phase kernel under the stream-K hybrid schedule (whole
 * rounds one tile per CU; the last, partial rounds as equal K-iteration ranges per group of 16 CUs, the two halves of a
 * shared tile combined in-kernel through a 256-KiB fp32 slab); 11 = the few-rows kernel (M <= 512).  A configuration that does
 * not take a shape falls back
 * inside the library.  All differ only in fp32 summation order.  81..89 = ablation / timeline builds of the phase kernel
 * (wrong results by design, tests/probes/gemm_probe.cpp).
 * Stream-K scratch: 64 MiB + flags per (device, stream), hipMalloc'ed on the first launch that uses it (never while the
 * stream is being captured -- such launches take the plain schedule) and kept; DVLA_GEMM_STREAMK=0 turns the schedule off. */
void dvla_set_gemm_variant(int variant);
/* 10 = the phase kernel under the FULL stream-K schedule: every tile belongs to the K-iteration ranges, so the groups' tile
 * boundaries (epilogue write bursts) fall at different times for the whole launch instead of in lockstep rounds.
 * What the last dvla_gemm_bf16 call of this process launched: 2 / 4 / 6 / 7 / 8 as above, 9 / 10 only when the stream-K
 * schedule actually ENGAGED (otherwise 8: the plain schedule of the same kernel); 0 before the first call.  Tests use it to
 * assert that a forced configuration ran instead of falling back. */
int dvla_last_gemm_variant(void);
/* How the persistent kernels spread tiles over the CUs (process-wide; env DVLA_GEMM_OVERSUBSCRIBE / DVLA_GEMM_STREAMK at load).
 *   oversubscribe = 1 (default): one workgroup per CU slot walks a fixed share of the tiles; the stream-K schedule may apply.
 *     Fastest on a GPU that runs nothing else (the default bench: 165 ms / step against 176 with k = 8).
 *   oversubscribe = k > 1: up to k times as many, shorter-lived workgroups with equal tile counts, handed out by the hardware
 *     dispatcher as slots free up.  For callers whose GEMMs run for long stretches next to another stream's kernel that holds
 *     CUs: with a fixed share per CU a launch waits for the workgroups that could not start (measured +65 % per launch for
 *     16 of 256 CUs taken; k = 8: +0 ... +9 %; profiles/r02_gemm_cu_contention.txt).  The data-parallel path keeps k = 1: its
 *     bucket all-reduces (0.99 GB per step) occupy CUs for a few milliseconds of a 165-ms step.
 *   stream_k = 0 / 1: forbid / allow the stream-K schedule (needs every workgroup co-resident); -1 / oversubscribe < 1: keep. */
void dvla_set_gemm_schedule(int oversubscribe, int stream_k);
/* the schedule in force (either pointer may be NULL).  dreamvla_amd.ddp.GradBucketReducer switches to (8, 0) -- equal-sized,
 * short-lived workgroups, no co-residency assumption -- while its RCCL all-reduces are outstanding (their kernels hold CUs for
 * the whole transfer) and back afterwards. */
void dvla_get_gemm_schedule(int* oversubscribe, int* stream_k);

/* ---------------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (rows x cols, bf16 in/out, fp32 statistics).
 * gamma/beta may be NULL (DiT elementwise_affine=False, action_model/models.py:129-131,148).
 * Replaces nn.LayerNorm at models/vit_mae.py:77,202-204 (eps 1e-6), models/gpt2.py:312-315,437
 * (eps 1e-5), models/dreamvla_model.py:279,352,374,393,412,433, models/perceiver_resampler.py:14,28-29,101.
 * fwd writes mean/rstd (fp32, one per row) when they are non-NULL (needed by bwd).
 * bwd: dx (NULL, round 6: the input needs no gradient -- parameter gradients only); dgamma/dbeta (fp32, cols) when non-NULL, using
 *      `partial` = fp32 workspace of
 *      2 * dvla_layernorm_bwd_partial_rows() * cols floats.
 */
int dvla_layernorm_fwd(const void* x, const void* gamma, const void* beta, int32_t param_dtype, void* y,
                       float* mean, float* rstd, int64_t rows, int64_t cols, float eps, void* stream);
int dvla_layernorm_bwd(const void* dy, const void* x, const void* gamma, int32_t param_dtype,
                       const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                       float* partial, int64_t rows, int64_t cols, void* stream);
/* same, plus the gradient of the residual stream that bypassed the LayerNorm (pre-LN blocks: h = x + f(LN(x)), so
 * dL/dx = dL/dh + LN'(...)): dx = dres + LN-backward(dy), one bf16 rounding -- replaces the separate autograd
 * accumulation add after every norm1 / norm2 / ln_1 / ln_2 backward (models/gpt2.py:312-339, timm Block).  dres must
 * not overlap dx; dres == NULL gives plain LayerNorm backward.  dgamma / dbeta are written in `grad_dtype` (0 = bf16:
 * the parameters' dtype, no cast kernel afterwards; 1 = fp32). */
int dvla_layernorm_bwd_add(const void* dy, const void* x, const void* gamma, int32_t param_dtype,
                           const float* mean, const float* rstd, const void* dres, void* dx, void* dgamma,
                           void* dbeta, int32_t grad_dtype, float* partial, int64_t rows, int64_t cols, void* stream);
int64_t dvla_layernorm_bwd_partial_rows(void);
/* (ABI 8) LayerNorm over the LAST `grp` rows of every `gstride`-row sequence of a (n_seq * gstride, cols) buffer, `goff` = gstride - grp
 * rows skipped in front of each group -- the dream-head decoders normalise only their mask-token rows before the prediction layer
 * (`x = self.image_decoder_norm(x[:, -n_mask:, :])`, models/dreamvla_model.py:812-816 and the depth / dino / sam / trajectory twins):
 * the slice is strided across sequences, and as an ATen copy (+ a zero-fill and a copy-back in backward) it was six launches and ~0.5 ms
 * per step.  rows = n_seq * grp logical rows; y, mean, rstd, dy are contiguous over them; x is read, and dx (the gradient of the WHOLE
 * buffer: zeros outside the groups, written by the kernel) is stored, in the buffer's rows.  General in goff: 0 <= goff, goff + grp <= gstride.
 * map_output != 0 is the mirror image: x (and dx) contiguous over the logical rows, the OUTPUT y (forward) and the incoming gradient dy
 * (backward) in the buffer's rows, nothing zero-filled -- two LayerNorms writing the two row ranges of one buffer replace
 * `torch.cat((norm_media(x), norm_latents(latents)), dim=-2)` of the PerceiverResampler (models/perceiver_resampler.py:44-49) and, in
 * backward, the copies of the cat's strided gradient slices. */
int dvla_layernorm_fwd_rows(const void* x, const void* gamma, const void* beta, int32_t param_dtype, void* y,
                            float* mean, float* rstd, int64_t rows, int64_t cols, float eps,
                            int32_t grp, int32_t gstride, int32_t goff, int32_t map_output, void* stream);
int dvla_layernorm_bwd_rows(const void* dy, const void* x, const void* gamma, int32_t param_dtype,
                            const float* mean, const float* rstd, void* dx, void* dgamma, void* dbeta,
                            int32_t grad_dtype, float* partial, int64_t rows, int64_t cols,
                            int32_t grp, int32_t gstride, int32_t goff, int32_t map_output, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused multi-head attention, head_dim = 64 (every attention in DreamVLA: ViT 768/12, trunk 1024/16,
 * dream-head decoders 1024/16, DiT 768/12, perceiver dim_head 64, CLIP text 512/8).
 *   O[b,i,h,:] = sum_j drop( softmax_j( scale * q[b,i,h,:].k[b,j,h,:] + mask[i,j] ) ) * v[b,j,h,:]
 * q/k/v/o element (b, token, head, d) lives at ptr[b*stride_b + token*stride_t + head*stride_h + d], so
 * the fused (B,N,3,h,d) timm qkv buffer, the GPT-2 c_attn (B,L,3H) buffer and the perceiver q / kv
 * buffers are read in place (no head-split copies).
 * Masks: the reference only ever builds 0 / -inf additive masks (generate_attention_mask,
 * models/dreamvla_model.py:25-66; CLIP's causal mask).  The host converts such a mask once into
 *   mask_bits_q (Lq x ceil(Lk/32) uint32): bit j of [i][kt] set <=> key 32*kt+j is visible to query i
 *   mask_bits_k (Lk x ceil(Lq/32) uint32): bit i of [j][qt] set <=> query 32*qt+i sees key j   (backward)
 *   tile_map    (ceil(Lq/32) x ceil(Lk/32) uint8): 0 = all -inf (tile skipped), 1 = all visible, 2 = mixed
 * so masked tiles cost nothing (64-81 % of the trunk's score matrix) and a mixed tile
 * costs one 32-bit load per lane.  key_index (Lk int32, optional) names the k/v (and dk/dv) token row that
 * holds key j (any order, every entry < 2^18): keys that no query can see are dropped from the key axis without
 * copying K/V, and the caller is free to ORDER the kept keys so that keys with the same audience share 32-key tiles
 * (dreamvla_amd.ops.build_mask_tables does); dk/dv rows that are not indexed are NOT written (the caller zero-fills them).
 * lse (B*H*Lq fp32, natural-log-sum-exp of the scaled+masked scores) is written when non-NULL.
 * Replaces: F.scaled_dot_product_attention in timm Attention (vit_mae.py:202-203, dreamvla_model.py:806-904,
 * action_model/models.py:137), GPT2Attention._attn / GPT2SdpaAttention (models/gpt2.py:61-84,267-274),
 * PerceiverAttention einsum-softmax-einsum (models/perceiver_resampler.py:55-60).
 */
typedef struct dvla_attn_params {
  const void* q; const void* k; const void* v; void* o;
  int64_t q_stride_b, q_stride_t, q_stride_h;
  int64_t k_stride_b, k_stride_t, k_stride_h;
  int64_t v_stride_b, v_stride_t, v_stride_h;
  int64_t o_stride_b, o_stride_t, o_stride_h;
  int32_t B, H, Lq, Lk;
  float scale;
  const int32_t* key_index;
  const uint32_t* mask_bits_q;
  const uint32_t* mask_bits_k;
  const uint8_t* tile_map;
  float dropout_p; uint32_t seed_lo; uint32_t seed_hi;
  float* lse;
  /* backward only */
  const void* dout; int64_t do_stride_b, do_stride_t, do_stride_h;
  float* delta; /* workspace B*H*Lq fp32: rowsum(dO * O) */
  void* dq; void* dk; void* dv;
  int64_t dq_stride_b, dq_stride_t, dq_stride_h;
  int64_t dk_stride_b, dk_stride_t, dk_stride_h;
  int64_t dv_stride_b, dv_stride_t, dv_stride_h;
} dvla_attn_params;
int dvla_attn_fwd(const dvla_attn_params* p, void* stream);
int dvla_attn_bwd(const dvla_attn_params* p, void* stream);
/* The same attention for SHORT sequences (Lq == Lk <= 64) with ANY head_dim <= 128 (multiple of 8), no mask, no dropout: one
 * wave per (batch, head).  Further limits (DVLA_ERR_UNSUPPORTED from both entry points): B <= 65535, and the backward kernel's
 * LDS, (4 L (head_dim + 1) + 2 L (L + 1)) * 4 bytes, must fit 160 KiB -- L = 64 with head_dim 128 does not (L <= 63 does).  The only head_dim != 64 the reference can produce is the DiT-S action head (models/action_model/
 * action_model.py:12-14: 384 / 4 = 96) on 6-token sequences.  Same parameter block; `delta` is not used. */
int dvla_attn_small_fwd(const dvla_attn_params* p, int32_t head_dim, void* stream);
int dvla_attn_small_bwd(const dvla_attn_params* p, int32_t head_dim, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Small HBM-bound helpers (bf16 unless noted).
 */
/* out[n] = sum_m x[m,n]  (bias gradients).  out fp32; `partial` = fp32 workspace of
 * dvla_colsum_partial_rows() * cols floats. */
int dvla_colsum(const void* x, int64_t ld, int64_t rows, int64_t cols, float* out, float* partial, void* stream);
/* same with `out` written in out_dtype (0 = bf16: the bias gradient in the parameter's dtype, no cast kernel; 1 = fp32) */
int dvla_colsum_dt(const void* x, int64_t ld, int64_t rows, int64_t cols, void* out, int32_t out_dtype, float* partial,
                   void* stream);
int64_t dvla_colsum_partial_rows(void);
/* y = dropout(x) * 1/(1-p) with the stateless hash RNG; rows x cols, row stride = cols.
 * GPT2Model.drop (models/gpt2.py:459) and its backward (same call on the gradient). */
int dvla_dropout(const void* x, void* y, int64_t rows, int64_t cols, float p, uint32_t seed_lo, uint32_t seed_hi,
                 void* stream);
/* dz = dy * act'(preact) (optionally after dropout mask of dy): backward of a fused activation. */
int dvla_act_bwd(const void* dy, const void* preact, void* dz, int64_t rows, int64_t cols, int32_t act,
                 float dropout_p, uint32_t seed_lo, uint32_t seed_hi, void* stream);
/* The same dz, plus colsum[c] = sum_r dz[r, c] (fp32 sums of the stored bf16 values, written as colsum_dtype) from the same pass:
 * the bias gradient of `y = dropout(act(x W + b))` (models/gpt2.py:160,172-173,329-339: c_proj and the MLP's second Conv1D),
 * which otherwise costs either a column-sum pass over dz or the summing code inside the weight-gradient GEMM.  partial:
 * dvla_colsum_partial_rows() x cols floats.  cols % 8 == 0 and 16-byte aligned operands, else DVLA_ERR_UNSUPPORTED. */
int dvla_act_bwd_colsum(const void* dy, const void* preact, void* dz, int64_t rows, int64_t cols, int32_t act, float dropout_p,
                        uint32_t seed_lo, uint32_t seed_hi, void* colsum, int32_t colsum_dtype, float* partial, void* stream);
/* y = act(x) */
int dvla_act_fwd(const void* x, void* y, int64_t n, int32_t act, void* stream);
/* One step of the action sampler's algebra, evaluation path (models/dreamvla_model.py:935-987): classifier-free guidance
 * (action_model/models.py:253-268: eps = uncond + s (cond - uncond), evaluated in the model dtype like the tensor expression
 * it replaces: three bf16 roundings) followed by the eta = 0 DDIM update (gaussian_diffusion.py:522-569):
 *     pred_x0 = a x - b eps;   eps' = (a x - pred_x0) / b;   x_next = pred_x0 sqrt_acp_prev + sqrt_1m_acp_prev eps'
 * in fp32, operation by operation (no contraction) -- ~20 elementwise ATen launches on (bs, 3, 7) tensors become one.
 * model_out: bf16, sample i of the guided half at model_out + i * sample_stride, of the unguided half at
 * model_out + (bs + i) * sample_stride, per_sample contiguous values each; x, x_next: fp32 (bs, per_sample) contiguous. */
int dvla_ddim_cfg_step(const void* model_out, int64_t sample_stride, const float* x, float* x_next, int64_t bs, int64_t per_sample,
                       float cfg_scale, float a, float b, float sqrt_acp_prev, float sqrt_1m_acp_prev, void* stream);
/* The WHOLE evaluation sampler of the DiT action head in one launch (models/dreamvla_model.py:935-987: ddim_sample_loop over
 * net.forward_with_cfg, eta = 0; models/action_model/models.py:162-268 DiT.forward / forward_with_cfg; gaussian_diffusion.py:
 * 522-569 ddim_sample): `steps` x { token embedding, `depth` DiT blocks, final layer, guidance + DDIM update } on 2 bs sequences
 * of 2 tokens tokens (the guided and the unguided half of every sample), as a persistent kernel on the 32 CUs of one XCD whose
 * workgroups exchange activations through that XCD's L2 (csrc/dit_team.hip; tests/probes/sync_probe.cpp has the measurements
 * behind the design).  Same arithmetic and rounding points as the launch-by-launch path (ActionModel.sample_ddim_cfg: few-rows
 * GEMMs with LayerNorm on the fly, flash attention, dvla_ddim_cfg_step), other fp32 summation order inside the products.
 *   blocks     device array of `depth` entries: bf16 weights in nn.Linear layout (out, in), bf16 biases
 *   xemb_*     x_embedder Linear(channels -> hidden); final_*: final_layer.linear (hidden -> channels); pos (2 tokens, hidden)
 *   cond       (steps, 2 bs, tokens, hidden) bf16: z_embedder([cond ; uncondition]) + t_embedder(timestep of step j), j = 0 the
 *              FIRST sampler step (the largest timestep)
 *   coef       (steps, 4) fp32 on the device: a, b, sqrt_acp_prev, sqrt_1m_acp_prev of dvla_ddim_cfg_step per sampler step
 *   noise/out  (bs, tokens, channels) fp32: start noise / samples (NaN if a wait inside the kernel timed out)
 *   workspace  dvla_dit_sample_workspace_bytes(hidden) bytes, 16-byte aligned, ZERO-INITIALISED by the caller once (the kernel
 *              leaves its counters at zero); 32-bit word 33 = number of launches so far in which a wait timed out (their outputs
 *              are NaN; the launch retires its own status word 32, so a timeout does not carry over to the next launch), word 34 =
 *              the last non-zero status, words 64 .. 95 = the XCC id each of the 32 team members ran on in the last launch (all
 *              equal = the fast case), valid after a launch
 * DVLA_ERR_UNSUPPORTED unless hidden = 768 (DiT-B), head_dim 64, 2 tokens <= 8, 4 bs tokens <= 16 rows (one episode),
 * channels <= 16, on a device with 256 CUs: the caller then runs the launch-by-launch sampler (which is the faster one for two
 * row blocks and for hidden 1024: profiles/r04_dit_team_perf.jsonl). */
typedef struct dvla_dit_block_weights {
  const void *qkv_w, *qkv_b, *proj_w, *proj_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} dvla_dit_block_weights;
typedef struct dvla_dit_sample_params {
  const dvla_dit_block_weights* blocks;
  const void *xemb_w, *xemb_b, *final_w, *final_b, *pos, *cond;
  const float* coef;
  const float* noise;
  float* out;
  void* workspace;
  int64_t workspace_bytes;
  float cfg_scale, ln_eps;
  int32_t depth, hidden, heads, channels, tokens, bs, steps, reserved;
} dvla_dit_sample_params;
int64_t dvla_dit_sample_workspace_bytes(int32_t hidden);
int dvla_dit_sample(const dvla_dit_sample_params* p, void* stream);
/* measurement: a device buffer of 2 x 8 x (steps x (1 + 5 depth) + 1) uint64 that later launches fill with wall-clock stamps
 * (100 MHz) of team members 0 and 17 -- per exchange: weights requested, producers arrived, operands landed, partial tiles in
 * LDS, results stored; NULL (the default) switches it off */
void dvla_dit_sample_set_stamps(void* device_buffer);
/* test hook: the next n dvla_dit_sample launches behave as if a wait inside the kernel had timed out (NaN output, workspace word 33
 * incremented); a launch captured into a hipGraph while the hook is armed times out on every replay */
void dvla_dit_sample_inject_timeouts(int32_t n);
/* dst(bf16) = src(fp32) / dst(fp32) = src(bf16) */
int dvla_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int dvla_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
/* out = a + b (bf16, n elements); b_period > 0 broadcasts b with period b_period elements */
int dvla_add(const void* a, const void* b, void* out, int64_t n, int64_t b_period, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Optimizer step over flat bf16 buffers (the caller's `clip_grad_norm_` + `torch.optim.AdamW.step`, reference
 * train.py:253-262 / utils/train_utils.py:600-608).
 * dvla_sumsq_bf16: out[0] (+)= sum x[i]^2 in fp32; `partial` holds dvla_sumsq_partial_len() floats.
 * dvla_adamw_bf16: one AdamW step on n elements; math in fp32, parameters and both moments stored in bf16 (torch's
 *   fused AdamW with bf16 parameters).  If grad_sumsq != NULL the gradient is first scaled by
 *   min(1, max_norm / (sqrt(*grad_sumsq) + 1e-6)) and rounded to bf16 (clip_grad_norm_ semantics; the scalar stays on
 *   the device, no host synchronisation).  `step` is the 1-based step count (bias corrections). */
int64_t dvla_sumsq_partial_len(void);
int dvla_sumsq_bf16(const void* x, int64_t n, float* partial, float* out, int32_t accumulate, void* stream);
int dvla_adamw_bf16(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int64_t step, const float* grad_sumsq, float max_norm,
                    void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Token assembly (models/dreamvla_model.py:739-759): out[b, s, t, :] = part_k[b, s, t - tok_begin_k, :] + pos[s, :]
 * for the part k that owns token t -- the per-frame conditioning tokens (text, state, resampled image tokens, cls tokens)
 * and the learned prediction-query tokens (stride_b = stride_s = 0: broadcast parameters) -- replacing torch.cat + the
 * window-position add.  Parts cover [0, T) in order; element strides; every base 16-byte aligned; H % 8 == 0.
 * pos may be NULL (no add) and is read at pos + s * pos_stride_s.  bf16.
 */
#define DVLA_MAX_TOKEN_SRCS 16
typedef struct dvla_token_src { const void* base; int64_t stride_b; int64_t stride_s; int32_t tok_begin; int32_t tok_count; } dvla_token_src;
int dvla_assemble_tokens(const dvla_token_src* srcs, int32_t n_src, const void* pos, int64_t pos_stride_s, void* out,
                         int32_t B, int32_t S, int32_t T, int32_t H, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Training-loss reductions (the caller-side loss block of the reference, utils/train_utils.py:172-450 + utils/sigloss.py:
 * patchify -> per-patch normalise -> (flow mask) -> MSE; 1 - cosine_similarity; SiLog on un-patchified depth).  bf16 data,
 * fp32 arithmetic, deterministic two-stage sums.  Tensors are addressed per FRAME through a view so the caller's slices
 * (prediction[:, view, 0], label[:, future : future + T]) are read in place: frame f lives at
 *   base + (f / T) * stride_b + (f % T) * stride_t          (element strides; the inner block of a frame is contiguous).
 * fwd: out2[0] = loss (fp32 scalar), out2[1] = auxiliary (SiLog: mean d, needed by bwd); `partial` = fp32 workspace of
 *      dvla_loss_partial_len() floats.   bwd: dpred (bf16, same addressing as pred) = grad_out[0] * dLoss/dpred.
 */
typedef struct dvla_frame_view { const void* base; int64_t stride_b; int64_t stride_t; int32_t T; } dvla_frame_view;
int64_t dvla_loss_partial_len(void);
/* pred frames (196, 768); image frames (3, 224, 224); patch_mask NULL or fp32 (n_frames, 196) of {0,1} */
int dvla_patch_mse_fwd(const dvla_frame_view* pred, const dvla_frame_view* image, const float* patch_mask, int64_t n_frames,
                       float* out2, float* partial, void* stream);
int dvla_patch_mse_bwd(const dvla_frame_view* pred, const dvla_frame_view* image, const float* patch_mask, int64_t n_frames,
                       const float* grad_out, const dvla_frame_view* dpred, void* stream);
/* pred / label frames (rows_per_frame, cols); cols % 64 == 0, cols <= 1024 */
int dvla_cosine_loss_fwd(const dvla_frame_view* pred, const dvla_frame_view* label, int32_t rows_per_frame, int32_t cols,
                         int64_t n_frames, float* out2, float* partial, void* stream);
int dvla_cosine_loss_bwd(const dvla_frame_view* pred, const dvla_frame_view* label, int32_t rows_per_frame, int32_t cols,
                         int64_t n_frames, const float* grad_out, const dvla_frame_view* dpred, void* stream);
/* pred frames (196, 256) = 16x16 depth patches; depth frames (1, 224, 224) */
int dvla_silog_loss_fwd(const dvla_frame_view* pred, const dvla_frame_view* depth, int64_t n_frames, float lambd, float* out2,
                        float* partial, void* stream);
int dvla_silog_loss_bwd(const dvla_frame_view* pred, const dvla_frame_view* depth, int64_t n_frames, float lambd, const float* out2,
                        const float* grad_out, const dvla_frame_view* dpred, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Attention-mask tables from the block-mask RULE, on the device.
 * Replaces the per-step host regeneration + upload of the (L, L) additive mask in the pretrain phase
 * (models/dreamvla_model.py:610-628 calling generate_attention_mask, 25-66) and the host-side derivation of the
 * kernels' tables from it.  `rule` are generate_attention_mask's own arguments; `drop` (device, [K][n_drop] int32) are
 * the obs-token indices that `mask_l_obs_ratio` drew for each window step with numpy's RNG on the host (n_drop =
 * int(ratio * num_obs); the draw stays on the host so the stream is consumed exactly as the reference consumes it).
 * Outputs (device, caller-allocated): key_index (Lk int32), bits_q (L x ceil(Lk/32) uint32), bits_k (Lk x ceil(L/32)),
 * tile_map (ceil(L/32) x ceil(Lk/32) uint8), where L = K (num_A + num_B) and Lk = K (num_A + kept obs columns) --
 * exactly the tables dvla_attn_fwd / dvla_attn_bwd take.  No host synchronisation, capturable. */
typedef struct dvla_mask_rule {
  int32_t K, num_A, num_B, num_obs, action_pred_steps;
  int32_t atten_goal, atten_goal_state, atten_only_obs, attn_robot_proprio_state;
  int32_t n_drop;
} dvla_mask_rule;
int dvla_mask_tables(const dvla_mask_rule* rule, const int32_t* drop, int32_t* key_index, uint32_t* bits_q, uint32_t* bits_k,
                     uint8_t* tile_map, void* stream);

/* ---------------------------------------------------------------------------------------------------
 * Camera-frame input pipeline on the device: uint8 HWC frames (already resized to the model's
 * resolution on the host) -> ToTensor (/255) -> CLIP Normalize (mean3 / std3, fp32, torch's operation order) ->
 * RandomShiftsAug as the integer-shift gather it is -> bf16 CHW.  Replaces, per frame, clip's `_transform` tail
 * (ToTensor + Normalize: utils/data_utils.py:175-178 via image_processor), RandomShiftsAug.forward / forward_traj
 * (utils/data_utils.py:326-383, applied by the collaters at 1337-1353) and the fp32 upload + cast of
 * utils/train_utils.py:118-123.  src: (n, H, W, 3) uint8; shift: (n, 2) int32 (sx, sy) in [0, 2 pad] or NULL (no
 * augmentation); out: (n, 3, H, W) bf16, 16-byte aligned; W % 8 == 0. */
int dvla_image_preprocess(const uint8_t* src, const int32_t* shift, void* out, int64_t n, int32_t height, int32_t width,
                          int32_t pad, const float* mean3, const float* std3, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DVLA_H_ */
