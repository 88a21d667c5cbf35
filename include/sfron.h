/* sfron.h -- C ABI of libsfron.so: the MI355X (gfx950) SFR-on unlearning hot path.
 *
 * The reference (K1nght/Unified-Unlearning-w-Remain-Geometry) is 100 % Python and has no FFI:
 * its "operator interface" for this path is the PyTorch op sequences inside
 *   DiT/forget.py:256-322, DiT/diffusion/gaussian_diffusion.py:715-787, DiT/models.py:233-248.
 * Each entry point below replaces one of those sequences (cited per function) and is what a
 * ctypes / cffi binding on the reference side would call (INTEGRATION.md shows the stubs).
 *
 * Conventions: plain pointers and sizes only (no torch types); every pointer is a DEVICE pointer
 * unless stated; `stream` is a hipStream_t passed as void* (NULL = default stream); all launches
 * are asynchronous on `stream`; return 0 on success, a hipError_t value or SFRON_ERR_* (>= 1001)
 * on failure; nothing throws.  bf16 tensors are passed as uint16_t*.
 *
 * State: every tensor, workspace, stream and event belongs to the caller or to a handle the caller creates (sfron_aux_create,
 * sfron_probe_create); entry points are re-entrant per stream.  Process-wide state is limited to (a) three SCHEDULE switches --
 * sfron_gemm_loader_waves, sfron_attn_fwd_form, sfron_attn_bwd_form: each picks between schedules that give the same bits (the tests
 * compare them), so no result depends on them (sfron_attn_bwd_form: the same values to fp32 rounding); not thread-safe, meant for tests
 * and A-B timing -- and (b) since ABI 15 a per-device FREE LIST of weight-gradient stream pairs: sfron_aux_destroy returns a handle's two HIP
 * streams to it (drained) and the next sfron_aux_create on that device takes them instead of creating new ones, so that a process that builds
 * one engine after another keeps running on the same streams (the runtime maps streams onto four hardware queues; a later handle with fresh
 * streams can land on a queue another stream of the step uses: measured 4 ms per step, profiles/r06_fp8_leg.txt).  Two live handles never
 * share a pair.  The fp8 activation-range counters (rounds 4-5: a __device__ global) are a caller-owned device array since ABI 15
 * (sfron_fp8_activation_amax).  One host thread per device drives the library.
 */
#ifndef SFRON_H
#define SFRON_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ version / probing */
int sfron_abi_version(void);            /* bumps when a signature changes */
const char* sfron_build_arch(void);     /* "gfx950" */

/* ------------------------------------------------------------------ parameter sweep (sweep.hip)
 * Flat fp32 arenas: params, grads, exp_avg, exp_avg_sq, ema share offsets; mask is 1 byte/element
 * (torch.bool storage, 0/1); w_bf16 is the bf16 shadow the GEMMs read. */

/* length (in doubles) the `partials` scratch of sfron_sumsq_masked must have */
int sfron_sweep_partials_len(void);

/* partial sums of (mask ? g : 0)^2; first half of clip_grad_norm_ after `grad *= mask`
 * (DiT/forget.py:289-298).  mask may be NULL.  *nblk_out (HOST int) = number of partials written.
 * g2 (may be NULL): a second gradient arena added element-wise first (g + g2): the two micro-batch chains of a
 * step write disjoint arenas so they never have to order their weight-gradient GEMMs against each other. */
int sfron_sumsq_masked(const float* g, const float* g2, const uint8_t* mask, int64_t n, double* partials, int* nblk_out,
                       void* stream);

/* the same partial sums over a table of element ranges: ranges (DEVICE) int64 [n_ranges][2] = {offset, length} into g / mask (multiples of 4;
 * a range is summed by ONE workgroup: split tensors above ~64 K elements); partials[r] = sum over range r of (mask ? g : 0)^2.  What the clip
 * norm still reads when the weight-gradient GEMMs leave their own sums (sfron_gemm_desc.sumsq_partials, sfron_aux_arm_sumsq). */
int sfron_sumsq_masked_ranges(const float* g, const uint8_t* mask, const int64_t* ranges, int n_ranges, double* partials, void* stream);

/* stats[0] = ||g||_2, stats[1] = min(1, max_norm / (norm + 1e-6)), stats[2] = sum of squares
 * (torch.nn.utils.clip_grad_norm_ semantics; DiT/forget.py:293-298) */
int sfron_clip_coef(const double* partials, int nblk, float max_norm, float* stats, void* stream);

/* fused: g' = (mask ? g : 0) * stats[1]; Adam/AdamW single-tensor update of (p, m, v) with
 * host-computed (double, rounded to fp32 exactly where torch rounds) step_size = lr / (1 - beta1^step),
 * bc2_sqrt = sqrt(1 - beta2^step),
 * decay_mul = 1 - lr * weight_decay (1.0 for the reference's wd = 0); optional bf16 shadow write;
 * optional fused EMA of the NEW p: ema_mode 0 none, 1 DiT (ema*d + (1-d)*p, DiT/forget.py:52-62),
 * 2 DDPM ((1-mu)*p + mu*shadow, DDPM/models/ema.py:17-24).  mask / stats / w_bf16 / ema may be NULL.
 * (DiT/forget.py:289-299,320; DDPM/runners/diffusion.py:1126-1138,1169-1180) */
int sfron_masked_clip_adam(float* p, const float* g, const float* g2 /* NULL or second arena, see above */, float* m, float* v,
                           const uint8_t* mask, const float* stats,
                           int64_t n, double beta1, double beta2, double eps, double step_size, double bc2_sqrt,
                           double decay_mul, uint16_t* w_bf16, float* ema, double ema_decay, int ema_mode, void* stream);

/* the same sweep on at most max_workgroups workgroups of 256 threads (0 = as many as the range offers, capped at 2048): a
 * sweep issued on a second stream BESIDE a GEMM chain takes a bounded share of the chip's wave slots and HBM queue */
int sfron_masked_clip_adam_wg(float* p, const float* g, const float* g2, float* m, float* v, const uint8_t* mask, const float* stats,
                              int64_t n, double beta1, double beta2, double eps, double step_size, double bc2_sqrt,
                              double decay_mul, uint16_t* w_bf16, float* ema, double ema_decay, int ema_mode, int max_workgroups,
                              void* stream);

/* config 5: the sweep over ONE weight tensor (+ its bias) that also writes the e4m3 shadow of the new values, w_e4m3[i] =
 * e4m3(p_new[i] * *w_e4m3_scale) (device scalar from sfron_fp8_update_scales): the optimizer step re-quantises what it touches, no
 * second pass over the masters.  n % 4 == 0; max_workgroups as sfron_masked_clip_adam_wg. */
int sfron_masked_clip_adam_q(float* p, const float* g, float* m, float* v, const uint8_t* mask, const float* stats, int64_t n, double beta1,
                             double beta2, double eps, double step_size, double bc2_sqrt, double decay_mul, uint16_t* w_bf16, float* ema,
                             double ema_decay, int ema_mode, uint8_t* w_e4m3, const float* w_e4m3_scale, int max_workgroups, void* stream);

/* The same two sweeps over a weight matrix W [NM][D] (p / m / v / mask / w_bf16 / ema point at ITS first element) whose gradient is
 * a rank-R product dW[n][k] = sum_{b < R} dmod[b][n] * sc[b][k] (bf16 factors, fp32 accumulation in index order), formed inside
 * the sweep instead of by a weight-gradient GEMM that writes NM*D floats for the sweep to read back: the adaLN_modulation Linears
 * of all DiT blocks (DiT/models.py:113-116,131-134: input silu(c) [batch][D], a third of DiT-XL/2's parameters; R = batch).
 * NM % 8 == 0, D % 4 == 0.  sumsq: *nblk_out partials (NM / 8), to be combined with the other ranges' by sfron_clip_coef. */
int sfron_sumsq_lowrank(const uint16_t* dmod, const uint16_t* sc, int R, int NM, int D, const uint8_t* mask, double* partials,
                        int* nblk_out, void* stream);
int sfron_adam_lowrank(float* p, float* m, float* v, const uint8_t* mask, const float* stats, const uint16_t* dmod, const uint16_t* sc, int R,
                       int NM, int D, double beta1, double beta2, double eps, double step_size, double bc2_sqrt, double decay_mul,
                       uint16_t* w_bf16, float* ema, double ema_decay, int ema_mode, void* stream);

/* stand-alone EMA (frozen parameters such as pos_embed; DiT/forget.py:60-62) */
int sfron_ema_update(float* ema, const float* p, int64_t n, double decay, int ema_mode, void* stream);

/* F += g^2 / n_iters (DiT/generate_fisher.py:236-239, on device instead of .cpu()) */
int sfron_fisher_accum(float* fisher, const float* g, int64_t n, float n_iters, void* stream);

/* F += ((g [+ g2]) * stats[1])^2 / n_iters: the DDPM variant squares the gradient AFTER clip_grad_norm_
 * (DDPM/runners/diffusion.py:1271-1281); stats from sfron_clip_coef (NULL = no clipping), g2 NULL or a second arena added first */
int sfron_fisher_accum_clipped(float* fisher, const float* g, const float* g2, const float* stats, int64_t n, float n_iters, void* stream);

/* mask = ((F_f + 1e-15) / (F_r + 1e-15)) >= th, IEEE fp32, bit-exact with torch
 * (DiT/generate_mask.py:34-35, DDPM/generate_fisher_mask.py:39-46) */
int sfron_mask_from_fisher(const float* forget_fisher, const float* remain_fisher, int64_t n, float th,
                           uint8_t* mask, void* stream);

/* fp32 -> bf16 (RNE) */
int sfron_cast_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);

/* ------------------------------------------------------------------ diffusion loss (loss.hip)
 * tab: [T][8] fp32 rows = { sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod,
 *   sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_mean_coef1,
 *   posterior_mean_coef2, posterior_log_variance_clipped, log(betas) }
 * (DiT/diffusion/gaussian_diffusion.py:167-201, gathered as fp32 like _extract_into_tensor :861-873) */
#define SFRON_TAB_COLS 8

/* x_t = sqrt_ac[t] * x0 + sqrt_1m_ac[t] * noise   (gaussian_diffusion.py:215-230); t is int64 [n] */
int sfron_q_sample(const float* x0, const float* noise, const int64_t* t, const float* tab, int n, int chw, float* x_t,
                   void* stream);

/* model_out [n, 2c, hw] -> mse[n], vb[n] (loss = mse + vb) and
 * d_model_out = d( grad_scale * sum_i loss_i ) / d model_out      (gaussian_diffusion.py:746-783, 682-713)
 * grad_scale = (+/- forget_alpha or 1) / global_batch reproduces `(alpha * -loss.mean()).backward()`
 * (DiT/forget.py:272,286-288,311). */
int sfron_dit_loss_fwd_bwd(const float* x0, const float* noise, const float* model_out, const int64_t* t,
                           const float* tab, int n, int c, int hw, float grad_scale, float* mse, float* vb,
                           float* d_model_out, void* stream);

/* One ancestral sampling step, gaussian_diffusion.py:376-421 over :254-332 (LEARNED_RANGE variance, EPSILON mean):
 * sample = posterior_mean(pred_x0, x) + (t != 0) * exp(0.5 * log_var) * noise; pred_xstart may be NULL.  t indexes `tab`
 * (the respaced schedule); model_out [n][2c][hw] was evaluated by the caller at the mapped original timestep. */
int sfron_p_sample(const float* x, const float* model_out, const int64_t* t, const float* tab, const float* noise, int n, int c,
                   int hw, int clip_denoised, float* sample, float* pred_xstart, void* stream);
/* classifier-free guidance mix of DiT.forward_with_cfg (DiT/models.py:258-266), in place on model_out [n_total][cout][hw]:
 * for the first n_guided channels, rows i and i + n_total/2 both become uncond + cfg_scale * (cond - uncond). */
int sfron_cfg_combine(float* model_out, int n_total, int cout, int hw, int n_guided, float cfg_scale, void* stream);

/* ---- DDPM (CIFAR-10) epsilon loss: DDPM/functions/losses.py:22-38 (sum over C,H,W, mean over the batch), :49-69 (adaptive
 * "adaga" weights), :32 (alphas_cumprod recomputed from fp32 betas). */
/* abar[t] = prod_{s<=t} (1 - betas[s]), running product in double, rounded to fp32 per entry (= torch CPU cumprod) */
int sfron_ddpm_alphas_cumprod(const float* betas, int T, float* abar, void* stream);
/* x_t = x0 * sqrt(abar[t]) + e * sqrt(1 - abar[t])   (correctly rounded sqrt; losses.py:33-34) */
int sfron_ddpm_q_sample(const float* x0, const float* e, const int64_t* t, const float* abar, int n, int chw, float* x_t,
                        void* stream);
/* per_sample[i] = sum_{chw} (e - model_out)^2         (losses.py:36-38 with keepdim=True) */
int sfron_ddpm_sample_loss(const float* e, const float* model_out, int n, int chw, float* per_sample, void* stream);
/* mode 0 "simple": loss = sum_local(per)/n_global, coef_i = 2*scale/n_global.
 * mode 1 "adaga":  w_i = 1/(per_i^lambd + 1e-8); W = sum_i w_i, written to *wsum (use_wsum = 0) or taken from *wsum
 *                  (use_wsum = 1: the caller all-reduced it over the data-parallel ranks); loss = sum_local(w_i per_i)/W,
 *                  coef_i = 2*scale*w_i/W.     d(scale*loss)/d model_out_i = coef_i * (model_out_i - e_i) */
int sfron_ddpm_loss_coef(const float* per_sample, int n, int mode, float lambd, float scale, int n_global, float* wsum,
                         int use_wsum, float* coef, float* loss, void* stream);
int sfron_ddpm_loss_bwd(const float* e, const float* model_out, const float* coef, int n, int chw, float* d_model_out,
                        void* stream);
/* one step of DDPM/functions/denoising.py:72-95 (generalized / DDIM update, same timestep for the whole batch):
 * x0_pred = (x - eps*s1)/s2, x_next = s3*x0_pred + c1*noise + c2*eps; noise may be NULL when c1 == 0, x0_pred may be NULL */
int sfron_ddim_step(const float* x, const float* eps, const float* noise, int64_t n, float s1, float s2, float s3, float c1,
                    float c2, float* x_next, float* x0_pred, void* stream);

/* ------------------------------------------------------------------ bf16 MFMA GEMM (gemm.hip)
 * C[M,N] = alpha * op(A)[M,K] · op(B)[K,N] (+ bias[N]) with a fused epilogue; fp32 accumulation.
 *   a_transposed = 0: A is [M][K] row-major (lda), contraction contiguous      (activations, forward / dgrad)
 *   a_transposed = 1: A is [K][M] row-major (lda), i.e. the operand is A^T of a row-major tensor (wgrad)
 *   b_transposed = 0: B is [N][K] row-major (ldb): nn.Linear weight layout, Y = X W^T
 *   b_transposed = 1: B is [K][N] row-major (ldb): dgrad (dX = dY W) and wgrad (dW = dY^T X)
 * Replaces the GEMMs behind nn.Linear / Conv2d(k=s=p) fwd+bwd of DiT/models.py:108-121,138-142,169.
 * Requirements: K % 8 == 0, N % 4 == 0, lda/ldb % 8 == 0, 16-byte aligned A/B; transposed operands need
 * their non-contraction extent % 8 == 0.  Rows/cols beyond M/N and k >= K are handled (zero-filled / masked). */
enum {
  SFRON_EPI_BF16 = 0,     /* c_bf16 = result                                                               */
  SFRON_EPI_F32 = 1,      /* c_f32 = result (+= if accumulate)                  -- wgrad into the grad arena */
  SFRON_EPI_GELU = 2,     /* aux = bf16(result) (pre-activation), c_bf16 = gelu_tanh(result)   -- Mlp.fc1+act */
  SFRON_EPI_GATE_RES = 3, /* aux = bf16(result); c_f32[row,col] = resid[row,col] + gate[row/tokens, col] * result
                             -- x = x + gate * branch(x)  (DiT/models.py:120-121)                          */
  SFRON_EPI_DGELU = 4,    /* c_bf16 = result * gelu_tanh'(aux)                           -- fc2 dgrad + act' */
  SFRON_EPI_POS = 5,      /* c_f32 = result + pos[row % tokens, col]    -- x_embedder(x) + pos_embed (:240) */
  /* Round 6: the GELU pair with `aux` = gelu_tanh'(result) as ONE BYTE per element instead of the bf16 pre-activation: `aux` then points to a
   * uint8 [M][ldaux] array, code = round((g' + 0.15) * 196), g' = code / 196 - 0.15 (GELU'_tanh lies in [-0.129, 1.129]; step 0.0051, i.e.
   * <= 0.0026 absolute -- the size of what the bf16 rounding of the pre-activation does to GELU').  fc1 writes 113 instead of 151 MB per
   * block at DiT-XL/2, the fc2 dgrad reads half as much and evaluates no exp.  Only where sfron_gemm_gelu_q_supported(M, N, K) (the 256 x 192
   * pipelined tile: M % 256 == 0, N % 192 == 0, K % 128 == 0, ldc_bf16 % 8 == 0, ldaux % 8 == 0), otherwise SFRON_ERR_UNSUPPORTED; a forward
   * pass that wrote codes must be followed by the _Q dgrad (the two arrays are not interchangeable). */
  SFRON_EPI_GELU_Q = 7,   /* aux(u8) = code(gelu_tanh'(result)), c_bf16 = gelu_tanh(result)          -- Mlp.fc1+act */
  SFRON_EPI_DGELU_Q = 8   /* c_bf16 = result * decode(aux(u8))                                 -- fc2 dgrad + act' */
};
typedef struct sfron_gemm_desc {
  const uint16_t* A; const uint16_t* B;
  int M, N, K, lda, ldb;
  int a_transposed, b_transposed;
  int epilogue;
  float alpha;
  const float* bias;                 /* [N] fp32 or NULL */
  uint16_t* c_bf16; int ldc_bf16;
  float* c_f32; int ldc_f32;
  uint16_t* aux; int ldaux;
  const float* gate; int ldgate;
  const float* pos;
  int tokens;                        /* tokens per sample (row -> sample / position) */
  int accumulate;
  const float* resid;                /* EPI_GATE_RES: c_f32 = resid + gate * result; NULL = in place (resid = c_f32) */
  int split_k;                       /* > 1 (EPI_F32 only, no bias): split s writes its partial product to
                                        c_f32 + s * split_stride; sum the slabs with sfron_reduce_chunks */
  long split_stride;
  int tile_hint;                     /* 0 = auto; -1 = generic kernel; fast tiles 1 = 128x128, 2 = 256x192, 3 = 256x256, 4 = 384x192 */
  float* a_rowsum;                   /* weight-gradient layout only (a_transposed = b_transposed = 1, EPI_F32, no split): fp32 [M],
                                        a_rowsum[m] = sum_k op(A)[m][k] -- the bias gradient sum_rows dY of the same nn.Linear
                                        (DiT/models.py:108-121 backward), taken from the dY tiles the GEMM already holds in LDS
                                        (extra MFMAs against a ones fragment, shared by the workgroups of a tile row).  Only shapes
                                        sfron_gemm_rowsum_supported accepts. */
  float* rowsum_ws;                  /* with a_rowsum: fp32 scratch [(N / 192) * M] for the per-tile-column partial sums, added in a
                                        fixed order by a small reduction launch on the same stream */
  float* col_partials;               /* SFRON_EPI_DGELU only: fp32 [sfron_gemm_dgelu_colpart_rows(M, N, K)][N]; row r = the column
                                        sums of output rows 256 r .. 256 r + 255 (fp32, before the bf16 rounding): partials of the fc1
                                        bias gradient sum_rows d_hpre, formed where d_hpre is produced instead of by a second pass
                                        over it (sum them with sfron_reduce_chunks).  NULL = not wanted. */
  const uint8_t* sumsq_mask;         /* with sumsq_partials: byte mask over the OUTPUT (same layout as c_f32: element (m, n) at m * ldc_f32 + n) or NULL */
  double* sumsq_partials;            /* weight-gradient layout only (a_transposed = b_transposed = 1, EPI_F32, no split, no a_rowsum): fp64
                                        [sfron_gemm_sumsq_partials(M, N, K)], partial t = sum over output tile t of (mask ? c : 0)^2 -- the masked
                                        sum of squares clip_grad_norm_ needs (DiT/forget.py:289-298), formed where the gradient is produced
                                        instead of by a pass over the gradient arena; combine with sfron_clip_coef.  NULL = not wanted. */
} sfron_gemm_desc;
int sfron_gemm_bf16(const sfron_gemm_desc* desc /* HOST pointer */, void* stream);
/* number of partial rows an EPI_DGELU product of this shape writes to col_partials (M / 256), 0 = shape unsupported: use sfron_colsum */
int sfron_gemm_dgelu_colpart_rows(int M, int N, int K);
/* 1 when a weight-gradient GEMM dW[M][N] = dY[K][M]^T X[K][N] of this shape can also produce a_rowsum (else use sfron_colsum) */
int sfron_gemm_gelu_q_supported(int M, int N, int K);     /* 1 = SFRON_EPI_GELU_Q / _DGELU_Q take this (rows, hidden, contraction) shape */
int sfron_gemm_rowsum_supported(int M, int N, int K);
/* Finish of a split-K product (sfron_gemm_desc.split_k > 1 leaves n_splits fp32 slabs, split_stride elements apart): the slabs summed in index
 * order (bitwise reproducible) and written (a) as bf16 [n] -- an input gradient that the next product reads as its operand (autograd of the
 * nn.Linear layers, DiT/models.py:108-121) -- or (b) through the gated-residual epilogue of the forward proj / fc2 products (models.py:120-121):
 * v = sum + bias; branch (bf16 [M][N]) = v; out (fp32 [M][N]) = resid + gate[(row / tokens) * ldgate + col] * v.  n_splits <= 8.  What the
 * few-tile products of a small batch * tokens (BASELINE config 2: DiT-B/4, 2048 token rows) use to fill the chip (csrc/dit_engine.hip). */
int sfron_split_sum_bf16(const float* slabs, int n_splits, int64_t n, int64_t split_stride, uint16_t* out, void* stream);
int sfron_split_gate_res(const float* slabs, int n_splits, int64_t split_stride, const float* bias, const float* gate, int ldgate, int tokens,
                         const float* resid, float* out, uint16_t* branch, int M, int N, void* stream);

/* number of fp64 partials a weight-gradient GEMM dW[M][N] = dY[K][M]^T X[K][N] of this shape writes to sumsq_partials, 0 = shape unsupported */
int sfron_gemm_sumsq_partials(int M, int N, int K);


/* ------------------------------------------------------------------ fp8 (e4m3) forward GEMMs (fp8.hip) -- BASELINE config 5
 * "DiT-XL/2 fp8 weights + bf16 activations on the CDNA4 fp8 MFMA": the four token GEMMs of a DiT block forward
 * (DiT/models.py:108-121) on v_mfma_scale_f32_16x16x128_f8f6f4.  The reference has no fp8 path; tolerance is stated against the bf16
 * path (config 3) in tests/test_gpu_fp8.py.  e4m3 = OCP e4m3fn, round to nearest even, saturating at +-448.
 * Weights: fp32 masters + e4m3 shadow arena (same offsets, 1 byte / element) + ONE power-of-two scale per tensor;
 * activations: quantised by their producers with static power-of-two scales. */

/* table [n_tensors][2] int64 (DEVICE): element offset and length (multiples of 8) of each quantised tensor inside the arena.
 * mode 0: amax_bits[t] = max(amax_bits[t], bits(max |p|)) only.  mode 1: dst[off + i] = e4m3(p[off + i] * scales[t]) and the same amax
 * (delayed scaling: quantise with the scale derived from the PREVIOUS pass's amax while collecting the next one). */
int sfron_fp8_quant_tensors(const float* params, const int64_t* table, int n_tensors, const float* scales, uint32_t* amax_bits,
                            uint8_t* dst, int mode, void* stream);
/* scales[t] = 2^floor(log2(224 / amax_t)) (2x headroom under 448; 1 for an all-zero tensor); clears amax_bits */
int sfron_fp8_update_scales(uint32_t* amax_bits, int n_tensors, float* scales, void* stream);
/* How much of the e4m3 range the ACTIVATIONS of config 5 used since the last reset: out3[site] (HOST out) = max |x * scale| over every value
 * quantised at site 0 = sfron_ln_modulate_fwd_q's output, 1 = sfron_cast_e4m3, 2 = the e4m3 GELU output of sfron_fp8_gemm (c_e4m3).  The
 * conversion saturates at 448: a value above it means values WERE clipped (the reference has no fp8 path: nothing there to mirror).  The
 * three words are the CALLER's (round 6): a zero-initialised DEVICE uint32 [3] handed as `act_amax` to every quantising entry point
 * (sfron_ln_modulate_fwd_q, sfron_cast_e4m3, sfron_fp8_gemm_desc.act_amax, sfron_dit_forward_fp8; NULL there = no tracking); this call
 * copies them to the host (synchronises `stream`) and, with reset != 0, clears them. */
int sfron_fp8_activation_amax(uint32_t* act_amax, float* out3 /* HOST out, 3 floats */, int reset, void* stream);

/* dst[i] = e4m3(src[i] * scale); src bf16 (src_is_bf16 = 1) or fp32; n % 8 == 0 */
int sfron_cast_e4m3(const void* src, int src_is_bf16, int64_t n, float scale, uint8_t* dst, uint32_t* act_amax /* DEVICE [3] or NULL */,
                    void* stream);
/* sfron_ln_modulate_fwd that also writes out_e4m3 = e4m3(value * e4m3_scale) from the fp32 value (the A operand of the next GEMM) */
int sfron_ln_modulate_fwd_q(const float* x, const float* shift, const float* scale, int ldmod, int tokens, int M, int D, uint16_t* out,
                            uint8_t* out_e4m3, float e4m3_scale, float* mean, float* rstd, uint32_t* act_amax /* DEVICE [3] or NULL */,
                            void* stream);

/* C[M][N] = (A8[M][K] . B8[N][K]^T) / (a_scale * *w_scale) + bias with one of the block's epilogues:
 *   SFRON_EPI_BF16      c_bf16 = result                                                   (qkv)
 *   SFRON_EPI_GELU      aux = result (pre-activation, bf16), c_bf16 = gelu_tanh(result),
 *                       c_e4m3 = e4m3(gelu_tanh(result) * c_e4m3_scale) [M][N]            (fc1: h for the backward pass AND for fc2)
 *   SFRON_EPI_GATE_RES  aux = result (bf16), c_f32 = resid + gate[row / tokens] * result  (proj, fc2)
 * Shapes: M % 256 == 0, N % 128 == 0 or N % 144 == 0 (256 x 128 / 256 x 144 tiles, the one that fills the CUs better), K % 128 == 0
 * (sfron_fp8_gemm_supported). */
typedef struct sfron_fp8_gemm_desc {
  const uint8_t* A; const uint8_t* B;
  int M, N, K;
  const float* w_scale;              /* DEVICE scalar: the scale B was quantised with (sfron_fp8_update_scales output) */
  float a_scale;                     /* the scale A was quantised with */
  int epilogue;
  const float* bias;
  uint16_t* c_bf16; int ldc_bf16;
  uint16_t* aux; int ldaux;
  uint8_t* c_e4m3; float c_e4m3_scale;
  float* c_f32; int ldc_f32;
  const float* resid;                /* NULL = in place */
  const float* gate; int ldgate;
  int tokens;
  uint32_t* act_amax;                /* DEVICE uint32 [3] or NULL: the caller's activation-range words (sfron_fp8_activation_amax); the e4m3
                                        GELU output (c_e4m3) raises word 2 */
  int aux_q;                         /* SFRON_EPI_GELU only, 1 = `aux` receives gelu_tanh'(result) as one byte per element (uint8 [M][ldaux],
                                        the code of SFRON_EPI_GELU_Q) instead of the bf16 pre-activation: for a backward pass on
                                        SFRON_EPI_DGELU_Q */
} sfron_fp8_gemm_desc;
int sfron_fp8_gemm_supported(int M, int N, int K);
int sfron_fp8_gemm(const sfron_fp8_gemm_desc* desc /* HOST pointer */, void* stream);

/* ------------------------------------------------------------------ convolutional U-Net blocks (conv.hip)
 * Replaces the Conv2d / GroupNorm / bmm-softmax sequences of DDPM/models/diffusion.py:43-192,283-413 (Conditional_Model) forward
 * and backward.  Activations are NHWC: a [batch * H * W][C] row-major matrix (bf16 where they feed a GEMM, fp32 elsewhere). */

/* batched GEMM on the generic 128x128 tile: C[z] = alpha * op(A[z]) op(B[z]) (+ bias) (+ resid) (+ per-sample row vector),
 * z < batch, operands advanced by the element strides; layouts as sfron_gemm_desc.  Exactly one of c_bf16 / c_f32 is set. */
typedef struct sfron_bgemm_desc {
  const uint16_t* A; const uint16_t* B;
  int M, N, K, lda, ldb;
  int a_transposed, b_transposed;
  int batch;
  long stride_a, stride_b, stride_c;
  int batch2;                      /* inner batch (0 / 1 = none): attention heads that are column slices of one matrix */
  long stride_a2, stride_b2, stride_c2;
  float alpha;
  const float* bias;               /* [N] or NULL */
  uint16_t* c_bf16; float* c_f32; int ldc;
  const float* resid;              /* fp32 [M][ldc] added to the result (c_f32 only), or NULL */
  const float* sample_vec;         /* fp32 vec[(row / rows_per_sample) * ld_vec + col] added (c_f32 only), or NULL */
  int ld_vec, rows_per_sample;
  int accumulate;                  /* c_f32 += result */
  float* split_ws; int split_ws_slabs;   /* optional: fp32 scratch [split_ws_slabs][M * N] -- a plain weight gradient (both operands
                                      transposed, batch 1, ldc == N, no epilogue extras) then splits its contraction over the
                                      chip and adds the slabs in a fixed order */
} sfron_bgemm_desc;
int sfron_bgemm_bf16(const sfron_bgemm_desc* desc /* HOST pointer */, void* stream);

/* implicit-GEMM convolution.  The GEMM rows are the pixels of an (h_out x w_out) grid per sample; row p = (b, ho, wo) reads, for
 * tap (kh, kw), the source pixel (ho * stride + kh - pad, wo * stride + kw - pad) of a [batch][h_src][w_src][c_src] bf16 image:
 *   upsample = 1: through a nearest x2 upsampling (F.interpolate(scale_factor=2) + conv, models/diffusion.py:56-60)
 *   dilate   = 1: as a zero-dilated x2 image (the input gradient of a stride-2 convolution)
 * and zero outside it (padding; the (0,1,0,1) pad of Downsample, :76-80, is the bottom / right edge of a pad-0 stride-2 conv). */
typedef struct sfron_conv_desc {
  int batch, h_src, w_src, c_src;   /* c_src % 8 == 0 */
  int h_out, w_out;
  int n_out;                        /* GEMM N: output channels (forward) / input channels (input gradient) */
  int taps, stride, pad, upsample, dilate;
  const float* bias;                /* [n_out] or NULL */
  const float* resid;               /* fp32 [rows][ld_out] added (out_f32 only) or NULL */
  const float* sample_vec;          /* fp32 [batch][ld_vec] added per sample (out_f32 only) or NULL */
  int ld_vec;
  uint16_t* out_bf16; float* out_f32; int ld_out;
  int accumulate;
  float* split_ws; int split_ws_slabs;   /* optional fp32 scratch [split_ws_slabs][rows * n_out]: a forward / input-gradient conv with few
                                       output tiles and a deep contraction then splits the contraction over the chip */
  int* split_pending;               /* optional HOST int (ABI 16): when the call splits its contraction (and ld_out == n_out) it leaves
                                       *split_pending (> 0) slabs [rows * n_out] in split_ws and does NOT launch the pass that adds them and
                                       applies bias / sample_vec / resid -- the caller hands them to a GroupNorm that finishes the sum itself
                                       (sfron_split_src, sfron_groupnorm_*_src) or to sfron_split_finish; 0 = the output is complete.
                                       NULL = always complete. */
} sfron_conv_desc;
/* out[p][n] = sum_{tap, c} src[src(p, tap)][c] * w[n][tap][c]; w bf16 [n_out][taps][c_src] (sfron_conv_wprep's "fwd" layout;
 * the input gradient calls this on dY with the "dgrad" layout) */
int sfron_conv_fwd(const sfron_conv_desc* desc, const uint16_t* src, const uint16_t* w, void* stream);
/* dw_gemm fp32 [splits][n_out][taps * c_src]: slab s = sum over the s-th range of pixels p of dy[p][n] * src[src(p, tap)][c]
 * (dy bf16 [rows][ld_dy]); splits = sfron_conv_wgrad_splits(desc) >= 1 = the number of slabs WRITTEN (the contraction over all pixels is split
 * over the chip; size dw_gemm for it and pass it to sfron_conv_wgrad_scatter) */
int sfron_conv_wgrad_splits(const sfron_conv_desc* desc);
int sfron_conv_wgrad(const sfron_conv_desc* desc, const uint16_t* dy, int ld_dy, const uint16_t* src, float* dw_gemm, void* stream);
/* fp32 OIHW master weights -> bf16 operands: w_fwd [c_out_p][taps][c_in_p] (zero padded), w_dgrad [c_in][taps flipped][c_out_p] or NULL */
int sfron_conv_wprep(const float* w_oihw, int c_out, int c_in, int taps, int c_out_p, int c_in_p, uint16_t* w_fwd, uint16_t* w_dgrad,
                     void* stream);
/* the same for every 3x3 kernel of a model in one launch: a device-resident table, item i owning the tile ids
 * [tile0, tile0 + sfron_conv_wprep_tiles(c_out_p, c_in_p)) of the grid (tile0 ascending, n_tiles = their total) */
typedef struct sfron_wprep_item {
  const float* w;         /* OIHW fp32 master [c_out][c_in][3][3] */
  uint16_t* fwd;          /* bf16 [c_out_p][9][c_in_p] */
  uint16_t* dgr;          /* bf16 [c_in][9 flipped][c_out_p], or NULL */
  int32_t co, ci, co_p, ci_p, tile0, pad_;
} sfron_wprep_item;
int sfron_conv_wprep_tiles(int c_out_p, int c_in_p);
int sfron_conv_wprep_batch(const sfron_wprep_item* items_dev, int n_items, int n_tiles, void* stream);
/* dw_gemm [n_slabs][c_out_p..][taps][c_in_p] -> OIHW gradient [c_out][c_in][taps] (overwrite; the slabs are added in order) */
int sfron_conv_wgrad_scatter(const float* dw_gemm, int c_out, int c_in, int taps, int c_in_p, int n_slabs, int64_t slab_stride,
                             float* dw_oihw, void* stream);
/* The scatters of MANY kernel gradients in ONE launch (round 6): item i is sfron_conv_wgrad_scatter(dw_gemm, c_out, c_in, taps, c_in_p, n_slabs,
 * slab_stride, dw_oihw), bit for bit.  `items` is a HOST array (the items travel by value in the kernel arguments, 80 per launch; capturable).
 * A U-Net backward pass keeps each layer's slabs in a buffer of the layer's own and issues the scatters once, behind its last product. */
typedef struct sfron_wgrad_scatter_item {
  const float* dw_gemm; float* dw_oihw; int64_t slab_stride; int c_out, c_in, taps, c_in_p, n_slabs, reserved;
} sfron_wgrad_scatter_item;
int sfron_conv_wgrad_scatter_batch(const sfron_wgrad_scatter_item* items /* HOST */, int n_items, void* stream);

int sfron_nchw_to_rows_bf16(const float* x, int B, int C, int HW, int c_pad, uint16_t* rows, void* stream);
int sfron_nchw_to_rows_f32(const float* x, int B, int C, int HW, int ld, float* rows, void* stream);
int sfron_rows_to_nchw(const float* rows, int ld, int B, int C, int HW, float* x, void* stream);

/* y = bf16( act(GroupNorm(x; groups, eps) * gamma + beta) [* drop_mask * drop_scale] ), act = swish when `swish`; x fp32 rows
 * [B * HW][ldx]; mean / rstd [B][groups] saved for the backward pass (models/diffusion.py:43-46,126-131) */
int sfron_groupnorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, int B, int HW, int C, int groups, float eps,
                        int swish, const uint8_t* drop_mask, float drop_scale, uint16_t* y, float* mean, float* rstd,
                        void* scratch /* sfron_groupnorm_scratch_bytes(), 16-B aligned; NULL = per-(sample, group) kernels */, void* stream);
/* dy fp32 [B * HW][C] = gradient wrt y; dx (+)= gradient wrt x; part_gamma / part_beta [B][C] per-sample partial sums */
int sfron_groupnorm_bwd(const float* dy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                        const float* rstd, int B, int HW, int C, int groups, int swish, const uint8_t* drop_mask, float drop_scale,
                        float* dx, int lddx, int accumulate, float* part_gamma, float* part_beta, void* scratch, void* stream);
/* the same with one more term: dx (+)= extra + gradient wrt x, extra fp32 [B * HW][ld_extra] (the residual branch's share of x's
 * gradient -- ResnetBlock `x + h`, DDPM/models/diffusion.py:145 -- that a separate pass would add; same bits as
 * sfron_copy_cols(extra -> dx, accumulate) followed by sfron_groupnorm_bwd(accumulate = 1)).  extra == NULL: sfron_groupnorm_bwd. */
int sfron_groupnorm_bwd_res(const float* dy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                            const float* rstd, int B, int HW, int C, int groups, int swish, const uint8_t* drop_mask, float drop_scale,
                            float* dx, int lddx, int accumulate, const float* extra, int ld_extra, float* part_gamma, float* part_beta,
                            void* scratch, void* stream);
/* the backward pass for an x with ONE consumer whose gradient feeds a GEMM: dx leaves as the bf16 operand [B * HW][C] (no fp32 copy),
 * with its column sums per (sample, pixel chunk): col_partials fp32 [B][sfron_groupnorm_chunks(B, HW)][C] -- summed over everything
 * the bias gradient of the layer that produced x, summed per sample the gradient of a per-sample vector added to x (the
 * temb / class-embedding projection of ResnetBlock, DDPM/models/diffusion.py:120-124): sfron_reduce_chunks finishes both.  Replaces
 * sfron_groupnorm_bwd + sfron_cast_rows_colsum + sfron_sample_colsum over an fp32 dx.  Needs sfron_groupnorm_bwd_cast_ok(). */
int sfron_groupnorm_bwd_cast(const float* dy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                             const float* rstd, int B, int HW, int C, int groups, int swish, const uint8_t* drop_mask, float drop_scale,
                             uint16_t* dx_bf16, float* col_partials, float* part_gamma, float* part_beta, void* scratch, void* stream);

/* The GroupNorm INPUT as an unfinished split-K result (round 6, ABI 16): element (row, c) = sum_{s < n_slabs} slabs[s * slab_stride + row * C + c]
 * (in index order) + bias[c] + sample_vec[(row / (rows per sample)) * ld_vec + c] + resid[row * ld_resid + c] -- what the finish launch of a
 * split convolution (sfron_conv_desc.split_pending) would have written.  The one-launch GroupNorm (sfron_groupnorm_one_launch(B, HW, C, groups)
 * != 0: a workgroup per sample and block of whole groups) forms each element in its first pass and STORES it to the tensor the finish would have
 * filled (x of the forward pass / dy of the backward pass, row stride C: every later reader finds it there), so that launch no longer exists:
 * the same bits as sfron_split_finish followed by sfron_groupnorm_fwd / _bwd_res / _bwd_cast.  Returns SFRON_ERR_UNSUPPORTED (nothing launched)
 * when the shape does not take the one-launch form: the caller then calls sfron_split_finish and the plain entry point. */
typedef struct sfron_split_src {
  const float* slabs; int n_slabs; int64_t slab_stride;
  const float* bias; const float* sample_vec; int ld_vec; const float* resid; int ld_resid;    /* each optional */
} sfron_split_src;
int sfron_split_finish(const sfron_split_src* src /* HOST */, int64_t rows, int C, int rows_per_sample, float* out_f32, uint16_t* out_bf16,
                       int ld_out, void* stream);
int sfron_groupnorm_one_launch(int B, int HW, int C, int groups);
int sfron_groupnorm_fwd_src(const sfron_split_src* src /* HOST */, float* x /* [B * HW][C], written */, const float* gamma, const float* beta, int B,
                            int HW, int C, int groups, float eps, int swish, const uint8_t* drop_mask, float drop_scale, uint16_t* y, float* mean,
                            float* rstd, void* stream);
int sfron_groupnorm_bwd_res_src(const sfron_split_src* src /* HOST */, float* dy /* [B * HW][C], written */, const float* x, int ldx,
                                const float* gamma, const float* beta, const float* mean, const float* rstd, int B, int HW, int C, int groups,
                                int swish, const uint8_t* drop_mask, float drop_scale, float* dx, int lddx, int accumulate, const float* extra,
                                int ld_extra, float* part_gamma, float* part_beta, void* stream);
int sfron_groupnorm_bwd_cast_src(const sfron_split_src* src /* HOST */, float* dy /* [B * HW][C], written */, const float* x, int ldx,
                                 const float* gamma, const float* beta, const float* mean, const float* rstd, int B, int HW, int C, int groups,
                                 int swish, const uint8_t* drop_mask, float drop_scale, uint16_t* dx_bf16, float* col_partials,
                                 float* part_gamma, float* part_beta, void* stream);
int sfron_groupnorm_bwd_cast_ok(int ldx, int C, int groups);
int sfron_groupnorm_chunks(int B, int HW);
int64_t sfron_groupnorm_scratch_bytes(int B, int HW, int C, int groups);
/* p = bf16(softmax(scale * s)) over rows of length n; ds = bf16(scale * p * (dp - sum(p * dp)))   (AttnBlock, :168-186) */
int sfron_softmax_fwd(const float* s, int64_t rows, int n, int n_valid /* keys; columns beyond get probability 0 */, float scale, uint16_t* p,
                      void* stream);
/* y = bf16(LayerNorm(x; eps) * gamma + beta) on fp32 rows [rows][D]; backward: dx (+)=, per-block partial sums of d gamma / d beta
 * [ceil(rows / sfron_layernorm_rows_per_block(rows))][D] (BasicTransformerBlock.norm1..3, SD/ldm/modules/attention.py:223-225) */
int sfron_layernorm_fwd(const float* x, const float* gamma, const float* beta, int64_t rows, int D, float eps, uint16_t* y, float* mean,
                        float* rstd, void* stream);
int sfron_layernorm_rows_per_block(int64_t rows);
int sfron_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, int64_t rows, int D, float* dx,
                        int accumulate, float* part_gamma, float* part_beta, void* stream);
/* the same with one more term of x's gradient, extra fp32 [rows][D] (the residual stream's share, x + attn(norm(x)),
 * SD/ldm/modules/attention.py:271-273): dx (+)= extra + gradient, the bits of a separate dx (+)= extra pass followed by sfron_layernorm_bwd */
int sfron_layernorm_bwd_res(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, int64_t rows, int D, float* dx,
                            int accumulate, const float* extra, float* part_gamma, float* part_beta, void* stream);
/* GEGLU (attention.py:37-45): h fp32 [rows][2F] = value || gate; out = bf16(value * gelu_erf(gate)); backward dh bf16 [rows][2F] */
int sfron_geglu_fwd(const float* h, int64_t rows, int F, uint16_t* out, void* stream);
int sfron_geglu_bwd(const float* d_out, const float* h, int64_t rows, int F, uint16_t* dh, void* stream);
int sfron_softmax_bwd(const uint16_t* p, const float* dp, int64_t rows, int n, float scale, uint16_t* ds, void* stream);
/* out[b][c] = sum over the HW rows of sample b of x[row][c] */
int sfron_sample_colsum(const float* x, int ld, int B, int HW, int C, float* out, int ld_out,
                        float* scratch /* optional row-chunk partials, >= B * 32 * C floats */, int64_t scratch_floats, void* stream);
/* out = alpha * a + beta * b (guidance mix (1 + s) * cond - s * null of _forward_with_cond_scale, :340-357) */
int sfron_axpby(const float* a, const float* b, float alpha, float beta, int64_t n, float* out, void* stream);
/* backward of nearest x2 upsampling: dx[b][h][w][c] (+)= sum of the 2x2 block of dy [B][2H][2W][C] */
int sfron_pool2_sum(const float* dy, int B, int H, int W, int C, float* dx, int accumulate, void* stream);
int sfron_cast_rows_bf16(const float* x, int ldx, int64_t rows, int C, uint16_t* y, void* stream);
/* y = bf16(x) and colsum[c] = sum over rows of x[.][c] in one pass (output gradient -> GEMM operand + bias gradient, `Conv2d` /
 * `Linear` backward); partials: scratch of max_partials * C floats */
int sfron_cast_rows_colsum(const float* x, int ldx, int64_t rows, int C, uint16_t* y, float* partials, int max_partials, float* colsum,
                           void* stream);
/* The first half of sfron_cast_rows_colsum alone (round 6): y and the column-sum partials [*chunks_out][C]; the caller finishes with
 * sfron_reduce_chunks(partials, 1, *chunks_out, C, colsum, C, 0) -- or collects that finish into a sfron_reduce_batch launch. */
int sfron_cast_rows_colsum_partials(const float* x, int ldx, int64_t rows, int C, uint16_t* y, float* partials, int max_partials,
                                    int* chunks_out /* HOST */, void* stream);
/* y[r][0..C) (+)= x[r][0..C): channel-slice copies of torch.cat(dim=1) and its backward (:404) */
/* keep mask of nn.Dropout(p) (DDPM/models/diffusion.py:131), 1 = kept with probability 1 - p: a counter-based draw keyed by
 * (seed, counter[0] on the device, salt, element); the caller advances the device counter once per pass */
int sfron_dropout_mask(uint64_t seed, const int64_t* counter, int64_t salt, int64_t n, float p, uint8_t* mask, void* stream);
/* The same masks for n_items (salt, n) pairs in one launch: items_dev = DEVICE int64 triples [salt, n, byte offset into mask_base]
 * (offsets multiples of 4), max_n = the largest n.  Bit for bit what n_items calls of sfron_dropout_mask write (the nn.Dropout of every
 * ResnetBlock of a pass, DDPM/models/diffusion.py:131). */
int sfron_dropout_mask_batch(uint64_t seed, const int64_t* counter, const int64_t* items_dev, int n_items, int64_t max_n, float p,
                             uint8_t* mask_base, void* stream);
int sfron_copy_cols(const float* x, int ldx, int64_t rows, int C, float* y, int ldy, int accumulate, void* stream);
/* Two such copies in one launch: y0[rows][ldy0] columns [0, C0) (+)= x0, y1 columns [0, C1) (+)= x1 -- the up path's torch.cat([h, skip], dim=1)
 * (DDPM/models/diffusion.py:403, openaimodel.py:791) with y0 / y1 the two column halves of the concatenated tensor, and its backward with x0 / x1
 * the halves of its gradient. */
int sfron_copy_cols2(const float* x0, int ldx0, int C0, float* y0, int ldy0, int accumulate0, const float* x1, int ldx1, int C1, float* y1,
                     int ldy1, int accumulate1, int64_t rows, void* stream);
/* get_timestep_embedding (:17-35): bf16 [n][dim] = sin(t f) || cos(t f), f_j = exp(-ln(1e4) j / (dim/2 - 1)); t float */
int sfron_ddpm_timestep_embed(const float* t, int n, int dim, uint16_t* out, void* stream);
/* out[b] = keep[b] ? table[c[b]] : null_emb (:370-376; keep NULL = all kept); backward accumulates into d_table (+=), writes d_null */
int sfron_class_embed_fwd(const float* table, const float* null_emb, const int64_t* c, const uint8_t* keep, int n_classes, int n, int D,
                          uint16_t* out, void* stream);
int sfron_class_embed_bwd(const float* d, const int64_t* c, const uint8_t* keep, int n_classes, int n, int D, float* d_table, float* d_null,
                          void* stream);

/* ------------------------------------------------------------------ adaLN-Zero elementwise (norm.hip)
 * mod buffers are fp32 [batch][ldmod]; shift/scale/gate pointers already include the column offset of the
 * chunk (DiT/models.py:119 .chunk(6, dim=1)).  `tokens` = tokens per sample (row / tokens = sample). */

/* rows per reduction chunk of the *_bwd kernels' partial buffers (one workgroup's rows: 16 when 16 | tokens) */
int sfron_rows_per_chunk(int tokens);

/* out = bf16( LayerNorm(x; eps 1e-6, no affine) * (1 + scale) + shift ), saves mean / rstd per row
 * (modulate(norm(x), shift, scale), DiT/models.py:19-20,120-121,139-140) */
int sfron_ln_modulate_fwd(const float* x, const float* shift, const float* scale, int ldmod, int tokens, int M, int D,
                          uint16_t* out, float* mean, float* rstd, void* stream);

/* backward of the above: dx (+)= dLN; p_shift/p_scale [M / rows_per_chunk][D] = per-chunk column sums of
 * d_out and d_out * xhat (finish with sfron_reduce_chunks -> d shift / d scale of the adaLN Linear) */
int sfron_ln_modulate_bwd(const uint16_t* d_out, const float* x, const float* mean, const float* rstd, const float* scale,
                          int ldmod, int tokens, int M, int D, float* dx, int dx_accumulate, float* p_shift,
                          float* p_scale, void* stream);

/* backward of x + gate * branch (DiT/models.py:120-121): d_branch = bf16(dy * gate);
 * p_gate / p_dy [M / rows_per_chunk][D] = per-chunk column sums of dy * branch and of dy */
int sfron_gate_bwd(const float* dy, const uint16_t* branch, const float* gate, int ldmod, int tokens, int M, int D,
                   uint16_t* d_branch, float* p_gate, float* p_dy, void* stream);

/* sfron_ln_modulate_bwd followed by sfron_gate_bwd of the NEXT (earlier) branch on the freshly accumulated dx rows, in
 * one pass (saves the fp32 re-read of dx): DiT/models.py:120-121 backward, branch k's LN + branch k-1's gate. */
int sfron_ln_gate_bwd(const uint16_t* d_out, const float* x, const float* mean, const float* rstd, const float* scale,
                      int ldmod, int tokens, int M, int D, float* dx, int dx_accumulate, float* p_shift, float* p_scale,
                      const uint16_t* branch, const float* gate, int ldgate, uint16_t* d_branch, float* p_gate, float* p_dy,
                      void* stream);

/* out[g * ldout + c] (+)= sum_{j < per_group} partials[(g * per_group + j) * D + c]   (fixed order, reproducible) */
int sfron_reduce_chunks(const float* partials, int groups, int per_group, int D, float* out, int ldout, int accumulate,
                        void* stream);
/* MANY fixed-order reductions in ONE launch (round 6): item i computes out[g * ldout + c] = sum_{j < per_group} partials[(g * per_group + j) * D + c]
 * for g < groups, c < D -- exactly what sfron_reduce_chunks(partials, groups, per_group, D, out, ldout, 0) computes, in the same order, bit for
 * bit.  `items` is a HOST array: the items travel by value in the kernel arguments (120 per launch), so there is no device table to build or to
 * keep alive and the call may be captured into a graph.  The parameter-gradient finishes of a U-Net backward pass (GroupNorm / LayerNorm affine
 * gradients, bias gradients, per-sample projection gradients: ~190 launches of ~5 us per DDPM step) are collected by the tape and issued once. */
typedef struct sfron_reduce_item { const float* partials; float* out; int groups, per_group, D, ldout; } sfron_reduce_item;
int sfron_reduce_batch(const sfron_reduce_item* items /* HOST */, int n_items, void* stream);
/* two partial buffers in one launch: out0[g*ld0 + c] = sum_j p0[(g*per_group + j)*D + c], out1 likewise from p1 */
int sfron_reduce2(const float* p0, const float* p1, int groups, int per_group, int D, float* out0, int ld0, float* out1,
                  int ld1, void* stream);
/* bias gradients behind the adaLN gates of all blocks at once (proj.bias: which = 0, fc2.bias: which = 1):
 * out[l*out_stride + out_which{0,1} + c] = sum_b gate[b*ldg + l*gate_stride + which*gate_which + c] * S[((2l+which)*B + b)*D + c]
 * where S holds the per-sample token sums of the upstream gradient (second output of sfron_gate_bwd, reduced). */
int sfron_gated_bias_grads(const float* S, const float* gate, int ldg, long gate_stride, long gate_which, int layers, int B,
                           int D, float* out, long out_stride, long out_which0, long out_which1, void* stream);
/* Deferred reduction of a whole backward pass in ONE launch.  parts holds n_slots slots of slot_stride floats; slot
 * s = layer*8 + kind*2 + buf is a [groups*per_group][D] partial buffer written by sfron_gate_bwd / sfron_ln_modulate_bwd.
 * out: dst_base[kind*2+buf][layer*dst_layer_stride[..] + g*dst_ld[..] + c] = sum_j slot[(g*per_group + j)*D + c].
 * The three arrays are HOST arrays of length 8. */
int sfron_reduce_slots(const float* parts, long slot_stride, int n_slots, int groups, int per_group, int D,
                       float* const* dst_base, const long* dst_layer_stride, const int* dst_ld, void* stream);
/* out[c] = sum_g w[g * ldw + c] * sum_j partials[(g * per_group + j) * D + c]      (bias grad behind a gate) */
int sfron_weighted_reduce(const float* partials, int groups, int per_group, int D, const float* w, int ldw, float* out,
                          void* stream);
/* out[c] = sum_r X[r][c]; X bf16 (is_bf16 = 1) or fp32; partials scratch [max_partials][N]      (bias grads) */
int sfron_colsum(const void* X, int is_bf16, int M, int N, int ld, float* partials, int max_partials, float* out,
                 void* stream);

/* ------------------------------------------------------------------ conditioning / layout (embed.hip) */
/* out[b] = bf16( cos(t*f) || sin(t*f) ), f_j = exp(-ln(1e4) j / (dim/2))    (DiT/models.py:41-59) */
int sfron_timestep_embed(const int64_t* t, int n, int dim, uint16_t* out, int ld, void* stream);
int sfron_silu_fwd(const float* x, int64_t n, uint16_t* y_bf16, void* stream);
int sfron_silu_bwd(const float* dy, const float* x, int64_t n, uint16_t* dx_bf16, float* dx_f32, void* stream);
/* c = t_emb + table[drop ? num_classes : y];  silu_c = bf16(silu(c))   (DiT/models.py:78-94,243; drop may be NULL) */
int sfron_cond_fwd(const float* t_emb, const float* table, const int64_t* y, const uint8_t* drop, int num_classes, int n,
                   int D, float* c, uint16_t* silu_c, void* stream);
/* d_c = d_silu_c * silu'(c); d_table[label] += d_c  (d_table must be zeroed by the caller) */
int sfron_cond_bwd(const float* d_silu_c, const float* c, const int64_t* y, const uint8_t* drop, int num_classes, int n,
                   int D, float* d_c, float* d_table, void* stream);
/* step guard (fail-loud checks without host syncs; guard.py polls `flags` behind an event).  flags: fp32 [4] =
 * {non-finite loss, non-finite gradient norm, label outside [0, num_classes), timestep outside [0, num_timesteps)}, accumulated.
 * guard_inputs: y_safe / t_safe = the inputs clamped into range (what the kernels then index with), flags[2] / [3] += number of
 * clamped entries -- where the reference's nn.Embedding / table gather raise IndexError (DiT/models.py:89-93,
 * gaussian_diffusion.py:861-873).  guard_finite: a (and optional b, c, d) are per-sample loss terms [n]; stats (optional) =
 * the clip statistics of sfron_clip_coef, stats[0] = gradient norm (the values DiT/forget.py:329-336 logs). */
int sfron_guard_inputs(const int64_t* y, const int64_t* t, int n, int num_classes, int num_timesteps, int64_t* y_safe, int64_t* t_safe,
                       float* flags, void* stream);
int sfron_guard_finite(const float* a, const float* b, const float* c, const float* d, int n, const float* stats, float* flags, void* stream);
/* latent front-end: out [n][c][hw] = (mean + exp(0.5 * clamp(logvar, -30, 20)) * eps) * scale from cached VAE posterior moments
 * [n][2c][hw] = mean || logvar -- vae.encode(x).latent_dist.sample().mul_(0.18215) of DiT/forget.py:265-267,305-307 with the
 * encoder's output cached offline (diffusers' DiagonalGaussianDistribution.sample; diffusers is absent here: parity unpinned) */
int sfron_latent_sample(const float* moments, const float* eps, int n, int c, int hw, float scale, float* out, void* stream);
/* NCHW fp32 image -> bf16 token rows [n*T][C*p*p]; chan_last 0: k = c*p*p + ph*p + pw (Conv2d weight order,
 * timm PatchEmbed); 1: k = (ph*p + pw)*C + c (unpatchify order, DiT/models.py:218-231) */
int sfron_patchify(const float* img, int n, int C, int H, int W, int p, int chan_last, uint16_t* rows, int ld, void* stream);
int sfron_unpatchify(const float* rows, int ld, int n, int C, int H, int W, int p, float* img, void* stream);

/* ------------------------------------------------------------------ attention (attn.hip)
 * qkv [B*T][3*H*hd] bf16 (column = which*D + head*hd + d), o / d_o [B*T][H*hd] bf16, lse [B][H][T] fp32.
 * softmax(q k^T * hd^-0.5) v, non-causal (timm Attention as used at DiT/models.py:108,120).
 * Supported: T % 64 == 0 with head_dim a multiple of 8 up to 80 (DiT 64 / 72; the LDM UNet's 40 / 80) on the tiled kernels; T < 64
 * (any T, head_dim <= 128: the patch-8 DiT models at 256 px have 16 tokens) on plain-FMA kernels, one workgroup per (batch, head). */
int sfron_attn_fwd(const uint16_t* qkv, uint16_t* o, float* lse, int B, int T, int H, int hd, void* stream);
/* dqkv [B*T][3*H*hd] bf16 = gradient wrt qkv; delta_scratch fp32 [B*H*T] (used by the two-kernel form only).
 * T = 128 / 256: ONE kernel, one 8-wave workgroup per (batch, head): S and dP are formed once per (query chunk, key block),
 * dV and dK accumulate from registers, dS crosses LDS once and dQ is formed from that image -- 5 products, no atomics.
 * Other T (multiples of 64): dQ kernel + dK/dV kernel (7 products). */
int sfron_attn_bwd(const uint16_t* qkv, const uint16_t* o, const uint16_t* d_o, const float* lse, float* delta_scratch,
                   uint16_t* dqkv, int B, int T, int H, int hd, void* stream);
/* the same backward pass (one-kernel form only: sfron_attn_bwd_bias_supported(T)) that also leaves the qkv.bias gradient partials:
 * bias_partials fp32 [B][3*H*hd], row b = sum over the T tokens of sample b of dqkv (fp32, before the bf16 rounding; fixed
 * summation order, no atomics) -- sum the B rows with sfron_reduce_chunks.  Replaces a second pass over dqkv (sfron_colsum):
 * the bias gradient of timm Attention.qkv (DiT/models.py:108,120 backward). */
int sfron_attn_bwd_bias_supported(int T);
int sfron_attn_bwd_bias(const uint16_t* qkv, const uint16_t* o, const uint16_t* d_o, const float* lse, uint16_t* dqkv, float* bias_partials,
                        int B, int T, int H, int hd, void* stream);

/* test hook: 2 = always the two-kernel backward, 0 = default; returns the previous setting (process-wide, not thread-safe) */
int sfron_attn_bwd_form(int form);
/* test / A-B hook: 0 (default) = by rule (eight waves of 16 query rows per workgroup for sequences of 512 tokens or more, four waves of 32
 * rows below), 4 / 8 / 16 = force the four-wave / eight-wave 16-row / eight-wave 32-row (one workgroup per 256 query rows; measured slower)
 * form, 2 (round 6) = four waves walking BOTH 128-row query blocks of a head as one chunk stream where the sequence is a multiple of 256 rows
 * (measured: not faster) -- each where the sequence length allows it (bit-identical results); returns the previous setting */
int sfron_attn_fwd_form(int form);
/* Process-wide form of the three-slot GEMM tiles (256 x 144 forward / dgrad, 192 x 192 weight gradient): 4 = four extra LOADER waves per
 * workgroup issue every LDS-DMA piece and the eight multiplying waves none (csrc/gemm.hip k_gemm_pipe NL; taken by the dgrad and
 * weight-gradient layouts, where it measured faster) -- the default; 0 = every wave issues its share (5 / 6: weight gradients / dgrad
 * only, 9: the fp8 tiles in their loader form too -- for A-B runs).  The pipelined convolution tiles (csrc/conv.hip k_cgemm / k_cgemm_t,
 * contractions of >= 8 K-tiles) follow: loader form unless n == 0 (10 / 11 / 12: as 4 with none / only k_cgemm / only k_cgemm_t of
 * them in the loader form).  Same results bit for bit (same products, same summation order).  Returns the previous value.
 * The product build accepts 0, 4, 9 and 10 (what the form-equivalence tests use) and ignores any other value; the A-B values 5..8, 11, 12
 * exist in the debug-knob build (make dbg: libsfron_dbg.so, tools/ only). */
int sfron_gemm_loader_waves(int n);

/* ------------------------------------------------------------------ whole-model DiT pass (dit_engine.hip)
 * Replaces autograd over DiT.forward (DiT/models.py:233-248) inside the SFR-on step (DiT/forget.py:271-288,310-319).
 * Parameters live in ONE arena (fp32 master, bf16 shadow and fp32 grads share offsets); sfron_dit_param_layout
 * reports the offsets so the host can expose every tensor under its reference state_dict key. */
typedef struct sfron_dit_cfg {
  int batch;          /* per-GPU batch */
  int in_channels;    /* 4 */
  int input_size;     /* latent side, 32 for 256 px */
  int patch;          /* 2 / 4 / 8 */
  int hidden;         /* D */
  int depth;          /* L */
  int heads;          /* H; head_dim = D / H must be 64 or 72 */
  int mlp_hidden;     /* 4 D */
  int num_classes;    /* table has num_classes + 1 rows (CFG null class) */
  int freq_dim;       /* 256 */
  int out_channels;   /* 8 with learn_sigma */
} sfron_dit_cfg;

/* out[SFRON_DIT_LAYOUT_LEN] (element offsets into the arena): total, trainable, pe_w, pe_b, t0_w, t0_b, t2_w, t2_b,
 * table, ada_w, ada_b, blocks, blk_stride, then offsets inside a block: qkv_w, qkv_b, proj_w, proj_b, fc1_w, fc1_b,
 * fc2_w, fc2_b, then fin_w, fin_b, pos.  ada_w is [(6*depth+2)*D][D]: block l rows [6lD, 6(l+1)D), final layer last. */
#define SFRON_DIT_LAYOUT_LEN 24
int sfron_dit_param_layout(const sfron_dit_cfg* cfg, int64_t* out, int n_out);
int64_t sfron_dit_workspace_bytes(const sfron_dit_cfg* cfg);   /* < 0 on unsupported cfg */

/* out [B][out_channels][S][S] fp32 = DiT(x_t, t, y); drop [B] uint8 (1 = replace label by the null class) or NULL.
 * Saves activations in `workspace` for sfron_dit_backward. */
int sfron_dit_forward(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* x_t,
                      const int64_t* t, const int64_t* y, const uint8_t* drop, void* workspace, float* out, void* stream);
/* sfron_dit_forward_probed whose block l first waits (hipStreamWaitEvent on `stream`) for block_ready[l] (hipEvent_t, [depth], NULL
 * entries skipped): the optimizer sweep of the previous stage may still be rewriting the LATER blocks' weights on another stream
 * while the first blocks already run (DiT/forget.py:299 -> :310: the remain forward needs block l's new weights only at block l). */
int sfron_dit_forward_after(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* x_t,
                            const int64_t* t, const int64_t* y, const uint8_t* drop, void* workspace, float* out,
                            void* const* block_ready, void* probe, void* stream);
/* Config 5: the same forward pass with the four block GEMMs on the fp8 matrix core.  params_e4m3: e4m3 shadow arena (same offsets as
 * params); w_scales: DEVICE fp32 [depth][4] = quantisation scales of {qkv, proj, fc1, fc2}.weight of each block; act_scales: HOST
 * fp32 [3] = static scales of {LayerNorm+modulate output, attention output, gelu(fc1)}; workspace_e4m3: sfron_dit_fp8_workspace_bytes.
 * Saves the same bf16 activations as sfron_dit_forward, so sfron_dit_backward follows unchanged (straight-through estimator).
 * Returns SFRON_ERR_UNSUPPORTED when a block GEMM shape is not a multiple of the fp8 tile (sfron_fp8_gemm_supported). */
int64_t sfron_dit_fp8_workspace_bytes(const sfron_dit_cfg* cfg);
int sfron_dit_forward_fp8(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const uint8_t* params_e4m3,
                          const float* w_scales, const float* act_scales, uint32_t* act_amax /* DEVICE [3] or NULL */, const float* x_t,
                          const int64_t* t, const int64_t* y,
                          const uint8_t* drop, void* workspace, void* workspace_e4m3, float* out,
                          void* const* block_ready /* as sfron_dit_forward_after, or NULL */, void* stream);
/* The forward pass in two calls (round 6).  phase 1 = only what stands in front of block 0 -- patch embedding, timestep / label embedders,
 * the adaLN modulation of every block: the "conditioning prologue", a chain of ten small dependent launches; phase 2 = only the blocks and
 * the final layer, on the workspace the phase-1 call filled (same arguments).  A caller whose optimizer sweep of the block ranges runs beside
 * the pass on another stream (block_ready) starts that sweep BETWEEN the two calls: beside a bandwidth-heavy sweep every boundary between two
 * small dependent launches costs 60-100 us instead of ~5 (profiles/r06_stage_boundary.txt).  phase 1 ignores block_ready / probe / out.
 * phase 3 = phase 1 without its last launch -- the adaLN product silu(c) W_ada^T + b, the only launch in front of block 0 that reads the
 * adaLN_modulation matrix (DiT/models.py:113-116,119) -- and phase 4 = that launch alone (ABI 16): a caller whose optimizer sweep of that
 * matrix runs on another stream issues phase 3 beside it, orders `stream` behind the sweep, then phase 4 and phase 2. */
int sfron_dit_forward_phase(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* x_t,
                            const int64_t* t, const int64_t* y, const uint8_t* drop, void* workspace, float* out,
                            void* const* block_ready /* or NULL */, void* probe /* or NULL */, int phase /* 1 | 2 | 3 | 4 */, void* stream);
int sfron_dit_forward_fp8_phase(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const uint8_t* params_e4m3,
                                const float* w_scales, const float* act_scales, uint32_t* act_amax /* DEVICE [3] or NULL */, const float* x_t,
                          const int64_t* t, const int64_t* y,
                                const uint8_t* drop, void* workspace, void* workspace_e4m3, float* out,
                                void* const* block_ready /* or NULL */, int phase /* 1 | 2 | 3 | 4 */, void* stream);
/* Same as sfron_dit_forward, with HIP events recorded (on `stream`) around the fc1 GEMM of block 0 -- the
 * dominant kernel class -- into `probe` (may be NULL).  Used by bench.py for the live roofline measurement. */
int sfron_dit_forward_probed(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* x_t,
                             const int64_t* t, const int64_t* y, const uint8_t* drop, void* workspace, float* out,
                             void* probe, void* stream);
int sfron_probe_create(int max_samples, void** probe /* HOST out */);
int sfron_probe_reset(void* probe);
int sfron_probe_read(void* probe, int* n_samples, double* total_ms);   /* synchronises on the recorded events */
int sfron_probe_destroy(void* probe);
/* grads (arena layout, trainable part fully overwritten) = d loss / d params given d_out = d loss / d out.
 * aux (from sfron_aux_create, may be NULL): a side HIP stream + events; the weight-gradient GEMMs and bias column
 * sums then run concurrently with the dgrad / elementwise chain and join `stream` before the call's work ends. */
int sfron_dit_backward(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* d_out,
                       const int64_t* y, const uint8_t* drop, void* workspace, float* grads, void* aux, void* stream);
/* Data-parallel form of the backward pass.  block_events: L hipEvent_t handles (entries may be NULL), event l is recorded on the
 * aux handle's weight-gradient stream once every gradient inside block l's arena range is final except proj.bias and fc2.bias;
 * late_bias: fp32 [L][2][D] that receives those two instead of the arena.  The host can then all-reduce block ranges while the
 * backward pass is still running, all-reduce late_bias and the non-block ranges at the end, and call
 * sfron_dit_scatter_late_bias to put the reduced biases into the arena.  ada_dmod_out / ada_sc_out (both or neither): the
 * adaLN_modulation weight gradient (all blocks + final layer, [(6L+2)D][D], a third of the arena) is NOT computed; instead its two
 * bf16 factors dmod [B][(6L+2)D] and silu(c) [B][D] are copied out, so that the host all-gathers them over the ranks and forms
 * dmod_all^T silu(c)_all with one sfron_gemm_bf16 over the global batch.  (Reference: nn.DataParallel reduces after
 * loss.backward(), DiT/forget.py:193,288.) */
int sfron_dit_backward_dp(const sfron_dit_cfg* cfg, const float* params, const uint16_t* params_bf16, const float* d_out,
                          const int64_t* y, const uint8_t* drop, void* workspace, float* grads, void* aux,
                          void* const* block_events, float* late_bias, uint16_t* ada_dmod_out, uint16_t* ada_sc_out, void* stream);
int sfron_dit_scatter_late_bias(const sfron_dit_cfg* cfg, const float* late_bias, float* grads, void* stream);
int sfron_aux_create(void** aux /* HOST out */);
/* attach a probe (sfron_probe_create; NULL detaches): the backward pass then records an event pair on the weight-gradient
 * stream around each of the four weight-gradient GEMMs (qkv, proj, fc1, fc2) of every 9th block -- the kernel that holds the
 * largest share of GPU time -- for bench.py's live roofline figure */
int sfron_aux_set_probe(void* aux, void* probe);
/* One-shot: the NEXT sfron_dit_backward / _dp call through this handle also leaves the masked sum of squares of every block weight gradient
 * (qkv / proj / fc1 / fc2 weights of all blocks) in `partials` -- fp64 [sfron_dit_sumsq_partials_len(cfg)], one per 192 x 192 output tile, every
 * entry rewritten by the pass -- taken from the accumulators of the weight-gradient GEMMs (sfron_gemm_desc.sumsq_partials).  mask_arena: byte mask
 * indexed like the parameter / gradient arena (NULL = no mask).  The clip norm of DiT/forget.py:289-298 then needs a pass only over what is left
 * (biases, embedders, final layer: sfron_sumsq_masked_ranges) plus sfron_clip_coef.  Single-process runs only: a data-parallel run must take the
 * norm of the REDUCED gradient.  sfron_dit_sumsq_partials_len returns 0 when a block shape does not run on the 192 x 192 weight-gradient tile.
 * partials == NULL DISARMS the handle (a caller whose pass failed between arming and the backward pass must not leave the pointer behind). */
int sfron_dit_sumsq_partials_len(const sfron_dit_cfg* cfg);
int sfron_aux_arm_sumsq(void* aux, const uint8_t* mask_arena, double* partials);
/* Orders `stream` behind the point of the LAST backward pass run with this handle after which the adaLN_modulation matrix (weights, bf16
 * shadow) is not read and its gradient factors (sfron_dit_backward_dp: ada_dmod_out / ada_sc_out) are complete: the dgrad through that
 * Linear.  What remains of the pass is the embedders' backward; the clip norm's share of the adaLN matrix (sfron_sumsq_lowrank) may run
 * beside it on another stream (the host mirror does: -0.12 ms per step; sweeping the matrix itself there measured slower).  No-op before
 * the first backward pass.  Replaces nothing in the reference: scheduling of DiT/forget.py:293-298. */
int sfron_aux_wait_ada(void* aux, void* stream);
/* Orders `stream` behind the point of the LAST sfron_dit_backward_dp call with ada_dmod_out / ada_sc_out through this handle at which those two
 * factors are complete (earlier than sfron_aux_wait_ada: the dgrad through the adaLN Linear is still to come).  sfron_sumsq_lowrank reads
 * nothing else.  No-op before the first such pass. */
int sfron_aux_wait_ada_factors(void* aux, void* stream);
/* The handle's two weight-gradient streams (hipStream_t, owned by the library: read-only use).  For callers that put work of their own beside a
 * backward pass -- a gradient exchange, a sweep -- and must pick a stream that does NOT share a hardware queue with these two: the runtime maps
 * streams onto four hardware queues and two streams on one queue serialise (profiles/r06_hw_queues.txt; the host mirror probes: streams.py). */
int sfron_aux_streams(void* aux, void** side /* HOST out */, void** side2 /* HOST out */);
int sfron_aux_destroy(void* aux);

#ifdef __cplusplus
}
#endif
#endif /* SFRON_H */
