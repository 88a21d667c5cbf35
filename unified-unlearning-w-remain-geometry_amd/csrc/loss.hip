// Diffusion loss of the DiT path: q_sample, and the fused (mse + vb) forward + backward.
//
// Replaces the PyTorch op sequences of /root/reference/DiT/diffusion:
//   q_sample ............. gaussian_diffusion.py:215-230 (tables gathered as fp32: :861-873)
//   training_losses ...... gaussian_diffusion.py:746-783 (MSE loss, EPSILON mean, LEARNED_RANGE var)
//   _vb_terms_bpd ........ gaussian_diffusion.py:682-713 with p_mean_variance :285-293,317-332,
//                          q_posterior_mean_variance :232-252, _predict_xstart_from_eps :334-339
//   normal_kl / cdf / nll  diffusion_utils.py:10-44,62-88
// One pass reads (x0, noise, model_output) and writes per-sample mse / vb plus
// dOut = d( grad_scale * sum_i loss_i ) / d model_output -- so the denoiser backward starts
// from dOut with no autograd graph on the loss.  eps_hat is detached inside vb (reference
// :758), so eps channels get only the mse gradient and var channels only the vb gradient.
//
// tab: [T][8] fp32 = sqrt_ac, sqrt_1m_ac, sqrt_recip_ac, sqrt_recipm1_ac, post_coef1, post_coef2,
//      post_logvar_clipped, log_beta   (each = the fp64 table value rounded once to fp32).
#include "common.h"
#include "../../include/sfron.h"

namespace {

constexpr int TPB = 256;

__global__ __launch_bounds__(TPB) void k_q_sample(const float* __restrict__ x0, const float* __restrict__ noise,
                                                  const int64_t* __restrict__ t, const float* __restrict__ tab,
                                                  int chw, float* __restrict__ xt) {
  const int n = blockIdx.y;
  const float* row = tab + (size_t)t[n] * 8;
  const float a = row[0], b = row[1];
  const size_t base = (size_t)n * chw;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < chw; i += gridDim.x * TPB)
    xt[base + i] = a * x0[base + i] + b * noise[base + i];
}

__device__ __forceinline__ float cdf_approx(float x, float& dcdf) {
  // 0.5*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3))) and its derivative
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float x2 = x * x;
  float u = k0 * (x + k1 * x2 * x);
  float th = tanhf(u);
  dcdf = 0.5f * (1.0f - th * th) * k0 * (1.0f + 3.0f * k1 * x2);
  return 0.5f * (1.0f + th);
}

// one workgroup per sample
__global__ __launch_bounds__(TPB) void k_dit_loss(const float* __restrict__ x0, const float* __restrict__ noise,
                                                  const float* __restrict__ out, const int64_t* __restrict__ t,
                                                  const float* __restrict__ tab, int C, int hw, float grad_scale,
                                                  float* __restrict__ mse_out, float* __restrict__ vb_out,
                                                  float* __restrict__ dout) {
  __shared__ double sh[2][TPB / 64];
  const int n = blockIdx.x;
  const int64_t tn = t[n];
  const float* row = tab + (size_t)tn * 8;
  const float sa = row[0], s1 = row[1], sra = row[2], srm1 = row[3], c1 = row[4], c2 = row[5];
  const float min_log = row[6], max_log = row[7];
  const int chw = C * hw;
  const float inv_chw = 1.0f / (float)chw;
  const float inv_ln2 = 1.4426950408889634f;
  const size_t xb = (size_t)n * chw;
  const size_t ob = (size_t)n * 2 * chw;
  float acc_mse = 0.f, acc_vb = 0.f;
  for (int i = threadIdx.x; i < chw; i += TPB) {
    const float x = x0[xb + i], e = noise[xb + i];
    const float eh = out[ob + i], vv = out[ob + chw + i];
    const float xt = sa * x + s1 * e;
    // ---- mse term
    const float d = e - eh;
    acc_mse += d * d;
    const float g_eps = 2.0f * (eh - e) * inv_chw * grad_scale;
    // ---- vb term
    const float frac = (vv + 1.0f) / 2.0f;
    const float lv = frac * max_log + (1.0f - frac) * min_log;
    const float pred_x0 = sra * xt - srm1 * eh;
    const float mu_t = c1 * pred_x0 + c2 * xt;      // model mean
    const float mu_q = c1 * x + c2 * xt;            // true posterior mean
    float term, dterm_dlv;
    if (tn == 0) {
      const float cx = x - mu_t;
      const float inv_std = expf(-0.5f * lv);
      const float plus_in = inv_std * (cx + 1.0f / 255.0f);
      const float min_in = inv_std * (cx - 1.0f / 255.0f);
      float dplus, dmin;
      const float cdf_plus = cdf_approx(plus_in, dplus);
      const float cdf_min = cdf_approx(min_in, dmin);
      // d plus_in / d lv = -0.5 * plus_in
      const float dp = dplus * (-0.5f * plus_in), dm = dmin * (-0.5f * min_in);
      float logp, dlogp;
      if (x < -0.999f) {
        const float a = fmaxf(cdf_plus, 1e-12f);
        logp = logf(a);
        dlogp = (cdf_plus >= 1e-12f) ? dp / a : 0.f;
      } else if (x > 0.999f) {
        const float q = 1.0f - cdf_min;
        const float a = fmaxf(q, 1e-12f);
        logp = logf(a);
        dlogp = (q >= 1e-12f) ? -dm / a : 0.f;
      } else {
        const float q = cdf_plus - cdf_min;
        const float a = fmaxf(q, 1e-12f);
        logp = logf(a);
        dlogp = (q >= 1e-12f) ? (dp - dm) / a : 0.f;
      }
      term = -logp;
      dterm_dlv = -dlogp;
    } else {
      const float dmu = mu_q - mu_t;
      const float e1 = expf(min_log - lv), e2 = expf(-lv);
      term = 0.5f * (-1.0f + lv - min_log + e1 + dmu * dmu * e2);
      dterm_dlv = 0.5f * (1.0f - e1 - dmu * dmu * e2);
    }
    acc_vb += term;
    const float g_var = dterm_dlv * (0.5f * (max_log - min_log)) * inv_chw * inv_ln2 * grad_scale;
    dout[ob + i] = g_eps;
    dout[ob + chw + i] = g_var;
  }
  double a = wave_sum_d((double)acc_mse), b = wave_sum_d((double)acc_vb);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0, tb = 0;
    for (int i = 0; i < TPB / 64; ++i) { ta += sh[0][i]; tb += sh[1][i]; }
    mse_out[n] = (float)(ta / chw);
    vb_out[n] = (float)(tb / chw * 1.4426950408889634);
  }
}


// ---------------------------------------------------------------- ancestral sampling step + classifier-free guidance
// p_sample (gaussian_diffusion.py:376-421) over p_mean_variance (:254-332) for the LEARNED_RANGE / EPSILON model:
//   log_var = frac*log(beta_t) + (1-frac)*posterior_log_variance_clipped_t, frac = (v+1)/2
//   pred_x0 = sqrt_recip_ac*x - sqrt_recipm1_ac*eps   (clamped to [-1,1] when clip_denoised)
//   sample  = coef1*pred_x0 + coef2*x + (t != 0) * exp(0.5*log_var) * noise
// t indexes the (possibly respaced) table; the model was evaluated at timestep_map[t] by the host.
__global__ __launch_bounds__(TPB) void k_p_sample(const float* __restrict__ x, const float* __restrict__ out,
                                                  const int64_t* __restrict__ t, const float* __restrict__ tab,
                                                  const float* __restrict__ noise, int chw, int clip,
                                                  float* __restrict__ sample, float* __restrict__ pred) {
  const int n = blockIdx.y;
  const float* row = tab + (size_t)t[n] * 8;
  const float sra = row[2], srm1 = row[3], c1 = row[4], c2 = row[5], min_log = row[6], max_log = row[7];
  const float nz = t[n] != 0 ? 1.0f : 0.0f;
  const size_t xb = (size_t)n * chw, ob = (size_t)n * 2 * chw;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < chw; i += gridDim.x * TPB) {
    const float xv = x[xb + i], eps = out[ob + i], v = out[ob + chw + i];
    const float frac = (v + 1.0f) / 2.0f;
    const float lv = frac * max_log + (1.0f - frac) * min_log;
    float px = sra * xv - srm1 * eps;
    if (clip) px = fminf(fmaxf(px, -1.0f), 1.0f);
    const float mean = c1 * px + c2 * xv;
    sample[xb + i] = mean + (nz * expf(0.5f * lv)) * noise[xb + i];
    if (pred) pred[xb + i] = px;
  }
}

// forward_with_cfg's guidance mix (DiT/models.py:258-266), in place on model_out [2*half][cout][hw]:
// channels < n_guided of sample i and i+half both become uncond + s*(cond - uncond); the rest is untouched
__global__ __launch_bounds__(TPB) void k_cfg_combine(float* __restrict__ out, int half, int cout, int hw, int n_guided, float s) {
  const int n = blockIdx.y;
  const size_t cb = (size_t)n * cout * hw, ub = (size_t)(n + half) * cout * hw;
  const int tot = n_guided * hw;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < tot; i += gridDim.x * TPB) {
    const float c = out[cb + i], u = out[ub + i];
    const float h = u + s * (c - u);
    out[cb + i] = h; out[ub + i] = h;
  }
}

// ---------------------------------------------------------------- DDPM (CIFAR-10) epsilon loss
// Replaces /root/reference/DDPM/functions/losses.py:
//   noise_estimation_loss_conditional :22-38  a = (1-b).cumprod(0)[t]; x = x0*sqrt(a) + e*sqrt(1-a);
//                                             loss_i = sum_{chw} (e - model(x, t, c))^2   (a SUM over pixels)
//   adaptive_loss :49-69                      w_i = 1/(loss_i^lambd + 1e-8) (detached); mean_i(w_i/sum(w) * loss_i * N)
// alphas_cumprod follows torch's CPU cumprod: the running product is kept in double and rounded to fp32 per entry
// (bit-exact with the oracle; the reference's device scan may differ in the last ulp).
__global__ void k_ddpm_abar(const float* __restrict__ betas, int T, float* __restrict__ abar) {
  if (threadIdx.x || blockIdx.x) return;
  double a = 1.0;
  for (int t = 0; t < T; ++t) { a *= (double)(1.0f - betas[t]); abar[t] = (float)a; }
}

__global__ __launch_bounds__(TPB) void k_ddpm_q_sample(const float* __restrict__ x0, const float* __restrict__ e,
                                                       const int64_t* __restrict__ t, const float* __restrict__ abar,
                                                       int chw, float* __restrict__ xt) {
  const int n = blockIdx.y;
  const float a = abar[t[n]];
  // sqrt in double, rounded to fp32: equals the correctly rounded fp32 sqrt (53 >= 2*24 + 2 bits)
  const float sa = (float)sqrt((double)a), sb = (float)sqrt((double)(1.0f - a));
  const size_t base = (size_t)n * chw;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < chw; i += gridDim.x * TPB) xt[base + i] = x0[base + i] * sa + e[base + i] * sb;
}

// one workgroup per sample: per[n] = sum_{chw} (e - out)^2
__global__ __launch_bounds__(TPB) void k_ddpm_sample_loss(const float* __restrict__ e, const float* __restrict__ out, int chw,
                                                          float* __restrict__ per) {
  __shared__ double sh[TPB / 64];
  const size_t base = (size_t)blockIdx.x * chw;
  float acc = 0.f;
  for (int i = threadIdx.x; i < chw; i += TPB) { const float d = e[base + i] - out[base + i]; acc += d * d; }
  const double w = wave_sum_d((double)acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0;
    for (int i = 0; i < TPB / 64; ++i) s += sh[i];
    per[blockIdx.x] = (float)s;
  }
}

// single workgroup.  mode 0 ("simple", :38): loss = sum_local(per)/n_global, coef_i = 2*scale/n_global.
// mode 1 (adaga): w_i = 1/(per_i^lambd + 1e-8); W = sum w (local, or *wsum when use_wsum: the all-reduced global sum);
//                 loss = sum_local(w_i*per_i)/W, coef_i = 2*scale*w_i/W.      d loss / d out_i = coef_i * (out_i - e_i)
__global__ __launch_bounds__(TPB) void k_ddpm_loss_coef(const float* __restrict__ per, int n, int mode, float lambd, float scale,
                                                        int n_global, float* __restrict__ wsum, int use_wsum,
                                                        float* __restrict__ coef, float* __restrict__ loss) {
  __shared__ double sh[2][TPB / 64];
  double sw = 0, swl = 0;
  for (int i = threadIdx.x; i < n; i += TPB) {
    const float l = per[i];
    const float w = mode == 1 ? 1.0f / (powf(l, lambd) + 1e-8f) : 1.0f;
    coef[i] = w;
    sw += (double)w; swl += (double)w * (double)l;
  }
  sw = wave_sum_d(sw); swl = wave_sum_d(swl);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = sw; sh[1][threadIdx.x >> 6] = swl; }
  __syncthreads();
  double W = 0, WL = 0;
  for (int i = 0; i < TPB / 64; ++i) { W += sh[0][i]; WL += sh[1][i]; }
  if (mode == 1) {
    if (use_wsum) W = (double)*wsum;
    else if (threadIdx.x == 0) *wsum = (float)W;
  } else {
    W = (double)n_global;
  }
  __syncthreads();
  const float fW = (float)W;
  for (int i = threadIdx.x; i < n; i += TPB) coef[i] = 2.0f * scale * (coef[i] / fW);
  if (threadIdx.x == 0) *loss = (float)(WL / W);
}

__global__ __launch_bounds__(TPB) void k_ddpm_loss_bwd(const float* __restrict__ e, const float* __restrict__ out,
                                                       const float* __restrict__ coef, int chw, float* __restrict__ dout) {
  const int n = blockIdx.y;
  const float c = coef[n];
  const size_t base = (size_t)n * chw;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < chw; i += gridDim.x * TPB) dout[base + i] = c * (out[base + i] - e[base + i]);
}


// DDIM-style step of DDPM/functions/denoising.py:72-95 (one timestep for the whole batch, scalars from the host):
//   x0 = (x - eps*s1)/s2;  x_next = s3*x0 + c1*noise + c2*eps      with s1 = sqrt(1-a_t), s2 = sqrt(a_t), s3 = sqrt(a_next)
__global__ __launch_bounds__(TPB) void k_ddim_step(const float* __restrict__ x, const float* __restrict__ eps,
                                                   const float* __restrict__ noise, long n, float s1, float s2, float s3, float c1,
                                                   float c2, float* __restrict__ x_next, float* __restrict__ x0_pred) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
    const float e = eps[i];
    const float x0 = (x[i] - e * s1) / s2;
    const float nz = noise ? noise[i] : 0.0f;
    x_next[i] = (s3 * x0 + c1 * nz) + c2 * e;
    if (x0_pred) x0_pred[i] = x0;
  }
}

}  // namespace

extern "C" {

int sfron_q_sample(const float* x0, const float* noise, const int64_t* t, const float* tab, int n, int chw, float* x_t,
                   void* stream) {
  SFRON_CHECK_ARG(x0 && noise && t && tab && x_t && n > 0 && chw > 0);
  int gx = cdiv(chw, TPB);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_q_sample, dim3(gx, n), dim3(TPB), 0, (hipStream_t)stream, x0, noise, t, tab, chw, x_t);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_dit_loss_fwd_bwd(const float* x0, const float* noise, const float* model_out, const int64_t* t,
                           const float* tab, int n, int c, int hw, float grad_scale, float* mse, float* vb,
                           float* d_model_out, void* stream) {
  SFRON_CHECK_ARG(x0 && noise && model_out && t && tab && mse && vb && d_model_out && n > 0 && c > 0 && hw > 0);
  hipLaunchKernelGGL(k_dit_loss, dim3(n), dim3(TPB), 0, (hipStream_t)stream, x0, noise, model_out, t, tab, c, hw,
                     grad_scale, mse, vb, d_model_out);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_ddpm_alphas_cumprod(const float* betas, int T, float* abar, void* stream) {
  SFRON_CHECK_ARG(betas && abar && T > 0);
  hipLaunchKernelGGL(k_ddpm_abar, dim3(1), dim3(64), 0, (hipStream_t)stream, betas, T, abar);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_ddpm_q_sample(const float* x0, const float* e, const int64_t* t, const float* abar, int n, int chw, float* x_t,
                        void* stream) {
  SFRON_CHECK_ARG(x0 && e && t && abar && x_t && n > 0 && chw > 0);
  int gx = cdiv(chw, TPB);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_ddpm_q_sample, dim3(gx, n), dim3(TPB), 0, (hipStream_t)stream, x0, e, t, abar, chw, x_t);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_ddpm_sample_loss(const float* e, const float* model_out, int n, int chw, float* per_sample, void* stream) {
  SFRON_CHECK_ARG(e && model_out && per_sample && n > 0 && chw > 0);
  hipLaunchKernelGGL(k_ddpm_sample_loss, dim3(n), dim3(TPB), 0, (hipStream_t)stream, e, model_out, chw, per_sample);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_ddpm_loss_coef(const float* per_sample, int n, int mode, float lambd, float scale, int n_global, float* wsum,
                         int use_wsum, float* coef, float* loss, void* stream) {
  SFRON_CHECK_ARG(per_sample && coef && loss && n > 0 && n_global >= n && (mode == 0 || mode == 1));
  SFRON_CHECK_ARG(mode == 0 || wsum);
  hipLaunchKernelGGL(k_ddpm_loss_coef, dim3(1), dim3(TPB), 0, (hipStream_t)stream, per_sample, n, mode, lambd, scale, n_global,
                     wsum, use_wsum, coef, loss);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_ddpm_loss_bwd(const float* e, const float* model_out, const float* coef, int n, int chw, float* d_model_out,
                        void* stream) {
  SFRON_CHECK_ARG(e && model_out && coef && d_model_out && n > 0 && chw > 0);
  int gx = cdiv(chw, TPB);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_ddpm_loss_bwd, dim3(gx, n), dim3(TPB), 0, (hipStream_t)stream, e, model_out, coef, chw, d_model_out);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_p_sample(const float* x, const float* model_out, const int64_t* t, const float* tab, const float* noise, int n, int c,
                   int hw, int clip_denoised, float* sample, float* pred_xstart, void* stream) {
  SFRON_CHECK_ARG(x && model_out && t && tab && noise && sample && n > 0 && c > 0 && hw > 0);
  const int chw = c * hw;
  int gx = cdiv(chw, TPB);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_p_sample, dim3(gx, n), dim3(TPB), 0, (hipStream_t)stream, x, model_out, t, tab, noise, chw, clip_denoised,
                     sample, pred_xstart);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_cfg_combine(float* model_out, int n_total, int cout, int hw, int n_guided, float cfg_scale, void* stream) {
  SFRON_CHECK_ARG(model_out && n_total > 0 && n_total % 2 == 0 && cout > 0 && hw > 0 && n_guided > 0 && n_guided <= cout);
  int gx = cdiv(n_guided * hw, TPB);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_cfg_combine, dim3(gx, n_total / 2), dim3(TPB), 0, (hipStream_t)stream, model_out, n_total / 2, cout, hw,
                     n_guided, cfg_scale);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_ddim_step(const float* x, const float* eps, const float* noise, int64_t n, float s1, float s2, float s3, float c1,
                    float c2, float* x_next, float* x0_pred, void* stream) {
  SFRON_CHECK_ARG(x && eps && x_next && n > 0 && s2 != 0.0f);
  long gx = (n + TPB - 1) / TPB;
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(k_ddim_step, dim3((int)gx), dim3(TPB), 0, (hipStream_t)stream, x, eps, noise, (long)n, s1, s2, s3, c1, c2, x_next,
                     x0_pred);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

}  // extern "C"
