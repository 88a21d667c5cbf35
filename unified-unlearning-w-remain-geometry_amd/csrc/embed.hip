// Small conditioning / layout kernels of the DiT denoiser (negligible FLOPs, HBM / latency bound).
//
// Replaces the PyTorch op sequences of /root/reference/DiT/models.py:
//   TimestepEmbedder.timestep_embedding .... :41-59 (cos || sin, freqs = exp(-ln(1e4) * i / half))
//   nn.SiLU inside t_embedder / adaLN ...... :34,114,132
//   LabelEmbedder (+ token_drop) ........... :78-94  (the drop mask is an explicit input)
//   c = t + y ............................... :243
//   PatchEmbed's im2col (Conv2d k = s = p) . :169,240 (timm PatchEmbed: flatten(2).transpose(1,2))
//   unpatchify .............................. :218-231 ('nhwpqc->nchpwq')
#include "common.h"
#include "../../include/sfron.h"

namespace {

constexpr int TPB = 256;

__global__ __launch_bounds__(TPB) void k_timestep_embed(const int64_t* __restrict__ t, int n, int dim, __bf16* __restrict__ out,
                                                        int ld) {
  const int half = dim / 2;
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n * half) return;
  const int b = i / half, j = i % half;
  const float freq = expf(-9.210340371976184f * (float)j / (float)half);    // -log(10000) * j / half, fp32 like torch
  const float arg = (float)t[b] * freq;
  out[(size_t)b * ld + j] = f2bf(cosf(arg));
  out[(size_t)b * ld + half + j] = f2bf(sinf(arg));
}

// y = silu(x): writes bf16 (GEMM operand) and optionally keeps nothing else; x fp32 [n]
__global__ __launch_bounds__(TPB) void k_silu_fwd(const float* __restrict__ x, int64_t n, __bf16* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) y[i] = f2bf(silu(x[i]));
}
// dx = dy * silu'(x);  dy fp32, out bf16 (next GEMM operand) and/or fp32
__global__ __launch_bounds__(TPB) void k_silu_bwd(const float* __restrict__ dy, const float* __restrict__ x, int64_t n,
                                                  __bf16* __restrict__ dx_bf16, float* __restrict__ dx_f32) {
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const float g = dy[i] * silu_grad(x[i]);
    if (dx_bf16) dx_bf16[i] = f2bf(g);
    if (dx_f32) dx_f32[i] = g;
  }
}

// c[b] = t_emb[b] + table[drop[b] ? num_classes : y[b]];  silu_c = bf16(silu(c))
__global__ __launch_bounds__(TPB) void k_cond_fwd(const float* __restrict__ t_emb, const float* __restrict__ table,
                                                  const int64_t* __restrict__ y, const uint8_t* __restrict__ drop,
                                                  int num_classes, int D, float* __restrict__ c, __bf16* __restrict__ silu_c) {
  const int b = blockIdx.y;
  const int col = blockIdx.x * TPB + threadIdx.x;
  if (col >= D) return;
  const int64_t yb = y[b];
  // labels outside [0, num_classes) never index the table (the caller's step guard reports them, as the reference's
  // nn.Embedding would raise): they read / accumulate the null-class row
  const int64_t lab = ((drop && drop[b]) || yb < 0 || yb >= num_classes) ? num_classes : yb;
  const float v = t_emb[(size_t)b * D + col] + table[(size_t)lab * D + col];
  c[(size_t)b * D + col] = v;
  silu_c[(size_t)b * D + col] = f2bf(silu(v));
}

// d_c = d_silu_c * silu'(c); d_table[label] += d_c (serial over the batch per column: deterministic);
// d_c is also the gradient of t_emb.
// Round 5: the batch rows of a column used to be one dependent read-modify-write chain through the table row in global memory (38 us for 32
// rows on the critical stream).  Now a thread requests all rows of its column (chunks of CB rows) before the first use, keeps the d_c values
// in registers and gives every DISTINCT label of the chunk one load + adds in batch order + one store: the same sums in the same order (same
// bits), a few independent memory latencies instead of n dependent ones.
constexpr int CB = 32;
__global__ __launch_bounds__(TPB) void k_cond_bwd(const float* __restrict__ d_silu_c, const float* __restrict__ c,
                                                  const int64_t* __restrict__ y, const uint8_t* __restrict__ drop,
                                                  int num_classes, int n, int D, float* __restrict__ d_c,
                                                  float* __restrict__ d_table) {
  __shared__ int s_lab[CB];
  const int col = blockIdx.x * TPB + threadIdx.x;
  for (int b0 = 0; b0 < n; b0 += CB) {
    const int nb = min(CB, n - b0);
    __syncthreads();
    if (threadIdx.x < nb) {
      const int b = b0 + threadIdx.x;
      const int64_t yb = y[b];
      // labels outside [0, num_classes) never index the table (the caller's step guard reports them, as the reference's
      // nn.Embedding would raise): they read / accumulate the null-class row
      s_lab[threadIdx.x] = ((drop && drop[b]) || yb < 0 || yb >= num_classes) ? num_classes : (int)yb;
    }
    __syncthreads();
    if (col >= D) continue;
    float dv[CB], cv[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int b = b0 + (i < nb ? i : 0);
      dv[i] = d_silu_c[(size_t)b * D + col];
      cv[i] = c[(size_t)b * D + col];
    }
    float g[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      g[i] = dv[i] * silu_grad(cv[i]);
      if (i < nb) d_c[(size_t)(b0 + i) * D + col] = g[i];
    }
    // every distinct label's table element is requested before the first store (a store to one label's row would otherwise order the next
    // label's load behind it: the compiler cannot know the rows differ)
    float tv[CB];
    bool first[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int lab = s_lab[i < nb ? i : 0];
      first[i] = i < nb;                                                    // block-uniform
#pragma unroll
      for (int j = 0; j < i; ++j) first[i] = first[i] && s_lab[j] != lab;
      tv[i] = first[i] ? d_table[(size_t)lab * D + col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      if (first[i]) {
        const int lab = s_lab[i];
        float v = tv[i];
#pragma unroll
        for (int j = i; j < CB; ++j)
          if (j < nb && s_lab[j] == lab) v += g[j];
        d_table[(size_t)lab * D + col] = v;
      }
    }
  }
}

// image [n][C][H][W] fp32 -> token rows [n*T][C*p*p] bf16.
// chan_last = 0: column k = c*p*p + ph*p + pw   (Conv2d weight.view(D, -1) order, input patchify)
// chan_last = 1: column k = (ph*p + pw)*C + c   (unpatchify's 'nhwpqc' order, used for d_out -> d_tokens)
__global__ __launch_bounds__(TPB) void k_patchify(const float* __restrict__ img, int n, int C, int H, int W, int p,
                                                  int chan_last, __bf16* __restrict__ rows, int ld) {
  const int gw = W / p, gh = H / p, T = gw * gh, K = C * p * p;
  const int64_t total = (int64_t)n * T * K;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
    const int k = (int)(i % K);
    const int64_t r = i / K;
    const int tok = (int)(r % T), b = (int)(r / T);
    const int th = tok / gw, tw = tok % gw;
    int c, ph, pw;
    if (chan_last) { c = k % C; const int pq = k / C; ph = pq / p; pw = pq % p; }
    else { c = k / (p * p); const int pq = k % (p * p); ph = pq / p; pw = pq % p; }
    rows[(size_t)r * ld + k] = f2bf(img[(((size_t)b * C + c) * H + th * p + ph) * W + tw * p + pw]);
  }
}

// token rows [n*T][p*p*C] fp32 (chan_last order) -> image [n][C][H][W] fp32
__global__ __launch_bounds__(TPB) void k_unpatchify(const float* __restrict__ rows, int ld, int n, int C, int H, int W, int p,
                                                    float* __restrict__ img) {
  const int gw = W / p, gh = H / p, T = gw * gh;
  const int64_t total = (int64_t)n * C * H * W;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
    const int w = (int)(i % W), h = (int)((i / W) % H), c = (int)((i / ((int64_t)W * H)) % C), b = (int)(i / ((int64_t)W * H * C));
    const int th = h / p, ph = h % p, tw = w / p, pw = w % p;
    img[i] = rows[((size_t)b * T + th * gw + tw) * ld + (ph * p + pw) * C + c];
  }
}

// latents = (mean + exp(0.5 * clamp(logvar, -30, 20)) * eps) * scale: DiagonalGaussianDistribution.sample() of the diffusers
// AutoencoderKL posterior followed by .mul_(0.18215) (DiT/forget.py:265-267,305-307); moments [n][2C][hw] = mean || logvar
__global__ __launch_bounds__(TPB) void k_latent_sample(const float* __restrict__ moments, const float* __restrict__ eps, int n, int c, int hw,
                                                       float scale, float* __restrict__ out) {
  const int64_t total = (int64_t)n * c * hw;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
    const int64_t b = i / ((int64_t)c * hw), r = i - b * c * hw;
    const float mean = moments[b * 2 * c * hw + r];
    float lv = moments[b * 2 * c * hw + (int64_t)c * hw + r];
    lv = fminf(fmaxf(lv, -30.0f), 20.0f);
    out[i] = (mean + expf(0.5f * lv) * eps[i]) * scale;
  }
}

inline int grid_for(int64_t n) {
  int64_t b = (n + TPB - 1) / TPB;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// ---- step guard (host side: guard.py): the fail-loud checks of one SFR-on iteration as two tiny launches instead of ~40 torch ops.
// Inputs of a pass: labels / timesteps clamped into range (the kernels downstream never index out of bounds) and the number of
// clamped entries added to flags[2] (labels) / flags[3] (timesteps) -- the reference raises an IndexError there
// (DiT/models.py:89-93, gaussian_diffusion.py:861-873); the host raises SfronError at its next poll.
__global__ __launch_bounds__(TPB) void k_guard_inputs(const int64_t* __restrict__ y, const int64_t* __restrict__ t, int n, int num_classes,
                                                      int num_timesteps, int64_t* __restrict__ y_safe, int64_t* __restrict__ t_safe,
                                                      float* __restrict__ flags) {
  __shared__ int bad[2];
  if (threadIdx.x < 2) bad[threadIdx.x] = 0;
  __syncthreads();
  int by = 0, bt = 0;
  for (int i = threadIdx.x; i < n; i += TPB) {
    const int64_t yy = y[i], tt = t[i];
    const int64_t yc = yy < 0 ? 0 : (yy >= num_classes ? num_classes - 1 : yy);
    const int64_t tc = tt < 0 ? 0 : (tt >= num_timesteps ? num_timesteps - 1 : tt);
    by += yc != yy; bt += tc != tt;
    y_safe[i] = yc; t_safe[i] = tc;
  }
  if (by) atomicAdd(&bad[0], by);       // integer LDS atomics: order-independent
  if (bt) atomicAdd(&bad[1], bt);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (bad[0]) flags[2] += (float)bad[0];
    if (bad[1]) flags[3] += (float)bad[1];
  }
}
// flags[0] += 1 if any of the per-sample loss terms is NaN / Inf, flags[1] += 1 if the gradient norm (stats[0]) is
__global__ __launch_bounds__(TPB) void k_guard_finite(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                      const float* __restrict__ d, int n, const float* __restrict__ stats,
                                                      float* __restrict__ flags) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  int nb = 0;
  for (int i = threadIdx.x; i < n; i += TPB) {
    nb += !isfinite(a[i]);
    if (b) nb += !isfinite(b[i]);
    if (c) nb += !isfinite(c[i]);
    if (d) nb += !isfinite(d[i]);
  }
  if (nb) atomicAdd(&bad, nb);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (bad) flags[0] += 1.0f;
    if (stats && !isfinite(stats[0])) flags[1] += 1.0f;
  }
}

}  // namespace

extern "C" {

int sfron_latent_sample(const float* moments, const float* eps, int n, int c, int hw, float scale, float* out, void* stream) {
  SFRON_CHECK_ARG(moments && eps && out && n > 0 && c > 0 && hw > 0);
  hipLaunchKernelGGL(k_latent_sample, dim3(grid_for((int64_t)n * c * hw)), dim3(TPB), 0, (hipStream_t)stream, moments, eps, n, c, hw, scale, out);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_timestep_embed(const int64_t* t, int n, int dim, uint16_t* out, int ld, void* stream) {
  SFRON_CHECK_ARG(t && out && n > 0 && dim > 0 && dim % 2 == 0 && ld >= dim);
  hipLaunchKernelGGL(k_timestep_embed, dim3(cdiv((long)n * (dim / 2), TPB)), dim3(TPB), 0, (hipStream_t)stream, t, n, dim,
                     (__bf16*)out, ld);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_silu_fwd(const float* x, int64_t n, uint16_t* y_bf16, void* stream) {
  SFRON_CHECK_ARG(x && y_bf16 && n > 0);
  hipLaunchKernelGGL(k_silu_fwd, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, x, n, (__bf16*)y_bf16);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_silu_bwd(const float* dy, const float* x, int64_t n, uint16_t* dx_bf16, float* dx_f32, void* stream) {
  SFRON_CHECK_ARG(dy && x && n > 0 && (dx_bf16 || dx_f32));
  hipLaunchKernelGGL(k_silu_bwd, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, dy, x, n, (__bf16*)dx_bf16, dx_f32);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_cond_fwd(const float* t_emb, const float* table, const int64_t* y, const uint8_t* drop, int num_classes, int n,
                   int D, float* c, uint16_t* silu_c, void* stream) {
  SFRON_CHECK_ARG(t_emb && table && y && c && silu_c && n > 0 && D > 0 && num_classes > 0);
  hipLaunchKernelGGL(k_cond_fwd, dim3(cdiv(D, TPB), n), dim3(TPB), 0, (hipStream_t)stream, t_emb, table, y, drop, num_classes,
                     D, c, (__bf16*)silu_c);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_cond_bwd(const float* d_silu_c, const float* c, const int64_t* y, const uint8_t* drop, int num_classes, int n,
                   int D, float* d_c, float* d_table, void* stream) {
  SFRON_CHECK_ARG(d_silu_c && c && y && d_c && d_table && n > 0 && D > 0);
  hipLaunchKernelGGL(k_cond_bwd, dim3(cdiv(D, TPB)), dim3(TPB), 0, (hipStream_t)stream, d_silu_c, c, y, drop, num_classes, n,
                     D, d_c, d_table);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_patchify(const float* img, int n, int C, int H, int W, int p, int chan_last, uint16_t* rows, int ld, void* stream) {
  SFRON_CHECK_ARG(img && rows && n > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0 && ld >= C * p * p);
  hipLaunchKernelGGL(k_patchify, dim3(grid_for((int64_t)n * C * H * W)), dim3(TPB), 0, (hipStream_t)stream, img, n, C, H, W, p,
                     chan_last, (__bf16*)rows, ld);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_unpatchify(const float* rows, int ld, int n, int C, int H, int W, int p, float* img, void* stream) {
  SFRON_CHECK_ARG(img && rows && n > 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0 && ld >= C * p * p);
  hipLaunchKernelGGL(k_unpatchify, dim3(grid_for((int64_t)n * C * H * W)), dim3(TPB), 0, (hipStream_t)stream, rows, ld, n, C, H,
                     W, p, img);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_guard_inputs(const int64_t* y, const int64_t* t, int n, int num_classes, int num_timesteps, int64_t* y_safe, int64_t* t_safe,
                       float* flags, void* stream) {
  SFRON_CHECK_ARG(y && t && y_safe && t_safe && flags && n > 0 && num_classes > 0 && num_timesteps > 0);
  hipLaunchKernelGGL(k_guard_inputs, dim3(1), dim3(TPB), 0, (hipStream_t)stream, y, t, n, num_classes, num_timesteps, y_safe, t_safe, flags);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_guard_finite(const float* a, const float* b, const float* c, const float* d, int n, const float* stats, float* flags, void* stream) {
  SFRON_CHECK_ARG(a && flags && n > 0);
  hipLaunchKernelGGL(k_guard_finite, dim3(1), dim3(TPB), 0, (hipStream_t)stream, a, b, c, d, n, stats, flags);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

}  // extern "C"
