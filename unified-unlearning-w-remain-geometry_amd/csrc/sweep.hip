// Parameter-side sweep of one SFR-on stage over flat fp32 arenas (HBM-bound kernels).
//
// Replaces the per-tensor PyTorch op sequences of the reference:
//   mask ⊙ grad ............ DiT/forget.py:289-292, DDPM/runners/diffusion.py:1126-1129
//   clip_grad_norm_ ........ DiT/forget.py:293-298, DDPM/runners/diffusion.py:1131-1136,1169-1174
//   Adam / AdamW .step() ... DiT/forget.py:199,299,320, DDPM/functions/__init__.py:9-18
//   update_ema / EMAHelper . DiT/forget.py:52-62, DDPM/models/ema.py:17-24
//   Fisher, saliency mask .. DiT/generate_fisher.py:236-239, DiT/generate_mask.py:34-35
//
// Layout: every parameter tensor of the model lives at a fixed offset of ONE flat fp32 arena;
// grad / exp_avg / exp_avg_sq / ema / byte-mask / bf16-shadow arenas share the same offsets, so a
// stage is two streaming launches (norm pre-pass, fused update) instead of ~10 per tensor.
// Arithmetic follows torch's single-tensor Adam: m = lerp(m, g, 1-b1); v = v*b2 + (1-b2)*g*g;
// p += -step_size * m / (sqrt(v)/bc2_sqrt + eps).  A masked-out element takes g = 0 (it is NOT
// skipped: momentum still moves it -- SURVEY.md section 9 Q7).
#include "common.h"
#include "../../include/sfron.h"

// gemm.hip (C++ linkage, not part of the C ABI): masked sum of squares of A^T B, one fp64 partial per 128 x 128 tile
int gemm_sumsq_lowrank(const uint16_t* a, const uint16_t* b, int R, int NM, int D, const uint8_t* mask, double* partials, int* nblk, void* stream);

namespace {

constexpr int TPB = 256;
constexpr int MAX_GRID = 2048;

__global__ __launch_bounds__(TPB) void k_sumsq_masked(const float* __restrict__ g, const float* __restrict__ g2,
                                                       const uint8_t* __restrict__ mask, int64_t n, double* __restrict__ partials) {
  __shared__ double sh[TPB / 64];
  const int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const uchar4* m4 = reinterpret_cast<const uchar4*>(mask);
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
    float4 x = g4[i];
    if (g2) { const float4 y = reinterpret_cast<const float4*>(g2)[i]; x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w; }
    if (mask) {
      uchar4 mk = m4[i];
      x.x = mk.x ? x.x : 0.f; x.y = mk.y ? x.y : 0.f; x.z = mk.z ? x.z : 0.f; x.w = mk.w ? x.w : 0.f;
    }
    acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = (n4 << 2) + threadIdx.x;
    float x = g[i];
    if (g2) x += g2[i];
    if (mask && !mask[i]) x = 0.f;
    acc += x * x;
  }
  double d = wave_sum_d((double)acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < TPB / 64; ++i) t += sh[i];
    partials[blockIdx.x] = t;
  }
}

// stats[0] = total L2 norm, stats[1] = clip coefficient min(1, max_norm/(norm+1e-6)), stats[2] = sum of squares
// The same partial sums over a TABLE of element ranges of the arenas (tab[r] = {offset, length}, multiples of 4, DEVICE memory): one workgroup and one
// partial per range.  For what the clip norm still has to read once the weight-gradient GEMMs leave their own masked sums of squares
// (sfron_gemm_desc.sumsq_partials): biases, embedders, the final layer -- a hundred small ranges in ONE launch.  Ranges of up to ~64 K elements
// (the host splits longer tensors).
__global__ __launch_bounds__(TPB) void k_sumsq_ranges(const float* __restrict__ g, const uint8_t* __restrict__ mask, const long long* __restrict__ tab,
                                                       double* __restrict__ partials) {
  __shared__ double sh[TPB / 64];
  const long long off = tab[2 * blockIdx.x], n4 = tab[2 * blockIdx.x + 1] >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g + off);
  const uchar4* m4 = reinterpret_cast<const uchar4*>(mask ? mask + off : nullptr);
  float acc = 0.f;
  for (long long i = threadIdx.x; i < n4; i += TPB) {
    float4 x = g4[i];
    if (mask) {
      const uchar4 mk = m4[i];
      x.x = mk.x ? x.x : 0.f; x.y = mk.y ? x.y : 0.f; x.z = mk.z ? x.z : 0.f; x.w = mk.w ? x.w : 0.f;
    }
    acc += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
  }
  const double d = wave_sum_d((double)acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < TPB / 64; ++i) t += sh[i];
    partials[blockIdx.x] = t;
  }
}

// 1024 threads, four independent partial sums per thread (the DiT-XL/2 forget stage hands over ~26 000 partials: 24 480 of the rank-(batch)
// range + the flat ranges'; 256 threads walking them with one dependent add per load took 29 us on the critical stream), fixed order.
constexpr int CC_TPB = 1024;
__global__ __launch_bounds__(CC_TPB) void k_clip_coef(const double* __restrict__ partials, int nblk, float max_norm,
                                                      float* __restrict__ stats) {
  __shared__ double sh[CC_TPB / 64];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int i = threadIdx.x;
  for (; i + 3 * CC_TPB < nblk; i += 4 * CC_TPB) {
    const double v0 = partials[i], v1 = partials[i + CC_TPB], v2 = partials[i + 2 * CC_TPB], v3 = partials[i + 3 * CC_TPB];
    a0 += v0; a1 += v1; a2 += v2; a3 += v3;
  }
  for (; i < nblk; i += CC_TPB) a0 += partials[i];
  double acc = wave_sum_d((a0 + a1) + (a2 + a3));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < CC_TPB / 64; ++w) t += sh[w];
    float norm = (float)sqrt(t);
    float coef = max_norm / (norm + 1e-6f);          // torch: clip_coef = max_norm / (total_norm + 1e-6)
    coef = coef > 1.0f ? 1.0f : coef;                // torch.clamp(clip_coef, max=1.0)
    stats[0] = norm; stats[1] = coef; stats[2] = (float)t;
  }
}

struct AdamArgs {
  float w1, beta2, w2, eps, step_size, bc2_sqrt, decay_mul;   // w1 = float(1-beta1), w2 = float(1-beta2) (from double, like torch)
  float ema_decay, ema_w;   // ema_w = float(1 - ema_decay)
  int ema_mode;      // 0 none, 1 DiT form ema*d + (1-d)*p, 2 DDPM form (1-mu)*p + mu*shadow (ema_decay = mu)
};

__device__ __forceinline__ float adam_one(float p, float g, float& m, float& v, const AdamArgs& a) {
  p = p * a.decay_mul;
  m = m + a.w1 * (g - m);                                   // exp_avg.lerp_(grad, 1-beta1), weight < 0.5 branch
  v = v * a.beta2 + (a.w2 * g) * g;                         // exp_avg_sq.mul_(b2).addcmul_(g, g, value=1-b2)
  float denom = sqrtf(v) / a.bc2_sqrt + a.eps;              // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
  return p + (-a.step_size * m) / denom;                    // param.addcdiv_(exp_avg, denom, value=-step_size)
}
__device__ __forceinline__ float ema_one(float e, float p, const AdamArgs& a) {
  if (a.ema_mode == 1) return e * a.ema_decay + a.ema_w * p;
  return a.ema_w * p + a.ema_decay * e;
}

__device__ __forceinline__ uint32_t pack4_e4m3(float a, float b, float c, float d) {      // OCP e4m3fn, RNE, saturating (as csrc/fp8.hip)
  const float mx = 448.0f;
  int r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(a, -mx), mx), fminf(fmaxf(b, -mx), mx), r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(c, -mx), mx), fminf(fmaxf(d, -mx), mx), r, true);
  return (uint32_t)r;
}

// w8 / w8_scale (optional, config 5): the e4m3 shadow of the NEW p, quantised with the device scalar *w8_scale -- the range is one
// weight tensor (+ its bias, whose e4m3 bytes nobody reads), so the re-quantisation pass over the masters disappears
// The sweep touches every word of its arenas exactly once, most of it beside the next forward pass: nontemporal (streaming) accesses keep
// it from evicting the GEMMs' operand panels from the XCDs' L2s (same-box A-B in the step: 63.2 / 63.3 -> 62.4 / 62.7 ms; alone the kernel
// streams at the same rate either way).  The bf16 / e4m3 shadow stores stay ordinary: the next forward pass reads them.
__device__ __forceinline__ float4 LD4(const float* b, int64_t i) {
  const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(b) + i); return make_float4(t[0], t[1], t[2], t[3]); }
__device__ __forceinline__ void ST4(float* b, int64_t i, float4 x) {
  __builtin_nontemporal_store(f32x4{x.x, x.y, x.z, x.w}, reinterpret_cast<f32x4*>(b) + i); }
__global__ __launch_bounds__(TPB) void k_masked_clip_adam(float* __restrict__ p, const float* __restrict__ g,
                                                          const float* __restrict__ g2, float* __restrict__ m, float* __restrict__ v,
                                                          const uint8_t* __restrict__ mask, const float* __restrict__ stats,
                                                          int64_t n, AdamArgs a, uint16_t* __restrict__ wbf,
                                                          float* __restrict__ ema, uint8_t* __restrict__ w8 = nullptr,
                                                          const float* __restrict__ w8_scale = nullptr) {
  const float coef = stats ? stats[1] : 1.0f;
  const float s8 = w8 ? *w8_scale : 1.0f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
    float4 pp = LD4(p, i);
    float4 gg = LD4(g, i);
    if (g2) { const float4 y = LD4(g2, i); gg.x += y.x; gg.y += y.y; gg.z += y.z; gg.w += y.w; }
    float4 mm = LD4(m, i);
    float4 vv = LD4(v, i);
    if (mask) {
      const uint32_t mk = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(mask) + i);
      gg.x = (mk & 0xffu) ? gg.x : 0.f; gg.y = (mk & 0xff00u) ? gg.y : 0.f; gg.z = (mk & 0xff0000u) ? gg.z : 0.f; gg.w = (mk & 0xff000000u) ? gg.w : 0.f;
    }
    gg.x *= coef; gg.y *= coef; gg.z *= coef; gg.w *= coef;
    pp.x = adam_one(pp.x, gg.x, mm.x, vv.x, a);
    pp.y = adam_one(pp.y, gg.y, mm.y, vv.y, a);
    pp.z = adam_one(pp.z, gg.z, mm.z, vv.z, a);
    pp.w = adam_one(pp.w, gg.w, mm.w, vv.w, a);
    ST4(p, i, pp);
    ST4(m, i, mm);
    ST4(v, i, vv);
    if (wbf) {
      bf16x4 b = {f2bf(pp.x), f2bf(pp.y), f2bf(pp.z), f2bf(pp.w)};
      reinterpret_cast<bf16x4*>(wbf)[i] = b;
    }
    if (w8) reinterpret_cast<uint32_t*>(w8)[i] = pack4_e4m3(pp.x * s8, pp.y * s8, pp.z * s8, pp.w * s8);
    if (a.ema_mode) {
      float4 ee = LD4(ema, i);
      ee.x = ema_one(ee.x, pp.x, a); ee.y = ema_one(ee.y, pp.y, a);
      ee.z = ema_one(ee.z, pp.z, a); ee.w = ema_one(ee.w, pp.w, a);
      ST4(ema, i, ee);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = (n4 << 2) + threadIdx.x;
    float gg = g[i];
    if (g2) gg += g2[i];
    if (mask && !mask[i]) gg = 0.f;
    gg *= coef;
    float mm = m[i], vv = v[i];
    float pp = adam_one(p[i], gg, mm, vv, a);
    p[i] = pp; m[i] = mm; v[i] = vv;
    if (wbf) reinterpret_cast<__bf16*>(wbf)[i] = f2bf(pp);
    if (a.ema_mode) ema[i] = ema_one(ema[i], pp, a);
  }
}

// ---- rank-R gradient: the adaLN_modulation weight of ALL blocks, W [NM][D] (a third of DiT-XL/2's parameters), has the gradient
// dW[n][k] = sum_{b < R} dmod[b][n] * sc[b][k] (R = batch rows; bf16 factors, fp32 accumulation -- the same numbers the
// weight-gradient GEMM would form from the same factors).  Forming it HERE, inside the sweep, removes the 892 MB fp32 write of that
// GEMM and the sweep's read of it (and, in the forget stage, the norm pre-pass's).  One thread owns 4 consecutive k of LR_ROWS (8; 16 measured slower) rows:
// per b one 8-byte load of sc and one 16-byte wave-uniform load of dmod feed LR_ROWS x 4 FMAs; b runs in index order (deterministic).
constexpr int LR_ROWS = 8;
__device__ __forceinline__ void lowrank_grad(const __bf16* __restrict__ dmod, const __bf16* __restrict__ sc, int R, int NM, int D, int n0,
                                             int c, f32x4 (&acc)[LR_ROWS]) {
#pragma unroll
  for (int r = 0; r < LR_ROWS; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
  // eight batch rows per trip: all sixteen loads of a trip are issued before its 256 FMAs (a load-use loop pays one L2 latency per row)
  for (int b0 = 0; b0 < R; b0 += 8) {
    bf16x4 s4[8];
    bf16x8 d8[8][LR_ROWS / 8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + u < R ? b0 + u : R - 1;                       // clamped: the tail's surplus rows get weight 0 below
      s4[u] = *reinterpret_cast<const bf16x4*>(sc + (size_t)b * D + 4 * c);
#pragma unroll
      for (int h = 0; h < LR_ROWS / 8; ++h)
        d8[u][h] = *reinterpret_cast<const bf16x8*>(dmod + (size_t)b * NM + n0 + 8 * h);    // wave-uniform address: scalar loads
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float on = b0 + u < R ? 1.0f : 0.0f;
      const f32x4 sv = {bf2f(s4[u][0]) * on, bf2f(s4[u][1]) * on, bf2f(s4[u][2]) * on, bf2f(s4[u][3]) * on};
#pragma unroll
      for (int r = 0; r < LR_ROWS; ++r) {           // explicit FMAs: the library is built with -ffp-contract=off (a * b + c would be two instructions)
        const float dr = bf2f(d8[u][r >> 3][r & 7]);
        acc[r] = __builtin_elementwise_fma(sv, f32x4{dr, dr, dr, dr}, acc[r]);
      }
    }
  }
}

// partials[blockIdx.x] = sum over this workgroup's LR_ROWS rows of (mask ? g : 0)^2
__global__ void k_sumsq_lowrank(const __bf16* __restrict__ dmod, const __bf16* __restrict__ sc, int R, int NM, int D,
                                const uint8_t* __restrict__ mask, double* __restrict__ partials) {
  __shared__ double sh[16];
  const int c = threadIdx.x, n0 = blockIdx.x * LR_ROWS;
  float a = 0.f;
  if (c < (D >> 2)) {
    f32x4 g[LR_ROWS];
    lowrank_grad(dmod, sc, R, NM, D, n0, c, g);
#pragma unroll
    for (int r = 0; r < LR_ROWS; ++r) {
      f32x4 x = g[r];
      if (mask) {
        const uchar4 mk = reinterpret_cast<const uchar4*>(mask + (size_t)(n0 + r) * D)[c];
        x[0] = mk.x ? x[0] : 0.f; x[1] = mk.y ? x[1] : 0.f; x[2] = mk.z ? x[2] : 0.f; x[3] = mk.w ? x[3] : 0.f;
      }
      a += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    }
  }
  double d = wave_sum_d((double)a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    partials[blockIdx.x] = t;
  }
}

__global__ void k_adam_lowrank(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const uint8_t* __restrict__ mask,
                               const float* __restrict__ stats, const __bf16* __restrict__ dmod, const __bf16* __restrict__ sc, int R,
                               int NM, int D, AdamArgs a, uint16_t* __restrict__ wbf, float* __restrict__ ema) {
  const int c = threadIdx.x, n0 = blockIdx.x * LR_ROWS;
  if (c >= (D >> 2)) return;
  const float coef = stats ? stats[1] : 1.0f;
  f32x4 g[LR_ROWS];
  lowrank_grad(dmod, sc, R, NM, D, n0, c, g);
#pragma unroll
  for (int r = 0; r < LR_ROWS; ++r) {
    const size_t i = (size_t)(n0 + r) * (D >> 2) + c;               // float4 index inside W
    float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    f32x4 gg = g[r];
    if (mask) {
      const uchar4 mk = reinterpret_cast<const uchar4*>(mask)[i];
      gg[0] = mk.x ? gg[0] : 0.f; gg[1] = mk.y ? gg[1] : 0.f; gg[2] = mk.z ? gg[2] : 0.f; gg[3] = mk.w ? gg[3] : 0.f;
    }
    gg = gg * coef;
    pp.x = adam_one(pp.x, gg[0], mm.x, vv.x, a); pp.y = adam_one(pp.y, gg[1], mm.y, vv.y, a);
    pp.z = adam_one(pp.z, gg[2], mm.z, vv.z, a); pp.w = adam_one(pp.w, gg[3], mm.w, vv.w, a);
    reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
    if (wbf) reinterpret_cast<bf16x4*>(wbf)[i] = bf16x4{f2bf(pp.x), f2bf(pp.y), f2bf(pp.z), f2bf(pp.w)};
    if (a.ema_mode) {
      float4 ee = reinterpret_cast<float4*>(ema)[i];
      ee.x = ema_one(ee.x, pp.x, a); ee.y = ema_one(ee.y, pp.y, a); ee.z = ema_one(ee.z, pp.z, a); ee.w = ema_one(ee.w, pp.w, a);
      reinterpret_cast<float4*>(ema)[i] = ee;
    }
  }
}

__global__ __launch_bounds__(TPB) void k_ema(float* __restrict__ ema, const float* __restrict__ p, int64_t n, AdamArgs a) {
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
    ema[i] = ema_one(ema[i], p[i], a);
}

__global__ __launch_bounds__(TPB) void k_fisher_accum(float* __restrict__ F, const float* __restrict__ g, int64_t n, float n_iters) {
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    float x = g[i];
    F[i] = F[i] + (x * x) / n_iters;             // forget_gradients[name] += (grad**2) / n_iters
  }
}

// the DDPM variant squares the CLIPPED gradient (DDPM/runners/diffusion.py:1271-1281: clip_grad_norm_ runs before grad**2):
// g * stats[1] with stats from sfron_clip_coef; g2 (optional) is a second gradient arena added first (the two guidance
// branches of the mode="test" forward, back-propagated one after the other)
__global__ __launch_bounds__(TPB) void k_fisher_accum_clipped(float* __restrict__ F, const float* __restrict__ g,
                                                              const float* __restrict__ g2, const float* __restrict__ stats, int64_t n,
                                                              float n_iters) {
  const float c = stats ? stats[1] : 1.0f;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    float x = g[i];
    if (g2) x += g2[i];
    x *= c;                                       // clip_grad_norm_: grad.mul_(clip_coef), then squared
    F[i] = F[i] + (x * x) / n_iters;
  }
}

__global__ __launch_bounds__(TPB) void k_mask_from_fisher(const float* __restrict__ ff, const float* __restrict__ rf,
                                                          int64_t n, float th, uint8_t* __restrict__ mask) {
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    // IEEE fp32 add / divide / compare, no fast-math: bit-exact with torch ((F_f+1e-15)/(F_r+1e-15)) >= th
    float a = __fadd_rn(ff[i], 1e-15f);
    float b = __fadd_rn(rf[i], 1e-15f);
    float q = __fdiv_rn(a, b);
    mask[i] = (q >= th) ? 1 : 0;
  }
}

__global__ __launch_bounds__(TPB) void k_cast_bf16(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
    float4 x = reinterpret_cast<const float4*>(src)[i];
    bf16x4 b = {f2bf(x.x), f2bf(x.y), f2bf(x.z), f2bf(x.w)};
    reinterpret_cast<bf16x4*>(dst)[i] = b;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = (n4 << 2) + threadIdx.x;
    reinterpret_cast<__bf16*>(dst)[i] = f2bf(src[i]);
  }
}

inline int grid_for(int64_t n_items) {
  int64_t b = (n_items + TPB - 1) / TPB;
  if (b < 1) b = 1;
  return (int)(b > MAX_GRID ? MAX_GRID : b);
}

}  // namespace

extern "C" {

int sfron_sweep_partials_len(void) { return MAX_GRID; }

int sfron_sumsq_masked(const float* g, const float* g2, const uint8_t* mask, int64_t n, double* partials, int* nblk_out,
                       void* stream) {
  SFRON_CHECK_ARG(g && partials && nblk_out && n >= 0);
  SFRON_CHECK_ARG(((uintptr_t)g & 15) == 0 && ((uintptr_t)g2 & 15) == 0 && (!mask || ((uintptr_t)mask & 3) == 0));
  int grid = grid_for(n >> 2);
  *nblk_out = grid;
  hipLaunchKernelGGL(k_sumsq_masked, dim3(grid), dim3(TPB), 0, (hipStream_t)stream, g, g2, mask, n, partials);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_sumsq_masked_ranges(const float* g, const uint8_t* mask, const int64_t* ranges, int n_ranges, double* partials, void* stream) {
  SFRON_CHECK_ARG(g && ranges && partials && n_ranges > 0);
  SFRON_CHECK_ARG(((uintptr_t)g & 15) == 0 && (!mask || ((uintptr_t)mask & 3) == 0));
  hipLaunchKernelGGL(k_sumsq_ranges, dim3(n_ranges), dim3(TPB), 0, (hipStream_t)stream, g, mask, (const long long*)ranges, partials);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_clip_coef(const double* partials, int nblk, float max_norm, float* stats, void* stream) {
  SFRON_CHECK_ARG(partials && stats && nblk > 0);
  hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(CC_TPB), 0, (hipStream_t)stream, partials, nblk, max_norm, stats);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_masked_clip_adam_wg(float* p, const float* g, const float* g2, float* m, float* v, const uint8_t* mask, const float* stats,
                              int64_t n, double beta1, double beta2, double eps, double step_size, double bc2_sqrt,
                              double decay_mul, uint16_t* w_bf16, float* ema, double ema_decay, int ema_mode, int max_workgroups,
                              void* stream) {
  SFRON_CHECK_ARG(p && g && m && v && n >= 0);
  SFRON_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)g2 | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  SFRON_CHECK_ARG(!mask || ((uintptr_t)mask & 3) == 0);
  SFRON_CHECK_ARG(!w_bf16 || ((uintptr_t)w_bf16 & 7) == 0);
  SFRON_CHECK_ARG(ema_mode == 0 || (ema && ((uintptr_t)ema & 15) == 0 && (ema_mode == 1 || ema_mode == 2)));
  AdamArgs a{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)step_size, (float)bc2_sqrt,
             (float)decay_mul, (float)ema_decay, (float)(1.0 - ema_decay), ema_mode};
  // three resident workgroups per CU (768 on 256 CUs), each striding through the arenas: measured 5.7-6.0 TB/s against 5.4-5.5 for
  // 2048 short-lived ones (tools/bench_sweep.py, 675 M parameters; fewer than 512 starve the memory system)
  constexpr int ADAM_GRID = 768;
  int grid = grid_for(n >> 2);
  if (grid > ADAM_GRID) grid = ADAM_GRID;
  if (max_workgroups > 0 && grid > max_workgroups) grid = max_workgroups;      // a sweep that runs BESIDE a GEMM chain: bounded share of the chip
  hipLaunchKernelGGL(k_masked_clip_adam, dim3(grid), dim3(TPB), 0, (hipStream_t)stream,
                     p, g, g2, m, v, mask, stats, n, a, w_bf16, ema);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_masked_clip_adam_q(float* p, const float* g, float* m, float* v, const uint8_t* mask, const float* stats, int64_t n, double beta1,
                             double beta2, double eps, double step_size, double bc2_sqrt, double decay_mul, uint16_t* w_bf16, float* ema,
                             double ema_decay, int ema_mode, uint8_t* w_e4m3, const float* w_e4m3_scale, int max_workgroups, void* stream) {
  SFRON_CHECK_ARG(p && g && m && v && n >= 0 && n % 4 == 0 && w_e4m3 && w_e4m3_scale);
  SFRON_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)w_e4m3 & 3) == 0);
  SFRON_CHECK_ARG(!mask || ((uintptr_t)mask & 3) == 0);
  SFRON_CHECK_ARG(!w_bf16 || ((uintptr_t)w_bf16 & 7) == 0);
  SFRON_CHECK_ARG(ema_mode == 0 || (ema && ((uintptr_t)ema & 15) == 0 && (ema_mode == 1 || ema_mode == 2)));
  AdamArgs a{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)step_size, (float)bc2_sqrt,
             (float)decay_mul, (float)ema_decay, (float)(1.0 - ema_decay), ema_mode};
  int grid = grid_for(n >> 2);
  if (grid > 768) grid = 768;
  if (max_workgroups > 0 && grid > max_workgroups) grid = max_workgroups;
  hipLaunchKernelGGL(k_masked_clip_adam, dim3(grid), dim3(TPB), 0, (hipStream_t)stream, p, g, (const float*)nullptr, m, v, mask, stats, n, a, w_bf16,
                     ema, w_e4m3, w_e4m3_scale);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_masked_clip_adam(float* p, const float* g, const float* g2, float* m, float* v, const uint8_t* mask, const float* stats,
                           int64_t n, double beta1, double beta2, double eps, double step_size, double bc2_sqrt,
                           double decay_mul, uint16_t* w_bf16, float* ema, double ema_decay, int ema_mode, void* stream) {
  return sfron_masked_clip_adam_wg(p, g, g2, m, v, mask, stats, n, beta1, beta2, eps, step_size, bc2_sqrt, decay_mul, w_bf16, ema,
                                   ema_decay, ema_mode, 0, stream);
}

// rank-R gradient sweeps (see k_adam_lowrank): W = p[0 .. NM*D), dW = dmod^T sc with dmod bf16 [R][NM], sc bf16 [R][D]
int sfron_sumsq_lowrank(const uint16_t* dmod, const uint16_t* sc, int R, int NM, int D, const uint8_t* mask, double* partials,
                        int* nblk_out, void* stream) {
  SFRON_CHECK_ARG(dmod && sc && partials && nblk_out && R > 0 && NM > 0 && D > 0 && NM % LR_ROWS == 0 && D % 4 == 0 && D <= 4096);
  SFRON_CHECK_ARG((((uintptr_t)dmod) & 15) == 0 && (((uintptr_t)sc) & 7) == 0 && (!mask || ((uintptr_t)mask & 3) == 0));
  // The norm pre-pass only needs sum (mask g)^2: on the matrix core the rank-R product is one 16 x 16 x 32 step per output tile and is
  // never stored (gemm.hip EPI_SUMSQ: 128 x 128 tiles, one fp64 partial each; was 32 vector FMAs per element: 480 -> ~100 us at DiT-XL/2).
  // Its partial count must fit the NM / LR_ROWS doubles the caller holds for this segment.
  if (NM % 128 == 0 && D % 128 == 0 && D <= 128 * (128 / LR_ROWS) && NM % 8 == 0 && D % 8 == 0)
    return gemm_sumsq_lowrank(dmod, sc, R, NM, D, mask, partials, nblk_out, stream);
  const int threads = (D / 4 + 63) / 64 * 64;
  *nblk_out = NM / LR_ROWS;
  hipLaunchKernelGGL(k_sumsq_lowrank, dim3(NM / LR_ROWS), dim3(threads), 0, (hipStream_t)stream, (const __bf16*)dmod, (const __bf16*)sc, R, NM, D,
                     mask, partials);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_adam_lowrank(float* p, float* m, float* v, const uint8_t* mask, const float* stats, const uint16_t* dmod, const uint16_t* sc, int R,
                       int NM, int D, double beta1, double beta2, double eps, double step_size, double bc2_sqrt, double decay_mul,
                       uint16_t* w_bf16, float* ema, double ema_decay, int ema_mode, void* stream) {
  SFRON_CHECK_ARG(p && m && v && dmod && sc && R > 0 && NM > 0 && D > 0 && NM % LR_ROWS == 0 && D % 4 == 0 && D <= 4096);
  SFRON_CHECK_ARG((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)dmod) & 15) == 0 && (((uintptr_t)sc) & 7) == 0);
  SFRON_CHECK_ARG((!mask || ((uintptr_t)mask & 3) == 0) && (!w_bf16 || ((uintptr_t)w_bf16 & 7) == 0));
  SFRON_CHECK_ARG(ema_mode == 0 || (ema && ((uintptr_t)ema & 15) == 0 && (ema_mode == 1 || ema_mode == 2)));
  AdamArgs a{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)step_size, (float)bc2_sqrt,
             (float)decay_mul, (float)ema_decay, (float)(1.0 - ema_decay), ema_mode};
  const int threads = (D / 4 + 63) / 64 * 64;
  hipLaunchKernelGGL(k_adam_lowrank, dim3(NM / LR_ROWS), dim3(threads), 0, (hipStream_t)stream, p, m, v, mask, stats, (const __bf16*)dmod,
                     (const __bf16*)sc, R, NM, D, a, w_bf16, ema);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_ema_update(float* ema, const float* p, int64_t n, double decay, int ema_mode, void* stream) {
  SFRON_CHECK_ARG(ema && p && n >= 0 && (ema_mode == 1 || ema_mode == 2));
  AdamArgs a{0, 0, 0, 0, 0, 1, 1, (float)decay, (float)(1.0 - decay), ema_mode};
  hipLaunchKernelGGL(k_ema, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, ema, p, n, a);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_fisher_accum(float* fisher, const float* g, int64_t n, float n_iters, void* stream) {
  SFRON_CHECK_ARG(fisher && g && n >= 0 && n_iters > 0);
  hipLaunchKernelGGL(k_fisher_accum, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, fisher, g, n, n_iters);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_fisher_accum_clipped(float* fisher, const float* g, const float* g2, const float* stats, int64_t n, float n_iters, void* stream) {
  SFRON_CHECK_ARG(fisher && g && n >= 0 && n_iters > 0);
  hipLaunchKernelGGL(k_fisher_accum_clipped, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, fisher, g, g2, stats, n, n_iters);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_mask_from_fisher(const float* forget_fisher, const float* remain_fisher, int64_t n, float th, uint8_t* mask,
                           void* stream) {
  SFRON_CHECK_ARG(forget_fisher && remain_fisher && mask && n >= 0);
  hipLaunchKernelGGL(k_mask_from_fisher, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, forget_fisher,
                     remain_fisher, n, th, mask);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_cast_bf16(const float* src, uint16_t* dst, int64_t n, void* stream) {
  SFRON_CHECK_ARG(src && dst && n >= 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0);
  hipLaunchKernelGGL(k_cast_bf16, dim3(grid_for(n >> 2)), dim3(TPB), 0, (hipStream_t)stream, src, dst, n);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

}  // extern "C"
