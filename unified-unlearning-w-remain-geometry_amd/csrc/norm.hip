// adaLN-Zero elementwise / reduction kernels of the DiT block (HBM-bound, one wave per token row).
//
// Replaces the PyTorch op sequences of /root/reference/DiT/models.py:
//   modulate(norm(x), shift, scale) ...... :19-20,103,109,120-121,139-140 (LayerNorm eps 1e-6, no affine)
//   x + gate.unsqueeze(1) * branch ....... :120-121 (forward is fused into the GEMM epilogue; the
//                                          backward's d_gate / d_branch is here)
//   and autograd's backward of both, including the per-sample token reductions that produce
//   d(shift, scale, gate) for the adaLN_modulation Linear (:113-118).
// Column reductions are written as per-row-chunk partials and summed by a second tiny kernel in a
// fixed order, so results are bitwise reproducible (no float atomics).
#include "common.h"
#include "../../include/sfron.h"

namespace {

constexpr int TPB = 256;          // 4 waves, one token row per wave at a time
constexpr int NCH = 5;            // float4 chunks per lane: supports D <= 64*4*NCH = 1280
constexpr float LN_EPS = 1e-6f;

struct RowRegs { float4 v[NCH]; };

__device__ __forceinline__ void load_row_f32(const float* __restrict__ p, int D4, int lane, RowRegs& r) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    r.v[i] = c < D4 ? reinterpret_cast<const float4*>(p)[c] : make_float4(0, 0, 0, 0);
  }
}
__device__ __forceinline__ void load_row_bf16(const __bf16* __restrict__ p, int D4, int lane, RowRegs& r) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < D4) {
      const bf16x4 b = reinterpret_cast<const bf16x4*>(p)[c];
      r.v[i] = make_float4(bf2f(b[0]), bf2f(b[1]), bf2f(b[2]), bf2f(b[3]));
    } else {
      r.v[i] = make_float4(0, 0, 0, 0);
    }
  }
}

// ---------------------------------------------------------------- LN + modulate forward
__global__ __launch_bounds__(TPB) void k_ln_mod_fwd(const float* __restrict__ x, const float* __restrict__ shift,
                                                    const float* __restrict__ scale, int ldmod, int T, int M, int D,
                                                    __bf16* __restrict__ out, float* __restrict__ mean_out,
                                                    float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  const int D4 = D >> 2;
  RowRegs r;
  load_row_f32(x + (size_t)row * D, D4, lane, r);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) s += r.v[i].x + r.v[i].y + r.v[i].z + r.v[i].w;
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (lane + 64 * i < D4) {
      const float a = r.v[i].x - mean, b = r.v[i].y - mean, c = r.v[i].z - mean, d = r.v[i].w - mean;
      q += a * a + b * b + c * c + d * d;
    }
  }
  const float var = wave_sum(q) / (float)D;
  const float rstd = 1.0f / sqrtf(var + LN_EPS);
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
  const int b = row / T;
  const float* sh = shift + (size_t)b * ldmod;
  const float* sc = scale + (size_t)b * ldmod;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    if (c < D4) {
      const float4 h = reinterpret_cast<const float4*>(sh)[c];
      const float4 g = reinterpret_cast<const float4*>(sc)[c];
      bf16x4 o = {f2bf((r.v[i].x - mean) * rstd * (1.0f + g.x) + h.x), f2bf((r.v[i].y - mean) * rstd * (1.0f + g.y) + h.y),
                  f2bf((r.v[i].z - mean) * rstd * (1.0f + g.z) + h.z), f2bf((r.v[i].w - mean) * rstd * (1.0f + g.w) + h.w)};
      reinterpret_cast<bf16x4*>(out + (size_t)row * D)[c] = o;
    }
  }
}

// each wave stores its own column partials (row-chunk = the rpw rows of one wave); the sums over the chunks of a
// sample are taken later in ONE launch for the whole backward pass (k_reduce_slots): no LDS, no barrier here
template <int NACC>
__device__ __forceinline__ void wave_partial_store(const float4 (&acc)[NACC][NCH], int D4, int lane, float* const (&dst)[NACC]) {
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < D4) reinterpret_cast<float4*>(dst[a])[c] = acc[a][i];
    }
}

// ---------------------------------------------------------------- LN + modulate backward
// dx (fp32, in/out) += d LN-path;  partials: p_shift[chunk][D] = sum_rows dxmod, p_scale[chunk][D] = sum_rows dxmod*xhat
__global__ __launch_bounds__(TPB) void k_ln_mod_bwd(const __bf16* __restrict__ dxmod, const float* __restrict__ x,
                                                    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                    const float* __restrict__ scale, int ldmod, int T, int M, int D,
                                                    int rpw, float* __restrict__ dx, int dx_accumulate,
                                                    float* __restrict__ p_shift, float* __restrict__ p_scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D4 = D >> 2;
  const int chunk = blockIdx.x * 4 + wave;       // one chunk = the rpw consecutive rows of this wave
  const int row0 = chunk * rpw;
  if (row0 >= M) return;
  const int b = row0 / T;                        // all rows of a chunk belong to one sample (T % rpw == 0)
  float4 gs[NCH];                                // 1 + scale
  const float* sc = scale + (size_t)b * ldmod;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    float4 g = c < D4 ? reinterpret_cast<const float4*>(sc)[c] : make_float4(0, 0, 0, 0);
    gs[i] = make_float4(1.0f + g.x, 1.0f + g.y, 1.0f + g.z, 1.0f + g.w);
  }
  float4 acc[2][NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) { acc[0][i] = make_float4(0, 0, 0, 0); acc[1][i] = make_float4(0, 0, 0, 0); }
  for (int rr = 0; rr < rpw; ++rr) {
    const int row = row0 + rr;
    RowRegs xr, dr;
    load_row_f32(x + (size_t)row * D, D4, lane, xr);
    load_row_bf16(dxmod + (size_t)row * D, D4, lane, dr);
    const float mean = mean_in[row], rstd = rstd_in[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (lane + 64 * i < D4) {
        float4& xv = xr.v[i];
        xv.x = (xv.x - mean) * rstd; xv.y = (xv.y - mean) * rstd; xv.z = (xv.z - mean) * rstd; xv.w = (xv.w - mean) * rstd;
        const float4 d = dr.v[i];
        acc[0][i].x += d.x; acc[0][i].y += d.y; acc[0][i].z += d.z; acc[0][i].w += d.w;
        acc[1][i].x += d.x * xv.x; acc[1][i].y += d.y * xv.y; acc[1][i].z += d.z * xv.z; acc[1][i].w += d.w * xv.w;
        float4& g = dr.v[i];                      // g = dxmod * (1 + scale)
        g.x *= gs[i].x; g.y *= gs[i].y; g.z *= gs[i].z; g.w *= gs[i].w;
        s1 += g.x + g.y + g.z + g.w;
        s2 += g.x * xv.x + g.y * xv.y + g.z * xv.z + g.w * xv.w;
      }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < D4) {
        const float4 g = dr.v[i], xh = xr.v[i];
        float4 o = make_float4(rstd * (g.x - m1 - xh.x * m2), rstd * (g.y - m1 - xh.y * m2),
                               rstd * (g.z - m1 - xh.z * m2), rstd * (g.w - m1 - xh.w * m2));
        float4* dp = reinterpret_cast<float4*>(dx + (size_t)row * D) + c;
        if (dx_accumulate) { const float4 p = *dp; o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w; }
        *dp = o;
      }
    }
  }
  float* const dst[2] = {p_shift + (size_t)chunk * D, p_scale + (size_t)chunk * D};
  wave_partial_store<2>(acc, D4, lane, dst);
}

// ---------------------------------------------------------------- gated-residual backward
// d_branch (bf16) = dy * gate[b];  partials: p_gate[chunk][D] = sum_rows dy * branch, p_dy[chunk][D] = sum_rows dy
__global__ __launch_bounds__(TPB) void k_gate_bwd(const float* __restrict__ dy, const __bf16* __restrict__ branch,
                                                  const float* __restrict__ gate, int ldmod, int T, int M, int D, int rpw,
                                                  __bf16* __restrict__ d_branch, float* __restrict__ p_gate,
                                                  float* __restrict__ p_dy) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D4 = D >> 2;
  const int chunk = blockIdx.x * 4 + wave;
  const int row0 = chunk * rpw;
  if (row0 >= M) return;
  const int b = row0 / T;
  float4 gt[NCH];
  const float* gp = gate + (size_t)b * ldmod;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    gt[i] = c < D4 ? reinterpret_cast<const float4*>(gp)[c] : make_float4(0, 0, 0, 0);
  }
  float4 acc[2][NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) { acc[0][i] = make_float4(0, 0, 0, 0); acc[1][i] = make_float4(0, 0, 0, 0); }
  for (int rr = 0; rr < rpw; ++rr) {
    const int row = row0 + rr;
    RowRegs dr, br;
    load_row_f32(dy + (size_t)row * D, D4, lane, dr);
    load_row_bf16(branch + (size_t)row * D, D4, lane, br);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + 64 * i;
      if (c < D4) {
        const float4 d = dr.v[i], a = br.v[i];
        acc[0][i].x += d.x * a.x; acc[0][i].y += d.y * a.y; acc[0][i].z += d.z * a.z; acc[0][i].w += d.w * a.w;
        acc[1][i].x += d.x; acc[1][i].y += d.y; acc[1][i].z += d.z; acc[1][i].w += d.w;
        bf16x4 o = {f2bf(d.x * gt[i].x), f2bf(d.y * gt[i].y), f2bf(d.z * gt[i].z), f2bf(d.w * gt[i].w)};
        reinterpret_cast<bf16x4*>(d_branch + (size_t)row * D)[c] = o;
      }
    }
  }
  float* const dst[2] = {p_gate + (size_t)chunk * D, p_dy + (size_t)chunk * D};
  wave_partial_store<2>(acc, D4, lane, dst);
}

// ---------------------------------------------------------------- small fixed-order reductions
// out[g * ldout + c] (+)= sum_{j < per_group} P[(g * per_group + j) * D + c]
__global__ __launch_bounds__(TPB) void k_reduce_chunks(const float* __restrict__ P, int per_group, int D, float* __restrict__ out,
                                                       int ldout, int accumulate) {
  const int g = blockIdx.y;
  const int c = blockIdx.x * TPB + threadIdx.x;
  if (c >= D) return;
  const float* p = P + (size_t)g * per_group * D + c;
  float s = 0.f;
  for (int j = 0; j < per_group; ++j) s += p[(size_t)j * D];
  float* o = out + (size_t)g * ldout + c;
  *o = accumulate ? *o + s : s;
}

// out[c] = sum_b w[b * ldw + c] * sum_{j < per_group} P[(b * per_group + j) * D + c]   (bias grad behind a gate)
__global__ __launch_bounds__(TPB) void k_weighted_reduce(const float* __restrict__ P, int groups, int per_group, int D,
                                                         const float* __restrict__ w, int ldw, float* __restrict__ out) {
  const int c = blockIdx.x * TPB + threadIdx.x;
  if (c >= D) return;
  float tot = 0.f;
  for (int b = 0; b < groups; ++b) {
    const float* p = P + (size_t)b * per_group * D + c;
    float s = 0.f;
    for (int j = 0; j < per_group; ++j) s += p[(size_t)j * D];
    tot += w[(size_t)b * ldw + c] * s;
  }
  out[c] = tot;
}

// two partial buffers at once: out0[g*ld0 + c] = sum_j P0[(g*per+j)*D + c]; out1 likewise from P1
__global__ __launch_bounds__(TPB) void k_reduce2(const float* __restrict__ P0, const float* __restrict__ P1, int per_group, int D,
                                                 float* __restrict__ out0, int ld0, float* __restrict__ out1, int ld1) {
  const int g = blockIdx.y;
  const int c = blockIdx.x * TPB + threadIdx.x;
  if (c >= D) return;
  const size_t base = (size_t)g * per_group * D + c;
  float s0 = 0.f, s1 = 0.f;
  for (int j = 0; j < per_group; ++j) { s0 += P0[base + (size_t)j * D]; s1 += P1[base + (size_t)j * D]; }
  out0[(size_t)g * ld0 + c] = s0;
  out1[(size_t)g * ld1 + c] = s1;
}

// bias gradients behind the gates of every block in one launch:
//   out[l * out_stride + which * out_which + c] = sum_b gate[b * ldg + l * gate_stride + which * gate_which + c] * S[((l*2+which)*B + b) * D + c]
__global__ __launch_bounds__(TPB) void k_gated_bias_grads(const float* __restrict__ S, const float* __restrict__ gate, int ldg,
                                                          long gate_stride, long gate_which, int B, int D, float* __restrict__ out,
                                                          long out_stride, long out_which0, long out_which1) {
  const int l = blockIdx.y >> 1, which = blockIdx.y & 1;
  const int c = blockIdx.x * TPB + threadIdx.x;
  if (c >= D) return;
  const float* s = S + ((size_t)(l * 2 + which) * B) * D + c;
  const float* gt = gate + (size_t)l * gate_stride + (which ? gate_which : 0) + c;
  float t = 0.f;
  for (int b = 0; b < B; ++b) t += gt[(size_t)b * ldg] * s[(size_t)b * D];
  out[(size_t)l * out_stride + (which ? out_which1 : out_which0) + c] = t;
}

// column sums of a [M][N] matrix: stage 1 writes partials[chunk][N]; the caller finishes with k_reduce_chunks
template <typename T>
__global__ __launch_bounds__(TPB) void k_colsum_partial(const T* __restrict__ X, int M, int N, int ld, int rows_per_block,
                                                        float* __restrict__ partials) {
  __shared__ float4 sh[3][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 64 + lane;            // group of 4 columns
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float4 acc = make_float4(0, 0, 0, 0);
  if (c4 * 4 < N) {
    for (int r = r0 + wave; r < r1; r += 4) {
      if constexpr (sizeof(T) == 2) {
        const bf16x4 b = *reinterpret_cast<const bf16x4*>(X + (size_t)r * ld + c4 * 4);
        acc.x += bf2f(b[0]); acc.y += bf2f(b[1]); acc.z += bf2f(b[2]); acc.w += bf2f(b[3]);
      } else {
        const float4 b = *reinterpret_cast<const float4*>(X + (size_t)r * ld + c4 * 4);
        acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
      }
    }
  }
  if (wave > 0) sh[wave - 1][lane] = acc;
  __syncthreads();
  if (wave == 0 && c4 * 4 < N) {
    for (int w = 0; w < 3; ++w) { acc.x += sh[w][lane].x; acc.y += sh[w][lane].y; acc.z += sh[w][lane].z; acc.w += sh[w][lane].w; }
    *reinterpret_cast<float4*>(partials + (size_t)blockIdx.y * N + c4 * 4) = acc;
  }
}

// rows per wave (= rows per partial chunk) of the backward elementwise kernels
inline int pick_rpw(int T) {
  if (T % 4 == 0) return 4;
  if (T % 2 == 0) return 2;
  return 1;
}

struct SlotDst { float* base; long layer_stride; int ld; };
struct SlotArgs { SlotDst dst[8]; };      // [kind 0..3][buf 0..1]

// One launch for the whole backward pass: slot s = (layer, kind, buf) holds per-chunk partials [B*per][D];
// out[dst(kind,buf).base + layer*stride + b*ld + c] = sum_j partial[(b*per + j)*D + c]
__global__ __launch_bounds__(TPB) void k_reduce_slots(const float* __restrict__ parts, long slot_stride, int per, int D, SlotArgs a) {
  const int slot = blockIdx.z, b = blockIdx.y;
  const int c = blockIdx.x * TPB + threadIdx.x;
  if (c >= D) return;
  const int layer = slot >> 3, kb = slot & 7;
  const float* p = parts + (size_t)slot * slot_stride + (size_t)b * per * D + c;
  float s = 0.f;
  for (int j = 0; j < per; ++j) s += p[(size_t)j * D];
  const SlotDst d = a.dst[kb];
  d.base[(size_t)layer * d.layer_stride + (size_t)b * d.ld + c] = s;
}

}  // namespace

extern "C" {

int sfron_rows_per_chunk(int tokens) { return tokens > 0 ? pick_rpw(tokens) : 0; }

int sfron_ln_modulate_fwd(const float* x, const float* shift, const float* scale, int ldmod, int tokens, int M, int D,
                          uint16_t* out, float* mean, float* rstd, void* stream) {
  SFRON_CHECK_ARG(x && shift && scale && out && mean && rstd && M > 0 && tokens > 0);
  SFRON_CHECK_ARG(D % 4 == 0 && D <= 64 * 4 * NCH && ldmod % 4 == 0);
  SFRON_CHECK_ARG((((uintptr_t)x | (uintptr_t)shift | (uintptr_t)scale) & 15) == 0 && ((uintptr_t)out & 7) == 0);
  hipLaunchKernelGGL(k_ln_mod_fwd, dim3(cdiv(M, 4)), dim3(TPB), 0, (hipStream_t)stream, x, shift, scale, ldmod, tokens, M, D,
                     (__bf16*)out, mean, rstd);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_ln_modulate_bwd(const uint16_t* d_out, const float* x, const float* mean, const float* rstd, const float* scale,
                          int ldmod, int tokens, int M, int D, float* dx, int dx_accumulate, float* p_shift,
                          float* p_scale, void* stream) {
  SFRON_CHECK_ARG(d_out && x && mean && rstd && scale && dx && p_shift && p_scale && M > 0);
  SFRON_CHECK_ARG(D % 4 == 0 && D <= 64 * 4 * NCH && ldmod % 4 == 0);
  const int rpw = pick_rpw(tokens);
  SFRON_CHECK_ARG(M % tokens == 0);
  hipLaunchKernelGGL(k_ln_mod_bwd, dim3(cdiv(M / rpw, 4)), dim3(TPB), 0, (hipStream_t)stream, (const __bf16*)d_out, x, mean,
                     rstd, scale, ldmod, tokens, M, D, rpw, dx, dx_accumulate, p_shift, p_scale);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_gate_bwd(const float* dy, const uint16_t* branch, const float* gate, int ldmod, int tokens, int M, int D,
                   uint16_t* d_branch, float* p_gate, float* p_dy, void* stream) {
  SFRON_CHECK_ARG(dy && branch && gate && d_branch && p_gate && p_dy && M > 0);
  SFRON_CHECK_ARG(D % 4 == 0 && D <= 64 * 4 * NCH && ldmod % 4 == 0);
  const int rpw = pick_rpw(tokens);
  SFRON_CHECK_ARG(M % tokens == 0);
  hipLaunchKernelGGL(k_gate_bwd, dim3(cdiv(M / rpw, 4)), dim3(TPB), 0, (hipStream_t)stream, dy, (const __bf16*)branch, gate,
                     ldmod, tokens, M, D, rpw, (__bf16*)d_branch, p_gate, p_dy);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_reduce_chunks(const float* partials, int groups, int per_group, int D, float* out, int ldout, int accumulate,
                        void* stream) {
  SFRON_CHECK_ARG(partials && out && groups > 0 && per_group > 0 && D > 0);
  hipLaunchKernelGGL(k_reduce_chunks, dim3(cdiv(D, TPB), groups), dim3(TPB), 0, (hipStream_t)stream, partials, per_group, D,
                     out, ldout, accumulate);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_reduce2(const float* p0, const float* p1, int groups, int per_group, int D, float* out0, int ld0, float* out1,
                  int ld1, void* stream) {
  SFRON_CHECK_ARG(p0 && p1 && out0 && out1 && groups > 0 && per_group > 0 && D > 0);
  hipLaunchKernelGGL(k_reduce2, dim3(cdiv(D, TPB), groups), dim3(TPB), 0, (hipStream_t)stream, p0, p1, per_group, D, out0, ld0,
                     out1, ld1);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_gated_bias_grads(const float* S, const float* gate, int ldg, long gate_stride, long gate_which, int layers, int B,
                           int D, float* out, long out_stride, long out_which0, long out_which1, void* stream) {
  SFRON_CHECK_ARG(S && gate && out && layers > 0 && B > 0 && D > 0);
  hipLaunchKernelGGL(k_gated_bias_grads, dim3(cdiv(D, TPB), 2 * layers), dim3(TPB), 0, (hipStream_t)stream, S, gate, ldg,
                     gate_stride, gate_which, B, D, out, out_stride, out_which0, out_which1);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_reduce_slots(const float* parts, long slot_stride, int n_slots, int groups, int per_group, int D,
                       float* const* dst_base, const long* dst_layer_stride, const int* dst_ld, void* stream) {
  SFRON_CHECK_ARG(parts && dst_base && dst_layer_stride && dst_ld && n_slots > 0 && groups > 0 && per_group > 0 && D > 0);
  SlotArgs a;
  for (int i = 0; i < 8; ++i) a.dst[i] = SlotDst{dst_base[i], dst_layer_stride[i], dst_ld[i]};
  hipLaunchKernelGGL(k_reduce_slots, dim3(cdiv(D, TPB), groups, n_slots), dim3(TPB), 0, (hipStream_t)stream, parts, slot_stride,
                     per_group, D, a);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_weighted_reduce(const float* partials, int groups, int per_group, int D, const float* w, int ldw, float* out,
                          void* stream) {
  SFRON_CHECK_ARG(partials && out && w && groups > 0 && per_group > 0 && D > 0);
  hipLaunchKernelGGL(k_weighted_reduce, dim3(cdiv(D, TPB)), dim3(TPB), 0, (hipStream_t)stream, partials, groups, per_group,
                     D, w, ldw, out);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_colsum(const void* X, int is_bf16, int M, int N, int ld, float* partials, int max_partials, float* out,
                 void* stream) {
  SFRON_CHECK_ARG(X && partials && out && M > 0 && N > 0 && N % 4 == 0 && ld % 4 == 0 && max_partials > 0);
  int chunks = cdiv(M, 128);
  if (chunks > max_partials) chunks = max_partials;
  const int rpb = cdiv(M, chunks);
  chunks = cdiv(M, rpb);
  dim3 grid(cdiv(N, 256), chunks);
  if (is_bf16)
    hipLaunchKernelGGL(k_colsum_partial<__bf16>, grid, dim3(TPB), 0, (hipStream_t)stream, (const __bf16*)X, M, N, ld, rpb, partials);
  else
    hipLaunchKernelGGL(k_colsum_partial<float>, grid, dim3(TPB), 0, (hipStream_t)stream, (const float*)X, M, N, ld, rpb, partials);
  SFRON_LAUNCH_STATUS();
  hipLaunchKernelGGL(k_reduce_chunks, dim3(cdiv(N, TPB), 1), dim3(TPB), 0, (hipStream_t)stream, partials, chunks, N, out, N, 0);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

}  // extern "C"
