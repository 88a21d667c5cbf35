// Fused multi-head self-attention (non-causal, no dropout) forward and backward for the DiT block.
//
// Replaces timm Attention's softmax(q * hd^-0.5 @ k^T) @ v (used at /root/reference/DiT/models.py:108,120;
// timm is un-vendored: behaviour per SURVEY.md section 8c) and autograd's backward of it.
// Layout: qkv is the fused-QKV GEMM output [B*T][3*D] bf16 with column = which*D + head*hd + d
// (== reshape(B,N,3,H,hd)); o / d_o are [B*T][D] with column = head*hd + d (== transpose(1,2).reshape).
//
// Flash-style: scores never touch HBM.  All three kernels put the SOFTMAX ROW on the MFMA lane
// (S^T = K·Q^T is what the matrix core computes), so row max / row sum / rescale are lane-local, the
// probabilities feed the next MFMA straight from registers (k-slot permutation, no LDS round trip)
// and V / K / Q / dO tiles are consumed both row-wise (ds_read_b128) and column-wise
// (ds_read_b64_tr_b16) from ONE swizzled LDS image.
//   forward : grid (T / (64*QT), B*H); wave = 16*QT queries, streams 64-key chunks of K,V
//   dQ      : same decomposition; recomputes P from LSE
//   dK,dV   : wave = 16*KT keys, streams 64-query chunks of Q,dO; no cross-workgroup reduction
// head_dim 64 -> HDP 64, 2 k-steps, 4 d-tiles; head_dim 72 (DiT-XL) -> HDP 128 image rows, 3 k-steps
// (zero padded to 96), 5 d-tiles (80).
#include "common.h"
#include "../../include/sfron.h"

namespace {

constexpr int NT = 256;
constexpr float LOG2E = 1.4426950408889634f;
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

template <int HDP> __device__ __forceinline__ int aswz(int row) {
  return HDP == 64 ? (((row >> 1) & 3) << 1) : ((row & 7) << 1);
}
template <int HDP> __device__ __forceinline__ int aoff(int row, int ch) { return row * HDP + ((ch ^ aswz<HDP>(row)) << 3); }

// stage a [64 rows][hd] bf16 chunk (row stride ld in global) into the swizzled image; columns >= hd stay zero
template <int HDP>
struct ChunkStager {
  static constexpr int NL = HDP / 32 + 1;
  uint4 r[NL];
  __device__ __forceinline__ void load(const __bf16* src, int ld, int ch_per_row, int tid) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int c = tid + NT * i;
      if (c < 64 * ch_per_row) {
        const int row = c / ch_per_row, ch = c % ch_per_row;
        r[i] = *reinterpret_cast<const uint4*>(src + (size_t)row * ld + ch * 8);
      }
    }
  }
  __device__ __forceinline__ void store(__bf16* img, int ch_per_row, int tid) const {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int c = tid + NT * i;
      if (c < 64 * ch_per_row) {
        const int row = c / ch_per_row, ch = c % ch_per_row;
        *reinterpret_cast<uint4*>(img + aoff<HDP>(row, ch)) = r[i];
      }
    }
  }
};

template <int HDP> __device__ __forceinline__ void zero_image(__bf16* img, int n_images, int tid) {
  uint4 z = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < n_images * 64 * HDP / 8; i += NT) reinterpret_cast<uint4*>(img)[i] = z;
}

// row fragment: lane holds X[row0 + (lane&15)][32*ks + 8*(lane>>4) + 0..7]
template <int HDP> __device__ __forceinline__ bf16x8 frag_rows(const __bf16* img, int row0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8*>(img + aoff<HDP>(row0 + (lane & 15), ks * 4 + (lane >> 4)));
}
// column fragment with the PERMUTED k-slot map used by the register-resident probabilities:
// lane holds X[rows r = rbase + 4g + j (j<4) and rbase + 16 + 4g + (j-4)][col0 + (lane&15)], g = lane>>4
template <int HDP> __device__ __forceinline__ bf16x8 frag_cols_perm(const __bf16* img, int rbase, int col0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int ch = (col0 >> 3) + (p >> 1);
  const int r0 = rbase + 4 * g + q;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + aoff<HDP>(r0, ch) + 4 * (p & 1)));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + aoff<HDP>(r0 + 16, ch) + 4 * (p & 1)));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
__device__ __forceinline__ bf16x8 pack_perm(const f32x4& a, const f32x4& b) {
  bf16x8 r;
  r[0] = f2bf(a[0]); r[1] = f2bf(a[1]); r[2] = f2bf(a[2]); r[3] = f2bf(a[3]);
  r[4] = f2bf(b[0]); r[5] = f2bf(b[1]); r[6] = f2bf(b[2]); r[7] = f2bf(b[3]);
  return r;
}
// wave's own rows straight from global: lane holds X[row0 + (lane&15)][32*ks + 8*(lane>>4) + 0..7], zero beyond hd
__device__ __forceinline__ bf16x8 frag_rows_global(const __bf16* base, int ld, int row0, int ks, int hd, int lane) {
  const int d = 32 * ks + 8 * (lane >> 4);
  bf16x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.0f;
  if (d >= hd) return z;
  return *reinterpret_cast<const bf16x8*>(base + (size_t)(row0 + (lane & 15)) * ld + d);
}
__device__ __forceinline__ float group_max(float v) {      // over the 4 lane groups holding one softmax row
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

// ------------------------------------------------------------------------------------- forward
template <int HDP, int KS, int NDT, int QT>
__global__ __launch_bounds__(NT) void k_attn_fwd(const __bf16* __restrict__ qkv, __bf16* __restrict__ o,
                                                 float* __restrict__ lse, int T, int H, int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  __bf16* sK = smem;                     // [2][64*HDP]
  __bf16* sV = smem + 2 * 64 * HDP;      // [2][64*HDP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int D = H * hd, ld = 3 * D, chpr = hd / 8;
  const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);
  const __bf16* base = qkv + (size_t)b * T * ld + h * hd;
  const float c = scale * LOG2E;

  zero_image<HDP>(smem, 4, tid);
  bf16x8 fq[QT][KS];
#pragma unroll
  for (int qi = 0; qi < QT; ++qi)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fq[qi][ks] = frag_rows_global(base, ld, q0 + 16 * qi, ks, hd, lane);
  __syncthreads();

  ChunkStager<HDP> stK, stV;
  stK.load(base + D, ld, chpr, tid);
  stV.load(base + 2 * D, ld, chpr, tid);
  stK.store(sK, chpr, tid);
  stV.store(sV, chpr, tid);
  __syncthreads();

  f32x4 oacc[QT][NDT];
  float m[QT], l[QT];
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) {
    m[qi] = -INFINITY; l[qi] = 0.f;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) oacc[qi][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int nchunk = T / 64;
  for (int kc = 0; kc < nchunk; ++kc) {
    const int cur = kc & 1;
    const bool more = kc + 1 < nchunk;
    if (more) {
      stK.load(base + D + (size_t)(kc + 1) * 64 * ld, ld, chpr, tid);
      stV.load(base + 2 * D + (size_t)(kc + 1) * 64 * ld, ld, chpr, tid);
    }
    const __bf16* iK = sK + cur * 64 * HDP;
    const __bf16* iV = sV + cur * 64 * HDP;
    f32x4 s[QT][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      bf16x8 fk[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fk[ks] = frag_rows<HDP>(iK, kt * 16, ks, lane);
#pragma unroll
      for (int qi = 0; qi < QT; ++qi) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[ks], fq[qi][ks], a, 0, 0, 0);
        s[qi][kt] = a;     // S[q = q0+16qi+(lane&15)][key = 64kc + 16kt + 4g + j]
      }
    }
    bf16x8 pf[QT][2];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[qi][kt][j]);
      mx = group_max(mx);
      const float mn = fmaxf(m[qi], mx);
      const float alpha = exp2f((m[qi] - mn) * c);
      m[qi] = mn;
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = exp2f((s[qi][kt][j] - mn) * c);
          s[qi][kt][j] = p;
          ps += p;
        }
      l[qi] = l[qi] * alpha + ps;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) oacc[qi][dt] *= alpha;
      pf[qi][0] = pack_perm(s[qi][0], s[qi][1]);
      pf[qi][1] = pack_perm(s[qi][2], s[qi][3]);
    }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 fv = frag_cols_perm<HDP>(iV, 32 * s2, dt * 16, lane);
#pragma unroll
        for (int qi = 0; qi < QT; ++qi)
          oacc[qi][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, pf[qi][s2], oacc[qi][dt], 0, 0, 0);
      }
    if (more) {
      stK.store(sK + (cur ^ 1) * 64 * HDP, chpr, tid);
      stV.store(sV + (cur ^ 1) * 64 * HDP, chpr, tid);
    }
    __syncthreads();
  }
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) {
    const float lt = group_sum(l[qi]);
    const float inv = 1.0f / lt;
    const int q = q0 + 16 * qi + (lane & 15);
    __bf16* orow = o + ((size_t)b * T + q) * D + h * hd;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = dt * 16 + 4 * g;
      if (d < hd) {
        const f32x4 v = oacc[qi][dt] * inv;
        bf16x4 ov = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *reinterpret_cast<bf16x4*>(orow + d) = ov;
      }
    }
    if (g == 0) lse[(size_t)bh * T + q] = m[qi] * scale + logf(lt);
  }
}

// delta[b,h,q] = sum_d dO[q, h, d] * O[q, h, d]
__global__ __launch_bounds__(NT) void k_attn_delta(const __bf16* __restrict__ o, const __bf16* __restrict__ d_o,
                                                   float* __restrict__ delta, int B, int T, int H, int hd) {
  const int64_t idx = (int64_t)blockIdx.x * NT + threadIdx.x;
  if (idx >= (int64_t)B * T * H) return;
  const int h = (int)(idx % H);
  const int64_t row = idx / H;
  const int b = (int)(row / T), q = (int)(row % T);
  const __bf16* po = o + row * (size_t)(H * hd) + h * hd;
  const __bf16* pd = d_o + row * (size_t)(H * hd) + h * hd;
  float s = 0.f;
  for (int d = 0; d < hd; d += 8) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(po + d);
    const bf16x8 e = *reinterpret_cast<const bf16x8*>(pd + d);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += bf2f(a[j]) * bf2f(e[j]);
  }
  delta[((size_t)b * H + h) * T + q] = s;
}

// ------------------------------------------------------------------------------------- dQ
template <int HDP, int KS, int NDT, int QT>
__global__ __launch_bounds__(NT) void k_attn_bwd_dq(const __bf16* __restrict__ qkv, const __bf16* __restrict__ d_o,
                                                    const float* __restrict__ lse, const float* __restrict__ delta,
                                                    __bf16* __restrict__ dqkv, int T, int H, int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  __bf16* sK = smem;
  __bf16* sV = smem + 2 * 64 * HDP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int D = H * hd, ld = 3 * D, chpr = hd / 8;
  const int q0 = blockIdx.x * (64 * QT) + wave * (16 * QT);
  const __bf16* base = qkv + (size_t)b * T * ld + h * hd;
  const __bf16* dob = d_o + (size_t)b * T * D + h * hd;
  const float c = scale * LOG2E;

  zero_image<HDP>(smem, 4, tid);
  bf16x8 fq[QT][KS], fdo[QT][KS];
  float nl[QT], dl[QT];
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      fq[qi][ks] = frag_rows_global(base, ld, q0 + 16 * qi, ks, hd, lane);
      fdo[qi][ks] = frag_rows_global(dob, D, q0 + 16 * qi, ks, hd, lane);
    }
    const int q = q0 + 16 * qi + (lane & 15);
    nl[qi] = -lse[(size_t)bh * T + q] * LOG2E;
    dl[qi] = delta[(size_t)bh * T + q];
  }
  __syncthreads();
  ChunkStager<HDP> stK, stV;
  stK.load(base + D, ld, chpr, tid);
  stV.load(base + 2 * D, ld, chpr, tid);
  stK.store(sK, chpr, tid);
  stV.store(sV, chpr, tid);
  __syncthreads();

  f32x4 dq[QT][NDT];
#pragma unroll
  for (int qi = 0; qi < QT; ++qi)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) dq[qi][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunk = T / 64;
  for (int kc = 0; kc < nchunk; ++kc) {
    const int cur = kc & 1;
    const bool more = kc + 1 < nchunk;
    if (more) {
      stK.load(base + D + (size_t)(kc + 1) * 64 * ld, ld, chpr, tid);
      stV.load(base + 2 * D + (size_t)(kc + 1) * 64 * ld, ld, chpr, tid);
    }
    const __bf16* iK = sK + cur * 64 * HDP;
    const __bf16* iV = sV + cur * 64 * HDP;
    bf16x8 ds[QT][2];
    f32x4 t[QT][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      bf16x8 fk[KS], fv[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { fk[ks] = frag_rows<HDP>(iK, kt * 16, ks, lane); fv[ks] = frag_rows<HDP>(iV, kt * 16, ks, lane); }
#pragma unroll
      for (int qi = 0; qi < QT; ++qi) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[ks], fq[qi][ks], a, 0, 0, 0);     // S
          p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[ks], fdo[qi][ks], p, 0, 0, 0);    // dP
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pr = exp2f(a[j] * c + nl[qi]);
          a[j] = pr * (p[j] - dl[qi]) * scale;                                             // dS (incl. softmax scale)
        }
        t[qi][kt] = a;
      }
    }
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) { ds[qi][0] = pack_perm(t[qi][0], t[qi][1]); ds[qi][1] = pack_perm(t[qi][2], t[qi][3]); }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 fkt = frag_cols_perm<HDP>(iK, 32 * s2, dt * 16, lane);
#pragma unroll
        for (int qi = 0; qi < QT; ++qi)
          dq[qi][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fkt, ds[qi][s2], dq[qi][dt], 0, 0, 0);
      }
    if (more) {
      stK.store(sK + (cur ^ 1) * 64 * HDP, chpr, tid);
      stV.store(sV + (cur ^ 1) * 64 * HDP, chpr, tid);
    }
    __syncthreads();
  }
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) {
    const int q = q0 + 16 * qi + (lane & 15);
    __bf16* row = dqkv + ((size_t)b * T + q) * ld + h * hd;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = dt * 16 + 4 * g;
      if (d < hd) {
        const f32x4 v = dq[qi][dt];
        bf16x4 ov = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *reinterpret_cast<bf16x4*>(row + d) = ov;
      }
    }
  }
}

// ------------------------------------------------------------------------------------- dK, dV
// wave owns 16*KT keys; S / dP are computed with the KEY on the lane: S[q = 4g+j][key = lane&15]
template <int HDP, int KS, int NDT, int KT>
__global__ __launch_bounds__(NT) void k_attn_bwd_dkv(const __bf16* __restrict__ qkv, const __bf16* __restrict__ d_o,
                                                     const float* __restrict__ lse, const float* __restrict__ delta,
                                                     __bf16* __restrict__ dqkv, int T, int H, int hd, float scale) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  __bf16* sQ = smem;
  __bf16* sO = smem + 2 * 64 * HDP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int D = H * hd, ld = 3 * D, chpr = hd / 8;
  const int k0 = blockIdx.x * (64 * KT) + wave * (16 * KT);
  const __bf16* base = qkv + (size_t)b * T * ld + h * hd;
  const __bf16* dob = d_o + (size_t)b * T * D + h * hd;
  const float* lrow = lse + (size_t)bh * T;
  const float* drow = delta + (size_t)bh * T;
  const float c = scale * LOG2E;

  zero_image<HDP>(smem, 4, tid);
  bf16x8 fk[KT][KS], fv[KT][KS];
#pragma unroll
  for (int ki = 0; ki < KT; ++ki)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      fk[ki][ks] = frag_rows_global(base + D, ld, k0 + 16 * ki, ks, hd, lane);
      fv[ki][ks] = frag_rows_global(base + 2 * D, ld, k0 + 16 * ki, ks, hd, lane);
    }
  __syncthreads();
  ChunkStager<HDP> stQ, stO;
  stQ.load(base, ld, chpr, tid);
  stO.load(dob, D, chpr, tid);
  stQ.store(sQ, chpr, tid);
  stO.store(sO, chpr, tid);
  __syncthreads();

  f32x4 dk[KT][NDT], dv[KT][NDT];
#pragma unroll
  for (int ki = 0; ki < KT; ++ki)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) { dk[ki][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[ki][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nchunk = T / 64;
  for (int qc = 0; qc < nchunk; ++qc) {
    const int cur = qc & 1;
    const bool more = qc + 1 < nchunk;
    if (more) {
      stQ.load(base + (size_t)(qc + 1) * 64 * ld, ld, chpr, tid);
      stO.load(dob + (size_t)(qc + 1) * 64 * D, D, chpr, tid);
    }
    const __bf16* iQ = sQ + cur * 64 * HDP;
    const __bf16* iO = sO + cur * 64 * HDP;
    f32x4 pt[KT][4], st[KT][4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      bf16x8 fqr[KS], fdr[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { fqr[ks] = frag_rows<HDP>(iQ, qt * 16, ks, lane); fdr[ks] = frag_rows<HDP>(iO, qt * 16, ks, lane); }
      const float4 l4 = *reinterpret_cast<const float4*>(lrow + qc * 64 + qt * 16 + 4 * g);
      const float4 d4 = *reinterpret_cast<const float4*>(drow + qc * 64 + qt * 16 + 4 * g);
      const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq_[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int ki = 0; ki < KT; ++ki) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, p = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqr[ks], fk[ki][ks], a, 0, 0, 0);    // S[q=4g+j][key=lane&15]
          p = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fdr[ks], fv[ki][ks], p, 0, 0, 0);    // dP
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pr = exp2f(a[j] * c - lq[j] * LOG2E);
          a[j] = pr;
          p[j] = pr * (p[j] - dq_[j]) * scale;
        }
        pt[ki][qt] = a;
        st[ki][qt] = p;
      }
    }
    bf16x8 pp[KT][2], sp[KT][2];
#pragma unroll
    for (int ki = 0; ki < KT; ++ki) {
      pp[ki][0] = pack_perm(pt[ki][0], pt[ki][1]); pp[ki][1] = pack_perm(pt[ki][2], pt[ki][3]);
      sp[ki][0] = pack_perm(st[ki][0], st[ki][1]); sp[ki][1] = pack_perm(st[ki][2], st[ki][3]);
    }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8 fot = frag_cols_perm<HDP>(iO, 32 * s2, dt * 16, lane);    // dO^T
        const bf16x8 fqt = frag_cols_perm<HDP>(iQ, 32 * s2, dt * 16, lane);    // Q^T
#pragma unroll
        for (int ki = 0; ki < KT; ++ki) {
          dv[ki][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fot, pp[ki][s2], dv[ki][dt], 0, 0, 0);
          dk[ki][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqt, sp[ki][s2], dk[ki][dt], 0, 0, 0);
        }
      }
    if (more) {
      stQ.store(sQ + (cur ^ 1) * 64 * HDP, chpr, tid);
      stO.store(sO + (cur ^ 1) * 64 * HDP, chpr, tid);
    }
    __syncthreads();
  }
#pragma unroll
  for (int ki = 0; ki < KT; ++ki) {
    const int key = k0 + 16 * ki + (lane & 15);
    __bf16* row = dqkv + ((size_t)b * T + key) * ld + h * hd;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const int d = dt * 16 + 4 * g;
      if (d < hd) {
        const f32x4 a = dk[ki][dt], v = dv[ki][dt];
        bf16x4 ka = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
        bf16x4 va = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        *reinterpret_cast<bf16x4*>(row + D + d) = ka;
        *reinterpret_cast<bf16x4*>(row + 2 * D + d) = va;
      }
    }
  }
}

template <int HDP> size_t lds_bytes() { return 4 * 64 * HDP * sizeof(__bf16); }

template <int HDP, int KS, int NDT>
int launch_fwd(const __bf16* qkv, __bf16* o, float* lse, int B, int T, int H, int hd, float scale, hipStream_t s) {
  if (T % 128 == 0)
    hipLaunchKernelGGL((k_attn_fwd<HDP, KS, NDT, 2>), dim3(T / 128, B * H), dim3(NT), lds_bytes<HDP>(), s, qkv, o, lse, T, H, hd, scale);
  else
    hipLaunchKernelGGL((k_attn_fwd<HDP, KS, NDT, 1>), dim3(T / 64, B * H), dim3(NT), lds_bytes<HDP>(), s, qkv, o, lse, T, H, hd, scale);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFRON_OK : (int)e;
}
template <int HDP, int KS, int NDT>
int launch_bwd(const __bf16* qkv, const __bf16* o, const __bf16* d_o, const float* lse, float* delta, __bf16* dqkv, int B,
               int T, int H, int hd, float scale, hipStream_t s) {
  hipLaunchKernelGGL(k_attn_delta, dim3(cdiv((long)B * T * H, NT)), dim3(NT), 0, s, o, d_o, delta, B, T, H, hd);
  if (T % 128 == 0)
    hipLaunchKernelGGL((k_attn_bwd_dq<HDP, KS, NDT, 2>), dim3(T / 128, B * H), dim3(NT), lds_bytes<HDP>(), s, qkv, d_o, lse, delta, dqkv, T, H, hd, scale);
  else
    hipLaunchKernelGGL((k_attn_bwd_dq<HDP, KS, NDT, 1>), dim3(T / 64, B * H), dim3(NT), lds_bytes<HDP>(), s, qkv, d_o, lse, delta, dqkv, T, H, hd, scale);
  hipLaunchKernelGGL((k_attn_bwd_dkv<HDP, KS, NDT, 1>), dim3(T / 64, B * H), dim3(NT), lds_bytes<HDP>(), s, qkv, d_o, lse, delta, dqkv, T, H, hd, scale);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFRON_OK : (int)e;
}

}  // namespace

extern "C" {

int sfron_attn_fwd(const uint16_t* qkv, uint16_t* o, float* lse, int B, int T, int H, int hd, void* stream) {
  SFRON_CHECK_ARG(qkv && o && lse && B > 0 && H > 0 && T > 0 && T % 64 == 0);
  SFRON_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)o) & 15) == 0);
  const float scale = 1.0f / sqrtf((float)hd);
  hipStream_t s = (hipStream_t)stream;
  if (hd == 64) return launch_fwd<64, 2, 4>((const __bf16*)qkv, (__bf16*)o, lse, B, T, H, hd, scale, s);
  if (hd == 72) return launch_fwd<128, 3, 5>((const __bf16*)qkv, (__bf16*)o, lse, B, T, H, hd, scale, s);
  return SFRON_ERR_UNSUPPORTED;
}

int sfron_attn_bwd(const uint16_t* qkv, const uint16_t* o, const uint16_t* d_o, const float* lse, float* delta_scratch,
                   uint16_t* dqkv, int B, int T, int H, int hd, void* stream) {
  SFRON_CHECK_ARG(qkv && o && d_o && lse && delta_scratch && dqkv && B > 0 && H > 0 && T > 0 && T % 64 == 0);
  SFRON_CHECK_ARG((((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dqkv) & 15) == 0);
  const float scale = 1.0f / sqrtf((float)hd);
  hipStream_t s = (hipStream_t)stream;
  if (hd == 64)
    return launch_bwd<64, 2, 4>((const __bf16*)qkv, (const __bf16*)o, (const __bf16*)d_o, lse, delta_scratch, (__bf16*)dqkv, B, T, H, hd, scale, s);
  if (hd == 72)
    return launch_bwd<128, 3, 5>((const __bf16*)qkv, (const __bf16*)o, (const __bf16*)d_o, lse, delta_scratch, (__bf16*)dqkv, B, T, H, hd, scale, s);
  return SFRON_ERR_UNSUPPORTED;
}

}  // extern "C"
