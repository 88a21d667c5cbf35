// BASELINE config 5: the four token GEMMs of a DiT block (DiT/models.py:108-121 forward) on the CDNA4 fp8 matrix core.
//
// Both MFMA operands must be fp8 (v_mfma_scale_f32_16x16x128_f8f6f4 has no bf16 x fp8 form, and a bf16 A tile of a 128-deep step
// would not fit a ring in LDS), so
//   weights      live as fp32 masters (the optimizer's arena) + an e4m3 shadow, one power-of-two scale per tensor
//                (scale = 2^floor(log2(224 / amax)): exact to apply and to undo, 2x headroom under the 448 maximum), re-quantised
//                after every optimizer step in ONE pass with the scale of the previous step's amax (delayed scaling: a weight moves
//                by at most lr per step) while the pass collects the new amax (integer atomicMax on the float bits: order-free);
//   activations  are quantised where they are PRODUCED, with static power-of-two scales per kind: LayerNorm+modulate writes its
//                bf16 output (the backward pass's operand) and the e4m3 copy, the fc1 epilogue writes h in bf16 and e4m3, the
//                attention output gets one cast kernel.
// The backward pass is unchanged (bf16 operands, bf16 weight shadow): the quantisation is a straight-through estimator, as in
// oracle/fp8_ref.py.  e4m3 = OCP e4m3fn (gfx950's v_cvt_pk_fp8_f32), round to nearest even, saturating at +-448.
//
// GEMM kernel: C[M][N] = deq * (A8[M][K] . B8[N][K]^T) with the 256 x 128 x 128-byte tile of tools/probes/fp8_gemm_probe.hip
// (three LDS slots, both operands staged by buffer_load ... lds with the XOR swizzle on the source address, 8 waves of 32 x 128;
// exact-integer verified there), plus the epilogues the block needs: bias -> bf16 (qkv); bias, GELU-tanh, pre-activation + bf16 + e4m3
// outputs (fc1); bias, gate, residual -> fp32 stream + bf16 branch output (proj, fc2).
#include "common.h"
#include <string.h>
#include "../../include/sfron.h"
#include <atomic>

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) void lptr8_t;

constexpr float E4M3_MAX = 448.0f;
constexpr int TPB = 256;

__device__ __forceinline__ float sat8(float x) { return fminf(fmaxf(x, -E4M3_MAX), E4M3_MAX); }

// ---- how much of the e4m3 range the ACTIVATIONS use: the activation scales are static (engine.py: 8 / 32 / 16), the conversion saturates, and
// nothing else would tell a run on real weights that its LN + modulate outputs left the range.  Every kernel that quantises an activation
// keeps the maximum of |x * scale| per thread (one v_max per value), a wave reduces it and ONE lane per wave raises the site's word with
// atomicMax on the bit pattern (non-negative floats order like unsigned integers).  sfron_fp8_activation_amax reads (and resets) the three
// words: a value above 448 means that values were clipped since the last reset.  The three words belong to the CALLER (round 6: a device
// array handed to every quantising entry point, NULL = no tracking; rounds 4-5 kept them in a __device__ global of the library).
// site 0 = LN + modulate output (the A operand of qkv / fc1), 1 = sfron_cast_e4m3 (attention output -> proj), 2 = GELU output (-> fc2)
__device__ __forceinline__ void amax4(float& m, float a, float b, float c, float d) {
  m = fmaxf(m, fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))));
}
__device__ __forceinline__ void amax_report(unsigned int* words, int site, float m) {
  if (!words) return;                          // kernel-uniform
  m = wave_max(m);
  // the atomic only when this wave would RAISE the word: thousands of waves hitting one address serialise in L2 (the unconditional form cost
  // config 5 six ms per step); a stale read of the word can only cause a surplus atomic, never a lost maximum
  if ((threadIdx.x & 63) == 0 && m > __uint_as_float(__builtin_nontemporal_load(&words[site])))
    atomicMax(&words[site], __float_as_uint(m));
}
// four fp32 -> four e4m3 bytes (little endian: a in bits 0..7)
__device__ __forceinline__ uint32_t pack_e4m3(float a, float b, float c, float d) {
  int v = 0;
  v = __builtin_amdgcn_cvt_pk_fp8_f32(sat8(a), sat8(b), v, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(sat8(c), sat8(d), v, true);
  return (uint32_t)v;
}

// ---------------------------------------------------------------- weights: per-tensor amax / quantisation
struct TensorRange { long long off, n; };   // element range of one tensor inside the arena (off, n multiples of 8)

// mode 0: amax only.  mode 1: dst = e4m3(p * scale[t]) AND amax.  grid (chunks, tensors)
__global__ __launch_bounds__(TPB) void k_fp8_quant_tensors(const float* __restrict__ p, const TensorRange* __restrict__ tab,
                                                           const float* __restrict__ scales, unsigned* __restrict__ amax_bits,
                                                           uint8_t* __restrict__ dst, int mode) {
  const int t = blockIdx.y;
  const TensorRange r = tab[t];
  const float s = mode ? scales[t] : 1.0f;
  const float4* src = reinterpret_cast<const float4*>(p + r.off);
  uint32_t* out = reinterpret_cast<uint32_t*>(dst + r.off);
  const long long n4 = r.n >> 2;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) {
    const float4 v = src[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    if (mode) out[i] = pack_e4m3(v.x * s, v.y * s, v.z * s, v.w * s);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax_bits + t, __float_as_uint(m));   // non-negative floats order like their bits
}

// scale = 2^floor(log2(224 / amax)) (1 for an all-zero tensor); amax is cleared for the next pass
__global__ void k_fp8_update_scales(unsigned* __restrict__ amax_bits, int n, float* __restrict__ scales) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float a = __uint_as_float(amax_bits[t]);
  float s = 1.0f;
  if (a > 0.f && isfinite(a)) {
    int e;
    (void)frexpf(224.0f / a, &e);            // 224 / a = f * 2^e, f in [0.5, 1)  ->  floor(log2) = e - 1
    e -= 1;
    e = e < -60 ? -60 : (e > 60 ? 60 : e);
    s = ldexpf(1.0f, e);
  }
  scales[t] = s;
  amax_bits[t] = 0u;
}

__global__ __launch_bounds__(TPB) void k_cast_e4m3_bf16(const __bf16* __restrict__ src, int64_t n, float scale, uint8_t* __restrict__ dst, unsigned int* amax) {
  const int64_t n8 = n >> 3;
  float am = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n8; i += (int64_t)gridDim.x * TPB) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(src)[i];
    const float f0 = bf2f(v[0]) * scale, f1 = bf2f(v[1]) * scale, f2 = bf2f(v[2]) * scale, f3 = bf2f(v[3]) * scale;
    const float f4 = bf2f(v[4]) * scale, f5 = bf2f(v[5]) * scale, f6 = bf2f(v[6]) * scale, f7 = bf2f(v[7]) * scale;
    amax4(am, f0, f1, f2, f3); amax4(am, f4, f5, f6, f7);
    uint2 o;
    o.x = pack_e4m3(f0, f1, f2, f3);
    o.y = pack_e4m3(f4, f5, f6, f7);
    reinterpret_cast<uint2*>(dst)[i] = o;
  }
  amax_report(amax, 1, am);
}
__global__ __launch_bounds__(TPB) void k_cast_e4m3_f32(const float* __restrict__ src, int64_t n, float scale, uint8_t* __restrict__ dst, unsigned int* amax) {
  const int64_t n4 = n >> 2;
  float am = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    amax4(am, v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    reinterpret_cast<uint32_t*>(dst)[i] = pack_e4m3(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
  }
  amax_report(amax, 1, am);
}

// ---------------------------------------------------------------- LayerNorm + modulate with the e4m3 copy (norm.hip's k_ln_mod_fwd + one store)
constexpr int NCHQ = 5;            // row chunks of 256 floats: D <= 1280
// (as norm.hip k_ln_mod_fwd: one-row buffer descriptors -- branch-free --, RPW rows of one sample per wave, every load before the first use,
// the sample's shift / scale rows fetched once per wave)
typedef unsigned int q_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t q_row_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <int RPW>
__global__ __launch_bounds__(TPB) void k_ln_mod_fwd_q(const float* __restrict__ x, const float* __restrict__ shift,
                                                      const float* __restrict__ scale, int ldmod, int T, int M, int D,
                                                      __bf16* __restrict__ out, uint8_t* __restrict__ out8, float s8,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out, unsigned int* amax) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = (blockIdx.x * 4 + wave) * RPW;
  if (row0 >= M) return;
  float am = 0.f;
  float4 v[RPW][NCHQ], sh[NCHQ], sc[NCHQ];
  {
    const int b = row0 / T;
    const __amdgpu_buffer_rsrc_t rh = q_row_rsrc(shift + (size_t)b * ldmod, D * 4), rc = q_row_rsrc(scale + (size_t)b * ldmod, D * 4);
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int row = row0 + k < M ? row0 + k : M - 1;
      const __amdgpu_buffer_rsrc_t rx = q_row_rsrc(x + (size_t)row * D, D * 4);
#pragma unroll
      for (int i = 0; i < NCHQ; ++i) v[k][i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, lane * 16 + 1024 * i, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < NCHQ; ++i) {
      sh[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rh, lane * 16 + 1024 * i, 0, 0));
      sc[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rc, lane * 16 + 1024 * i, 0, 0));
    }
  }
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int row = row0 + k;
    if (row >= M) break;                           // wave-uniform
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCHQ; ++i) s += v[k][i].x + v[k][i].y + v[k][i].z + v[k][i].w;      // zeros past the row end
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCHQ; ++i) {
      const float live = lane + 64 * i < (D >> 2) ? 1.0f : 0.0f;
      const float a = v[k][i].x - mean, b = v[k][i].y - mean, c = v[k][i].z - mean, d = v[k][i].w - mean;
      q += live * (a * a + b * b + c * c + d * d);
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    const __amdgpu_buffer_rsrc_t ro = q_row_rsrc(out + (size_t)row * D, D * 2), r8 = q_row_rsrc(out8 + (size_t)row * D, D);
#pragma unroll
    for (int i = 0; i < NCHQ; ++i) {
      const float4 h = sh[i], g = sc[i];
      const float o0 = (v[k][i].x - mean) * rstd * (1.0f + g.x) + h.x, o1 = (v[k][i].y - mean) * rstd * (1.0f + g.y) + h.y;
      const float o2 = (v[k][i].z - mean) * rstd * (1.0f + g.z) + h.z, o3 = (v[k][i].w - mean) * rstd * (1.0f + g.w) + h.w;
      const bf16x4 o = {f2bf(o0), f2bf(o1), f2bf(o2), f2bf(o3)};
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(q_u32x2, o), ro, lane * 8 + 512 * i, 0, 0);
      if (lane + 64 * i < (D >> 2)) amax4(am, o0 * s8, o1 * s8, o2 * s8, o3 * s8);        // (lanes past the row end hold the shift, not data)
      __builtin_amdgcn_raw_buffer_store_b32(pack_e4m3(o0 * s8, o1 * s8, o2 * s8, o3 * s8), r8, lane * 4 + 256 * i, 0, 0);
    }
  }
  amax_report(amax, 0, am);
}

// ---------------------------------------------------------------- fp8 x fp8 GEMM
struct Gemm8Args {
  const uint8_t* A; const uint8_t* B;     // [M][K], [N][K] e4m3
  int M, N, K;
  const float* w_scale;                   // device scalar: the weight tensor's quantisation scale
  float a_scale;                          // the activation's (static)
  const float* bias;                      // [N] fp32 or null
  __bf16* Cb; int ldcb;                   // bf16 output (EPI_BF16: result; EPI_GELU: gelu(result))
  __bf16* aux; int ldaux;                 // EPI_GELU: pre-activation; EPI_GATE_RES: branch output
  uint8_t* C8; float c8_scale;            // EPI_GELU: e4m3(gelu(result) * c8_scale) -- the A operand of fc2
  float* Cf; int ldcf; const float* resid;      // EPI_GATE_RES: Cf = resid + gate * result
  const float* gate; int ldgate; int T;
  unsigned int* act_amax;                 // the caller's activation-range words (site 2 is raised by EPI_GELU's e4m3 output), or null
  int aux_q;                              // EPI_GELU: aux = GELU'(pre-activation) as one byte per element (common.h geluq_pack4), uint8 [M][ldaux]
};
enum { E8_BF16 = 0, E8_GELU = 2, E8_GATE_RES = 3 };

constexpr int FBM = 256, BKB = 128, NW = 8, NSLOT8 = 3;
constexpr int A_BYTES = FBM * BKB, NA = FBM * 8 / 64 / NW;
// NTL = 16-column n-tiles per workgroup tile: 8 (256 x 128) or 9 (256 x 144 -- a [8192 x 1152] output is then exactly 256 tiles, as
// for the bf16 kernels; its 18 B-tile DMA instructions do not divide by the 8 waves: the waves without a third share send theirs
// through a zero-length descriptor to a dummy LDS kilobyte, so every wave issues the same number and the counted vmcnt waits hold)
template <int NTL> struct Tile8 {
  static constexpr int FBN = NTL * 16, B_BYTES = FBN * BKB, SLOT = A_BYTES + B_BYTES;
  static constexpr int NB_TOTAL = FBN * 8 / 64, NB = (NB_TOTAL + NW - 1) / NW, NDMA = NA + NB;
  static constexpr bool EVEN = NB_TOTAL % NW == 0;
  static constexpr size_t LDS = (size_t)NSLOT8 * SLOT + (EVEN ? 0 : 1024);
};

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma16b(__amdgpu_buffer_rsrc_t rsrc, uint8_t* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr8_t*)dst, 16, voff, soff, 0, 0);
}
// 32 consecutive k-bytes of row `row`, k group g: chunks 2g and 2g + 1 of the 128-byte row, chunk ^= row & 7
__device__ __forceinline__ i32x8 frag32(const uint8_t* img, int row, int g) {
  const i32x4 lo = *reinterpret_cast<const i32x4*>(img + row * 128 + (((2 * g) ^ (row & 7)) << 4));
  const i32x4 hi = *reinterpret_cast<const i32x4*>(img + row * 128 + (((2 * g + 1) ^ (row & 7)) << 4));
  return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

}  // namespace

// NL > 0: NL extra LOADER waves issue every LDS-DMA piece of the ring; the eight multiplying waves none (as csrc/gemm.hip k_gemm_pipe NL;
// tools/probes/persist_gemm_probe.hip measured this compiler-scheduled loop 13-22 % faster that way in bf16).  Twelve waves: <= 168
// registers, so the consumers fetch the bias after the main loop.  With e4m3 operands a K-step carries half the DMA pieces per FLOP
// and the split bought nothing in the step (g_fp8_loader_waves): kept as a switchable form, off by default.
template <int EPI, int NTL, int NL = 0>
__global__ __launch_bounds__(512 + 64 * NL) void k_gemm8(Gemm8Args g) {
  using TL = Tile8<NTL>;
  constexpr int FBN = TL::FBN, SLOT = TL::SLOT, NT8 = NTL;
  constexpr int NWD = NL > 0 ? NL : NW;                                  // waves that issue LDS-DMA
  constexpr int NA_ = FBM * 8 / 64 / NWD, NB = (TL::NB_TOTAL + NWD - 1) / NWD, NDMA8 = NA_ + NB;
  constexpr bool EVEN_ = TL::NB_TOTAL % NWD == 0;
  constexpr size_t DUMMY_AT = (size_t)NSLOT8 * SLOT;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem8[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = g.N / FBN;
  int id;
  {
    const int nblk = gridDim.x, b = blockIdx.x, q = nblk >> 3, r = nblk & 7, x = b & 7, y = b >> 3;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  const int tm = id / ntn, tn = id - tm * ntn, m0 = tm * FBM, n0 = tn * FBN;
  const int K = g.K, nk = K / BKB;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, g.M * K, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, g.N * K, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsNull = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, 0, 0x00020000);
  uint8_t* const dummy = smem8 + DUMMY_AT;
  const int lc16 = ((lane & 7) ^ ((lane >> 3) & 7)) << 4;
  const int dwave = NL > 0 ? wave - NW : wave;                       // (consumers of the loader form never use their plan)
  int a_off[NA_], b_off[NB];
#pragma unroll
  for (int i = 0; i < NA_; ++i) a_off[i] = (m0 + (dwave + i * NWD) * 8 + (lane >> 3)) * K + lc16;
#pragma unroll
  for (int i = 0; i < NB; ++i) b_off[i] = (n0 + (dwave + i * NWD) * 8 + (lane >> 3)) * K + lc16;
  auto issue = [&](int slot, int k0) {
    uint8_t* iA = smem8 + slot * SLOT;
    uint8_t* iB = iA + A_BYTES;
#pragma unroll
    for (int i = 0; i < NA_; ++i) dma16b(rsA, iA + (dwave + i * NWD) * 1024, a_off[i], k0);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if constexpr (EVEN_) dma16b(rsB, iB + (dwave + i * NWD) * 1024, b_off[i], k0);
      else {
        const bool ok = dwave + i * NWD < TL::NB_TOTAL;             // wave-uniform
        dma16b(ok ? rsB : rsNull, ok ? iB + (dwave + i * NWD) * 1024 : dummy, b_off[i], k0);
      }
    }
  };
  if constexpr (NL > 0) {
    if (wave >= NW) {                                                // ---- loader wave
      if (nk > 0) issue(0, 0);
      if (nk > 1) issue(1, BKB);
      int slot2 = 2;
      for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vm<NDMA8>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(slot2, (kt + 2) * BKB);
        slot2 = slot2 == 2 ? 0 : slot2 + 1;
      }
      return;
    }
  }
  f32x4 acc[2][NT8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  // this lane's output columns: bias fetched ahead of the main loop (loader form: behind it, the registers are short)
  float4 bias_v[NT8];
  auto load_bias = [&]() {
#pragma unroll
    for (int nt = 0; nt < NT8; ++nt)
      bias_v[nt] = g.bias ? *reinterpret_cast<const float4*>(g.bias + n0 + nt * 16 + 4 * fg) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  if constexpr (NL == 0) load_bias();
  const float deq = 1.0f / (g.a_scale * *g.w_scale);
  if constexpr (NL == 0) {
    if (nk > 0) issue(0, 0);
    if (nk > 1) issue(1, BKB);
  }
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (NL == 0) { if (kt + 1 < nk) wait_vm<NDMA8>(); else wait_vm<0>(); }
    __builtin_amdgcn_s_barrier();
    if constexpr (NL == 0) { if (kt + 2 < nk) issue(slot >= 1 ? slot - 1 : 2, (kt + 2) * BKB); }
    const uint8_t* iA = smem8 + slot * SLOT;
    const uint8_t* iB = iA + A_BYTES;
    const i32x8 fa0 = frag32(iA, wave * 32 + fr, fg), fa1 = frag32(iA, wave * 32 + 16 + fr, fg);
#pragma unroll
    for (int nt = 0; nt < NT8; ++nt) {
      const i32x8 fb = frag32(iB, nt * 16 + fr, fg);
      // cbsz = blgp = 0: both operands e4m3; block scales 2^(127 - 127) = 1 (the tensor scales are undone in the epilogue)
      acc[0][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa0, acc[0][nt], 0, 0, 0, 127, 0, 127);
      acc[1][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa1, acc[1][nt], 0, 0, 0, 127, 0, 127);
    }
    slot = slot == 2 ? 0 : slot + 1;
  }
  constexpr bool BIAS_BY_HALF = NL > 0 && EPI == E8_GATE_RES;       // (that epilogue also holds gate + residual words: bias with them, half a row at a time)
  if constexpr (NL > 0 && !BIAS_BY_HALF) load_bias();
  // lane holds C[m0 + 32 wave + 16 mt + fr][n0 + 16 nt + 4 fg .. +3]
  [[maybe_unused]] float act_am = 0.f;          // E8_GELU: max |gelu * c8_scale| of this thread's values (g_act_amax site 2)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = m0 + wave * 32 + mt * 16 + fr;
    // gated residual: the row's gate and residual words are fetched for all NT8 tiles before the first store (resid may alias Cf, so hipcc
    // keeps each load behind the previous store: one exposed latency per tile otherwise)
    // (in two halves: 2 x NT8 float4 on top of the accumulators do not fit the 168 registers of the loader form)
    constexpr int NH = (NT8 + 1) / 2;
    float4 gtv[EPI == E8_GATE_RES ? NH : 1], xrv[EPI == E8_GATE_RES ? NH : 1];
    bf16x4 ob[NT8], ab[NT8];
    unsigned c8[NT8];
    [[maybe_unused]] unsigned q8[EPI == E8_GELU ? NT8 : 1];
#pragma unroll
    for (int nt = 0; nt < NT8; ++nt) {
      if constexpr (EPI == E8_GATE_RES) {
        if (nt % NH == 0) {
#pragma unroll
          for (int u = 0; u < NH; ++u) {
            if (nt + u < NT8) {
              const int colu = n0 + (nt + u) * 16 + 4 * fg;
              gtv[u] = *reinterpret_cast<const float4*>(g.gate + (size_t)(row / g.T) * g.ldgate + colu);
              xrv[u] = *reinterpret_cast<const float4*>(g.resid + (size_t)row * g.ldcf + colu);
              if constexpr (BIAS_BY_HALF) bias_v[u] = g.bias ? *reinterpret_cast<const float4*>(g.bias + colu) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
        }
      }
      const int col = n0 + nt * 16 + 4 * fg;
      f32x4 v = acc[mt][nt] * deq;
      const float4 bv = bias_v[BIAS_BY_HALF ? nt % NH : nt];
      v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      // bf16 / e4m3 results are kept per tile and leave in PAIRS of tiles after the loop (common.h pair_pack: 16-byte bf16 pieces, 8-byte
      // e4m3 pieces; the store tail is issue-bound)
      if constexpr (EPI == E8_BF16) {
        ob[nt] = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
      } else if constexpr (EPI == E8_GELU) {
        float h0, h1, h2, h3;
        if (g.aux_q) {                              // kernel-uniform: the backward pass reads GELU'(v) as bytes (SFRON_EPI_DGELU_Q)
          f32x4 y, dy;
          gelu_tanh_both4(v, y, dy);
          h0 = y[0]; h1 = y[1]; h2 = y[2]; h3 = y[3];
          q8[nt] = geluq_pack4(dy);
        } else {
          ab[nt] = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
          h0 = gelu_tanh(v[0]); h1 = gelu_tanh(v[1]); h2 = gelu_tanh(v[2]); h3 = gelu_tanh(v[3]);
        }
        ob[nt] = bf16x4{f2bf(h0), f2bf(h1), f2bf(h2), f2bf(h3)};
        amax4(act_am, h0 * g.c8_scale, h1 * g.c8_scale, h2 * g.c8_scale, h3 * g.c8_scale);
        c8[nt] = pack_e4m3(h0 * g.c8_scale, h1 * g.c8_scale, h2 * g.c8_scale, h3 * g.c8_scale);
      } else {
        ab[nt] = bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        const float4 gt = gtv[nt % NH];
        float4 x = xrv[nt % NH];
        x.x += gt.x * v[0]; x.y += gt.y * v[1]; x.z += gt.z * v[2]; x.w += gt.w * v[3];
        *reinterpret_cast<float4*>(g.Cf + (size_t)row * g.ldcf + col) = x;
      }
    }
    const int colp = n0 + pair_col(fg);
    auto put = [&](__bf16* base, int ld, const bf16x4 (&t)[NT8]) {
      __bf16* const r = base + (size_t)row * ld;
      if ((ld & 7) == 0) {
#pragma unroll
        for (int np = 0; np < NT8 / 2; ++np) *reinterpret_cast<uint4*>(r + colp + np * 32) = pair_pack(t[2 * np], t[2 * np + 1]);
        if constexpr (NT8 & 1) *reinterpret_cast<bf16x4*>(r + n0 + (NT8 - 1) * 16 + 4 * fg) = t[NT8 - 1];
      } else {
#pragma unroll
        for (int nt = 0; nt < NT8; ++nt) *reinterpret_cast<bf16x4*>(r + n0 + nt * 16 + 4 * fg) = t[nt];
      }
    };
    if constexpr (EPI == E8_BF16 || EPI == E8_GELU) put(g.Cb, g.ldcb, ob);
    if constexpr (EPI == E8_GATE_RES) put(g.aux, g.ldaux, ab);
    if constexpr (EPI == E8_GELU) {
      if (g.aux_q) {
        uint8_t* const rq = reinterpret_cast<uint8_t*>(g.aux) + (size_t)row * g.ldaux;
#pragma unroll
        for (int np = 0; np < NT8 / 2; ++np) *reinterpret_cast<uint2*>(rq + colp + np * 32) = pair_pack8(q8[2 * np], q8[2 * np + 1]);
        if constexpr (NT8 & 1) *reinterpret_cast<uint32_t*>(rq + n0 + (NT8 - 1) * 16 + 4 * fg) = q8[NT8 - 1];
      } else put(g.aux, g.ldaux, ab);
    }
    if constexpr (EPI == E8_GELU) {
      uint8_t* const r8 = g.C8 + (size_t)row * g.N;
#pragma unroll
      for (int np = 0; np < NT8 / 2; ++np) {
        unsigned x = c8[2 * np], y = c8[2 * np + 1];
        permlane16_swap(x, y);                       // this lane: 8 contiguous e4m3 columns of the pair (the same piece map as the bf16 form)
        *reinterpret_cast<uint2*>(r8 + colp + np * 32) = make_uint2(x, y);
      }
      if constexpr (NT8 & 1) *reinterpret_cast<uint32_t*>(r8 + n0 + (NT8 - 1) * 16 + 4 * fg) = c8[NT8 - 1];
    }
  }
  if constexpr (EPI == E8_GELU) amax_report(g.act_amax, 2, act_am);
}
template __global__ void k_gemm8<E8_BF16, 8>(Gemm8Args);
template __global__ void k_gemm8<E8_GELU, 8>(Gemm8Args);
template __global__ void k_gemm8<E8_GATE_RES, 8>(Gemm8Args);
template __global__ void k_gemm8<E8_BF16, 9>(Gemm8Args);
template __global__ void k_gemm8<E8_GELU, 9>(Gemm8Args);
template __global__ void k_gemm8<E8_GATE_RES, 9>(Gemm8Args);
template __global__ void k_gemm8<E8_BF16, 8, 4>(Gemm8Args);
template __global__ void k_gemm8<E8_GELU, 8, 4>(Gemm8Args);
template __global__ void k_gemm8<E8_GATE_RES, 8, 4>(Gemm8Args);
template __global__ void k_gemm8<E8_BF16, 9, 4>(Gemm8Args);
template __global__ void k_gemm8<E8_GELU, 9, 4>(Gemm8Args);
template __global__ void k_gemm8<E8_GATE_RES, 9, 4>(Gemm8Args);

int g_fp8_loader_waves = 0;     // process-wide form of the fp8 tiles: 4 = loader waves (sfron_gemm_loader_waves(9)); measured in the config-5 step:
                                // 65.2 / 64.0 / 64.8 ms against 63.5 / 64.1 / 64.1 for the shared-wave form -- off

namespace {
template <int EPI, int NTL, int NL>
int launch8t(const Gemm8Args& g, hipStream_t s) {
  const size_t lds = (size_t)NSLOT8 * Tile8<NTL>::SLOT + 1024;      // (+ the dummy kilobyte of uneven DMA plans)
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if ((done.fetch_or(bit) & bit) == 0) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm8<EPI, NTL, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return (int)hipGetLastError();
  }
  hipLaunchKernelGGL((k_gemm8<EPI, NTL, NL>), dim3((g.M / FBM) * (g.N / (NTL * 16))), dim3(512 + 64 * NL), lds, s, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFRON_OK : (int)e;
}
// tile width: the one that fills the last round of 256 CUs better (ties: the wider tile, fewer operand bytes per FLOP)
inline int pick_ntl(int M, int N) {
  auto eff = [&](int w) {
    if (N % w) return 0.0;
    const long tiles = (long)(M / FBM) * (N / w);
    return (double)tiles / (double)(((tiles + 255) / 256) * 256);
  };
  const double e8 = eff(128), e9 = eff(144);
  return e9 >= e8 && e9 > 0.0 ? 9 : 8;
}
template <int EPI>
int launch8(const Gemm8Args& g, hipStream_t s) {
  if (g_fp8_loader_waves == 4) return pick_ntl(g.M, g.N) == 9 ? launch8t<EPI, 9, 4>(g, s) : launch8t<EPI, 8, 4>(g, s);
  return pick_ntl(g.M, g.N) == 9 ? launch8t<EPI, 9, 0>(g, s) : launch8t<EPI, 8, 0>(g, s);
}
inline int grid_for(int64_t items) { int64_t b = (items + TPB - 1) / TPB; return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b)); }
}  // namespace

extern "C" {

int sfron_fp8_gemm_supported(int M, int N, int K) {
  return (M > 0 && N > 0 && K > 0 && M % FBM == 0 && (N % 128 == 0 || N % 144 == 0) && K % BKB == 0) ? 1 : 0;
}

int sfron_fp8_gemm(const sfron_fp8_gemm_desc* d, void* stream) {
  SFRON_CHECK_ARG(d && d->A && d->B && d->w_scale && d->a_scale > 0.f);
  if (!sfron_fp8_gemm_supported(d->M, d->N, d->K)) return SFRON_ERR_UNSUPPORTED;
  SFRON_CHECK_ARG((long)d->M * d->K < (1L << 31) && (long)d->N * d->K < (1L << 31));        // 32-bit buffer offsets
  SFRON_CHECK_ARG((((uintptr_t)d->A | (uintptr_t)d->B) & 15) == 0);
  Gemm8Args g{};
  g.A = d->A; g.B = d->B; g.M = d->M; g.N = d->N; g.K = d->K; g.w_scale = d->w_scale; g.a_scale = d->a_scale; g.bias = d->bias;
  g.Cb = (__bf16*)d->c_bf16; g.ldcb = d->ldc_bf16; g.aux = (__bf16*)d->aux; g.ldaux = d->ldaux; g.C8 = d->c_e4m3; g.c8_scale = d->c_e4m3_scale;
  g.Cf = d->c_f32; g.ldcf = d->ldc_f32; g.resid = d->resid ? d->resid : d->c_f32; g.gate = d->gate; g.ldgate = d->ldgate;
  g.T = d->tokens > 0 ? d->tokens : 1;
  g.act_amax = d->act_amax;
  g.aux_q = d->aux_q;
  if (d->aux_q) SFRON_CHECK_ARG(d->epilogue == SFRON_EPI_GELU && d->ldaux % 8 == 0 && ((uintptr_t)d->aux & 7) == 0);
  hipStream_t s = (hipStream_t)stream;
  switch (d->epilogue) {
    case SFRON_EPI_BF16:
      SFRON_CHECK_ARG(g.Cb && g.ldcb % 4 == 0);
      return launch8<E8_BF16>(g, s);
    case SFRON_EPI_GELU:
      SFRON_CHECK_ARG(g.Cb && g.aux && g.C8 && g.c8_scale > 0.f && g.ldcb % 4 == 0 && g.ldaux % 4 == 0);
      return launch8<E8_GELU>(g, s);
    case SFRON_EPI_GATE_RES:
      SFRON_CHECK_ARG(g.Cf && g.aux && g.gate && g.ldcf % 4 == 0 && g.ldaux % 4 == 0 && g.ldgate % 4 == 0);
      return launch8<E8_GATE_RES>(g, s);
    default:
      return SFRON_ERR_UNSUPPORTED;
  }
}

int sfron_fp8_quant_tensors(const float* params, const int64_t* table, int n_tensors, const float* scales, uint32_t* amax_bits,
                            uint8_t* dst, int mode, void* stream) {
  SFRON_CHECK_ARG(params && table && amax_bits && n_tensors > 0 && (mode == 0 || (mode == 1 && scales && dst)));
  SFRON_CHECK_ARG((((uintptr_t)params) & 15) == 0 && (!dst || ((uintptr_t)dst & 3) == 0));
  static_assert(sizeof(TensorRange) == 2 * sizeof(int64_t), "table layout");
  hipLaunchKernelGGL(k_fp8_quant_tensors, dim3(64, n_tensors), dim3(TPB), 0, (hipStream_t)stream, params, (const TensorRange*)table, scales,
                     amax_bits, dst, mode);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_fp8_update_scales(uint32_t* amax_bits, int n_tensors, float* scales, void* stream) {
  SFRON_CHECK_ARG(amax_bits && scales && n_tensors > 0);
  hipLaunchKernelGGL(k_fp8_update_scales, dim3(cdiv(n_tensors, 64)), dim3(64), 0, (hipStream_t)stream, amax_bits, n_tensors, scales);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_fp8_activation_amax(uint32_t* act_amax, float* out3, int reset, void* stream) {
  SFRON_CHECK_ARG(act_amax && out3);
  hipStream_t s = (hipStream_t)stream;
  unsigned int bits[3] = {0u, 0u, 0u};
  if (hipMemcpyAsync(bits, act_amax, sizeof(bits), hipMemcpyDeviceToHost, s) != hipSuccess) return (int)hipGetLastError();
  if (hipStreamSynchronize(s) != hipSuccess) return (int)hipGetLastError();
  for (int i = 0; i < 3; ++i) { float f; memcpy(&f, &bits[i], sizeof(f)); out3[i] = f; }
  if (reset && hipMemsetAsync(act_amax, 0, sizeof(bits), s) != hipSuccess) return (int)hipGetLastError();
  return SFRON_OK;
}

int sfron_cast_e4m3(const void* src, int src_is_bf16, int64_t n, float scale, uint8_t* dst, uint32_t* act_amax, void* stream) {
  SFRON_CHECK_ARG(src && dst && n >= 0 && n % 8 == 0 && scale > 0.f && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
  if (src_is_bf16)
    hipLaunchKernelGGL(k_cast_e4m3_bf16, dim3(grid_for(n >> 3)), dim3(TPB), 0, (hipStream_t)stream, (const __bf16*)src, n, scale, dst, act_amax);
  else
    hipLaunchKernelGGL(k_cast_e4m3_f32, dim3(grid_for(n >> 2)), dim3(TPB), 0, (hipStream_t)stream, (const float*)src, n, scale, dst, act_amax);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_ln_modulate_fwd_q(const float* x, const float* shift, const float* scale, int ldmod, int tokens, int M, int D, uint16_t* out,
                            uint8_t* out_e4m3, float e4m3_scale, float* mean, float* rstd, uint32_t* act_amax, void* stream) {
  SFRON_CHECK_ARG(x && shift && scale && out && out_e4m3 && mean && rstd && M > 0 && tokens > 0 && e4m3_scale > 0.f);
  SFRON_CHECK_ARG(D % 4 == 0 && D <= 64 * 4 * NCHQ && ldmod % 4 == 0);
  SFRON_CHECK_ARG((((uintptr_t)x | (uintptr_t)shift | (uintptr_t)scale) & 15) == 0 && ((uintptr_t)out & 7) == 0 && ((uintptr_t)out_e4m3 & 3) == 0);
  if (M >= 8192 && tokens % 4 == 0)
    hipLaunchKernelGGL(k_ln_mod_fwd_q<4>, dim3(cdiv(M, 16)), dim3(TPB), 0, (hipStream_t)stream, x, shift, scale, ldmod, tokens, M, D, (__bf16*)out,
                     out_e4m3, e4m3_scale, mean, rstd, act_amax);
  else if (M >= 4096 && tokens % 2 == 0)
    hipLaunchKernelGGL(k_ln_mod_fwd_q<2>, dim3(cdiv(M, 8)), dim3(TPB), 0, (hipStream_t)stream, x, shift, scale, ldmod, tokens, M, D, (__bf16*)out,
                     out_e4m3, e4m3_scale, mean, rstd, act_amax);
  else
    hipLaunchKernelGGL(k_ln_mod_fwd_q<1>, dim3(cdiv(M, 4)), dim3(TPB), 0, (hipStream_t)stream, x, shift, scale, ldmod, tokens, M, D, (__bf16*)out,
                     out_e4m3, e4m3_scale, mean, rstd, act_amax);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

}  // extern "C"
