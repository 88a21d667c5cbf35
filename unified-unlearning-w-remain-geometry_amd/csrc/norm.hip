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
#include <cstdlib>
#include "../../include/sfron.h"

namespace {

constexpr int TPB = 256;          // 4 waves, one token row per wave at a time
constexpr int NCH = 5;            // float4 chunks per lane: supports D <= 64*4*NCH = 1280
constexpr float LN_EPS = 1e-6f;

struct RowRegs { float4 v[NCH]; };

// Row loads are UNCONDITIONAL (column clamped to the last chunk; lanes past the row end are zeroed by a select): a load inside an
// `if (c < D4)` body is followed by its first use in the same basic block, and hipcc then waits for EACH load before it issues the next
// (measured on the row-backward kernel: one exposed memory latency per 16-byte chunk, ~12 per row).
__device__ __forceinline__ void load_row_f32(const float* __restrict__ p, int D4, int lane, RowRegs& r) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    r.v[i] = reinterpret_cast<const float4*>(p)[c < D4 ? c : D4 - 1];
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (lane + 64 * i >= D4) r.v[i] = make_float4(0, 0, 0, 0);
}
// Buffer forms: the descriptor covers exactly ONE row, so a lane past the row end (the fifth 16-byte chunk of a 1152-wide row has 32
// live lanes) loads zeros and its stores are dropped by the hardware bounds check -- no exec masks, no branches, addresses = one
// scalar descriptor + one lane offset + immediates.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const void* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
struct RowRaw { bf16x4 v[NCH]; };     // a bf16 row as loaded (converted where it is used: half the registers while in flight)
__device__ __forceinline__ void load_raw_bf16(const __bf16* __restrict__ p, int D4, int lane, RowRaw& r) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + 64 * i;
    r.v[i] = reinterpret_cast<const bf16x4*>(p)[c < D4 ? c : D4 - 1];
  }
}
__device__ __forceinline__ float4 raw4(const RowRaw& r, int i) { return make_float4(bf2f(r.v[i][0]), bf2f(r.v[i][1]), bf2f(r.v[i][2]), bf2f(r.v[i][3])); }
__device__ __forceinline__ void load_row_bf16(const __bf16* __restrict__ p, int D4, int lane, RowRegs& r) {
  RowRaw raw;
  load_raw_bf16(p, D4, lane, raw);
#pragma unroll
  for (int i = 0; i < NCH; ++i) r.v[i] = lane + 64 * i < D4 ? raw4(raw, i) : make_float4(0, 0, 0, 0);
}

// ---------------------------------------------------------------- LN + modulate forward
// RPW rows per wave: both rows' loads are issued before the first reduction (a wave that lives for one 4.6 KB row spends most of its life
// in launch + one exposed memory latency)
template <int RPW>
__global__ __launch_bounds__(TPB) void k_ln_mod_fwd(const float* __restrict__ x, const float* __restrict__ shift,
                                                    const float* __restrict__ scale, int ldmod, int T, int M, int D,
                                                    __bf16* __restrict__ out, float* __restrict__ mean_out,
                                                    float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row0 = (blockIdx.x * 4 + wave) * RPW;
  if (row0 >= M) return;
  // branch-free: one-row buffer descriptors (see row_rsrc), every load of the wave's rows before the first use.  The shift / scale rows
  // of the sample (2 x 4.6 KB against 4.6 KB of x per token row) are fetched ONCE per wave: the launcher picks RPW so that the wave's
  // rows belong to one sample (RPW | tokens).
  RowRegs r[RPW], sh, sc;
  {
    const int b = row0 / T;
    const __amdgpu_buffer_rsrc_t rh = row_rsrc(shift + (size_t)b * ldmod, D * 4), rc = row_rsrc(scale + (size_t)b * ldmod, D * 4);
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int row = row0 + k < M ? row0 + k : M - 1;
      const __amdgpu_buffer_rsrc_t rx = row_rsrc(x + (size_t)row * D, D * 4);
#pragma unroll
      for (int i = 0; i < NCH; ++i) r[k].v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, lane * 16 + 1024 * i, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      sh.v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rh, lane * 16 + 1024 * i, 0, 0));
      sc.v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rc, lane * 16 + 1024 * i, 0, 0));
    }
  }
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int row = row0 + k;
    if (row >= M) break;                          // wave-uniform
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += r[k].v[i].x + r[k].v[i].y + r[k].v[i].z + r[k].v[i].w;     // zeros past the row end
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const float live = lane + 64 * i < (D >> 2) ? 1.0f : 0.0f;
      const float a = r[k].v[i].x - mean, b = r[k].v[i].y - mean, c = r[k].v[i].z - mean, d = r[k].v[i].w - mean;
      q += live * (a * a + b * b + c * c + d * d);
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + LN_EPS);
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    const __amdgpu_buffer_rsrc_t ro = row_rsrc(out + (size_t)row * D, D * 2);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const float4 h = sh.v[i], g = sc.v[i];
      const bf16x4 o = {f2bf((r[k].v[i].x - mean) * rstd * (1.0f + g.x) + h.x), f2bf((r[k].v[i].y - mean) * rstd * (1.0f + g.y) + h.y),
                        f2bf((r[k].v[i].z - mean) * rstd * (1.0f + g.z) + h.z), f2bf((r[k].v[i].w - mean) * rstd * (1.0f + g.w) + h.w)};
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), ro, lane * 8 + 512 * i, 0, 0);
    }
  }
}

// ---------------------------------------------------------------- row-wise backward (LN+modulate, gated residual, or both)
// One wave walks `rpw` consecutive token rows; a workgroup = 4 waves = one partial chunk of 4*rpw rows when `combine`
// (the four waves' column partials are summed through LDS in a fixed order), else one chunk per wave.  The sums over the
// chunks of a sample are taken later in ONE launch for the whole backward pass (k_reduce_slots): no float atomics.
//   LN   : dx (fp32, in/out) (+)= d LN-path of dxmod;  p_shift[chunk][D] = sum_rows dxmod, p_scale = sum_rows dxmod*xhat
//   GATE : d_branch (bf16) = dx * gate[b];              p_gate[chunk][D] = sum_rows dx*branch, p_dy = sum_rows dx
// LN && GATE is the fusion of "LN backward of branch k" with "gate backward of branch k-1" on the freshly accumulated dx
// row, which saves the re-read of dx (fp32 [M][D]) between the two.
struct RowBwdArgs {
  const __bf16* dxmod; const float* x; const float* mean; const float* rstd; const float* scale;
  float* p_shift; float* p_scale; int dx_accumulate;
  const __bf16* branch; const float* gate; __bf16* d_branch; float* p_gate; float* p_dy;
  float* dx;                      // LN: in/out; GATE only: read
  int ldmod_ln, ldmod_gate, T, M, D, rpw, combine;
};

// At most 192 registers (amdgpu_num_vgpr counts half of the unified file on gfx90a+): one workgroup (one wave per SIMD) still fits beside a
// weight-gradient workgroup (2 x 160 registers per SIMD) on the same CU -- left alone hipcc takes 256 and the kernel then runs only on
// CUs without one
template <bool LN, bool GATE, bool COMBINE>
#ifndef SFRON_ROWBWD_VGPR
#define SFRON_ROWBWD_VGPR 96
#endif
__global__ __launch_bounds__(TPB) __attribute__((amdgpu_num_vgpr(SFRON_ROWBWD_VGPR))) void k_row_bwd(const RowBwdArgs a) {
  constexpr int NACC = (LN ? 2 : 0) + (GATE ? 2 : 0);
  // 12 KB of LDS, so that a workgroup still fits on a CU next to a weight-gradient GEMM workgroup (3 x 48 KB slots):
  // these kernels run beside the side stream's GEMMs and are memory-bound, extra resident waves are what they need
  constexpr int HALF = 3;                          // chunks 0..2 (192 float4 per wave) then chunks 3..4
  __shared__ float4 sh[4][64 * HALF];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int D = a.D, D4 = D >> 2;
  const int wchunk = blockIdx.x * 4 + wave;      // the rpw consecutive rows of this wave
  const int row0 = wchunk * a.rpw;
  // combine: the 16 rows of the workgroup belong to ONE sample -- its scale / gate rows (4.6 KB each, read by every row) live in LDS
  // for the row loop (the buffer is free until the final reduction); otherwise (token counts not a multiple of 4) each wave reads
  // its own sample's rows from global memory
  float4* const s_sc = &sh[0][0];
  float4* const s_gt = s_sc + 64 * NCH;
  static_assert(2 * 64 * NCH <= 4 * 64 * HALF, "scale + gate rows fit the reduction buffer");
  if constexpr (COMBINE) {
    const int bw = (blockIdx.x * 4 * a.rpw) / a.T;
    for (int c = threadIdx.x; c < 64 * NCH; c += TPB) {        // zero past the row end: lanes there multiply it with zeros
      const int cl = c < D4 ? c : D4 - 1;
      const float keep = c < D4 ? 1.0f : 0.0f;
      if constexpr (LN) {
        const float4 v = reinterpret_cast<const float4*>(a.scale + (size_t)bw * a.ldmod_ln)[cl];
        s_sc[c] = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
      }
      if constexpr (GATE) {
        const float4 v = reinterpret_cast<const float4*>(a.gate + (size_t)bw * a.ldmod_gate)[cl];
        s_gt[c] = make_float4(v.x * keep, v.y * keep, v.z * keep, v.w * keep);
      }
    }
    __syncthreads();
  }
  if (!COMBINE && row0 >= a.M) return;         // combine: M % (4*rpw) == 0, every wave has rows
  const int b = row0 / a.T;                      // all rows of a chunk belong to one sample
  float4 acc[NACC][NCH];
#pragma unroll
  for (int k = 0; k < NACC; ++k)
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc[k][i] = make_float4(0, 0, 0, 0);
  const float* sc = LN ? a.scale + (size_t)b * a.ldmod_ln : nullptr;
  const float* gp = GATE ? a.gate + (size_t)b * a.ldmod_gate : nullptr;
  // The row loop is BRANCH-FREE (buffer loads / stores, see row_rsrc).  With `if (c < D4)` bodies every chunk was its own basic block
  // and hipcc's wait-count pass drained vmcnt(0) in front of each use: every store of a row waited for the store before it, every
  // bf16 load for the load before it (~12 exposed memory latencies per row; now one).
  auto ldf = [&](const float* p, RowRegs& r) {
    const __amdgpu_buffer_rsrc_t rs = row_rsrc(p, D * 4);
#pragma unroll
    for (int i = 0; i < NCH; ++i) r.v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + 1024 * i, 0, 0));
  };
  auto ldb = [&](const __bf16* p, RowRaw& r) {
    const __amdgpu_buffer_rsrc_t rs = row_rsrc(p, D * 2);
#pragma unroll
    for (int i = 0; i < NCH; ++i) r.v[i] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8 + 512 * i, 0, 0));
  };
  // scale / gate element of chunk i (a lane past the row end reads a defined word: the LDS buffer is 64 * NCH chunks long; global: clamped)
  auto mod4 = [&](const float4* lds_row, const float* grow, int i) {
    const int c = lane + 64 * i;
    if constexpr (COMBINE) return lds_row[c]; else return reinterpret_cast<const float4*>(grow)[c < D4 ? c : D4 - 1];
  };
  for (int rr = 0; rr < a.rpw; ++rr) {
    const int row = row0 + rr;
    // every load of the row goes out before the first use: ONE exposed memory latency per row
    RowRaw braw, draw;
    RowRegs xr, prev;                            // LN: x, then xhat | the dx row before this kernel (LN: if accumulate)
    float mean = 0.f, rstd = 0.f, m1 = 0.f, m2 = 0.f;
    if constexpr (LN) { mean = a.mean[row]; rstd = a.rstd[row]; }     // first: loads return in order, and these are needed first
#ifdef SFRON_TUNE_NO_BRANCH                        // timing experiment only (wrong d gate)
    if constexpr (GATE) { for (int i = 0; i < NCH; ++i) braw.v[i] = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f}; }
#else
    if constexpr (GATE) ldb(a.branch + (size_t)row * D, braw);
#endif
    const __amdgpu_buffer_rsrc_t rs_dx = row_rsrc(a.dx + (size_t)row * D, D * 4);
    if constexpr (LN) {
      ldf(a.x + (size_t)row * D, xr);
      ldb(a.dxmod + (size_t)row * D, draw);
      if (a.dx_accumulate) ldf(a.dx + (size_t)row * D, prev);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        float4& xv = xr.v[i];
        xv.x = (xv.x - mean) * rstd; xv.y = (xv.y - mean) * rstd; xv.z = (xv.z - mean) * rstd; xv.w = (xv.w - mean) * rstd;
        float4 g = raw4(draw, i);                 // zero past the row end: no share in the sums
        acc[0][i].x += g.x; acc[0][i].y += g.y; acc[0][i].z += g.z; acc[0][i].w += g.w;
        acc[1][i].x += g.x * xv.x; acc[1][i].y += g.y * xv.y; acc[1][i].z += g.z * xv.z; acc[1][i].w += g.w * xv.w;
        const float4 s4 = mod4(s_sc, sc, i);      // g = dxmod * (1 + scale)
        g.x *= 1.0f + s4.x; g.y *= 1.0f + s4.y; g.z *= 1.0f + s4.z; g.w *= 1.0f + s4.w;
        s1 += g.x + g.y + g.z + g.w;
        s2 += g.x * xv.x + g.y * xv.y + g.z * xv.z + g.w * xv.w;
        __builtin_amdgcn_sched_barrier(0);        // chunk by chunk: hoisting all the LDS reads of the row costs 40 registers (spills)
      }
      m1 = wave_sum(s1) / (float)D; m2 = wave_sum(s2) / (float)D;
    } else {
      ldf(a.dx + (size_t)row * D, prev);
    }
    const __amdgpu_buffer_rsrc_t rs_db = row_rsrc(GATE ? a.d_branch + (size_t)row * D : nullptr, D * 2);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      float4 d;                                  // the dx element the gate part sees
      if constexpr (LN) {
        float4 g = raw4(draw, i);                // (recomputed, not kept across the row sums: 20 registers)
        const float4 s4 = mod4(s_sc, sc, i), xh = xr.v[i];
        g.x *= 1.0f + s4.x; g.y *= 1.0f + s4.y; g.z *= 1.0f + s4.z; g.w *= 1.0f + s4.w;
        d = make_float4(rstd * (g.x - m1 - xh.x * m2), rstd * (g.y - m1 - xh.y * m2),
                        rstd * (g.z - m1 - xh.z * m2), rstd * (g.w - m1 - xh.w * m2));
        if (a.dx_accumulate) { const float4 p = prev.v[i]; d.x += p.x; d.y += p.y; d.z += p.z; d.w += p.w; }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d), rs_dx, lane * 16 + 1024 * i, 0, 0);
      } else {
        d = prev.v[i];
      }
      if constexpr (GATE) {
        constexpr int G0 = LN ? 2 : 0;
        const float4 h = raw4(braw, i);
        acc[G0][i].x += d.x * h.x; acc[G0][i].y += d.y * h.y; acc[G0][i].z += d.z * h.z; acc[G0][i].w += d.w * h.w;
        acc[G0 + 1][i].x += d.x; acc[G0 + 1][i].y += d.y; acc[G0 + 1][i].z += d.z; acc[G0 + 1][i].w += d.w;
        const float4 gt = mod4(s_gt, gp, i);
        const bf16x4 o = {f2bf(d.x * gt.x), f2bf(d.y * gt.y), f2bf(d.z * gt.z), f2bf(d.w * gt.w)};
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rs_db, lane * 8 + 512 * i, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float* dst[4];
  if constexpr (LN) { dst[0] = a.p_shift; dst[1] = a.p_scale; }
  if constexpr (GATE) { dst[LN ? 2 : 0] = a.p_gate; dst[LN ? 3 : 1] = a.p_dy; }
  if constexpr (!COMBINE) {
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < D4) reinterpret_cast<float4*>(dst[k] + (size_t)wchunk * D)[c] = acc[k][i];
      }
    return;
  }
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {                  // two passes through the 12 KB buffer
      constexpr int I0[2] = {0, HALF}, I1[2] = {HALF, NCH};
      __syncthreads();                             // (first pass: every wave is done with the scale / gate rows)
#pragma unroll
      for (int i = I0[h]; i < I1[h]; ++i) sh[wave][lane + 64 * (i - I0[h])] = acc[k][i];
      __syncthreads();
      const int c0 = 64 * I0[h], c1 = min(D4, 64 * I1[h]);
      for (int c = c0 + threadIdx.x; c < c1; c += TPB) {
        const float4 p0 = sh[0][c - c0], p1 = sh[1][c - c0], p2 = sh[2][c - c0], p3 = sh[3][c - c0];
        reinterpret_cast<float4*>(dst[k] + (size_t)blockIdx.x * D)[c] =
            make_float4(((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z,
                        ((p0.w + p1.w) + p2.w) + p3.w);
      }
    }
  }
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
  int j = 0;
  for (; j + 16 <= per_group; j += 16) {          // 16 loads in flight, summed in index order
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(j + u) * D];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  for (; j + 4 <= per_group; j += 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = p[(size_t)(j + u) * D];
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u];
  }
  for (; j < per_group; ++j) s += p[(size_t)j * D];
  float* o = out + (size_t)g * ldout + c;
  *o = accumulate ? *o + s : s;
}

// the same for MANY partials of FEW columns (bias gradients: 512 row chunks of a 128-wide d_out): one thread per column walks
// per_group dependent rounds of loads on one or two waves while the rest of the chip idles (6.7 us for 512 x 128).  Here a
// workgroup of 16 waves serves 64 columns, wave w sums the contiguous range [per_group w / 16, per_group (w + 1) / 16) with 8 loads
// in flight, and the 16 range sums meet in LDS in wave order: a fixed order again (not the order of k_reduce_chunks)
constexpr int RC_WAVES = 16;
__global__ __launch_bounds__(64 * RC_WAVES) void k_reduce_chunks_wide(const float* __restrict__ P, int per_group, int D, float* __restrict__ out,
                                                                      int ldout, int accumulate) {
  __shared__ float sh[RC_WAVES][64];
  const int g = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int j0 = (int)((long)per_group * wave / RC_WAVES), j1 = (int)((long)per_group * (wave + 1) / RC_WAVES);
  float s = 0.f;
  if (c < D) {
    const float* p = P + (size_t)g * per_group * D + c;
    int j = j0;
    for (; j + 8 <= j1; j += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(j + u) * D];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; j < j1; ++j) s += p[(size_t)j * D];
  }
  sh[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < D) {
    float a = sh[0][lane];
#pragma unroll
    for (int w = 1; w < RC_WAVES; ++w) a += sh[w][lane];
    float* o = out + (size_t)g * ldout + c;
    *o = accumulate ? *o + a : a;
  }
}
// which of the two forms a reduction takes (a pure function of its shape: the same call always sums in the same order)
static inline bool reduce_chunks_wide(int groups, int per_group, int D) { return per_group >= 64 && (long)cdiv(D, 64) * groups <= 512; }
__device__ __forceinline__ bool reduce_chunks_wide_d(int groups, int per_group, int D) { return per_group >= 64 && (long)((D + 63) / 64) * groups <= 512; }
static inline void launch_reduce_chunks(const float* partials, int groups, int per_group, int D, float* out, int ldout, int accumulate,
                                        hipStream_t hs) {
  if (reduce_chunks_wide(groups, per_group, D))
    hipLaunchKernelGGL(k_reduce_chunks_wide, dim3(cdiv(D, 64), groups), dim3(64 * RC_WAVES), 0, hs, partials, per_group, D, out, ldout, accumulate);
  else
    hipLaunchKernelGGL(k_reduce_chunks, dim3(cdiv(D, TPB), groups), dim3(TPB), 0, hs, partials, per_group, D, out, ldout, accumulate);
}

// MANY such reductions in one launch (round 6): the items travel BY VALUE in the kernel arguments (no table in memory: nothing to build, copy
// or keep alive, and a captured graph node holds its own copy), blockIdx.y picks the item.  Each item is summed in the order its own launch
// would use -- the rule above picks the form from the item's shape -- so a batch gives, bit for bit, what the separate launches give.
constexpr int RB_ITEMS = 120;                    // 120 x 32 bytes of the 4 KB a kernel may take as arguments
struct ReducePack { sfron_reduce_item it[RB_ITEMS]; };
__global__ __launch_bounds__(64 * RC_WAVES) void k_reduce_batch(ReducePack pack) {
  __shared__ float sh[RC_WAVES][64];
  const sfron_reduce_item it = pack.it[blockIdx.y];
  const int per = it.per_group, D = it.D;
  if (reduce_chunks_wide_d(it.groups, per, D)) {
    const int ncb = (D + 63) / 64;
    if ((int)blockIdx.x >= it.groups * ncb) return;                  // uniform per workgroup
    const int g = blockIdx.x / ncb, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = (blockIdx.x - g * ncb) * 64 + lane;
    const int j0 = (int)((long)per * wave / RC_WAVES), j1 = (int)((long)per * (wave + 1) / RC_WAVES);
    float s = 0.f;
    if (c < D) {
      const float* p = it.partials + (size_t)g * per * D + c;
      int j = j0;
      for (; j + 8 <= j1; j += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(j + u) * D];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; j < j1; ++j) s += p[(size_t)j * D];
    }
    sh[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < D) {
      float a = sh[0][lane];
#pragma unroll
      for (int w = 1; w < RC_WAVES; ++w) a += sh[w][lane];
      it.out[(size_t)g * it.ldout + c] = a;
    }
    return;
  }
  const long idx = (long)blockIdx.x * (64 * RC_WAVES) + threadIdx.x;
  if (idx >= (long)it.groups * D) return;
  const int g = (int)(idx / D), c = (int)(idx - (long)g * D);
  const float* p = it.partials + (size_t)g * per * D + c;
  float s = 0.f;
  int j = 0;
  for (; j + 16 <= per; j += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(j + u) * D];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  for (; j + 4 <= per; j += 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = p[(size_t)(j + u) * D];
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u];
  }
  for (; j < per; ++j) s += p[(size_t)j * D];
  it.out[(size_t)g * it.ldout + c] = s;
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
  int j = 0;
  for (; j + 8 <= per_group; j += 8) {            // 16 loads in flight, summed in index order
    float v0[8], v1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v0[u] = P0[base + (size_t)(j + u) * D]; v1[u] = P1[base + (size_t)(j + u) * D]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 += v0[u]; s1 += v1[u]; }
  }
  for (; j < per_group; ++j) { s0 += P0[base + (size_t)j * D]; s1 += P1[base + (size_t)j * D]; }
  out0[(size_t)g * ld0 + c] = s0;
  out1[(size_t)g * ld1 + c] = s1;
}

// the same on 16 waves per 64 columns when many partial rows meet few columns (LayerNorm parameter gradients of the LDM transformer
// blocks: up to 512 partial rows of 320 .. 1280 columns) -- the split of k_reduce_chunks_wide, same selection rule
__global__ __launch_bounds__(64 * RC_WAVES) void k_reduce2_wide(const float* __restrict__ P0, const float* __restrict__ P1, int per_group, int D,
                                                                float* __restrict__ out0, int ld0, float* __restrict__ out1, int ld1) {
  __shared__ float sh[2][RC_WAVES][64];
  const int g = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int j0 = (int)((long)per_group * wave / RC_WAVES), j1 = (int)((long)per_group * (wave + 1) / RC_WAVES);
  float s0 = 0.f, s1 = 0.f;
  if (c < D) {
    const size_t base = (size_t)g * per_group * D + c;
    int j = j0;
    for (; j + 4 <= j1; j += 4) {
      float v0[4], v1[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { v0[u] = P0[base + (size_t)(j + u) * D]; v1[u] = P1[base + (size_t)(j + u) * D]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { s0 += v0[u]; s1 += v1[u]; }
    }
    for (; j < j1; ++j) { s0 += P0[base + (size_t)j * D]; s1 += P1[base + (size_t)j * D]; }
  }
  sh[0][wave][lane] = s0; sh[1][wave][lane] = s1;
  __syncthreads();
  if (wave < 2 && c < D) {
    float a = sh[wave][0][lane];
#pragma unroll
    for (int w = 1; w < RC_WAVES; ++w) a += sh[wave][w][lane];
    if (wave == 0) out0[(size_t)g * ld0 + c] = a;
    else           out1[(size_t)g * ld1 + c] = a;
  }
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

// column sums of a [M][N] matrix: stage 1 writes partials[chunk][N]; the caller finishes with k_reduce_chunks.
// A lane owns VEC consecutive columns (one 16-byte load per row for bf16 VEC = 8 / fp32 VEC = 4), the 4 waves of a
// workgroup take rows r0+wave, +4, ... with 4 loads in flight each, and combine through LDS in a fixed order.
template <typename T, int VEC>
__global__ __launch_bounds__(TPB) void k_colsum_partial(const T* __restrict__ X, int M, int N, int ld, int rows_per_block,
                                                        float* __restrict__ partials) {
  __shared__ float sh[3][VEC][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + lane) * VEC;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  typedef __attribute__((ext_vector_type(VEC))) __bf16 bvec;
  typedef __attribute__((ext_vector_type(VEC))) float fvec;
  if (col < N) {
    const T* p = X + col;
    int r = r0 + wave;
    for (; r + 12 < r1; r += 16) {
      if constexpr (sizeof(T) == 2) {
        bvec q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const bvec*>(p + (size_t)(r + 4 * u) * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] += bf2f(q[u][v]);
      } else {
        fvec q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const fvec*>(p + (size_t)(r + 4 * u) * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] += q[u][v];
      }
    }
    for (; r < r1; r += 4) {
      if constexpr (sizeof(T) == 2) {
        const bvec q = *reinterpret_cast<const bvec*>(p + (size_t)r * ld);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += bf2f(q[v]);
      } else {
        const fvec q = *reinterpret_cast<const fvec*>(p + (size_t)r * ld);
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += q[v];
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) sh[wave - 1][v][lane] = acc[v];
  }
  __syncthreads();
  if (wave == 0 && col < N) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      acc[v] = ((acc[v] + sh[0][v][lane]) + sh[1][v][lane]) + sh[2][v][lane];
      partials[(size_t)blockIdx.y * N + col + v] = acc[v];
    }
  }
}

// rows per wave and whether the 4 waves of a workgroup combine their partials (needs 4*rpw | T)
struct ChunkPlan { int rpw, combine, rows; };
inline ChunkPlan pick_chunk(int T) {
#ifdef SFRON_DEBUG_KNOBS
  static const int knob = getenv("SFRON_ROW_RPW") ? atoi(getenv("SFRON_ROW_RPW")) : 0;      // A-B of the rows per wave (tools/bench_dbg.py)
  if (knob > 0 && T % (4 * knob) == 0) return {knob, 1, 4 * knob};
#endif
  // long sequences: 8 rows per wave = 32 per workgroup -- half the partial rows for the slot reduction to read, and the step measures
  // 0.2-0.3 ms faster at DiT-XL/2 (same box, three alternations: 62.49 / 62.45 / 62.66 -> 62.27 / 62.20 / 62.07; 16 rows per wave: 62.30 / 62.62 / 62.30;
  // 2: +0.5 ms).  Short ones (DiT-B/4: 64 tokens) keep 4: their grids are small already
  if (T % 32 == 0 && T >= 256) return {8, 1, 32};
  if (T % 16 == 0) return {4, 1, 16};
  if (T % 8 == 0) return {2, 1, 8};
  if (T % 4 == 0) return {1, 1, 4};
  if (T % 2 == 0) return {2, 0, 2};
  return {1, 0, 1};
}

template <bool LN, bool GATE>
int launch_row_bwd(RowBwdArgs a, void* stream) {
  const ChunkPlan cp = pick_chunk(a.T);
  a.rpw = cp.rpw; a.combine = cp.combine;
  if (cp.combine) SFRON_LAUNCH_EV((k_row_bwd<LN, GATE, true>), dim3(cdiv(a.M / cp.rpw, 4)), dim3(TPB), 0, (hipStream_t)stream, a);
  else SFRON_LAUNCH_EV((k_row_bwd<LN, GATE, false>), dim3(cdiv(a.M / cp.rpw, 4)), dim3(TPB), 0, (hipStream_t)stream, a);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

struct SlotDst { float* base; long layer_stride; int ld; };
struct SlotArgs { SlotDst dst[8]; };      // [kind 0..3][buf 0..1]

// One launch for the whole backward pass: slot s = (layer, kind, buf) holds per-chunk partials [B*per][D];
// out[dst(kind,buf).base + layer*stride + b*ld + c] = sum_j partial[(b*per + j)*D + c]
// A workgroup = one (slot, sample) row: blockDim = D / 4 threads rounded up to whole waves (DiT-XL/2: 288 of 320), a thread owns four columns.
// Round 5: every partial row of the sample (per <= RS_MAX: 16 at 256 tokens) is requested before the first add -- 16 x 4.6 KB in flight per
// workgroup instead of 8 per thread behind a 1024-column split that left every second workgroup with 32 live threads: 178 -> ~100 us for the
// 528 MB of a DiT-XL/2 pass (profiles/r05_stage_boundary.txt).  Summed in index order as before: the same bits.
constexpr int RS_MAX = 16;
__global__ __launch_bounds__(512) void k_reduce_slots(const float* __restrict__ parts, long slot_stride, int per, int D, SlotArgs a) {
  const int slot = blockIdx.y, b = blockIdx.x;
  const int layer = slot >> 3, kb = slot & 7;
  // four columns per thread (D % 4 == 0), 16-byte loads; one trip for D <= 4 * blockDim (every registry model), a column-group loop beyond
  // (a wider model must not fail in the middle of a backward pass)
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
  const float* p = parts + (size_t)slot * slot_stride + (size_t)b * per * D + c;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int j = 0;
  for (; j + RS_MAX <= per; j += RS_MAX) {
    float4 v[RS_MAX];
#pragma unroll
    for (int u = 0; u < RS_MAX; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(j + u) * D);
#pragma unroll
    for (int u = 0; u < RS_MAX; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; j + 4 <= per; j += 4) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(j + u) * D);
#pragma unroll
    for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  for (; j < per; ++j) { const float4 v = *reinterpret_cast<const float4*>(p + (size_t)j * D); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  const SlotDst d = a.dst[kb];
  *reinterpret_cast<float4*>(d.base + (size_t)layer * d.layer_stride + (size_t)b * d.ld + c) = s;
  }
}

// ---------------------------------------------------------------- finish of a split-K product (few-tile GEMMs: small batch * tokens)
// out = sum of the S fp32 slabs in index order (fixed: bitwise reproducible), written as bf16 (an input gradient that the next product
// reads as its operand) ...
constexpr int SPLIT_MAX = 8;
__global__ __launch_bounds__(TPB) void k_split_sum_bf16(const float* __restrict__ slabs, int S, long n4, long stride, __bf16* __restrict__ out) {
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long)gridDim.x * TPB) {
    float4 v[SPLIT_MAX];
#pragma unroll
    for (int s = 0; s < SPLIT_MAX; ++s) v[s] = s < S ? reinterpret_cast<const float4*>(slabs + (size_t)s * stride)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a = v[0];
#pragma unroll
    for (int s = 1; s < SPLIT_MAX; ++s)
      if (s < S) { a.x += v[s].x; a.y += v[s].y; a.z += v[s].z; a.w += v[s].w; }
    reinterpret_cast<bf16x4*>(out)[i] = bf16x4{f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w)};
  }
}
// ... or through the gated-residual epilogue of the forward proj / fc2 products (DiT/models.py:120-121): v = sum + bias; branch (bf16) = v;
// x_next = resid + gate[sample] * v -- operation for operation gemm.hip's EPI_GATE_RES on the summed accumulator
__global__ __launch_bounds__(TPB) void k_split_gate_res(const float* __restrict__ slabs, int S, long stride, const float* __restrict__ bias,
                                                        const float* __restrict__ gate, int ldgate, int T, const float* __restrict__ resid,
                                                        float* __restrict__ out, __bf16* __restrict__ branch, int M, int N) {
  const int n4 = N >> 2;
  const long tot = (long)M * n4;
  for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < tot; i += (long)gridDim.x * TPB) {
    const int row = (int)(i / n4), c4 = (int)(i - (long)row * n4);
    float4 v[SPLIT_MAX];
#pragma unroll
    for (int s = 0; s < SPLIT_MAX; ++s) v[s] = s < S ? reinterpret_cast<const float4*>(slabs + (size_t)s * stride)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 x = reinterpret_cast<const float4*>(resid)[i];
    const float4 gt = reinterpret_cast<const float4*>(gate + (size_t)(row / T) * ldgate)[c4];
    float4 a = v[0];
#pragma unroll
    for (int s = 1; s < SPLIT_MAX; ++s)
      if (s < S) { a.x += v[s].x; a.y += v[s].y; a.z += v[s].z; a.w += v[s].w; }
    if (bias) { const float4 b = reinterpret_cast<const float4*>(bias)[c4]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
    reinterpret_cast<bf16x4*>(branch)[i] = bf16x4{f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w)};
    reinterpret_cast<float4*>(out)[i] = make_float4(x.x + gt.x * a.x, x.y + gt.y * a.y, x.z + gt.z * a.z, x.w + gt.w * a.w);
  }
}

}  // namespace

extern "C" {

int sfron_split_sum_bf16(const float* slabs, int n_splits, int64_t n, int64_t split_stride, uint16_t* out, void* stream) {
  SFRON_CHECK_ARG(slabs && out && n_splits >= 1 && n_splits <= SPLIT_MAX && n > 0 && n % 4 == 0 && split_stride % 4 == 0 && split_stride >= n);
  SFRON_CHECK_ARG(((uintptr_t)slabs & 15) == 0 && ((uintptr_t)out & 7) == 0);
  const long n4 = n >> 2;
  const int grid = (int)(n4 / TPB < 1 ? 1 : (n4 / TPB > 2048 ? 2048 : n4 / TPB));
  hipLaunchKernelGGL(k_split_sum_bf16, dim3(grid), dim3(TPB), 0, (hipStream_t)stream, slabs, n_splits, n4, (long)split_stride, (__bf16*)out);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_split_gate_res(const float* slabs, int n_splits, int64_t split_stride, const float* bias, const float* gate, int ldgate, int tokens,
                         const float* resid, float* out, uint16_t* branch, int M, int N, void* stream) {
  SFRON_CHECK_ARG(slabs && gate && resid && out && branch && n_splits >= 1 && n_splits <= SPLIT_MAX && M > 0 && N > 0 && N % 4 == 0 && tokens > 0);
  SFRON_CHECK_ARG(split_stride % 4 == 0 && split_stride >= (int64_t)M * N && ldgate % 4 == 0);
  SFRON_CHECK_ARG((((uintptr_t)slabs | (uintptr_t)gate | (uintptr_t)resid | (uintptr_t)out | (uintptr_t)bias) & 15) == 0 && ((uintptr_t)branch & 7) == 0);
  const long tot = (long)M * (N >> 2);
  const int grid = (int)(tot / TPB < 1 ? 1 : (tot / TPB > 2048 ? 2048 : tot / TPB));
  hipLaunchKernelGGL(k_split_gate_res, dim3(grid), dim3(TPB), 0, (hipStream_t)stream, slabs, n_splits, (long)split_stride, bias, gate, ldgate, tokens,
                     resid, out, (__bf16*)branch, M, N);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_rows_per_chunk(int tokens) { return tokens > 0 ? pick_chunk(tokens).rows : 0; }

int sfron_ln_modulate_fwd(const float* x, const float* shift, const float* scale, int ldmod, int tokens, int M, int D,
                          uint16_t* out, float* mean, float* rstd, void* stream) {
  SFRON_CHECK_ARG(x && shift && scale && out && mean && rstd && M > 0 && tokens > 0);
  SFRON_CHECK_ARG(D % 4 == 0 && D <= 64 * 4 * NCH && ldmod % 4 == 0);
  SFRON_CHECK_ARG((((uintptr_t)x | (uintptr_t)shift | (uintptr_t)scale) & 15) == 0 && ((uintptr_t)out & 7) == 0);
  // rows per wave: as many as divide the token count (one sample per wave) while the launch still fills the chip
  if (M >= 8192 && tokens % 4 == 0)
    hipLaunchKernelGGL(k_ln_mod_fwd<4>, dim3(cdiv(M, 16)), dim3(TPB), 0, (hipStream_t)stream, x, shift, scale, ldmod, tokens, M, D,
                       (__bf16*)out, mean, rstd);
  else if (M >= 4096 && tokens % 2 == 0)
    hipLaunchKernelGGL(k_ln_mod_fwd<2>, dim3(cdiv(M, 8)), dim3(TPB), 0, (hipStream_t)stream, x, shift, scale, ldmod, tokens, M, D,
                       (__bf16*)out, mean, rstd);
  else
    hipLaunchKernelGGL(k_ln_mod_fwd<1>, dim3(cdiv(M, 4)), dim3(TPB), 0, (hipStream_t)stream, x, shift, scale, ldmod, tokens, M, D,
                       (__bf16*)out, mean, rstd);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_ln_modulate_bwd(const uint16_t* d_out, const float* x, const float* mean, const float* rstd, const float* scale,
                          int ldmod, int tokens, int M, int D, float* dx, int dx_accumulate, float* p_shift,
                          float* p_scale, void* stream) {
  SFRON_CHECK_ARG(d_out && x && mean && rstd && scale && dx && p_shift && p_scale && M > 0 && tokens > 0);
  SFRON_CHECK_ARG(D % 4 == 0 && D <= 64 * 4 * NCH && ldmod % 4 == 0 && M % tokens == 0);
  RowBwdArgs a{};
  a.dxmod = (const __bf16*)d_out; a.x = x; a.mean = mean; a.rstd = rstd; a.scale = scale; a.p_shift = p_shift; a.p_scale = p_scale;
  a.dx_accumulate = dx_accumulate; a.dx = dx; a.ldmod_ln = ldmod; a.T = tokens; a.M = M; a.D = D;
  return launch_row_bwd<true, false>(a, stream);
}

int sfron_gate_bwd(const float* dy, const uint16_t* branch, const float* gate, int ldmod, int tokens, int M, int D,
                   uint16_t* d_branch, float* p_gate, float* p_dy, void* stream) {
  SFRON_CHECK_ARG(dy && branch && gate && d_branch && p_gate && p_dy && M > 0 && tokens > 0);
  SFRON_CHECK_ARG(D % 4 == 0 && D <= 64 * 4 * NCH && ldmod % 4 == 0 && M % tokens == 0);
  RowBwdArgs a{};
  a.branch = (const __bf16*)branch; a.gate = gate; a.d_branch = (__bf16*)d_branch; a.p_gate = p_gate; a.p_dy = p_dy;
  a.dx = const_cast<float*>(dy); a.ldmod_gate = ldmod; a.T = tokens; a.M = M; a.D = D;
  return launch_row_bwd<false, true>(a, stream);
}

int sfron_ln_gate_bwd(const uint16_t* d_out, const float* x, const float* mean, const float* rstd, const float* scale,
                      int ldmod, int tokens, int M, int D, float* dx, int dx_accumulate, float* p_shift, float* p_scale,
                      const uint16_t* branch, const float* gate, int ldgate, uint16_t* d_branch, float* p_gate, float* p_dy,
                      void* stream) {
  SFRON_CHECK_ARG(d_out && x && mean && rstd && scale && dx && p_shift && p_scale && M > 0 && tokens > 0);
  SFRON_CHECK_ARG(branch && gate && d_branch && p_gate && p_dy);
  SFRON_CHECK_ARG(D % 4 == 0 && D <= 64 * 4 * NCH && ldmod % 4 == 0 && ldgate % 4 == 0 && M % tokens == 0);
  RowBwdArgs a{};
  a.dxmod = (const __bf16*)d_out; a.x = x; a.mean = mean; a.rstd = rstd; a.scale = scale; a.p_shift = p_shift; a.p_scale = p_scale;
  a.dx_accumulate = dx_accumulate; a.dx = dx; a.ldmod_ln = ldmod; a.T = tokens; a.M = M; a.D = D;
  a.branch = (const __bf16*)branch; a.gate = gate; a.d_branch = (__bf16*)d_branch; a.p_gate = p_gate; a.p_dy = p_dy;
  a.ldmod_gate = ldgate;
  return launch_row_bwd<true, true>(a, stream);
}

int sfron_reduce_chunks(const float* partials, int groups, int per_group, int D, float* out, int ldout, int accumulate,
                        void* stream) {
  SFRON_CHECK_ARG(partials && out && groups > 0 && per_group > 0 && D > 0);
  launch_reduce_chunks(partials, groups, per_group, D, out, ldout, accumulate, (hipStream_t)stream);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_reduce_batch(const sfron_reduce_item* items, int n_items, void* stream) {
  SFRON_CHECK_ARG(items && n_items > 0);
  for (int i = 0; i < n_items; ++i)
    SFRON_CHECK_ARG(items[i].partials && items[i].out && items[i].groups > 0 && items[i].per_group > 0 && items[i].D > 0 && items[i].ldout >= items[i].D);
  for (int i0 = 0; i0 < n_items; i0 += RB_ITEMS) {
    const int n = n_items - i0 < RB_ITEMS ? n_items - i0 : RB_ITEMS;
    ReducePack pack;
    long gx = 1;
    for (int i = 0; i < n; ++i) {
      const sfron_reduce_item& it = items[i0 + i];
      pack.it[i] = it;
      const long need = reduce_chunks_wide(it.groups, it.per_group, it.D) ? (long)it.groups * cdiv(it.D, 64)
                                                                           : ((long)it.groups * it.D + 64 * RC_WAVES - 1) / (64 * RC_WAVES);
      gx = need > gx ? need : gx;
    }
    for (int i = n; i < RB_ITEMS; ++i) pack.it[i] = pack.it[0];      // never indexed (blockIdx.y < n); defined bytes for the argument copy
    SFRON_CHECK_ARG(gx <= 0x7fffffffL);
    hipLaunchKernelGGL(k_reduce_batch, dim3((unsigned)gx, n), dim3(64 * RC_WAVES), 0, (hipStream_t)stream, pack);
    SFRON_LAUNCH_STATUS();
  }
  return SFRON_OK;
}

int sfron_reduce2(const float* p0, const float* p1, int groups, int per_group, int D, float* out0, int ld0, float* out1,
                  int ld1, void* stream) {
  SFRON_CHECK_ARG(p0 && p1 && out0 && out1 && groups > 0 && per_group > 0 && D > 0);
  if (reduce_chunks_wide(groups, per_group, D))
    hipLaunchKernelGGL(k_reduce2_wide, dim3(cdiv(D, 64), groups), dim3(64 * RC_WAVES), 0, (hipStream_t)stream, p0, p1, per_group, D, out0, ld0, out1,
                       ld1);
  else
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
  SFRON_CHECK_ARG(D % 4 == 0);
  for (int i = 0; i < 8; ++i) SFRON_CHECK_ARG(((uintptr_t)dst_base[i] & 15) == 0 && dst_layer_stride[i] % 4 == 0 && dst_ld[i] % 4 == 0);
  const int threads = D / 4 <= 512 ? cdiv(D / 4, 64) * 64 : 512;          // wider rows: the kernel loops over column groups of 2048
  hipLaunchKernelGGL(k_reduce_slots, dim3(groups, n_slots), dim3(threads), 0, (hipStream_t)stream, parts, slot_stride,
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
  SFRON_CHECK_ARG(((uintptr_t)X & 15) == 0);
  int chunks = cdiv(M, 32);
  if (chunks > max_partials) chunks = max_partials;
  const int rpb = cdiv(M, chunks);
  chunks = cdiv(M, rpb);
  hipStream_t hs = (hipStream_t)stream;
  if (is_bf16 && N % 8 == 0 && ld % 8 == 0)
    hipLaunchKernelGGL((k_colsum_partial<__bf16, 8>), dim3(cdiv(N, 512), chunks), dim3(TPB), 0, hs, (const __bf16*)X, M, N, ld, rpb, partials);
  else if (is_bf16)
    hipLaunchKernelGGL((k_colsum_partial<__bf16, 4>), dim3(cdiv(N, 256), chunks), dim3(TPB), 0, hs, (const __bf16*)X, M, N, ld, rpb, partials);
  else
    hipLaunchKernelGGL((k_colsum_partial<float, 4>), dim3(cdiv(N, 256), chunks), dim3(TPB), 0, hs, (const float*)X, M, N, ld, rpb, partials);
  SFRON_LAUNCH_STATUS();
  launch_reduce_chunks(partials, 1, chunks, N, out, N, 0, hs);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

}  // extern "C"
