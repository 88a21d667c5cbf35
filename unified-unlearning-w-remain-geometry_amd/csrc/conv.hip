// Convolutional U-Net building blocks (DDPM Conditional_Model; the same blocks serve the LDM UNetModel).
//
// Replaces, for /root/reference/DDPM/models/diffusion.py:
//   Conv2d 3x3 / 1x1 forward, input gradient and weight gradient (:49-82 Upsample / Downsample, :85-145 ResnetBlock,
//   :148-192 AttnBlock, :283-327 conv_in / conv_out)              -> k_bgemm: implicit-GEMM convolution on NHWC bf16
//   GroupNorm(32, eps 1e-6) (+ swish, + dropout) forward / backward (:43-46,38-40,126-131)   -> k_gn_fwd / k_gn_bwd
//   single-head attention bmm / softmax (:168-186)                -> batched k_bgemm + k_softmax_fwd / k_softmax_bwd
//   nearest-neighbour upsampling (:56-57) and the (0,1,0,1) pad + stride-2 conv (:76-80) are address arithmetic of the
//   im2col loader (no copies); get_timestep_embedding (:17-35); classes_emb / null_classes_emb select (:370-376).
//
// Activations are NHWC: a [B*H*W][C] row-major matrix, so a 3x3 convolution is the GEMM
//   Y[p][co] = sum_{tap, ci} X[src(p, tap)][ci] * W[co][tap][ci]
// whose A operand is never materialised: the staging loader computes src(p, tap) (stride, padding, nearest-upsampled or
// zero-dilated source) per 16-byte chunk and zero-fills what falls outside the image.
//   forward   A = X (im2col),  B = W   [Cout][taps*Cin]
//   dgrad     A = dY (im2col, flipped taps; stride-2 convs read a zero-dilated dY),  B = W^T re-laid [Cin][taps*Cout]
//   wgrad     A = dY^T [pixels][Cout] (transposed read),  B = X (im2col, transposed read)  -> fp32 [Cout][taps*Cin]
// Tile 128x128x64, 4 waves, mfma_f32_16x16x32_bf16 issued as D^T = B^T A^T (lane owns 4 consecutive output columns), LDS
// images XOR-swizzled as in gemm.hip.  The U-Net is 5 % of the DiT step's FLOPs per sample: this kernel is the generic
// (register-staged) tile, not the LDS-DMA pipeline of gemm.hip.
#include "common.h"
#include <atomic>
#include <cstdlib>
#include <utility>
#include "../../include/sfron.h"

namespace {

struct ConvGeom {
  int Hs, Ws, C;          // source image (per sample) and its channel count
  int Ho, Wo;             // output grid that indexes the GEMM rows (forward output pixels / dgrad input pixels)
  int stride, pad;
  int up;                 // source is read through a nearest x2 upsampling (virtual extent 2Hs x 2Ws)
  int dil;                // source is read as a zero-dilated x2 image (transposed stride-2 convolution)
  int taps;               // 9 or 1
  int flip;               // dgrad: tap index runs over the flipped kernel (the weights were re-laid to match)
};

struct BGemmArgs {
  const __bf16* A; const __bf16* B;
  int M, N, K, lda, ldb;
  long sA, sB, sC;        // batch strides (elements); grid.y = batch
  long sA2, sB2, sC2;     // inner batch strides; grid.z = inner batch (attention heads: column offsets inside one matrix)
  __bf16* Cb; int ldcb;
  float* Cf; int ldcf;
  const float* bias;      // [N] or null
  const float* resid;     // [M][ldcf] fp32 added to the result, or null
  const float* vec;       // per-sample row vector vec[(row / T) * ldvec + col], or null
  int ldvec, T;
  float alpha;
  int accumulate;
  int kchunk;             // split-K (weight gradients: few output tiles, a contraction over all pixels): grid.y = split, split s
                          // takes k in [s * kchunk, (s + 1) * kchunk) and writes its own fp32 slab Cf + s * sC (0 = no split)
  int ksplit;             // split-K of a BATCHED product (k_bgemm only): grid.x = tiles * ksplit; split s of batch (y, z) writes the
                          // contiguous fp32 slab Cf[((s * gridDim.y + y) * gridDim.z + z)][M][N]; k_bsplit_finish adds the splits
  ConvGeom cg;
};

constexpr int BM = 128, BN = 128, BK = 64, NT = 256, TILE_ELEMS = 128 * 64;
enum { EPI_BF16 = 0, EPI_RES = 1 };
enum { CONV_NONE = 0, CONV_A = 1, CONV_B = 2 };

__device__ __forceinline__ int off_direct(int row, int ch) { return row * 64 + ((ch ^ (row & 7)) << 3); }
__device__ __forceinline__ int swz_tr(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ int off_tr(int krow, int ch) { return krow * 128 + ((ch ^ swz_tr(krow)) << 3); }
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
__device__ __forceinline__ bf16x8 frag_direct(const __bf16* img, int row, int kchunk) {
  return *reinterpret_cast<const bf16x8*>(img + off_direct(row, kchunk));
}
__device__ __forceinline__ bf16x8 frag_tr(const __bf16* img, int col0, int kr0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int ch = (col0 >> 3) + (p >> 1);
  const int r0 = kr0 + 8 * g + q;
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + off_tr(r0, ch) + 4 * (p & 1)));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + off_tr(r0 + 4, ch) + 4 * (p & 1)));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// source pixel (row index of the NHWC matrix) of output pixel (b, ho, wo) under tap `tap`; false = zero padding
__device__ __forceinline__ bool conv_src(const ConvGeom& c, int b, int ho, int wo, int tap, int& pix) {
  int kh = 0, kw = 0;
  if (c.taps == 9) { kh = tap / 3; kw = tap - 3 * kh; }
  int hi = ho * c.stride + kh - c.pad, wi = wo * c.stride + kw - c.pad;
  const int Hv = (c.up | c.dil) ? 2 * c.Hs : c.Hs, Wv = (c.up | c.dil) ? 2 * c.Ws : c.Ws;
  if (hi < 0 || wi < 0 || hi >= Hv || wi >= Wv) return false;
  if (c.dil) { if ((hi | wi) & 1) return false; }
  if (c.up | c.dil) { hi >>= 1; wi >>= 1; }
  pix = (b * c.Hs + hi) * c.Ws + wi;
  return true;
}

// ---- global -> register staging (128 x 64 direct image or 64 x 128 transposed-read image, 4 chunks of 16 B per thread)
template <bool TR, bool CONV>
struct Stager {
  const __bf16* P;
  int ld, dim;
  int lds_off[4];
  // plain: element offset of the chunk at k = 0 (direct: row * ld + ch * 8; TR: col), validity
  long base[4];
  bool valid[4];
  // conv direct: (b, ho, wo) of the 4 rows, k offset of this thread's chunk; conv TR: (tap, ci) of the 4... one column chunk
  int pb[4], ph[4], pw[4];
  int kch;                 // direct: k offset inside the tile (same for the 4 chunks); TR: first k-row of the thread
  int tap, ci;             // conv TR: fixed per thread
  __device__ __forceinline__ void init(const __bf16* P_, int ld_, int dim_, int d0, int tid, const ConvGeom& cg) {
    P = P_; ld = ld_; dim = dim_;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + NT * i;
      if (!TR) {
        const int row = c >> 3, ch = c & 7;
        const int gr = d0 + row;
        valid[i] = gr < dim;
        lds_off[i] = off_direct(row, ch);
        kch = ch * 8;
        if (!CONV) base[i] = (long)(valid[i] ? gr : 0) * ld + ch * 8;
        else {
          const int g2 = valid[i] ? gr : 0;
          pw[i] = g2 % cg.Wo; const int t = g2 / cg.Wo; ph[i] = t % cg.Ho; pb[i] = t / cg.Ho;
        }
      } else {
        const int krow = c >> 4, ch = c & 15;
        const int col = d0 + ch * 8;
        valid[i] = col < dim;
        lds_off[i] = off_tr(krow, ch);
        if (i == 0) kch = krow;
        if (!CONV) base[i] = (long)krow * ld + (valid[i] ? col : 0);
        else { const int cc = valid[i] ? col : 0; tap = cc / cg.C; ci = cc - tap * cg.C; }
      }
    }
  }
  __device__ __forceinline__ void load(uint4 (&r)[4], int k0, int K, const ConvGeom& cg) const {
    const uint4 z = make_uint4(0, 0, 0, 0);
    if (!TR) {
      const int k = k0 + kch;
      int t = 0, c0 = 0;
      if (CONV) { t = k / cg.C; c0 = k - t * cg.C; if (cg.flip) t = cg.taps - 1 - t; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bool ok = valid[i] && k < K;
        const __bf16* p = P;
        if (!CONV) p = P + base[i] + k0;
        else {
          int pix = 0;
          ok = ok && conv_src(cg, pb[i], ph[i], pw[i], t, pix);
          p = P + (long)pix * ld + c0;
        }
        r[i] = ok ? *reinterpret_cast<const uint4*>(p) : z;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kr = k0 + kch + 16 * i;            // contraction index = token / pixel row
        bool ok = valid[i] && kr < K;
        const __bf16* p = P;
        if (!CONV) p = P + base[i] + (long)k0 * ld;
        else {
          const int kk = ok ? kr : 0;
          const int wo = kk % cg.Wo; const int t2 = kk / cg.Wo; const int ho = t2 % cg.Ho; const int b = t2 / cg.Ho;
          int pix = 0;
          ok = ok && conv_src(cg, b, ho, wo, tap, pix);
          p = P + (long)pix * ld + ci;
        }
        r[i] = ok ? *reinterpret_cast<const uint4*>(p) : z;
      }
    }
  }
  __device__ __forceinline__ void store(__bf16* img, const uint4 (&r)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(img + lds_off[i]) = r[i];
  }
};

template <int EPI>
__device__ __forceinline__ void epi_store(const BGemmArgs& g, int row, int col, f32x4 v) {
  v = v * g.alpha;
  if (g.bias) { const float4 b = *reinterpret_cast<const float4*>(g.bias + col); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
  if (EPI == EPI_BF16) {
    bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    *reinterpret_cast<bf16x4*>(g.Cb + (size_t)row * g.ldcb + col) = o;
  } else {
    if (g.vec) { const float4 b = *reinterpret_cast<const float4*>(g.vec + (size_t)(row / g.T) * g.ldvec + col); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
    if (g.resid) { const float4 b = *reinterpret_cast<const float4*>(g.resid + (size_t)row * g.ldcf + col); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
    float4* dst = reinterpret_cast<float4*>(g.Cf + (size_t)row * g.ldcf + col);
    float4 o = make_float4(v[0], v[1], v[2], v[3]);
    if (g.accumulate) { const float4 c = *dst; o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
    *dst = o;
  }
}

}  // namespace

template <bool A_TR, bool B_TR, int EPI, int CONV>
__global__ __launch_bounds__(NT) void k_bgemm(BGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  __bf16* sA = smem;
  __bf16* sB = smem + 2 * TILE_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (g.N + BN - 1) / BN;
  int tile = blockIdx.x, split = 0;
  if (g.ksplit > 1) { const int nt = ((g.M + BM - 1) / BM) * ntn; split = tile / nt; tile -= split * nt; }
  const int tm = tile / ntn, tn = tile % ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const long bz = blockIdx.y, bi = blockIdx.z;
  int kbeg = 0, kend = g.K;
  if (g.ksplit > 1) { kbeg = split * g.kchunk; kend = min(g.K, kbeg + g.kchunk); g.A += bz * g.sA; g.B += bz * g.sB; }
  else if (g.kchunk > 0) { kbeg = (int)bz * g.kchunk; kend = min(g.K, kbeg + g.kchunk); }
  else { g.A += bz * g.sA; g.B += bz * g.sB; }
  g.A += bi * g.sA2; g.B += bi * g.sB2;
  if (g.ksplit > 1) {
    g.Cf += (((long)split * gridDim.y + bz) * gridDim.z + bi) * ((long)g.M * g.N);
  } else {
    const long co = bz * g.sC + bi * g.sC2;
    if (g.Cb) g.Cb += co;
    if (g.Cf) g.Cf += co;
    if (g.resid) g.resid += co;
  }

  Stager<A_TR, CONV == CONV_A> stA;
  Stager<B_TR, CONV == CONV_B> stB;
  stA.init(g.A, g.lda, g.M, m0, tid, g.cg);
  stB.init(g.B, g.ldb, g.N, n0, tid, g.cg);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (kend - kbeg + BK - 1) / BK;
  uint4 ra[4], rb[4];
  stA.load(ra, kbeg, kend, g.cg);
  stB.load(rb, kbeg, kend, g.cg);
  stA.store(sA, ra);
  stB.store(sB, rb);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      stA.load(ra, kbeg + (kt + 1) * BK, kend, g.cg);
      stB.load(rb, kbeg + (kt + 1) * BK, kend, g.cg);
    }
    const __bf16* iA = sA + cur * TILE_ELEMS;
    const __bf16* iB = sB + cur * TILE_ELEMS;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if (!A_TR) fa[mt] = frag_direct(iA, wm * 64 + mt * 16 + (lane & 15), ks * 4 + (lane >> 4));
        else       fa[mt] = frag_tr(iA, wm * 64 + mt * 16, ks * 32, lane);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (!B_TR) fb[nt] = frag_direct(iB, wn * 64 + nt * 16 + (lane & 15), ks * 4 + (lane >> 4));
        else       fb[nt] = frag_tr(iB, wn * 64 + nt * 16, ks * 32, lane);
      }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
    }
    if (more) {
      stA.store(sA + (cur ^ 1) * TILE_ELEMS, ra);
      stB.store(sB + (cur ^ 1) * TILE_ELEMS, rb);
    }
    __syncthreads();
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int row = m0 + wm * 64 + mt * 16 + (lane & 15);
    if (row >= g.M) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int col = n0 + wn * 64 + nt * 16 + 4 * (lane >> 4);
      if (col >= g.N) continue;
      epi_store<EPI>(g, row, col, acc[mt][nt]);
    }
  }
}

template __global__ void k_bgemm<false, false, EPI_BF16, CONV_NONE>(BGemmArgs);
template __global__ void k_bgemm<false, false, EPI_RES, CONV_NONE>(BGemmArgs);
template __global__ void k_bgemm<false, true, EPI_BF16, CONV_NONE>(BGemmArgs);
template __global__ void k_bgemm<false, true, EPI_RES, CONV_NONE>(BGemmArgs);
template __global__ void k_bgemm<true, true, EPI_BF16, CONV_NONE>(BGemmArgs);
template __global__ void k_bgemm<true, true, EPI_RES, CONV_NONE>(BGemmArgs);
template __global__ void k_bgemm<false, false, EPI_BF16, CONV_A>(BGemmArgs);
template __global__ void k_bgemm<false, false, EPI_RES, CONV_A>(BGemmArgs);
template __global__ void k_bgemm<true, true, EPI_RES, CONV_B>(BGemmArgs);

// =================================================================================================
// LDS-DMA pipelined tile for the large direct-operand products (3x3 / 1x1 convolution forward and input gradient, Linear forward):
// 256 x (NT_ * 16) x 64, 8 waves as 4 x 2 (64 x NT_*8 each), THREE LDS slots (two K-tiles in flight per CU, the
// operands of a U-Net pass are cold in L2), operands staged by `buffer_load_dwordx4 ... lds` with the XOR swizzle applied on the
// SOURCE address (the LDS image is lane-linear).  The im2col gather is address arithmetic of the A-operand DMA: a lane owns four
// output pixels (b, ho, wo); per K-tile the tap (kh, kw) and the channel offset are wave-uniform (C % 64 == 0), the lane adds the
// tap to its pixel, and a source outside the image (zero padding, odd positions of a zero-dilated gradient) becomes an
// out-of-range buffer offset, for which the DMA writes zeros.  NT_ = 10 (160 columns) tiles the 320 / 640 / 1280-wide outputs of
// the LDM UNet exactly, NT_ = 8 serves the DDPM widths (128 / 256) and ragged N (rows of B past N read as zeros through the
// descriptor's size, the epilogue skips their columns).  Needs M % 256 == 0 and K % 64 == 0; everything else stays on k_bgemm.
typedef __attribute__((address_space(3))) void lptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// one wave-instruction of LDS-DMA: 64 lanes x 16 B from rsrc[voff + soff] to the lane-linear KiB at `dst` (a __device__ helper:
// the builtin inside a kernel template's own lambda breaks hipcc's host pass)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, __bf16* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t*)dst, 16, voff, soff, 0, 0);
}

template <int NT_>
struct CTile {
  static constexpr int FBM = 256, FBN = NT_ * 16, NW = 8, NSLOT = 3;
  static constexpr int A_ELEMS = FBM * 64, B_ELEMS = FBN * 64, SLOT = A_ELEMS + B_ELEMS;
  static constexpr int NA = FBM * 8 / 64 / NW;           // A wave-instructions per wave and tile (4)
  static constexpr int NB_TOT = FBN * 8 / 64;            // B wave-instructions per tile
  static constexpr int NB = (NB_TOT + NW - 1) / NW;
  static constexpr bool EVEN = NB_TOT % NW == 0;         // NT_ = 10: 20 instructions on 8 waves -- waves 4..7 send a no-op third
  static constexpr int NDMA = NA + NB;
  static constexpr size_t LDS = (size_t)NSLOT * SLOT * sizeof(__bf16) + (EVEN ? 0 : 1024);
};

// NL > 0: NL extra waves (8 .. 8 + NL - 1) issue every LDS-DMA and do the im2col address arithmetic; the 8 MFMA waves only wait at
// the barrier (the loader / consumer split of gemm.hip k_gemm_pipe)
template <int NT_, int EPI, bool CONV, int NL = 0>
__global__ __launch_bounds__(512 + 64 * NL) void k_cgemm(BGemmArgs g, unsigned a_bytes, unsigned b_bytes) {
  using CT = CTile<NT_>;
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = (g.N + CT::FBN - 1) / CT::FBN;
  // workgroups of one XCD (blockIdx % 8) take consecutive tile ids: the column tiles of one row panel share that XCD's L2
  int id;
  {
    const int nblk = gridDim.x, b = blockIdx.x, q = nblk >> 3, r = nblk & 7, x = b & 7, y = b >> 3;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  const int tm = id / ntn, tn = id - tm * ntn;
  const int m0 = tm * CT::FBM, n0 = tn * CT::FBN;
  int kbeg = 0, kend = g.K;
  if (g.kchunk > 0) {
    kbeg = (int)blockIdx.y * g.kchunk; kend = min(g.K, kbeg + g.kchunk);
    const long co = (long)blockIdx.y * g.sC;
    if (g.Cf) g.Cf += co;
  }
  const int nk = (kend - kbeg) / BK;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, (int)b_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, 0, 0x00020000);
  __bf16* dummy = smem + CT::NSLOT * CT::SLOT;

  constexpr int NWD = NL ? NL : CT::NW;                       // waves that issue DMA
  constexpr int NA_ = CT::FBM * 8 / 64 / NWD, NB_ = (CT::NB_TOT + NWD - 1) / NWD, NDMA_ = NA_ + NB_;
  constexpr bool EVEN_ = CT::NB_TOT % NWD == 0;
  static_assert(NL == 0 || EVEN_, "loader waves: no dummy slot");
  const bool loader = NL && wave >= CT::NW;
  const int dwave = NL ? (wave - CT::NW) & (NWD - 1) : wave;
  const int lc8 = ((lane & 7) ^ ((lane >> 3) & 7)) << 3;      // logical 8-element chunk this lane fetches (rows start at multiples of 8)
  int a_off[NA_], a_h0[NA_], a_w0[NA_], a_pb[NA_], b_off[NB_];
#pragma unroll
  for (int i = 0; i < NA_; ++i) {
    const int gr = m0 + (dwave + i * NWD) * 8 + (lane >> 3);
    if (!CONV) a_off[i] = 2 * (gr * g.lda + lc8);
    else {
      const int wo = gr % g.cg.Wo, t = gr / g.cg.Wo, ho = t % g.cg.Ho, b = t / g.cg.Ho;
      a_h0[i] = ho * g.cg.stride - g.cg.pad; a_w0[i] = wo * g.cg.stride - g.cg.pad; a_pb[i] = b * g.cg.Hs * g.cg.Ws;
    }
  }
#pragma unroll
  for (int i = 0; i < NB_; ++i) b_off[i] = 2 * ((n0 + (dwave + i * NWD) * 8 + (lane >> 3)) * g.ldb + lc8);

  const int Hv = (g.cg.up | g.cg.dil) ? 2 * g.cg.Hs : g.cg.Hs, Wv = (g.cg.up | g.cg.dil) ? 2 * g.cg.Ws : g.cg.Ws;
  // tile kt -> slot: the A source of a convolution advances tap by tap, channel block by channel block
  int tap = 0, c0 = 0;
  if (CONV) { tap = kbeg / g.cg.C; c0 = kbeg - tap * g.cg.C; }
  auto issue = [&](int slot, int k0) {
    __bf16* iA = smem + slot * CT::SLOT;
    __bf16* iB = iA + CT::A_ELEMS;
    int kh = 0, kw = 0;
    if (CONV && g.cg.taps == 9) { const int tt = g.cg.flip ? 8 - tap : tap; kh = (tt * 11) >> 5; kw = tt - 3 * kh; }
#pragma unroll
    for (int i = 0; i < NA_; ++i) {
      int voff = 0;
      if (!CONV) voff = a_off[i];
      else {
        int hi = a_h0[i] + kh, wi = a_w0[i] + kw;
        bool ok = (unsigned)hi < (unsigned)Hv && (unsigned)wi < (unsigned)Wv;
        if (g.cg.dil) ok = ok && !((hi | wi) & 1);
        if (g.cg.up | g.cg.dil) { hi >>= 1; wi >>= 1; }
        voff = ok ? 2 * ((a_pb[i] + hi * g.cg.Ws + wi) * g.lda + lc8) : 0x7ffffff0;
      }
      dma16(rsA, iA + (dwave + i * NWD) * 512, voff, 2 * (CONV ? c0 : k0));
    }
#pragma unroll
    for (int i = 0; i < NB_; ++i) {
      if (EVEN_ || i + 1 < NB_) {
        dma16(rsB, iB + (dwave + i * NWD) * 512, b_off[i], 2 * k0);
      } else {
        const bool ok = dwave + i * NWD < CT::NB_TOT;     // wave-uniform
        dma16(ok ? rsB : rs0, ok ? iB + (dwave + i * NWD) * 512 : dummy, b_off[i], 2 * k0);
      }
    }
    if (CONV) { c0 += BK; if (c0 >= g.cg.C) { c0 = 0; ++tap; } }
  };

  // 4 x 2 waves of 64 x (NT_ / 2 * 16): per K-step a wave reads (4 + NT_ / 2) fragments for 4 * NT_ / 2 MFMAs -- a quarter less LDS traffic
  // per FLOP than 8 x 1 waves of 32 x NT_ * 16 (measured 6-7 % on every shape, tools/probes/persist_gemm_probe.hip)
  constexpr int WNT = NT_ / 2;
  static_assert(NT_ % 2 == 0, "two wave columns");
  const int wm = wave >> 1, wn = wave & 1;
  f32x4 acc[4][WNT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < WNT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](int slot) {
    const __bf16* iA = smem + slot * CT::SLOT;
    const __bf16* iB = iA + CT::A_ELEMS;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) fa[mt] = frag_direct(iA, wm * 64 + mt * 16 + fr, ks * 4 + fg);
#pragma unroll
      for (int nt = 0; nt < WNT; ++nt) {
        const bf16x8 fb = frag_direct(iB, wn * WNT * 16 + nt * 16 + fr, ks * 4 + fg);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[mt], acc[mt][nt], 0, 0, 0);
      }
    }
  };

  if (NL) {
    if (loader) {
      if (nk > 0) issue(0, kbeg);
      if (nk > 1) issue(1, kbeg + BK);
      int slot = 0;
      for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vmcnt<NDMA_>();
        else             wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(slot >= 1 ? slot - 1 : 2, kbeg + (kt + 2) * BK);
        slot = slot == 2 ? 0 : slot + 1;
      }
      return;
    }
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_barrier();
      compute(slot);
      slot = slot == 2 ? 0 : slot + 1;
    }
  } else {
    if (nk > 0) issue(0, kbeg);
    if (nk > 1) issue(1, kbeg + BK);
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vmcnt<NDMA_>();
      else             wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nk) issue(slot >= 1 ? slot - 1 : 2, kbeg + (kt + 2) * BK);       // (kt + 2) % 3
      compute(slot);
      slot = slot == 2 ? 0 : slot + 1;
    }
  }

#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int row = m0 + wm * 64 + mt * 16 + fr;
#pragma unroll
    for (int nt = 0; nt < WNT; ++nt) {
      const int col = n0 + wn * WNT * 16 + nt * 16 + 4 * fg;
      if (col < g.N) epi_store<EPI>(g, row, col, acc[mt][nt]);
    }
  }
}

template __global__ void k_cgemm<8, EPI_BF16, false>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<8, EPI_RES, false>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<8, EPI_BF16, true>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<8, EPI_RES, true>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<10, EPI_BF16, false>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<10, EPI_RES, false>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<10, EPI_BF16, true>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<10, EPI_RES, true>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<8, EPI_BF16, false, 4>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<8, EPI_RES, false, 4>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<8, EPI_BF16, true, 4>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<8, EPI_RES, true, 4>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<10, EPI_BF16, false, 4>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<10, EPI_RES, false, 4>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<10, EPI_BF16, true, 4>(BGemmArgs, unsigned, unsigned);
template __global__ void k_cgemm<10, EPI_RES, true, 4>(BGemmArgs, unsigned, unsigned);

// ---- the same pipeline for the products with a TRANSPOSED-READ operand (contraction index = matrix row):
//   T[p][q] = sum_k P[.][.] Q[k][q],  tile 256 (p) x 128 (q) x 64 (k), 8 waves as 4 x 2 (64 x 64 each), three slots
//   P_TR = false: P direct [p][k] (Linear input gradient  dX = dY W : P = dY, Q = W [k = out][q = in])
//   P_TR = true : P read transposed [k][p] (weight gradients; contraction over pixels / tokens)
//     CONVP: P is the im2col view of an NHWC image ([k = output pixel][p = (tap, ci)]): a lane's column chunk fixes its tap, per
//            K-tile it decodes its pixel rows (magic-number division) and shifts them by the tap
//     SWAP : the result is stored transposed, C[q][p] (convolution weight gradient [Co][taps * Ci] with the 256-wide tile on
//            the long (tap, ci) axis and the 128-wide one on Co: 320 / 640 / 1280 and 128 / 256 outputs tile without waste)
// Transposed fragments are ds_read_b64_tr_b16 through inline asm (hipcc would drain vmcnt(0) before the builtin while an
// LDS-DMA is in flight, gemm.hip); their completion is waited for explicitly.
__device__ __forceinline__ unsigned lds_addr_of(const __bf16* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(const void*)p;
}
template <int OFF>
__device__ __forceinline__ bf16x4 asm_tr_off(unsigned addr) {
  bf16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ bf16x8 asm_b128_off(unsigned addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ void lds_reads_done() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

struct TTile {
  static constexpr int FBP = 256, FBQ = 128, NW = 8, NSLOT = 3;
  static constexpr int P_ELEMS = FBP * 64, Q_ELEMS = FBQ * 64, SLOT = P_ELEMS + Q_ELEMS;
  static constexpr int NP = 4, NQ = 2, NDMA = NP + NQ;
  static constexpr size_t LDS = (size_t)NSLOT * SLOT * sizeof(__bf16);
};

template <bool P_TR, bool CONVP, bool SWAP, int EPI, int NL = 0>        // NL: loader waves, as k_cgemm
__global__ __launch_bounds__(512 + 64 * NL) void k_cgemm_t(BGemmArgs g, unsigned p_bytes, unsigned q_bytes, unsigned magic_w, unsigned magic_h) {
  using TT = TTile;
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = (wave >> 1) & 3, wq = wave & 1;
  constexpr int NWD = NL ? NL : TT::NW;                       // waves that issue DMA
  constexpr int NP_ = TT::NP * TT::NW / NWD, NQ_ = TT::NQ * TT::NW / NWD, NDMA_ = NP_ + NQ_;
  const bool loader = NL && wave >= TT::NW;
  const int dwave = NL ? (wave - TT::NW) & (NWD - 1) : wave;
  // P / Q roles: the descriptor's A is P unless SWAP (then its B -- the im2col operand -- is P and the output is stored transposed)
  const __bf16* Pp = SWAP ? g.B : g.A;
  const __bf16* Qp = SWAP ? g.A : g.B;
  const int ldp = SWAP ? g.ldb : g.lda, ldq = SWAP ? g.lda : g.ldb;
  const int NPd = SWAP ? g.N : g.M, NQd = SWAP ? g.M : g.N;        // extents of the p and q axes
  const int ntq = (NQd + TT::FBQ - 1) / TT::FBQ;
  int id;
  {
    const int nblk = gridDim.x, b = blockIdx.x, qq = nblk >> 3, r = nblk & 7, x = b & 7, y = b >> 3;
    id = (x < r ? x * (qq + 1) : r * (qq + 1) + (x - r) * qq) + y;
  }
  const int tp = id / ntq, tq = id - tp * ntq;
  const int p0 = tp * TT::FBP, q0 = tq * TT::FBQ;
  int kbeg = 0, kend = g.K;
  if (g.kchunk > 0) {
    kbeg = (int)blockIdx.y * g.kchunk; kend = min(g.K, kbeg + g.kchunk);
    if (g.Cf) g.Cf += (long)blockIdx.y * g.sC;
  }
  const int nk = (kend - kbeg) / BK;

  const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)Pp, 0, (int)p_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qp, 0, (int)q_bytes, 0x00020000);

  // ---- DMA plan.  Direct P image [256 p][64 k] (8 chunks per row, chunk ^= row & 7); transposed-read images [64 k][256 | 128]
  // (32 | 16 chunks per row, low four chunk bits ^= swz_tr(k row)).  Out-of-range columns of a ragged last tile read the next
  // row's data or zeros; their outputs are never stored.
  int p_off[NP_], q_off[NQ_];
  int p_kh[NP_], p_kw[NP_], p_ci[NP_];           // CONVP: tap and channel offset of the lane's column chunk (tap < 0: none)
#pragma unroll
  for (int i = 0; i < NP_; ++i) {
    const int j = dwave + i * NWD;
    if (!P_TR) {
      const int row = j * 8 + (lane >> 3);
      p_off[i] = 2 * ((p0 + row) * ldp + (((lane & 7) ^ ((lane >> 3) & 7)) << 3));
    } else {
      const int krow = 2 * j + (lane >> 5), pch = lane & 31;
      const int lc = (pch & ~15) | ((pch & 15) ^ swz_tr(krow));
      const int col = p0 + (lc << 3);
      if (!CONVP) p_off[i] = 2 * (krow * ldp + col);
      else {
        const int tap = col / g.cg.C;
        p_ci[i] = col - tap * g.cg.C;
        const int tt = tap < g.cg.taps ? tap : -64;
        const int kh = g.cg.taps == 9 ? (tt >= 0 ? (tt * 11) >> 5 : -64) : (tt >= 0 ? 0 : -64);
        p_kh[i] = kh; p_kw[i] = g.cg.taps == 9 ? tt - 3 * ((tt * 11) >> 5) : 0;
        p_off[i] = krow;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NQ_; ++i) {
    const int j = dwave + i * NWD;
    const int krow = 4 * j + (lane >> 4), pch = lane & 15;
    q_off[i] = 2 * (krow * ldq + q0 + ((pch ^ swz_tr(krow)) << 3));
  }
  const int Hv = (g.cg.up | g.cg.dil) ? 2 * g.cg.Hs : g.cg.Hs, Wv = (g.cg.up | g.cg.dil) ? 2 * g.cg.Ws : g.cg.Ws;
  auto issue = [&](int slot, int k0) {
    __bf16* iP = smem + slot * TT::SLOT;
    __bf16* iQ = iP + TT::P_ELEMS;
#pragma unroll
    for (int i = 0; i < NP_; ++i) {
      int voff = p_off[i], soff = 0;
      if (!P_TR) soff = 2 * k0;
      else if (!CONVP) soff = 2 * k0 * ldp;
      else {
        const unsigned px = (unsigned)(k0 + p_off[i]);                  // output pixel of this lane's k row
        const unsigned t = __umulhi(px, magic_w), wo = px - t * g.cg.Wo;
        const unsigned b = __umulhi(t, magic_h), ho = t - b * g.cg.Ho;
        int hi = (int)ho * g.cg.stride + p_kh[i] - g.cg.pad, wi = (int)wo * g.cg.stride + p_kw[i] - g.cg.pad;
        bool ok = (unsigned)hi < (unsigned)Hv && (unsigned)wi < (unsigned)Wv;
        if (g.cg.dil) ok = ok && !((hi | wi) & 1);
        if (g.cg.up | g.cg.dil) { hi >>= 1; wi >>= 1; }
        voff = ok ? 2 * (((int)b * g.cg.Hs * g.cg.Ws + hi * g.cg.Ws + wi) * ldp + p_ci[i]) : 0x7ffffff0;
      }
      dma16(rsP, iP + (dwave + i * NWD) * 512, voff, soff);
    }
#pragma unroll
    for (int i = 0; i < NQ_; ++i) dma16(rsQ, iQ + (dwave + i * NWD) * 512, q_off[i], 2 * k0 * ldq);
  };

  // ---- fragment addresses (bytes, relative to the slot): transposed-read fragments of 16 columns x 32 k
  const int fg = lane >> 4, fi = lane & 15, fq = fi >> 2, fp = fi & 3;
  const int r0 = 8 * fg + fq;                               // k row (+4 for the upper half, + 32 for the second k-step)
  const int sw_lo = swz_tr(r0), sw_hi = swz_tr(r0 + 4);
  unsigned pa[4][2], qa[4][2];                              // P: [mt][lo/hi] (direct P: [.][k-step]); Q: [nt][lo/hi]
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (P_TR) {
      const int ch = ((wp * 64 + t * 16) >> 3) + (fp >> 1);
      pa[t][0] = 2 * (r0 * 256 + (((ch & ~15) | ((ch & 15) ^ sw_lo)) << 3) + 4 * (fp & 1));
      pa[t][1] = 2 * ((r0 + 4) * 256 + (((ch & ~15) | ((ch & 15) ^ sw_hi)) << 3) + 4 * (fp & 1));
    } else {
      const int row = wp * 64 + t * 16 + fi;
      pa[t][0] = 2 * (row * 64 + (((0 + fg) ^ (row & 7)) << 3));
      pa[t][1] = 2 * (row * 64 + (((4 + fg) ^ (row & 7)) << 3));
    }
    const int chq = ((wq * 64 + t * 16) >> 3) + (fp >> 1);
    qa[t][0] = 2 * TT::P_ELEMS + 2 * (r0 * 128 + ((chq ^ sw_lo) << 3) + 4 * (fp & 1));
    qa[t][1] = 2 * TT::P_ELEMS + 2 * ((r0 + 4) * 128 + ((chq ^ sw_hi) << 3) + 4 * (fp & 1));
  }
  const unsigned smem_base = lds_addr_of(smem);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int slot) {
    const unsigned sb = smem_base + (unsigned)slot * (TT::SLOT * 2);
    bf16x8 fpv[2][4], fqv[2][4];
    static_for<2>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      static_for<4>([&](auto T) {
        constexpr int t = decltype(T)::value;
        if constexpr (P_TR) {
          const bf16x4 lo = asm_tr_off<ks * 32 * 256 * 2>(sb + pa[t][0]);
          const bf16x4 hi = asm_tr_off<ks * 32 * 256 * 2>(sb + pa[t][1]);
          fpv[ks][t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        } else {
          fpv[ks][t] = asm_b128_off<0>(sb + pa[t][ks]);
        }
        const bf16x4 lo = asm_tr_off<ks * 32 * 128 * 2>(sb + qa[t][0]);
        const bf16x4 hi = asm_tr_off<ks * 32 * 128 * 2>(sb + qa[t][1]);
        fqv[ks][t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      });
      lds_reads_done();
      static_for<4>([&](auto MT) {
        constexpr int mt = decltype(MT)::value;
        static_for<4>([&](auto NTI) {
          constexpr int nt = decltype(NTI)::value;
          if constexpr (SWAP) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fpv[ks][mt], fqv[ks][nt], acc[mt][nt], 0, 0, 0);
          else                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fqv[ks][nt], fpv[ks][mt], acc[mt][nt], 0, 0, 0);
        });
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  if (NL) {
    if (loader) {
      if (nk > 0) issue(0, kbeg);
      if (nk > 1) issue(1, kbeg + BK);
      int slot = 0;
      for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) wait_vmcnt<NDMA_>();
        else             wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue(slot >= 1 ? slot - 1 : 2, kbeg + (kt + 2) * BK);
        slot = slot == 2 ? 0 : slot + 1;
      }
      return;
    }
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_barrier();
      compute(slot);
      slot = slot == 2 ? 0 : slot + 1;
    }
  } else {
    if (nk > 0) issue(0, kbeg);
    if (nk > 1) issue(1, kbeg + BK);
    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vmcnt<NDMA_>();
      else             wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (kt + 2 < nk) issue(slot >= 1 ? slot - 1 : 2, kbeg + (kt + 2) * BK);
      compute(slot);
      slot = slot == 2 ? 0 : slot + 1;
    }
  }

  if constexpr (SWAP) {
    // lane holds T[p = 4 fg + r][q = fi] of each 16 x 16 block: four consecutive p -> one float4 of row q of the transposed result
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int pc = p0 + wp * 64 + mt * 16 + 4 * fg;
      if (pc >= NPd) continue;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int qr = q0 + wq * 64 + nt * 16 + fi;
        if (qr >= NQd) continue;
        float4* dst = reinterpret_cast<float4*>(g.Cf + (size_t)qr * g.ldcf + pc);
        const f32x4 v = acc[mt][nt] * g.alpha;
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (g.accumulate) { const float4 c = *dst; o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
        *dst = o;
      }
    }
  } else {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = p0 + wp * 64 + mt * 16 + fi;
      if (row >= NPd) continue;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int col = q0 + wq * 64 + nt * 16 + 4 * fg;
        if (col < NQd) epi_store<EPI>(g, row, col, acc[mt][nt]);
      }
    }
  }
}

template __global__ void k_cgemm_t<false, false, false, EPI_BF16>(BGemmArgs, unsigned, unsigned, unsigned, unsigned);
template __global__ void k_cgemm_t<false, false, false, EPI_RES>(BGemmArgs, unsigned, unsigned, unsigned, unsigned);
template __global__ void k_cgemm_t<true, false, false, EPI_RES>(BGemmArgs, unsigned, unsigned, unsigned, unsigned);
template __global__ void k_cgemm_t<true, true, true, EPI_RES>(BGemmArgs, unsigned, unsigned, unsigned, unsigned);
template __global__ void k_cgemm_t<false, false, false, EPI_BF16, 4>(BGemmArgs, unsigned, unsigned, unsigned, unsigned);
template __global__ void k_cgemm_t<false, false, false, EPI_RES, 4>(BGemmArgs, unsigned, unsigned, unsigned, unsigned);
template __global__ void k_cgemm_t<true, false, false, EPI_RES, 4>(BGemmArgs, unsigned, unsigned, unsigned, unsigned);
template __global__ void k_cgemm_t<true, true, true, EPI_RES, 4>(BGemmArgs, unsigned, unsigned, unsigned, unsigned);

int g_conv_loader_waves = 3;       // sfron_gemm_loader_waves(): bit 0 k_cgemm, bit 1 k_cgemm_t in the loader-wave form (0 = every wave issues its own DMA)
namespace {

constexpr int TPB = 256;
// rows per workgroup of the row-vectorised elementwise kernels: about 2048 workgroups over (column blocks x row chunks), >= 4 rows
inline int rows_chunk(int64_t rows, int C) {
  const int64_t cb = (C + 255) / 256;
  int64_t chunks = 2048 / cb; if (chunks < 1) chunks = 1;
  int64_t r = (rows + chunks - 1) / chunks;
  r = (r + 3) / 4 * 4;
  return (int)(r < 4 ? 4 : r);
}
inline int grid_for(int64_t n, int per = TPB) {
  int64_t b = (n + per - 1) / per;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---- weights: fp32 OIHW master -> bf16 GEMM operands.
// fwd  [Co_p][taps][Ci_p]   (rows beyond Cout and channels beyond Cin are zero)
// dgr  [Ci][taps][Co_p]     with the tap index flipped (tap' = taps - 1 - tap): B operand of the input-gradient GEMM
__global__ __launch_bounds__(TPB) void k_conv_wprep(const float* __restrict__ w, int Co, int Ci, int taps, int Co_p, int Ci_p,
                                                    __bf16* __restrict__ fwd, __bf16* __restrict__ dgr) {
  const int64_t nf = (int64_t)Co_p * taps * Ci_p;
  const int64_t nd = dgr ? (int64_t)Ci * taps * Co_p : 0;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nf + nd; i += (int64_t)gridDim.x * TPB) {
    if (i < nf) {
      const int ci = (int)(i % Ci_p); const int t = (int)((i / Ci_p) % taps); const int co = (int)(i / ((int64_t)Ci_p * taps));
      fwd[i] = (co < Co && ci < Ci) ? f2bf(w[((int64_t)co * Ci + ci) * taps + t]) : (__bf16)0.0f;
    } else {
      const int64_t j = i - nf;
      const int co = (int)(j % Co_p); const int t = (int)((j / Co_p) % taps); const int ci = (int)(j / ((int64_t)Co_p * taps));
      dgr[j] = co < Co ? f2bf(w[((int64_t)co * Ci + ci) * taps + (taps - 1 - t)]) : (__bf16)0.0f;
    }
  }
}
// the same for the large 3x3 kernels of the LDM UNet through an LDS tile of 32 output x 32 input channels (bf16 [32][32][9]): the OIHW
// master is read in contiguous runs of 32 * 9 floats, the forward operand is written in runs of 32 input channels per (co, tap),
// the input-gradient operand in runs of 32 output channels per (ci, flipped tap).  grid = (ceil(Ci_p / 32), ceil(Co_p / 32)).
__device__ __forceinline__ void wprep9_tile(const float* __restrict__ w, int Co, int Ci, int Co_p, int Ci_p, __bf16* __restrict__ fwd,
                                            __bf16* __restrict__ dgr, int ci0, int co0, __bf16 (*t)[32 * 9 + 2]) {
  for (int i = threadIdx.x; i < 32 * 288; i += TPB) {
    const int co = i / 288, k = i - co * 288, ci = k / 9;
    const bool in = co0 + co < Co && ci0 + ci < Ci;
    t[co][k] = in ? f2bf(w[((int64_t)(co0 + co) * Ci + ci0) * 9 + k]) : (__bf16)0.0f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 9 * 32; i += TPB) {          // fwd [Co_p][9][Ci_p]
    const int ci = i & 31, tp = (i >> 5) % 9, co = i / 288;
    if (co0 + co < Co_p && ci0 + ci < Ci_p) fwd[((int64_t)(co0 + co) * 9 + tp) * Ci_p + ci0 + ci] = t[co][ci * 9 + tp];
  }
  if (dgr)
    for (int i = threadIdx.x; i < 32 * 9 * 32; i += TPB) {        // dgr [Ci][9 flipped][Co_p]
      const int co = i & 31, tp = (i >> 5) % 9, ci = i / 288;
      if (co0 + co < Co_p && ci0 + ci < Ci) dgr[((int64_t)(ci0 + ci) * 9 + tp) * Co_p + co0 + co] = t[co][ci * 9 + (8 - tp)];
    }
}
__global__ __launch_bounds__(TPB) void k_conv_wprep9(const float* __restrict__ w, int Co, int Ci, int Co_p, int Ci_p, __bf16* __restrict__ fwd,
                                                     __bf16* __restrict__ dgr) {
  __shared__ __bf16 t[32][32 * 9 + 2];
  wprep9_tile(w, Co, Ci, Co_p, Ci_p, fwd, dgr, blockIdx.x * 32, blockIdx.y * 32, t);
}
// every 3x3 kernel of a model in ONE launch: block b serves tile (b - item.tile0) of the item whose tile range holds it
__global__ __launch_bounds__(TPB) void k_conv_wprep9_batch(const sfron_wprep_item* __restrict__ items, int n_items) {
  __shared__ __bf16 t[32][32 * 9 + 2];
  int lo = 0, hi = n_items - 1;
  const int b = blockIdx.x;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (items[mid].tile0 <= b) lo = mid; else hi = mid - 1; }
  const sfron_wprep_item it = items[lo];
  const int local = b - it.tile0, tx = (it.ci_p + 31) / 32;
  wprep9_tile(it.w, it.co, it.ci, it.co_p, it.ci_p, (__bf16*)it.fwd, (__bf16*)it.dgr, (local % tx) * 32, (local / tx) * 32, t);
}
// weight gradient fp32 [nslab][Co_p][taps][Ci_p] (GEMM output, split-K slabs) -> OIHW gradient (overwrite; slabs added in order)
__device__ __forceinline__ void wgrad_scatter_block(const float* __restrict__ g, int Co, int Ci, int taps, int Ci_p, int nslab,
                                                    int64_t slab_stride, float* __restrict__ dw, int co, int ci0, float* sh) {
  // one workgroup per (64 input channels, output channel): thread (tap group tg, channel c) sums the slabs of taps tg, tg + 4, tg + 8
  // (reads coalesced along ci), the [ci][tap] block is turned through LDS (row stride = taps, odd: no bank conflicts) and leaves as
  // one contiguous run of the OIHW gradient
  constexpr int CB = 64, TG = TPB / CB;
  const int c = threadIdx.x % CB, tg = threadIdx.x / CB;
  const int nci = min(CB, Ci - ci0);
  if (c < nci)
    for (int t = tg; t < taps; t += TG) {
      const float* p = g + ((int64_t)co * taps + t) * Ci_p + ci0 + c;
      float a = 0.f;
      int sl = 0;
      for (; sl + 8 <= nslab; sl += 8) {          // eight strided loads in flight, added in slab order (a load-use loop pays one
        float v8[8];                              // memory latency per slab: 38 us for 64 slabs of a 128 x 128 kernel)
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = p[(int64_t)(sl + u) * slab_stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v8[u];
      }
      for (; sl < nslab; ++sl) a += p[(int64_t)sl * slab_stride];
      sh[c * taps + t] = a;
    }
  __syncthreads();
  float* o = dw + ((int64_t)co * Ci + ci0) * taps;
  for (int k = threadIdx.x; k < nci * taps; k += TPB) o[k] = sh[k];
}
__global__ __launch_bounds__(TPB) void k_conv_wgrad_scatter(const float* __restrict__ g, int Co, int Ci, int taps, int Ci_p, int nslab,
                                                            int64_t slab_stride, float* __restrict__ dw) {
  __shared__ float sh[64 * 9];
  wgrad_scatter_block(g, Co, Ci, taps, Ci_p, nslab, slab_stride, dw, blockIdx.y, blockIdx.x * 64, sh);
}
// the scatters of MANY kernel gradients in one launch (round 6): the items travel by value in the kernel arguments, as k_reduce_batch's
constexpr int WS_ITEMS = 80;                     // 80 x 48 bytes of the 4 KB of kernel arguments
struct ScatterPack { sfron_wgrad_scatter_item it[WS_ITEMS]; };
__global__ __launch_bounds__(TPB) void k_conv_wgrad_scatter_batch(ScatterPack pack) {
  __shared__ float sh[64 * 9];
  const sfron_wgrad_scatter_item it = pack.it[blockIdx.y];
  const int ncb = (it.c_in + 63) / 64;
  if ((int)blockIdx.x >= ncb * it.c_out) return;                  // uniform per workgroup
  const int co = blockIdx.x / ncb, ci0 = (blockIdx.x - co * ncb) * 64;
  wgrad_scatter_block(it.dw_gemm, it.c_out, it.c_in, it.taps, it.c_in_p, it.n_slabs, it.slab_stride, it.dw_oihw, co, ci0, sh);
}

// ---- layout: NCHW fp32 image <-> NHWC rows
__global__ __launch_bounds__(TPB) void k_nchw_to_rows(const float* __restrict__ x, int B, int C, int HW, int Cp, __bf16* __restrict__ rows) {
  const int64_t n = (int64_t)B * HW * Cp;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int c = (int)(i % Cp); const int64_t p = i / Cp; const int hw = (int)(p % HW); const int b = (int)(p / HW);
    rows[i] = c < C ? f2bf(x[((int64_t)b * C + c) * HW + hw]) : (__bf16)0.0f;
  }
}
__global__ __launch_bounds__(TPB) void k_rows_to_nchw(const float* __restrict__ rows, int ld, int B, int C, int HW, float* __restrict__ x) {
  const int64_t n = (int64_t)B * C * HW;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int hw = (int)(i % HW); const int c = (int)((i / HW) % C); const int b = (int)(i / ((int64_t)HW * C));
    x[i] = rows[((int64_t)b * HW + hw) * ld + c];
  }
}
__global__ __launch_bounds__(TPB) void k_nchw_to_rows_f32(const float* __restrict__ x, int B, int C, int HW, int ld, float* __restrict__ rows) {
  const int64_t n = (int64_t)B * HW * ld;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int c = (int)(i % ld); const int64_t p = i / ld; const int hw = (int)(p % HW); const int b = (int)(p / HW);
    rows[i] = c < C ? x[((int64_t)b * C + c) * HW + hw] : 0.f;
  }
}

// ---- GroupNorm(G groups, eps) (+ swish) (+ dropout mask) on NHWC rows, one workgroup per (sample, group)
// y = bf16( act(xhat * gamma + beta) * (mask ? mask * drop_scale : 1) ), xhat = (x - mean) * rstd; saves mean / rstd
__global__ __launch_bounds__(TPB) void k_gn_fwd(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, int HW, int C, int G, float eps, int swish,
                                                const uint8_t* __restrict__ mask, float drop_scale, __bf16* __restrict__ y,
                                                float* __restrict__ mean, float* __restrict__ rstd) {
  __shared__ double red[2][TPB / 64];
  __shared__ float stat[2];
  const int b = blockIdx.x / G, gi = blockIdx.x % G, cg = C / G;
  const float* xb = x + (size_t)b * HW * ldx + gi * cg;
  const int n = HW * cg;
  // thread (slot, c) = (tid / cg, tid % cg) walks the pixels slot, slot + tpc, ... of its channel (no per-element division)
  const int tpc = TPB / cg, slot = threadIdx.x / cg, c = threadIdx.x - slot * cg;
  double s = 0.0, ss = 0.0;
  if (slot < tpc)
    for (int p = slot; p < HW; p += tpc) {
      const float v = xb[(size_t)p * ldx + c];
      s += v; ss += (double)v * v;
    }
  s = wave_sum_d(s); ss = wave_sum_d(ss);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, q = 0;
    for (int w = 0; w < TPB / 64; ++w) { a += red[0][w]; q += red[1][w]; }
    const double m = a / n;
    double var = q / n - m * m;                 // biased variance, as torch.nn.GroupNorm
    var = var < 0 ? 0 : var;
    stat[0] = (float)m; stat[1] = (float)(1.0 / sqrt(var + (double)eps));
    mean[blockIdx.x] = stat[0]; rstd[blockIdx.x] = stat[1];
  }
  __syncthreads();
  const float m = stat[0], r = stat[1];
  __bf16* yb = y + (size_t)b * HW * C + gi * cg;
  const uint8_t* mb = mask ? mask + (size_t)b * HW * C + gi * cg : nullptr;
  if (slot < tpc) {
    const float ga = gamma[gi * cg + c], be = beta[gi * cg + c];
    for (int p = slot; p < HW; p += tpc) {
      float z = (xb[(size_t)p * ldx + c] - m) * r * ga + be;
      if (swish) z = silu(z);
      if (mb) z = mb[(size_t)p * C + c] ? z * drop_scale : 0.f;
      yb[(size_t)p * C + c] = f2bf(z);
    }
  }
}
// backward: dy = gradient wrt the bf16 output (fp32 rows, ld = C); dx (+)= d GroupNorm; per-sample partial parameter
// gradients pg[b][C], pb[b][C] (summed over the batch by sfron_reduce_chunks: fixed order)
__global__ __launch_bounds__(TPB) void k_gn_bwd(const float* __restrict__ dy, const float* __restrict__ x, int ldx,
                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                const float* __restrict__ mean, const float* __restrict__ rstd, int HW, int C, int G,
                                                int swish, const uint8_t* __restrict__ mask, float drop_scale,
                                                float* __restrict__ dx, int lddx, int accumulate, float* __restrict__ pg,
                                                float* __restrict__ pb) {
  extern __shared__ float sh[];                 // [TPB/64][2] wave partials + per-channel [2][cg] accumulators
  const int b = blockIdx.x / G, gi = blockIdx.x % G, cg = C / G;
  float* chs = sh + 2 * (TPB / 64);             // [2][cg]: sum dz * xhat, sum dz   per channel
  const float m = mean[blockIdx.x], r = rstd[blockIdx.x];
  const float* xb = x + (size_t)b * HW * ldx + gi * cg;
  const float* dyb = dy + (size_t)b * HW * C + gi * cg;
  const uint8_t* mb = mask ? mask + (size_t)b * HW * C + gi * cg : nullptr;
  const int n = HW * cg;
  // pass 1: dz = dy * act'(z) (* dropout); group sums of dz * gamma and dz * gamma * xhat; per-channel sums.
  // Thread (slot, c) = (tid / cg, tid % cg) owns channel c and visits the pixels slot, slot + tpc, ... (tpc = TPB / cg slots; the
  // TPB - tpc * cg surplus threads idle here): consecutive threads still read consecutive addresses of a pixel's cg-channel run, and
  // every per-channel sum is formed in a fixed order (no atomics: bitwise reproducible for any channels-per-group <= TPB).
  float s1 = 0.f, s2 = 0.f;
  float ca = 0.f, cb = 0.f;
  const int tpc = TPB / cg, slot = threadIdx.x / cg, c = threadIdx.x - slot * cg;
  if (slot < tpc) {
    const float ga = gamma[gi * cg + c], be = beta[gi * cg + c];
    for (int p = slot; p < HW; p += tpc) {
      const float xh = (xb[(size_t)p * ldx + c] - m) * r;
      float d = dyb[(size_t)p * C + c];
      if (mb) d = mb[(size_t)p * C + c] ? d * drop_scale : 0.f;
      if (swish) d *= silu_grad(xh * ga + be);
      s1 += d * ga; s2 += d * ga * xh;
      ca += d * xh; cb += d;
    }
  }
  float* tmp = sh + 2 * (TPB / 64) + 2 * cg;     // [2][TPB]
  tmp[threadIdx.x] = ca; tmp[TPB + threadIdx.x] = cb;
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s1; sh[TPB / 64 + (threadIdx.x >> 6)] = s2; }
  __syncthreads();
  if (threadIdx.x < cg) {
    float a = 0.f, q = 0.f;
    for (int j = threadIdx.x; j < tpc * cg; j += cg) { a += tmp[j]; q += tmp[TPB + j]; }
    chs[threadIdx.x] = a; chs[cg + threadIdx.x] = q;
  }
  float t1 = 0.f, t2 = 0.f;
  for (int w = 0; w < TPB / 64; ++w) { t1 += sh[w]; t2 += sh[TPB / 64 + w]; }
  __syncthreads();
  if (threadIdx.x < cg) {
    pg[(size_t)b * C + gi * cg + threadIdx.x] = chs[threadIdx.x];
    pb[(size_t)b * C + gi * cg + threadIdx.x] = chs[cg + threadIdx.x];
  }
  const float inv = 1.0f / (float)n;
  const float k1 = t1 * inv, k2 = t2 * inv;
  float* dxb = dx + (size_t)b * HW * lddx + gi * cg;
  if (slot < tpc) {
    const float ga = gamma[gi * cg + c], be = beta[gi * cg + c];
    for (int p = slot; p < HW; p += tpc) {
      const float xh = (xb[(size_t)p * ldx + c] - m) * r;
      float d = dyb[(size_t)p * C + c];
      if (mb) d = mb[(size_t)p * C + c] ? d * drop_scale : 0.f;
      if (swish) d *= silu_grad(xh * ga + be);
      const float v = r * (d * ga - k1 - xh * k2);
      float* o = dxb + (size_t)p * lddx + c;
      *o = accumulate ? *o + v : v;
    }
  }
}

// ---- GroupNorm, row-coalesced two-phase form (the production path; the per-(sample, group) kernels above serve odd shapes).
// NHWC rows: a (sample, group) slab is HW runs of cg floats, 4 * C bytes apart -- a workgroup per slab touches 16..40 useful bytes of
// every 128-B line.  Here a workgroup owns a CHUNK OF PIXELS of one sample and all C channels: threads read whole rows as float4
// (thread = (row replica r, quad lane q); C / 4 quads per row, up to three quads per thread for C > 1024), so every load is a
// full-line stream.  Phase 1 leaves per-chunk partial sums, phase 2 (same chunking) combines them in a fixed order in its
// prologue and applies the normalisation; all per-channel / per-group sums are formed in a fixed order (bitwise reproducible).
constexpr int GNB = 1024;                       // 16 waves per workgroup: with one workgroup per (sample, chunk) the chip is only full this way
constexpr int GN_MAXQ = 1;                      // quads per thread: C <= 4 * GNB = 4096
struct GnMap {
  int qpr, qw, rpp, r, ql, nq;                  // quads per row, quad lanes, rows per pass, this thread's row replica / lane, its quads
  __device__ __forceinline__ void init(int C) {
    qpr = C >> 2; qw = qpr < GNB ? qpr : GNB; rpp = GNB / qw;
    r = threadIdx.x / qw; ql = threadIdx.x - r * qw; nq = (qpr - ql + qw - 1) / qw;
    if (r >= rpp) nq = 0;
  }
};
__host__ __device__ inline int gn_chunks(int B, int HW) {
  int n = (512 + B - 1) / B;
  if (n > 32) n = 32;
  if (n > HW / 8) n = HW / 8;
  return n < 1 ? 1 : n;
}

// phase 1 forward: partial (sum, sum of squares) per (sample, chunk, group), fp64
__global__ __launch_bounds__(GNB) void k_gn2_stats(const float* __restrict__ x, int ldx, int HW, int C, int G, int nchunk, double* __restrict__ part) {
  extern __shared__ double shd[];               // [rpp][C][2]
  const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk, cg = C / G;
  const int p0 = (int)((long)HW * ch / nchunk), p1 = (int)((long)HW * (ch + 1) / nchunk);
  GnMap m; m.init(C);
  double s[GN_MAXQ][4], ss[GN_MAXQ][4];
#pragma unroll
  for (int j = 0; j < GN_MAXQ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[j][e] = 0.0; ss[j][e] = 0.0; }
  const float* xb = x + (size_t)b * HW * ldx;
  for (int p = p0 + m.r; p < p1; p += m.rpp) {
#pragma unroll
    for (int j = 0; j < GN_MAXQ; ++j)
      if (j < m.nq) {
        const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)p * ldx + 4 * (m.ql + j * m.qw));
        s[j][0] += v.x; ss[j][0] += (double)v.x * v.x; s[j][1] += v.y; ss[j][1] += (double)v.y * v.y;
        s[j][2] += v.z; ss[j][2] += (double)v.z * v.z; s[j][3] += v.w; ss[j][3] += (double)v.w * v.w;
      }
  }
  if (m.r < m.rpp) {
#pragma unroll
    for (int j = 0; j < GN_MAXQ; ++j)
      if (j < m.nq)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 4 * (m.ql + j * m.qw) + e;
          shd[((size_t)m.r * C + c) * 2] = s[j][e]; shd[((size_t)m.r * C + c) * 2 + 1] = ss[j][e];
        }
  }
  __syncthreads();
  // one wave per group at a time: lane l adds the words l, l + 64, ... of the group's [rpp][cg] block in index order, then a butterfly
  // over the lanes -- a fixed order (bitwise reproducible).  (Was: G threads walking rpp * cg words each, 256 dependent LDS reads
  // = most of the kernel at the small levels.)
  const int lane = threadIdx.x & 63, nw = GNB / 64, per = m.rpp * cg;
  for (int g = threadIdx.x >> 6; g < G; g += nw) {
    double a = 0.0, q = 0.0;
    for (int i = lane; i < per; i += 64) {
      const int r = i / cg, c = g * cg + (i - r * cg);
      a += shd[((size_t)r * C + c) * 2]; q += shd[((size_t)r * C + c) * 2 + 1];
    }
    a = wave_sum_d(a); q = wave_sum_d(q);
    if (lane == 0) { double* o = part + (((size_t)b * nchunk + ch) * G + g) * 2; o[0] = a; o[1] = q; }
  }
}
// phase 2 forward: mean / rstd from the partials (every workgroup of a sample forms them the same way; chunk 0 stores them), apply
__global__ __launch_bounds__(GNB) void k_gn2_apply(const float* __restrict__ x, int ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   int HW, int C, int G, float eps, int swish, const uint8_t* __restrict__ mask, float drop_scale,
                                                   int nchunk, const double* __restrict__ part, __bf16* __restrict__ y, float* __restrict__ mean,
                                                   float* __restrict__ rstd) {
  __shared__ float st[2][64];
  __shared__ double sp[32 * 64 * 2];            // [chunk][group][2]: every partial of the sample, fetched by nchunk * G threads at once
  const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk, cg = C / G;
  for (int i = threadIdx.x; i < nchunk * G * 2; i += GNB) sp[i] = part[(size_t)b * nchunk * G * 2 + i];
  __syncthreads();
  if (threadIdx.x < G) {
    double a = 0.0, q = 0.0;
    for (int k = 0; k < nchunk; ++k) { const double* o = sp + ((size_t)k * G + threadIdx.x) * 2; a += o[0]; q += o[1]; }
    const double n = (double)HW * cg, mu = a / n;
    double var = q / n - mu * mu;                 // biased variance, as torch.nn.GroupNorm
    var = var < 0 ? 0 : var;
    st[0][threadIdx.x] = (float)mu; st[1][threadIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    if (ch == 0) { mean[b * G + threadIdx.x] = st[0][threadIdx.x]; rstd[b * G + threadIdx.x] = st[1][threadIdx.x]; }
  }
  __syncthreads();
  const int p0 = (int)((long)HW * ch / nchunk), p1 = (int)((long)HW * (ch + 1) / nchunk);
  GnMap m; m.init(C);
  float mu[GN_MAXQ][4], rs[GN_MAXQ][4], ga[GN_MAXQ][4], be[GN_MAXQ][4];
#pragma unroll
  for (int j = 0; j < GN_MAXQ; ++j)
    if (j < m.nq)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * (m.ql + j * m.qw) + e, gi = c / cg;
        mu[j][e] = st[0][gi]; rs[j][e] = st[1][gi]; ga[j][e] = gamma[c]; be[j][e] = beta[c];
      }
  const float* xb = x + (size_t)b * HW * ldx;
  __bf16* yb = y + (size_t)b * HW * C;
  const uint8_t* mb = mask ? mask + (size_t)b * HW * C : nullptr;
  for (int p = p0 + m.r; p < p1; p += m.rpp) {
#pragma unroll
    for (int j = 0; j < GN_MAXQ; ++j)
      if (j < m.nq) {
        const int c0 = 4 * (m.ql + j * m.qw);
        const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)p * ldx + c0);
        float z[4] = {v.x, v.y, v.z, v.w};
        uchar4 mk = make_uchar4(1, 1, 1, 1);
        if (mb) mk = *reinterpret_cast<const uchar4*>(mb + (size_t)p * C + c0);
        const uint8_t mke[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = (z[e] - mu[j][e]) * rs[j][e] * ga[j][e] + be[j][e];
          if (swish) t = silu(t);
          if (mb) t = mke[e] ? t * drop_scale : 0.f;
          z[e] = t;
        }
        *reinterpret_cast<bf16x4*>(yb + (size_t)p * C + c0) = bf16x4{f2bf(z[0]), f2bf(z[1]), f2bf(z[2]), f2bf(z[3])};
      }
  }
}
// phase 1 backward: per (sample, chunk, channel) partial sums of dz * xhat and dz, dz = dy * act'(z) (* dropout)
__global__ __launch_bounds__(GNB) void k_gn2_bwd_stats(const float* __restrict__ dy, const float* __restrict__ x, int ldx,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, int HW, int C, int G,
                                                       int swish, const uint8_t* __restrict__ mask, float drop_scale, int nchunk,
                                                       float* __restrict__ part) {
  extern __shared__ float shf[];                // [rpp][C][2]
  const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk, cg = C / G;
  const int p0 = (int)((long)HW * ch / nchunk), p1 = (int)((long)HW * (ch + 1) / nchunk);
  GnMap m; m.init(C);
  float mu[GN_MAXQ][4], rs[GN_MAXQ][4], ga[GN_MAXQ][4], be[GN_MAXQ][4], ca[GN_MAXQ][4], cb[GN_MAXQ][4];
#pragma unroll
  for (int j = 0; j < GN_MAXQ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ca[j][e] = 0.f; cb[j][e] = 0.f;
      if (j < m.nq) {
        const int c = 4 * (m.ql + j * m.qw) + e, gi = c / cg;
        mu[j][e] = mean[b * G + gi]; rs[j][e] = rstd[b * G + gi]; ga[j][e] = gamma[c]; be[j][e] = beta[c];
      }
    }
  const float* xb = x + (size_t)b * HW * ldx;
  const float* dyb = dy + (size_t)b * HW * C;
  const uint8_t* mb = mask ? mask + (size_t)b * HW * C : nullptr;
  for (int p = p0 + m.r; p < p1; p += m.rpp) {
#pragma unroll
    for (int j = 0; j < GN_MAXQ; ++j)
      if (j < m.nq) {
        const int c0 = 4 * (m.ql + j * m.qw);
        const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)p * ldx + c0);
        const float4 dv = *reinterpret_cast<const float4*>(dyb + (size_t)p * C + c0);
        uchar4 mk = make_uchar4(1, 1, 1, 1);
        if (mb) mk = *reinterpret_cast<const uchar4*>(mb + (size_t)p * C + c0);
        const float xv[4] = {v.x, v.y, v.z, v.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
        const uint8_t mke[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xv[e] - mu[j][e]) * rs[j][e];
          float d = dd[e];
          if (mb) d = mke[e] ? d * drop_scale : 0.f;
          if (swish) d *= silu_grad(xh * ga[j][e] + be[j][e]);
          ca[j][e] += d * xh; cb[j][e] += d;
        }
      }
  }
  if (m.r < m.rpp) {
#pragma unroll
    for (int j = 0; j < GN_MAXQ; ++j)
      if (j < m.nq)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 4 * (m.ql + j * m.qw) + e;
          shf[((size_t)m.r * C + c) * 2] = ca[j][e]; shf[((size_t)m.r * C + c) * 2 + 1] = cb[j][e];
        }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += GNB) {
    float a = 0.f, q = 0.f;
    for (int r = 0; r < m.rpp; ++r) { a += shf[((size_t)r * C + c) * 2]; q += shf[((size_t)r * C + c) * 2 + 1]; }
    float* o = part + (((size_t)b * nchunk + ch) * C + c) * 2;
    o[0] = a; o[1] = q;
  }
}
// phase 2 backward: per-channel sums over the chunks -> (chunk 0) the per-sample parameter-gradient partials, the group means
// k1 = mean(dz gamma), k2 = mean(dz gamma xhat); dx (+)= rstd (dz gamma - k1 - xhat k2)
__global__ __launch_bounds__(GNB) void k_gn2_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x, int ldx,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, int HW, int C, int G,
                                                       int swish, const uint8_t* __restrict__ mask, float drop_scale, int nchunk,
                                                       const float* __restrict__ part, float* __restrict__ dx, int lddx, int accumulate,
                                                       float* __restrict__ pg, float* __restrict__ pb, const float* __restrict__ extra,
                                                       int ldextra, __bf16* __restrict__ dx16, float* __restrict__ colpart) {
  // dx16 / colpart (sfron_groupnorm_bwd_cast): the gradient also (or only: dx == nullptr) as the bf16 GEMM operand [rows][C] of the
  // convolution that produced x, and its column sums per (sample, chunk) -- that layer's bias gradient and, per sample, the gradient
  // of a per-sample vector added to x -- which a cast pass and two column-sum passes over an fp32 dx would form otherwise
  extern __shared__ float shf[];                // [C][2] channel sums, then [G][2] group means
  const int b = blockIdx.x / nchunk, ch = blockIdx.x % nchunk, cg = C / G;
  float* kk = shf + 2 * C;
  float* sg = kk + 2 * G;                       // gamma, staged for the group sums below
  for (int c = threadIdx.x; c < C; c += GNB) {
    float a = 0.f, q = 0.f;
    sg[c] = gamma[c];
    for (int k0 = 0; k0 < nchunk; k0 += 8) {    // eight partial pairs in flight (a load-use loop pays one L2 latency per chunk)
      float2 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + u < nchunk ? k0 + u : nchunk - 1;
        t[u] = *reinterpret_cast<const float2*>(part + (((size_t)b * nchunk + k) * C + c) * 2);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u < nchunk) { a += t[u].x; q += t[u].y; }
    }
    shf[2 * c] = a; shf[2 * c + 1] = q;
    if (ch == 0) { pg[(size_t)b * C + c] = a; pb[(size_t)b * C + c] = q; }
  }
  __syncthreads();
  if (threadIdx.x < G) {
    float t1 = 0.f, t2 = 0.f;
    for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; ++c) { const float g1 = sg[c]; t1 += g1 * shf[2 * c + 1]; t2 += g1 * shf[2 * c]; }
    const float inv = 1.0f / ((float)HW * (float)cg);
    kk[2 * threadIdx.x] = t1 * inv; kk[2 * threadIdx.x + 1] = t2 * inv;
  }
  __syncthreads();
  const int p0 = (int)((long)HW * ch / nchunk), p1 = (int)((long)HW * (ch + 1) / nchunk);
  GnMap m; m.init(C);
  float mu[GN_MAXQ][4], rs[GN_MAXQ][4], ga[GN_MAXQ][4], be[GN_MAXQ][4], k1[GN_MAXQ][4], k2[GN_MAXQ][4];
#pragma unroll
  for (int j = 0; j < GN_MAXQ; ++j)
    if (j < m.nq)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = 4 * (m.ql + j * m.qw) + e, gi = c / cg;
        mu[j][e] = mean[b * G + gi]; rs[j][e] = rstd[b * G + gi]; ga[j][e] = gamma[c]; be[j][e] = beta[c];
        k1[j][e] = kk[2 * gi]; k2[j][e] = kk[2 * gi + 1];
      }
  const float* xb = x + (size_t)b * HW * ldx;
  const float* dyb = dy + (size_t)b * HW * C;
  const uint8_t* mb = mask ? mask + (size_t)b * HW * C : nullptr;
  float* dxb = dx ? dx + (size_t)b * HW * lddx : nullptr;
  __bf16* d16b = dx16 ? dx16 + (size_t)b * HW * C : nullptr;
  const float* exb = extra ? extra + (size_t)b * HW * ldextra : nullptr;
  float cs[GN_MAXQ][4];
#pragma unroll
  for (int j = 0; j < GN_MAXQ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[j][e] = 0.f;
  for (int p = p0 + m.r; p < p1; p += m.rpp) {
#pragma unroll
    for (int j = 0; j < GN_MAXQ; ++j)
      if (j < m.nq) {
        const int c0 = 4 * (m.ql + j * m.qw);
        const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)p * ldx + c0);
        const float4 dv = *reinterpret_cast<const float4*>(dyb + (size_t)p * C + c0);
        uchar4 mk = make_uchar4(1, 1, 1, 1);
        if (mb) mk = *reinterpret_cast<const uchar4*>(mb + (size_t)p * C + c0);
        const float xv[4] = {v.x, v.y, v.z, v.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
        const uint8_t mke[4] = {mk.x, mk.y, mk.z, mk.w};
        float4* op = dxb ? reinterpret_cast<float4*>(dxb + (size_t)p * lddx + c0) : nullptr;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (accumulate) { const float4 c = *op; o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w; }
        if (exb) {             // the term a separate dx (+)= extra pass would have added first: (dx + extra) + this layer's gradient
          const float4 c = *reinterpret_cast<const float4*>(exb + (size_t)p * ldextra + c0);
          o[0] += c.x; o[1] += c.y; o[2] += c.z; o[3] += c.w;
        }
        const bool acc = accumulate || exb;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xv[e] - mu[j][e]) * rs[j][e];
          float d = dd[e];
          if (mb) d = mke[e] ? d * drop_scale : 0.f;
          if (swish) d *= silu_grad(xh * ga[j][e] + be[j][e]);
          const float vv = rs[j][e] * (d * ga[j][e] - k1[j][e] - xh * k2[j][e]);
          o[e] = acc ? o[e] + vv : vv;
        }
        if (op) *op = make_float4(o[0], o[1], o[2], o[3]);
        if (d16b) *reinterpret_cast<bf16x4*>(d16b + (size_t)p * C + c0) = bf16x4{f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
        if (colpart) {
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[j][e] += o[e];
        }
      }
  }
  if (colpart) {                                // the row replicas' column sums meet in LDS in replica order (fixed order)
    float* red = shf + 3 * C + 2 * G;           // [rpp][C]
    if (m.r < m.rpp) {
#pragma unroll
      for (int j = 0; j < GN_MAXQ; ++j)
        if (j < m.nq)
#pragma unroll
          for (int e = 0; e < 4; ++e) red[(size_t)m.r * C + 4 * (m.ql + j * m.qw) + e] = cs[j][e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += GNB) {
      float a = 0.f;
      for (int r = 0; r < m.rpp; ++r) a += red[(size_t)r * C + c];
      colpart[((size_t)b * nchunk + ch) * C + c] = a;
    }
  }
}

// ---- GroupNorm in ONE launch per direction (round 6): a workgroup owns one sample and a block of `gpb` whole groups = Cs = gpb * C / G
// consecutive channels over ALL pixels -- statistics and normalisation never leave the workgroup, so the two phases of the k_gn2 pair (and the
// ~5 us boundary between two dependent launches, which at the 16 x 16 ... 4 x 4 levels of the DDPM U-Net is as long as either phase) become
// two passes of one kernel over a slab of at most 128 KB that the second pass finds in L2.  B * G / gpb workgroups (DDPM batch 64: 256 .. 1024):
// the chip stays full, which one workgroup per sample (round 5, profiles/r05_ab_log.txt) did not manage.  Rows are runs of Cs floats
// (>= 64 bytes), read as float4 by (row replica r, quad q) threads.  The same arithmetic per element as k_gn2_*; the sums run over this
// workgroup's rows in a fixed order (replica r takes rows r, r + rpp, ...; replicas, then channels, are added in index order): bitwise
// reproducible, not bit-identical to the chunked pair.
constexpr int GN3_T = 1024;
// the GroupNorm INPUT (forward: x, backward: dy) as the still unfinished result of a split-K product (round 6): n slabs [rows][C] to be added in
// index order, + bias[c] + vec[sample][c] + resid[row][c] -- element for element what k_split_finish4 writes.  The one-launch kernels form each
// element in their first pass, store it to the tensor the finish launch would have filled (their own second pass and every later reader find
// it there), and the finish launch between the product and the GroupNorm no longer exists.  slabs == nullptr: the input is a finished tensor.
struct Gn3Src {
  const float* slabs; int n; int64_t stride;
  const float* bias; const float* vec; int ldvec; const float* resid; int ldresid;
};
__device__ __forceinline__ float4 gn3_src_quad(const Gn3Src& s, int64_t row, int b, int c, int C) {
  const float* p = s.slabs + row * C + c;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  int sl = 0;
  for (; sl + 4 <= s.n; sl += 4) {                // four slab loads in flight, added in slab order
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(p + (sl + u) * s.stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
  }
  for (; sl < s.n; ++sl) { const float4 v = *reinterpret_cast<const float4*>(p + sl * s.stride); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (s.bias) b4 = *reinterpret_cast<const float4*>(s.bias + c);
  a.x += b4.x; a.y += b4.y; a.z += b4.z; a.w += b4.w;
  if (s.vec) { const float4 v = *reinterpret_cast<const float4*>(s.vec + (int64_t)b * s.ldvec + c); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  if (s.resid) { const float4 v = *reinterpret_cast<const float4*>(s.resid + row * s.ldresid + c); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  return a;
}
struct Gn3Map {
  int b, g0, c0, Cs, qpr, rpp, q, r;
  bool on;
  __device__ __forceinline__ void init(int C, int G, int gpb) {
    const int ncb = G / gpb, cg = C / G;
    b = blockIdx.x / ncb;
    g0 = (blockIdx.x - b * ncb) * gpb; c0 = g0 * cg; Cs = gpb * cg; qpr = Cs >> 2; rpp = GN3_T / qpr;
    r = threadIdx.x / qpr; q = threadIdx.x - r * qpr; on = r < rpp;
  }
};
template <bool SRC>
__global__ __launch_bounds__(GN3_T) void k_gn3_fwd(const float* __restrict__ x, int ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   int HW, int C, int G, int gpb, float eps, int swish, const uint8_t* __restrict__ mask,
                                                   float drop_scale, __bf16* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                   Gn3Src src) {
  __shared__ double shd[GN3_T * 4 * 2];         // [rpp][Cs][2], rpp * Cs <= 4 * GN3_T
  __shared__ float st[2][64];
  Gn3Map m; m.init(C, G, gpb);
  const int cg = C / G;
  const float* xb = x + (size_t)m.b * HW * ldx + m.c0 + 4 * m.q;
  double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
  if (SRC && m.on) {                              // x is still S slabs: finish it here (ldx == C), this thread's quads, for its own second pass
    float* xw = const_cast<float*>(xb);
#pragma unroll 2
    for (int p = m.r; p < HW; p += m.rpp) {
      const float4 v = gn3_src_quad(src, (int64_t)m.b * HW + p, m.b, m.c0 + 4 * m.q, C);
      *reinterpret_cast<float4*>(xw + (size_t)p * ldx) = v;
      s[0] += v.x; ss[0] += (double)v.x * v.x; s[1] += v.y; ss[1] += (double)v.y * v.y;
      s[2] += v.z; ss[2] += (double)v.z * v.z; s[3] += v.w; ss[3] += (double)v.w * v.w;
    }
  } else if (!SRC && m.on) {
#pragma unroll 8
    for (int p = m.r; p < HW; p += m.rpp) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)p * ldx);
      s[0] += v.x; ss[0] += (double)v.x * v.x; s[1] += v.y; ss[1] += (double)v.y * v.y;
      s[2] += v.z; ss[2] += (double)v.z * v.z; s[3] += v.w; ss[3] += (double)v.w * v.w;
    }
  }
  if (m.on) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      shd[((size_t)m.r * m.Cs + 4 * m.q + e) * 2] = s[e]; shd[((size_t)m.r * m.Cs + 4 * m.q + e) * 2 + 1] = ss[e];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, per = m.rpp * cg;
  for (int g = threadIdx.x >> 6; g < gpb; g += GN3_T / 64) {      // one wave per group at a time, as k_gn2_stats
    double a = 0.0, q = 0.0;
    for (int i = lane; i < per; i += 64) {
      const int r = i / cg, c = g * cg + (i - r * cg);
      a += shd[((size_t)r * m.Cs + c) * 2]; q += shd[((size_t)r * m.Cs + c) * 2 + 1];
    }
    a = wave_sum_d(a); q = wave_sum_d(q);
    if (lane == 0) {
      const double n = (double)HW * cg, mu = a / n;
      double var = q / n - mu * mu;                 // biased variance, as torch.nn.GroupNorm
      var = var < 0 ? 0 : var;
      st[0][g] = (float)mu; st[1][g] = (float)(1.0 / sqrt(var + (double)eps));
      mean[m.b * G + m.g0 + g] = st[0][g]; rstd[m.b * G + m.g0 + g] = st[1][g];
    }
  }
  __syncthreads();
  if (!m.on) return;
  float mu[4], rs[4], ga[4], be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * m.q + e, gi = c / cg;
    mu[e] = st[0][gi]; rs[e] = st[1][gi]; ga[e] = gamma[m.c0 + c]; be[e] = beta[m.c0 + c];
  }
  __bf16* yb = y + (size_t)m.b * HW * C + m.c0 + 4 * m.q;
  const uint8_t* mb = mask ? mask + (size_t)m.b * HW * C + m.c0 + 4 * m.q : nullptr;
#pragma unroll 8
  for (int p = m.r; p < HW; p += m.rpp) {
    const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)p * ldx);
    float z[4] = {v.x, v.y, v.z, v.w};
    uchar4 mk = make_uchar4(1, 1, 1, 1);
    if (mb) mk = *reinterpret_cast<const uchar4*>(mb + (size_t)p * C);
    const uint8_t mke[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = (z[e] - mu[e]) * rs[e] * ga[e] + be[e];
      if (swish) t = silu(t);
      if (mb) t = mke[e] ? t * drop_scale : 0.f;
      z[e] = t;
    }
    *reinterpret_cast<bf16x4*>(yb + (size_t)p * C) = bf16x4{f2bf(z[0]), f2bf(z[1]), f2bf(z[2]), f2bf(z[3])};
  }
}
// backward of the same decomposition: pass 1 = per-channel sums of dz xhat and dz over the sample (the parameter-gradient partials pg / pb of
// this sample, complete here) -> the group means k1, k2; pass 2 = dx (+)= rstd (dz gamma - k1 - xhat k2), every output form of k_gn2_bwd_apply
// (fp32 with accumulate / extra, bf16 operand copy, per-sample column sums: colpart keeps the caller's [B][nchunk][C] layout -- chunk 0 gets the
// sample's sum, the other chunks zeros)
template <bool SRC>
__global__ __launch_bounds__(GN3_T) void k_gn3_bwd(const float* __restrict__ dy, const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   int HW, int C, int G, int gpb, int swish, const uint8_t* __restrict__ mask, float drop_scale,
                                                   float* __restrict__ dx, int lddx, int accumulate, float* __restrict__ pg, float* __restrict__ pb,
                                                   const float* __restrict__ extra, int ldextra, __bf16* __restrict__ dx16,
                                                   float* __restrict__ colpart, int nchunk, Gn3Src src) {
  __shared__ float shf[GN3_T * 4 * 2];          // [rpp][Cs][2]; afterwards [rpp][Cs] column sums
  __shared__ float chs[GN3_T * 4 * 2 / (GN3_T / 256)];          // [Cs][2] channel sums, Cs <= 1024
  __shared__ float kk[2 * 64];
  Gn3Map m; m.init(C, G, gpb);
  const int cg = C / G;
  float mu[4], rs[4], ga[4], be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * m.q + e, gi = m.g0 + c / cg;
    mu[e] = 0.f; rs[e] = 0.f; ga[e] = 0.f; be[e] = 0.f;
    if (m.on) { mu[e] = mean[m.b * G + gi]; rs[e] = rstd[m.b * G + gi]; ga[e] = gamma[m.c0 + c]; be[e] = beta[m.c0 + c]; }
  }
  const float* xb = x + (size_t)m.b * HW * ldx + m.c0 + 4 * m.q;
  const float* dyb = dy + (size_t)m.b * HW * C + m.c0 + 4 * m.q;
  const uint8_t* mb = mask ? mask + (size_t)m.b * HW * C + m.c0 + 4 * m.q : nullptr;
  if (m.on) {
    float ca[4] = {0.f, 0.f, 0.f, 0.f}, cb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll SRC ? 2 : 8
    for (int p = m.r; p < HW; p += m.rpp) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)p * ldx);
      float4 dv;
      if constexpr (SRC) {                          // dy is still S slabs (an input-gradient convolution): finished here, stored for pass 2
        dv = gn3_src_quad(src, (int64_t)m.b * HW + p, m.b, m.c0 + 4 * m.q, C);
        *reinterpret_cast<float4*>(const_cast<float*>(dyb) + (size_t)p * C) = dv;
      } else
        dv = *reinterpret_cast<const float4*>(dyb + (size_t)p * C);
      uchar4 mk = make_uchar4(1, 1, 1, 1);
      if (mb) mk = *reinterpret_cast<const uchar4*>(mb + (size_t)p * C);
      const float xv[4] = {v.x, v.y, v.z, v.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
      const uint8_t mke[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[e] - mu[e]) * rs[e];
        float d = dd[e];
        if (mb) d = mke[e] ? d * drop_scale : 0.f;
        if (swish) d *= silu_grad(xh * ga[e] + be[e]);
        ca[e] += d * xh; cb[e] += d;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      shf[((size_t)m.r * m.Cs + 4 * m.q + e) * 2] = ca[e]; shf[((size_t)m.r * m.Cs + 4 * m.q + e) * 2 + 1] = cb[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < m.Cs; c += GN3_T) {
    float a = 0.f, q = 0.f;
    for (int r = 0; r < m.rpp; ++r) { a += shf[((size_t)r * m.Cs + c) * 2]; q += shf[((size_t)r * m.Cs + c) * 2 + 1]; }
    chs[2 * c] = a; chs[2 * c + 1] = q;
    pg[(size_t)m.b * C + m.c0 + c] = a; pb[(size_t)m.b * C + m.c0 + c] = q;
  }
  __syncthreads();
  if (threadIdx.x < gpb) {
    float t1 = 0.f, t2 = 0.f;
    for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; ++c) { const float g1 = gamma[m.c0 + c]; t1 += g1 * chs[2 * c + 1]; t2 += g1 * chs[2 * c]; }
    const float inv = 1.0f / ((float)HW * (float)cg);
    kk[2 * threadIdx.x] = t1 * inv; kk[2 * threadIdx.x + 1] = t2 * inv;
  }
  __syncthreads();
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  if (m.on) {
    float k1[4], k2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int gi = (4 * m.q + e) / cg; k1[e] = kk[2 * gi]; k2[e] = kk[2 * gi + 1]; }
    float* dxb = dx ? dx + (size_t)m.b * HW * lddx + m.c0 + 4 * m.q : nullptr;
    __bf16* d16b = dx16 ? dx16 + (size_t)m.b * HW * C + m.c0 + 4 * m.q : nullptr;
    const float* exb = extra ? extra + (size_t)m.b * HW * ldextra + m.c0 + 4 * m.q : nullptr;
#pragma unroll 4
    for (int p = m.r; p < HW; p += m.rpp) {
      const float4 v = *reinterpret_cast<const float4*>(xb + (size_t)p * ldx);
      const float4 dv = *reinterpret_cast<const float4*>(dyb + (size_t)p * C);
      uchar4 mk = make_uchar4(1, 1, 1, 1);
      if (mb) mk = *reinterpret_cast<const uchar4*>(mb + (size_t)p * C);
      const float xv[4] = {v.x, v.y, v.z, v.w}, dd[4] = {dv.x, dv.y, dv.z, dv.w};
      const uint8_t mke[4] = {mk.x, mk.y, mk.z, mk.w};
      float4* op = dxb ? reinterpret_cast<float4*>(dxb + (size_t)p * lddx) : nullptr;
      float o[4] = {0.f, 0.f, 0.f, 0.f};
      if (accumulate) { const float4 c = *op; o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w; }
      if (exb) {
        const float4 c = *reinterpret_cast<const float4*>(exb + (size_t)p * ldextra);
        o[0] += c.x; o[1] += c.y; o[2] += c.z; o[3] += c.w;
      }
      const bool acc = accumulate || exb;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[e] - mu[e]) * rs[e];
        float d = dd[e];
        if (mb) d = mke[e] ? d * drop_scale : 0.f;
        if (swish) d *= silu_grad(xh * ga[e] + be[e]);
        const float vv = rs[e] * (d * ga[e] - k1[e] - xh * k2[e]);
        o[e] = acc ? o[e] + vv : vv;
      }
      if (op) *op = make_float4(o[0], o[1], o[2], o[3]);
      if (d16b) *reinterpret_cast<bf16x4*>(d16b + (size_t)p * C) = bf16x4{f2bf(o[0]), f2bf(o[1]), f2bf(o[2]), f2bf(o[3])};
      if (colpart) {
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[e] += o[e];
      }
    }
  }
  if (colpart) {
    __syncthreads();                            // every reader of shf's channel partials is done
    if (m.on) {
#pragma unroll
      for (int e = 0; e < 4; ++e) shf[(size_t)m.r * m.Cs + 4 * m.q + e] = cs[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < m.Cs; c += GN3_T) {
      float a = 0.f;
      for (int r = 0; r < m.rpp; ++r) a += shf[(size_t)r * m.Cs + c];
      colpart[((size_t)m.b * nchunk) * C + m.c0 + c] = a;
      for (int k = 1; k < nchunk; ++k) colpart[((size_t)m.b * nchunk + k) * C + m.c0 + c] = 0.f;
    }
  }
}

// ---- softmax over rows of length n (fp32 in, bf16 out), one wave per row; backward dS = scale * P * (dP - sum(P dP))
// rows of `n` stored elements of which the first `nv` are keys (the rest is padding of the context length: probability 0)
__global__ __launch_bounds__(TPB) void k_softmax_fwd(const float* __restrict__ s, int64_t rows, int n, int nv, float scale, __bf16* __restrict__ p) {
  const int64_t row = (int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* sr = s + row * n;
  float mx = -INFINITY;
  for (int i = lane; i < nv; i += 64) mx = fmaxf(mx, sr[i] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int i = lane; i < nv; i += 64) sum += __expf(sr[i] * scale - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int i = lane; i < n; i += 64) p[row * n + i] = i < nv ? f2bf(__expf(sr[i] * scale - mx) * inv) : (__bf16)0.0f;
}
__global__ __launch_bounds__(TPB) void k_softmax_bwd(const __bf16* __restrict__ p, const float* __restrict__ dp, int64_t rows, int n,
                                                     float scale, __bf16* __restrict__ ds) {
  const int64_t row = (int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float dot = 0.f;
  for (int i = lane; i < n; i += 64) dot += bf2f(p[row * n + i]) * dp[row * n + i];
  dot = wave_sum(dot);
  for (int i = lane; i < n; i += 64) ds[row * n + i] = f2bf(scale * bf2f(p[row * n + i]) * (dp[row * n + i] - dot));
}

// ---- LayerNorm(D, eps, affine) on rows, one wave per row (BasicTransformerBlock.norm1..3, SD/ldm/modules/attention.py:223-225)
__global__ __launch_bounds__(TPB) void k_layernorm_fwd(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       int64_t rows, int D, float eps, __bf16* __restrict__ y, float* __restrict__ mean,
                                                       float* __restrict__ rstd) {
  const int64_t row = (int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i];
  const float m = wave_sum(s) / D;
  float q = 0.f;
  for (int i = lane; i < D; i += 64) { const float d = xr[i] - m; q += d * d; }
  const float r = rsqrtf(wave_sum(q) / D + eps);
  for (int i = lane; i < D; i += 64) y[row * D + i] = f2bf((xr[i] - m) * r * gamma[i] + beta[i]);
  if (lane == 0) { mean[row] = m; rstd[row] = r; }
}
// the same with the row in registers as float4 chunks (one 16-byte load per chunk, D <= 256 * NC, D % 4 == 0)
template <int NC>
__global__ __launch_bounds__(TPB) void k_layernorm_fwd_r(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         int64_t rows, int D, float eps, __bf16* __restrict__ y, float* __restrict__ mean,
                                                         float* __restrict__ rstd) {
  const int64_t row = (int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, D4 = D >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + row * D);
  float4 v[NC];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = lane + 64 * j;
    v[j] = c < D4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
  const float m = wave_sum(s) / D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NC; ++j)
    if (lane + 64 * j < D4) {
      const float a = v[j].x - m, b = v[j].y - m, c2 = v[j].z - m, d = v[j].w - m;
      q += (a * a + b * b) + (c2 * c2 + d * d);
    }
  const float r = rsqrtf(wave_sum(q) / D + eps);
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = lane + 64 * j;
    if (c < D4) {
      const float4 g4 = reinterpret_cast<const float4*>(gamma)[c], b4 = reinterpret_cast<const float4*>(beta)[c];
      reinterpret_cast<bf16x4*>(y + row * D)[c] = bf16x4{f2bf((v[j].x - m) * r * g4.x + b4.x), f2bf((v[j].y - m) * r * g4.y + b4.y),
                                                         f2bf((v[j].z - m) * r * g4.z + b4.z), f2bf((v[j].w - m) * r * g4.w + b4.w)};
    }
  }
  if (lane == 0) { mean[row] = m; rstd[row] = r; }
}
// dx (+)= d LayerNorm; pg / pb [nblk][D]: per-workgroup (4 rows) partial sums of dy * xhat and dy, written with plain stores
__global__ __launch_bounds__(TPB) void k_layernorm_bwd(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, int64_t rows, int D,
                                                       float* __restrict__ dx, int accumulate, float* __restrict__ pg, float* __restrict__ pb,
                                                       int rows_per_block, const float* __restrict__ extra) {
  extern __shared__ float sh[];                     // [4 waves][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* ag = sh + (size_t)wave * 2 * D;
  float* ab = ag + D;
  for (int i = lane; i < D; i += 64) { ag[i] = 0.f; ab[i] = 0.f; }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  for (int64_t row = r0 + wave; row < r0 + rows_per_block && row < rows; row += TPB / 64) {
    const float m = mean[row], r = rstd[row];
    const float* xr = x + row * D;
    const float* dr = dy + row * D;
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < D; i += 64) {
      const float xh = (xr[i] - m) * r, d = dr[i] * gamma[i];
      s1 += d; s2 += d * xh;
      ag[i] += dr[i] * xh; ab[i] += dr[i];          // a lane owns its columns: no conflict inside the wave
    }
    s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
    for (int i = lane; i < D; i += 64) {
      const float xh = (xr[i] - m) * r;
      const float v = r * (dr[i] * gamma[i] - s1 - xh * s2);
      float* o = dx + row * D + i;
      if (extra) { const float e = extra[row * D + i]; *o = v + (accumulate ? *o + e : e); }     // (dx + extra) + this layer's gradient
      else *o = accumulate ? *o + v : v;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += TPB) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < TPB / 64; ++w) { a += sh[(size_t)w * 2 * D + i]; b += sh[(size_t)w * 2 * D + D + i]; }
    pg[(size_t)blockIdx.x * D + i] = a; pb[(size_t)blockIdx.x * D + i] = b;
  }
}
// the same with the row and the per-column accumulators in registers, a lane owning float4 column chunks (16-byte loads): x and dy are
// read once, nothing but the final per-wave sums goes through LDS.
template <int NC>      // float4 chunks per lane: D <= 256 * NC, D % 4 == 0
__global__ __launch_bounds__(TPB) void k_layernorm_bwd_r(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd, int64_t rows, int D,
                                                         float* __restrict__ dx, int accumulate, float* __restrict__ pg, float* __restrict__ pb,
                                                         int rows_per_block, const float* __restrict__ extra) {
  extern __shared__ float sh[];                     // [4 waves][2][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, D4 = D >> 2;
  float4 ag[NC], ab[NC], gm[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    ag[j] = make_float4(0.f, 0.f, 0.f, 0.f); ab[j] = ag[j];
    const int c = lane + 64 * j;
    gm[j] = c < D4 ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  for (int64_t row = r0 + wave; row < r0 + rows_per_block && row < rows; row += TPB / 64) {
    const float m = mean[row], r = rstd[row];
    const float4* xr = reinterpret_cast<const float4*>(x + row * D);
    const float4* dr = reinterpret_cast<const float4*>(dy + row * D);
    float4 xh[NC], dv[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int c = lane + 64 * j;
      const bool in = c < D4;
      const float4 xv = in ? xr[c] : make_float4(m, m, m, m);
      dv[j] = in ? dr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      xh[j] = make_float4((xv.x - m) * r, (xv.y - m) * r, (xv.z - m) * r, (xv.w - m) * r);
      // the order of k_layernorm_bwd within a lane differs (columns grouped by fours): sums are fixed-order, results reproducible
      const float d0 = dv[j].x * gm[j].x, d1 = dv[j].y * gm[j].y, d2 = dv[j].z * gm[j].z, d3 = dv[j].w * gm[j].w;
      s1 += (d0 + d1) + (d2 + d3);
      s2 += (d0 * xh[j].x + d1 * xh[j].y) + (d2 * xh[j].z + d3 * xh[j].w);
      ag[j].x += dv[j].x * xh[j].x; ag[j].y += dv[j].y * xh[j].y; ag[j].z += dv[j].z * xh[j].z; ag[j].w += dv[j].w * xh[j].w;
      ab[j].x += dv[j].x; ab[j].y += dv[j].y; ab[j].z += dv[j].z; ab[j].w += dv[j].w;
    }
    s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int c = lane + 64 * j;
      if (c < D4) {
        float4 v = make_float4(r * (dv[j].x * gm[j].x - s1 - xh[j].x * s2), r * (dv[j].y * gm[j].y - s1 - xh[j].y * s2),
                               r * (dv[j].z * gm[j].z - s1 - xh[j].z * s2), r * (dv[j].w * gm[j].w - s1 - xh[j].w * s2));
        float4* o = reinterpret_cast<float4*>(dx + row * D) + c;
        if (extra) {                 // (dx + extra) + this layer's gradient: the bits of a separate dx (+)= extra pass before this one
          float4 e = reinterpret_cast<const float4*>(extra + row * D)[c];
          if (accumulate) { const float4 p = *o; e.x += p.x; e.y += p.y; e.z += p.z; e.w += p.w; }
          v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
        } else if (accumulate) { const float4 p = *o; v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
        *o = v;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    const int c = lane + 64 * j;
    if (c < D4) {
      reinterpret_cast<float4*>(sh + (size_t)wave * 2 * D)[c] = ag[j];
      reinterpret_cast<float4*>(sh + (size_t)wave * 2 * D + D)[c] = ab[j];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += TPB) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < TPB / 64; ++w) { a += sh[(size_t)w * 2 * D + i]; b += sh[(size_t)w * 2 * D + D + i]; }
    pg[(size_t)blockIdx.x * D + i] = a; pb[(size_t)blockIdx.x * D + i] = b;
  }
}
// GEGLU (attention.py:37-45): h [rows][2F] = value || gate -> out = bf16(value * gelu(gate)), exact (erf) GELU as F.gelu
__global__ __launch_bounds__(TPB) void k_geglu_fwd(const float* __restrict__ h, int64_t rows, int F, __bf16* __restrict__ out) {
  const int64_t n = rows * F;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int64_t r = i / F; const int c = (int)(i % F);
    const float a = h[r * 2 * F + c], gt = h[r * 2 * F + F + c];
    out[i] = f2bf(a * 0.5f * gt * (1.0f + erff(gt * 0.70710678118654752f)));
  }
}
// dh [rows][2F] bf16: d value = d_out * gelu(gate), d gate = d_out * value * gelu'(gate)
__global__ __launch_bounds__(TPB) void k_geglu_bwd(const float* __restrict__ d_out, const float* __restrict__ h, int64_t rows, int F,
                                                   __bf16* __restrict__ dh) {
  const int64_t n = rows * F;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int64_t r = i / F; const int c = (int)(i % F);
    const float a = h[r * 2 * F + c], gt = h[r * 2 * F + F + c], d = d_out[i];
    const float cdf = 0.5f * (1.0f + erff(gt * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * expf(-0.5f * gt * gt);
    dh[r * 2 * F + c] = f2bf(d * gt * cdf);
    dh[r * 2 * F + F + c] = f2bf(d * a * (cdf + gt * pdf));
  }
}
// the same on float4 column quads, rows walked by the workgroup's y index (no per-element 64-bit division): F % 4 == 0
__global__ __launch_bounds__(TPB) void k_geglu_fwd4(const float* __restrict__ h, int64_t rows, int F, int rows_per_block, __bf16* __restrict__ out) {
  const int c = (blockIdx.x * TPB + threadIdx.x) * 4;
  if (c >= F) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int64_t r = r0; r < r1; ++r) {
    const float4 a = *reinterpret_cast<const float4*>(h + r * 2 * F + c), gt = *reinterpret_cast<const float4*>(h + r * 2 * F + F + c);
    const float av[4] = {a.x, a.y, a.z, a.w}, gv[4] = {gt.x, gt.y, gt.z, gt.w};
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = f2bf(av[e] * 0.5f * gv[e] * (1.0f + erff(gv[e] * 0.70710678118654752f)));
    *reinterpret_cast<bf16x4*>(out + r * F + c) = o;
  }
}
__global__ __launch_bounds__(TPB) void k_geglu_bwd4(const float* __restrict__ d_out, const float* __restrict__ h, int64_t rows, int F, int rows_per_block,
                                                    __bf16* __restrict__ dh) {
  const int c = (blockIdx.x * TPB + threadIdx.x) * 4;
  if (c >= F) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int64_t r = r0; r < r1; ++r) {
    const float4 a = *reinterpret_cast<const float4*>(h + r * 2 * F + c), gt = *reinterpret_cast<const float4*>(h + r * 2 * F + F + c);
    const float4 d4 = *reinterpret_cast<const float4*>(d_out + r * F + c);
    const float av[4] = {a.x, a.y, a.z, a.w}, gv[4] = {gt.x, gt.y, gt.z, gt.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
    bf16x4 o1, o2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float cdf = 0.5f * (1.0f + erff(gv[e] * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * expf(-0.5f * gv[e] * gv[e]);
      o1[e] = f2bf(dv[e] * gv[e] * cdf);
      o2[e] = f2bf(dv[e] * av[e] * (cdf + gv[e] * pdf));
    }
    *reinterpret_cast<bf16x4*>(dh + r * 2 * F + c) = o1;
    *reinterpret_cast<bf16x4*>(dh + r * 2 * F + F + c) = o2;
  }
}

// finish of a batched split-K product: out[y][z][m][n] (bf16, strides sC / sC2 / ldc) = sum_s slab[s][y][z][m][n]
__global__ __launch_bounds__(TPB) void k_bsplit_finish(const float* __restrict__ slabs, int nsplit, int ny, int nz, int M, int N, __bf16* __restrict__ out,
                                                       long sC, long sC2, int ldc) {
  const int64_t per = (int64_t)M * N, n = (int64_t)ny * nz * per;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int64_t b = i / per, r = i - b * per;
    const int y = (int)(b / nz), z = (int)(b - (int64_t)y * nz), m = (int)(r / N), c = (int)(r - (int64_t)m * N);
    float a = 0.f;
    for (int sl = 0; sl < nsplit; ++sl) a += slabs[sl * n + i];
    out[y * sC + z * sC2 + (int64_t)m * ldc + c] = f2bf(a);
  }
}
// split-K finish: out[row][col] = sum_s slab[s][row][col] (+ bias) (+ per-sample vector) (+ resid), bf16 or fp32 -- the epilogue of
// k_bgemm applied after the slabs of a split contraction are added in order
__global__ __launch_bounds__(TPB) void k_split_finish(const float* __restrict__ slabs, int nsl, int64_t slab_stride, int M, int N,
                                                      const float* __restrict__ bias, const float* __restrict__ vec, int ldvec, int T,
                                                      const float* __restrict__ resid, float* __restrict__ of, __bf16* __restrict__ ob, int ldo) {
  const int64_t n = (int64_t)M * N;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int col = (int)(i % N); const int64_t row = i / N;
    float a = 0.f;
    for (int sl = 0; sl < nsl; ++sl) a += slabs[sl * slab_stride + i];
    if (bias) a += bias[col];
    if (vec) a += vec[(row / T) * ldvec + col];
    if (resid) a += resid[row * ldo + col];
    if (of) of[row * ldo + col] = a; else ob[row * ldo + col] = f2bf(a);
  }
}

// the same on float4 column quads with the rows walked per workgroup (no per-element 64-bit division): N % 4 == 0, ldo % 4 == 0
__global__ __launch_bounds__(TPB) void k_split_finish4(const float* __restrict__ slabs, int nsl, int64_t slab_stride, int M, int N,
                                                       const float* __restrict__ bias, const float* __restrict__ vec, int ldvec, int T,
                                                       const float* __restrict__ resid, float* __restrict__ of, __bf16* __restrict__ ob, int ldo,
                                                       int rows_per_block) {
  const int c = (blockIdx.x * TPB + threadIdx.x) * 4;
  if (c >= N) return;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) b4 = *reinterpret_cast<const float4*>(bias + c);
  for (int r = r0; r < r1; ++r) {
    const float* p = slabs + (int64_t)r * N + c;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sl = 0; sl < nsl; ++sl) { const float4 v = *reinterpret_cast<const float4*>(p + sl * slab_stride); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    a.x += b4.x; a.y += b4.y; a.z += b4.z; a.w += b4.w;
    if (vec) { const float4 v = *reinterpret_cast<const float4*>(vec + (int64_t)(r / T) * ldvec + c); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (resid) { const float4 v = *reinterpret_cast<const float4*>(resid + (int64_t)r * ldo + c); a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (of) *reinterpret_cast<float4*>(of + (int64_t)r * ldo + c) = a;
    else *reinterpret_cast<bf16x4*>(ob + (int64_t)r * ldo + c) = bf16x4{f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w)};
  }
}

// ---- small pieces
// out[b][c] = sum_{p < HW} x[(b * HW + p) * ld + c]  (per-sample column sums: gradient of a per-sample broadcast vector)
// one workgroup per (64-column slice, sample): the 4 waves stride over the rows (coalesced 256-B row reads), partial sums meet in LDS
// in a fixed order
// gridDim.z row chunks per sample (chunk z takes rows [HW z / nz, HW (z + 1) / nz)): with nz > 1 `out` is the partial buffer
// [B][nz][C] (ldo = C) that k_reduce_chunks adds per sample in a fixed order
__global__ __launch_bounds__(TPB) void k_sample_colsum(const float* __restrict__ x, int ld, int HW, int C, float* __restrict__ out, int ldo) {
  __shared__ float sh[TPB / 64][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int nz = gridDim.z, z = blockIdx.z;
  const int p0 = (int)((long)HW * z / nz), p1 = (int)((long)HW * (z + 1) / nz);
  float s = 0.f;
  if (c < C) {
    const float* xb = x + (size_t)b * HW * ld + c;
    for (int p = p0 + wave; p < p1; p += TPB / 64) s += xb[(size_t)p * ld];
  }
  sh[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < C) {
    float a = 0.f;
    for (int w = 0; w < TPB / 64; ++w) a += sh[w][lane];
    out[((size_t)b * nz + z) * ldo + c] = a;
  }
}
// out = alpha * a + beta * b   (classifier-free guidance mix (1 + s) cond - s null, models/diffusion.py:340-357)
__global__ __launch_bounds__(TPB) void k_axpby(const float* __restrict__ a, const float* __restrict__ b, float alpha, float beta, int64_t n,
                                               float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) out[i] = alpha * a[i] + beta * b[i];
}
// nearest x2 upsampling backward: dx[b][h][w][c] (+)= sum of the 2x2 block of dy ([B][2H][2W][C])
__global__ __launch_bounds__(TPB) void k_pool2_sum(const float* __restrict__ dy, int B, int H, int W, int C, float* __restrict__ dx, int accumulate) {
  const int64_t n = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int c = (int)(i % C); int64_t p = i / C; const int w = (int)(p % W); p /= W; const int h = (int)(p % H); const int b = (int)(p / H);
    const float* s = dy + (((int64_t)b * 2 * H + 2 * h) * 2 * W + 2 * w) * C + c;
    const float v = s[0] + s[C] + s[(int64_t)2 * W * C] + s[(int64_t)2 * W * C + C];
    dx[i] = accumulate ? dx[i] + v : v;
  }
}
__global__ __launch_bounds__(TPB) void k_cast_rows(const float* __restrict__ x, int ldx, int64_t rows, int C, __bf16* __restrict__ y) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
    y[i] = f2bf(x[(i / C) * ldx + (i % C)]);
}
// the same with a lane per float4 of a row (no per-element division), optionally with the column sums of x (the bias gradient of
// the layer whose output gradient is being cast): grid = (ceil(C / 256), row chunks); the 4 waves of a workgroup take rows
// r0 + wave, + 4, ... and meet in LDS in a fixed order; partials[chunk][C] are added by k_reduce_chunks
template <bool SUM>
__global__ __launch_bounds__(TPB) void k_cast_rows4(const float* __restrict__ x, int ldx, int64_t rows, int C, int rows_per_block,
                                                    __bf16* __restrict__ y, float* __restrict__ partials) {
  __shared__ float sh[3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + lane) * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < C)
    for (int64_t r = r0 + wave; r < r1; r += TPB / 64) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + col);
      *reinterpret_cast<bf16x4*>(y + r * C + col) = bf16x4{f2bf(v.x), f2bf(v.y), f2bf(v.z), f2bf(v.w)};
      if (SUM) { a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w; }
    }
  if (!SUM) return;
  if (wave > 0) { sh[wave - 1][0][lane] = a0; sh[wave - 1][1][lane] = a1; sh[wave - 1][2][lane] = a2; sh[wave - 1][3][lane] = a3; }
  __syncthreads();
  if (wave == 0 && col < C) {
    float* o = partials + (size_t)blockIdx.y * C + col;
    o[0] = ((a0 + sh[0][0][lane]) + sh[1][0][lane]) + sh[2][0][lane];
    o[1] = ((a1 + sh[0][1][lane]) + sh[1][1][lane]) + sh[2][1][lane];
    o[2] = ((a2 + sh[0][2][lane]) + sh[1][2][lane]) + sh[2][2][lane];
    o[3] = ((a3 + sh[0][3][lane]) + sh[1][3][lane]) + sh[2][3][lane];
  }
}
__global__ __launch_bounds__(TPB) void k_copy_cols4(const float* __restrict__ x, int ldx, int64_t rows, int C, float* __restrict__ y, int ldy,
                                                    int accumulate, int rows_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + lane) * 4;
  if (col >= C) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int64_t r = r0 + wave; r < r1; r += TPB / 64) {
    float4 v = *reinterpret_cast<const float4*>(x + r * ldx + col);
    float4* o = reinterpret_cast<float4*>(y + r * ldy + col);
    if (accumulate) { const float4 c = *o; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
    *o = v;
  }
}
// Two column copies in ONE launch (blockIdx.z = piece): the channel concatenation torch.cat([h, skip], dim=1) of the up path (two sources into
// the column halves of one destination) and its backward (the column halves of one source into two gradient buffers, each with its own
// accumulate flag) were two launches of k_copy_cols4 each -- 5 us apiece on a chain that is bound by its launch count.
struct CopyPiece { const float* x; int ldx; float* y; int ldy; int C; int accumulate; };
__global__ __launch_bounds__(TPB) void k_copy_cols4x2(CopyPiece p0, CopyPiece p1, int64_t rows, int rows_per_block) {
  const CopyPiece p = blockIdx.z ? p1 : p0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = (blockIdx.x * 64 + lane) * 4;
  if (col >= p.C) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int64_t r = r0 + wave; r < r1; r += TPB / 64) {
    float4 v = *reinterpret_cast<const float4*>(p.x + r * p.ldx + col);
    float4* o = reinterpret_cast<float4*>(p.y + r * p.ldy + col);
    if (p.accumulate) { const float4 c = *o; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
    *o = v;
  }
}
// Bernoulli keep mask of nn.Dropout(p) in one launch (torch.rand >= p -> uint8 takes three): element i of mask (seed, call, salt) is
// kept when 16 bits of splitmix64(seed, counter, salt, i / 4) reach p * 65536.  `counter` lives on the device and is advanced by the
// caller once per pass, so a captured graph draws fresh masks at every replay.
__global__ __launch_bounds__(TPB) void k_dropout_mask(uint64_t seed, const int64_t* __restrict__ counter, int64_t salt, int64_t n, unsigned thresh16,
                                                      uint8_t* __restrict__ out) {
  const uint64_t base = seed ^ ((uint64_t)counter[0] * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)salt * 0xD1B54A32D192ED03ull);
  const int64_t nq = (n + 3) >> 2;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nq; i += (int64_t)gridDim.x * TPB) {
    uint64_t z = base + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    uchar4 m;
    m.x = (unsigned)(z & 0xffff) >= thresh16; m.y = (unsigned)((z >> 16) & 0xffff) >= thresh16;
    m.z = (unsigned)((z >> 32) & 0xffff) >= thresh16; m.w = (unsigned)(z >> 48) >= thresh16;
    if (4 * i + 3 < n) *reinterpret_cast<uchar4*>(out + 4 * i) = m;
    else { const uint8_t e[4] = {m.x, m.y, m.z, m.w}; for (int k = 0; 4 * i + k < n; ++k) out[4 * i + k] = e[k]; }
  }
}
// Every dropout mask of a pass in ONE launch: item j = mask (salt, n elements) at byte offset `off` of one buffer (int64 triples [salt, n, off],
// off % 4 == 0); blockIdx.y = item.  Bit for bit the masks k_dropout_mask draws for the same (seed, counter, salt): a U-Net pass asked for one
// launch per ResnetBlock (44 launches of 5 us per DDPM SFR-on step).
__global__ __launch_bounds__(TPB) void k_dropout_mask_batch(uint64_t seed, const int64_t* __restrict__ counter, const int64_t* __restrict__ items,
                                                            unsigned thresh16, uint8_t* __restrict__ base_out) {
  const int64_t salt = items[3 * blockIdx.y], n = items[3 * blockIdx.y + 1];
  uint8_t* const out = base_out + items[3 * blockIdx.y + 2];
  const uint64_t base = seed ^ ((uint64_t)counter[0] * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)salt * 0xD1B54A32D192ED03ull);
  const int64_t nq = (n + 3) >> 2;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nq; i += (int64_t)gridDim.x * TPB) {
    uint64_t z = base + (uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    uchar4 m;
    m.x = (unsigned)(z & 0xffff) >= thresh16; m.y = (unsigned)((z >> 16) & 0xffff) >= thresh16;
    m.z = (unsigned)((z >> 32) & 0xffff) >= thresh16; m.w = (unsigned)(z >> 48) >= thresh16;
    if (4 * i + 3 < n) *reinterpret_cast<uchar4*>(out + 4 * i) = m;
    else { const uint8_t e[4] = {m.x, m.y, m.z, m.w}; for (int k = 0; 4 * i + k < n; ++k) out[4 * i + k] = e[k]; }
  }
}
// y[rows][ld_y] column slice <- x[rows][C] (channel concatenation) and back (+=)
__global__ __launch_bounds__(TPB) void k_copy_cols(const float* __restrict__ x, int ldx, int64_t rows, int C, float* __restrict__ y, int ldy,
                                                   int accumulate) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int64_t r = i / C; const int c = (int)(i % C);
    float* o = y + r * ldy + c;
    const float v = x[r * ldx + c];
    *o = accumulate ? *o + v : v;
  }
}
// DDPM get_timestep_embedding (models/diffusion.py:17-35): sin || cos, freq_j = exp(-ln(1e4) * j / (half - 1)); t is float
__global__ __launch_bounds__(TPB) void k_ddpm_temb(const float* __restrict__ t, int n, int dim, __bf16* __restrict__ out) {
  const int half = dim / 2;
  const float e = logf(10000.0f) / (float)(half - 1);
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n * dim; i += gridDim.x * TPB) {
    const int b = i / dim, j = i % dim;
    float v = 0.f;
    if (j < 2 * half) {
      const int jj = j < half ? j : j - half;
      const float a = t[b] * expf(-e * (float)jj);
      v = j < half ? sinf(a) : cosf(a);
    }
    out[i] = f2bf(v);
  }
}
// cemb_in[b] = keep[b] ? table[c[b]] : null_emb   (models/diffusion.py:370-376); labels outside the table read the null row
__global__ __launch_bounds__(TPB) void k_class_embed(const float* __restrict__ table, const float* __restrict__ null_emb,
                                                     const int64_t* __restrict__ c, const uint8_t* __restrict__ keep, int n_classes,
                                                     int n, int D, __bf16* __restrict__ out) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n * D; i += gridDim.x * TPB) {
    const int b = i / D, j = i % D;
    const int64_t lab = c[b];
    const bool use = (!keep || keep[b]) && lab >= 0 && lab < n_classes;
    out[i] = f2bf(use ? table[lab * D + j] : null_emb[j]);
  }
}
// d_table[c[b]] += d[b] (kept samples), d_null += d[b] (dropped): serial over the batch per column (deterministic)
__global__ __launch_bounds__(TPB) void k_class_embed_bwd(const float* __restrict__ d, const int64_t* __restrict__ c,
                                                         const uint8_t* __restrict__ keep, int n_classes, int n, int D,
                                                         float* __restrict__ d_table, float* __restrict__ d_null) {
  const int j = blockIdx.x * TPB + threadIdx.x;
  if (j >= D) return;
  float nul = 0.f;
  for (int b = 0; b < n; ++b) {
    const int64_t lab = c[b];
    const bool use = (!keep || keep[b]) && lab >= 0 && lab < n_classes;
    const float g = d[(size_t)b * D + j];
    if (use) d_table[lab * D + j] += g; else nul += g;
  }
  d_null[j] = nul;
}

// splits of the contraction for a weight-gradient GEMM: enough workgroups for the chip (about two rounds of 256 CUs), at least
// 512 contraction rows per split, at most `cap`
inline int plan_splits(int M, int N, int K, int cap) {
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int s = 512 / (tiles > 0 ? tiles : 1);
  if (s > K / 512) s = K / 512;
  if (s > cap) s = cap;
  return s < 1 ? 1 : s;
}

template <bool A_TR, bool B_TR, int EPI, int CONV>
int launch_bgemm(const BGemmArgs& g, int nbatch, hipStream_t s, int ninner = 1) {
  const size_t lds = 4 * TILE_ELEMS * sizeof(__bf16);
  const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
  hipLaunchKernelGGL((k_bgemm<A_TR, B_TR, EPI, CONV>), dim3(ntm * ntn * (g.ksplit > 1 ? g.ksplit : 1), nbatch, ninner), dim3(NT), lds, s, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFRON_OK : (int)e;
}

// debug build only (tools/ A-B runs): SFRON_NO_CGEMM=1 keeps every product on the generic k_bgemm tile; bit i of SFRON_NO_CGEMM
// disables: 1 direct tile, 2 Linear input gradient, 4 plain weight gradient, 8 convolution weight gradient
inline int cgemm_off() {
#ifdef SFRON_DEBUG_KNOBS
  static const int v = [] { const char* e = getenv("SFRON_NO_CGEMM"); return e ? atoi(e) : 0; }();
  return v;
#else
  return 0;
#endif
}
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per function AND per device
inline bool need_attr(std::atomic<uint64_t>& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  return (mask.fetch_or(bit) & bit) == 0;
}
template <int NT_, int EPI, bool CONV, int NL>
int launch_cgemm_nl(const BGemmArgs& g, unsigned a_bytes, unsigned b_bytes, int nsplit, hipStream_t s) {
  using CT = CTile<NT_>;
  static std::atomic<uint64_t> done{0};        // per instantiation, one bit per device
  if (need_attr(done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cgemm<NT_, EPI, CONV, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT::LDS) !=
        hipSuccess)
      return (int)hipGetLastError();
  }
  const int ntm = g.M / CT::FBM, ntn = (g.N + CT::FBN - 1) / CT::FBN;
  hipLaunchKernelGGL((k_cgemm<NT_, EPI, CONV, NL>), dim3(ntm * ntn, nsplit), dim3(512 + 64 * NL), CT::LDS, s, g, a_bytes, b_bytes);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFRON_OK : (int)e;
}
template <int NT_, int EPI, bool CONV>
int launch_cgemm_t(const BGemmArgs& g, unsigned a_bytes, unsigned b_bytes, int nsplit, hipStream_t s) {
  // a short contraction is all prologue: four loaders take twice as long to put the first two tiles in flight as eight waves do
  const int nk = (g.kchunk > 0 ? g.kchunk : g.K) / BK;
  return (g_conv_loader_waves & 1) && nk >= 8 ? launch_cgemm_nl<NT_, EPI, CONV, 4>(g, a_bytes, b_bytes, nsplit, s)
                                              : launch_cgemm_nl<NT_, EPI, CONV, 0>(g, a_bytes, b_bytes, nsplit, s);
}
// the pipelined tile when the product qualifies (-1: it does not, the caller launches k_bgemm)
int try_cgemm(const BGemmArgs& g, bool conv, size_t a_rows, int nsplit, hipStream_t s) {
  if (cgemm_off() & 1) return -1;
  if (g.M % 256 || g.K % BK || g.N % 4 || (g.kchunk > 0 && g.kchunk % BK)) return -1;
  if (conv && (g.cg.C % BK || (g.cg.taps != 9 && g.cg.taps != 1))) return -1;
  if ((((uintptr_t)g.A | (uintptr_t)g.B) & 15) || g.lda % 8 || g.ldb % 8) return -1;
  const size_t a_bytes = conv ? a_rows * g.lda * 2 : ((size_t)(g.M - 1) * g.lda + g.K) * 2;
  const size_t b_bytes = ((size_t)(g.N - 1) * g.ldb + g.K) * 2;
  if (a_bytes >= 0x7ffffff0ull || b_bytes >= 0x7ffffff0ull || (size_t)(g.N + 160) * g.ldb * 2 >= 0x7ffffff0ull) return -1;
  const bool bf = g.Cb != nullptr;
  const bool wide = g.N % 160 == 0;
#define SFRON_CG(NTV, CV) (bf ? launch_cgemm_t<NTV, EPI_BF16, CV>(g, (unsigned)a_bytes, (unsigned)b_bytes, nsplit, s) \
                              : launch_cgemm_t<NTV, EPI_RES, CV>(g, (unsigned)a_bytes, (unsigned)b_bytes, nsplit, s))
  if (conv) return wide ? SFRON_CG(10, true) : SFRON_CG(8, true);
  return wide ? SFRON_CG(10, false) : SFRON_CG(8, false);
#undef SFRON_CG
}

template <bool P_TR, bool CONVP, bool SWAP, int EPI, int NL>
int launch_cgemm_tt_nl(const BGemmArgs& g, size_t p_bytes, size_t q_bytes, unsigned mw, unsigned mh, int nsplit, hipStream_t s) {
  static std::atomic<uint64_t> done{0};
  if (need_attr(done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cgemm_t<P_TR, CONVP, SWAP, EPI, NL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)TTile::LDS) != hipSuccess)
      return (int)hipGetLastError();
  }
  const int np = SWAP ? g.N : g.M, nq = SWAP ? g.M : g.N;
  const int tiles = ((np + TTile::FBP - 1) / TTile::FBP) * ((nq + TTile::FBQ - 1) / TTile::FBQ);
  hipLaunchKernelGGL((k_cgemm_t<P_TR, CONVP, SWAP, EPI, NL>), dim3(tiles, nsplit), dim3(512 + 64 * NL), TTile::LDS, s, g, (unsigned)p_bytes, (unsigned)q_bytes, mw, mh);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFRON_OK : (int)e;
}
template <bool P_TR, bool CONVP, bool SWAP, int EPI>
int launch_cgemm_tt(const BGemmArgs& g, size_t p_bytes, size_t q_bytes, unsigned mw, unsigned mh, int nsplit, hipStream_t s) {
  const int nk = (g.kchunk > 0 ? g.kchunk : g.K) / BK;
  return (g_conv_loader_waves & 2) && nk >= 8 ? launch_cgemm_tt_nl<P_TR, CONVP, SWAP, EPI, 4>(g, p_bytes, q_bytes, mw, mh, nsplit, s)
                                              : launch_cgemm_tt_nl<P_TR, CONVP, SWAP, EPI, 0>(g, p_bytes, q_bytes, mw, mh, nsplit, s);
}
inline bool fits31(size_t b) { return b < 0x7ffffff0ull; }
// Linear input gradient (A direct [M][K], B read transposed [K][N]) on the pipelined tile; -1 = not eligible
int try_cgemm_dt(const BGemmArgs& g, hipStream_t s) {
  if (cgemm_off() & 2) return -1;
  if (g.M % 256 || g.K % BK || g.N % 8 || g.lda % 8 || g.ldb % 8 || (((uintptr_t)g.A | (uintptr_t)g.B) & 15)) return -1;
  const size_t pb = ((size_t)(g.M - 1) * g.lda + g.K) * 2, qb = ((size_t)(g.K - 1) * g.ldb + g.N) * 2;
  if (!fits31(pb) || !fits31(qb) || !fits31((size_t)g.K * g.ldb * 2 + 4096)) return -1;
  return g.Cb ? launch_cgemm_tt<false, false, false, EPI_BF16>(g, pb, qb, 0, 0, 1, s) : launch_cgemm_tt<false, false, false, EPI_RES>(g, pb, qb, 0, 0, 1, s);
}
// plain weight gradient (both operands read transposed, contraction over the rows), fp32 result; -1 = not eligible
bool tt_ok(const BGemmArgs& g) {
  if (cgemm_off() & 4) return false;
  if (g.K % BK || g.M % 8 || g.N % 8 || g.lda % 8 || g.ldb % 8 || (((uintptr_t)g.A | (uintptr_t)g.B) & 15) || !g.Cf) return false;
  if (g.kchunk > 0 && g.kchunk % BK) return false;
  return fits31(((size_t)g.K * g.lda + 4096) * 2) && fits31(((size_t)g.K * g.ldb + 4096) * 2);
}
int try_cgemm_tt(const BGemmArgs& g, int nsplit, hipStream_t s) {
  if (!tt_ok(g)) return -1;
  const size_t pb = ((size_t)(g.K - 1) * g.lda + g.M) * 2, qb = ((size_t)(g.K - 1) * g.ldb + g.N) * 2;
  return launch_cgemm_tt<true, false, false, EPI_RES>(g, pb, qb, 0, 0, nsplit, s);
}
// workgroups of the weight-gradient tile (256 on the long axis, 128 on the other)
inline int tt_tiles(int np, int nq) { return ((np + TTile::FBP - 1) / TTile::FBP) * ((nq + TTile::FBQ - 1) / TTile::FBQ); }
inline int tt_splits(int tiles, int K, int cap, int64_t out_elems) {
  // time model (us): rounds of workgroups x K-tiles each walks (1.4 us per 64 contraction rows) + the fp32 slabs every split
  // writes and the scatter / reduction reads back (8 B per output element at ~4 TB/s)
  int best = 1;
  double best_t = 1e30;
  const int lim = K / 512 < cap ? K / 512 : cap;
  for (int sp = 1; sp <= (lim < 1 ? 1 : lim); ++sp) {
    const int rounds = (tiles * sp + 255) / 256;
    const double t = rounds * ((double)K / sp) * (1.4 / 64.0) + (sp > 1 ? sp : 0) * (double)out_elems * 2e-6;
    if (t < best_t) { best_t = t; best = sp; }
  }
  return best;
}

}  // namespace

extern "C" {

int sfron_bgemm_bf16(const sfron_bgemm_desc* d, void* stream) {
  SFRON_CHECK_ARG(d && d->A && d->B && d->M > 0 && d->N > 0 && d->K > 0 && d->batch >= 1);
  SFRON_CHECK_ARG(d->N % 4 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0 && (((uintptr_t)d->A | (uintptr_t)d->B) & 15) == 0);
  SFRON_CHECK_ARG(d->stride_a % 8 == 0 && d->stride_b % 8 == 0 && d->stride_c % 4 == 0);
  if (!d->a_transposed || !d->b_transposed) SFRON_CHECK_ARG(d->K % 8 == 0);
  if (d->a_transposed) SFRON_CHECK_ARG(d->M % 8 == 0);
  if (d->b_transposed) SFRON_CHECK_ARG(d->N % 8 == 0);
  BGemmArgs g{};
  g.A = (const __bf16*)d->A; g.B = (const __bf16*)d->B;
  g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb;
  g.sA = d->stride_a; g.sB = d->stride_b; g.sC = d->stride_c;
  g.sA2 = d->stride_a2; g.sB2 = d->stride_b2; g.sC2 = d->stride_c2;
  const int ni = d->batch2 > 0 ? d->batch2 : 1;
  SFRON_CHECK_ARG(d->stride_a2 % 8 == 0 && d->stride_b2 % 8 == 0 && d->stride_c2 % 4 == 0);
  g.Cb = (__bf16*)d->c_bf16; g.ldcb = d->ldc; g.Cf = d->c_f32; g.ldcf = d->ldc;
  g.bias = d->bias; g.resid = d->resid; g.vec = d->sample_vec; g.ldvec = d->ld_vec; g.T = d->rows_per_sample > 0 ? d->rows_per_sample : 1;
  g.alpha = d->alpha; g.accumulate = d->accumulate;
  SFRON_CHECK_ARG((g.Cb != nullptr) != (g.Cf != nullptr) && d->ldc % 4 == 0);
  hipStream_t s = (hipStream_t)stream;
  const bool bf = g.Cb != nullptr;
  if (!d->a_transposed && !d->b_transposed) {
    if (d->batch == 1 && ni == 1) { const int rc = try_cgemm(g, false, 0, 1, s); if (rc >= 0) return rc; }
    return bf ? launch_bgemm<false, false, EPI_BF16, CONV_NONE>(g, d->batch, s, ni) : launch_bgemm<false, false, EPI_RES, CONV_NONE>(g, d->batch, s, ni);
  }
  if (!d->a_transposed && d->b_transposed) {
    if (d->batch == 1 && ni == 1) { const int rc = try_cgemm_dt(g, s); if (rc >= 0) return rc; }
    // a few rows against a deep contraction (the embedding projection's input gradient: 8 x 1280 over K = 28 000): split K over the chip
    if (!bf && d->split_ws && d->batch == 1 && ni == 1 && d->ldc == d->N && !d->bias && !d->resid && !d->sample_vec && !d->accumulate &&
        d->K >= 4096) {
      const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
      int sp = 256 / (tiles > 0 ? tiles : 1);
      if (sp > g.K / 512) sp = g.K / 512;
      if (sp > d->split_ws_slabs) sp = d->split_ws_slabs;
      if (sp > 1) {
        g.kchunk = ((g.K + sp - 1) / sp + BK - 1) / BK * BK;
        const int used = (g.K + g.kchunk - 1) / g.kchunk;
        g.Cf = d->split_ws; g.sC = (long)d->M * d->N;
        const int rc = launch_bgemm<false, true, EPI_RES, CONV_NONE>(g, used, s, 1);
        if (rc) return rc;
        return sfron_reduce_chunks(d->split_ws, 1, used, d->M * d->N, d->c_f32, d->M * d->N, 0, stream);
      }
    }
    return bf ? launch_bgemm<false, true, EPI_BF16, CONV_NONE>(g, d->batch, s, ni) : launch_bgemm<false, true, EPI_RES, CONV_NONE>(g, d->batch, s, ni);
  }
  if (d->a_transposed && d->b_transposed) {
    const bool fast = !bf && d->batch == 1 && ni == 1 && tt_ok(g) && d->M >= 64 && d->N >= 64 && d->K >= 256;
    // a plain weight gradient (fp32 [M][N] contiguous, contraction over all rows): split-K through the caller's slab scratch
    if (!bf && d->split_ws && d->batch == 1 && ni == 1 && d->ldc == d->N && !d->bias && !d->resid && !d->sample_vec && !d->accumulate) {
      const int sp = fast ? tt_splits(tt_tiles(d->M, d->N), d->K, d->split_ws_slabs, (int64_t)d->M * d->N) : plan_splits(d->M, d->N, d->K, d->split_ws_slabs);
      if (sp > 1) {
        g.kchunk = ((g.K + sp - 1) / sp + BK - 1) / BK * BK;
        const int used = (g.K + g.kchunk - 1) / g.kchunk;
        g.Cf = d->split_ws; g.sC = (long)d->M * d->N;
        int rc = fast ? try_cgemm_tt(g, used, s) : -1;
        if (rc < 0) rc = launch_bgemm<true, true, EPI_RES, CONV_NONE>(g, used, s, 1);
        if (rc) return rc;
        return sfron_reduce_chunks(d->split_ws, 1, used, d->M * d->N, d->c_f32, d->M * d->N, 0, stream);
      }
    }
    if (fast) { const int rc = try_cgemm_tt(g, 1, s); if (rc >= 0) return rc; }
    // head-batched products with a single output tile each and a contraction over all tokens (dK, dV of the cross-attention): a
    // workgroup per (sample, head) leaves the chip three quarters empty and walks K = 4096 alone -- split the contraction
    if (bf && d->split_ws && (d->batch > 1 || ni > 1) && !d->bias && !d->resid && !d->sample_vec && !d->accumulate && d->alpha == 1.0f && g.K >= 1024) {
      const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * d->batch * ni;
      int sp = 512 / (tiles > 0 ? tiles : 1);
      if (sp > g.K / 256) sp = g.K / 256;
      if (sp > 16) sp = 16;
      const int64_t per = (int64_t)d->batch * ni * d->M * d->N;
      if ((int64_t)sp * per > (int64_t)d->split_ws_slabs * d->M * d->N) sp = (int)((int64_t)d->split_ws_slabs * d->M * d->N / per);
      if (sp > 1) {
        BGemmArgs q = g;
        q.kchunk = ((g.K + sp - 1) / sp + BK - 1) / BK * BK;
        q.ksplit = (g.K + q.kchunk - 1) / q.kchunk;
        q.Cb = nullptr; q.Cf = d->split_ws; q.ldcf = d->N;
        int rc = launch_bgemm<true, true, EPI_RES, CONV_NONE>(q, d->batch, s, ni);
        if (rc) return rc;
        hipLaunchKernelGGL(k_bsplit_finish, dim3(grid_for(per)), dim3(TPB), 0, s, d->split_ws, q.ksplit, d->batch, ni, d->M, d->N, (__bf16*)d->c_bf16,
                           (long)d->stride_c, (long)d->stride_c2, d->ldc);
        SFRON_LAUNCH_STATUS();
        return SFRON_OK;
      }
    }
    return bf ? launch_bgemm<true, true, EPI_BF16, CONV_NONE>(g, d->batch, s, ni) : launch_bgemm<true, true, EPI_RES, CONV_NONE>(g, d->batch, s, ni);
  }
  return SFRON_ERR_UNSUPPORTED;
}

static int conv_geom(const sfron_conv_desc* d, ConvGeom& c, int src_c) {
  SFRON_CHECK_ARG(d->batch > 0 && d->h_src > 0 && d->w_src > 0 && d->h_out > 0 && d->w_out > 0 && (d->taps == 9 || d->taps == 1));
  SFRON_CHECK_ARG(d->stride == 1 || d->stride == 2);
  SFRON_CHECK_ARG(!(d->upsample && d->dilate) && src_c % 8 == 0);
  c.Hs = d->h_src; c.Ws = d->w_src; c.C = src_c; c.Ho = d->h_out; c.Wo = d->w_out;
  c.stride = d->stride; c.pad = d->pad; c.up = d->upsample; c.dil = d->dilate; c.taps = d->taps; c.flip = 0;
  return SFRON_OK;
}

/* forward (and, with re-laid weights + flipped taps, input gradient): out[p][n] = sum_{tap, c} src[src(p, tap)][c] w[n][tap][c] */
int sfron_conv_fwd(const sfron_conv_desc* d, const uint16_t* src, const uint16_t* w, void* stream) {
  SFRON_CHECK_ARG(d && src && w && d->n_out % 4 == 0);
  if (d->split_pending) *d->split_pending = 0;
  BGemmArgs g{};
  int rc = conv_geom(d, g.cg, d->c_src); if (rc) return rc;
  g.A = (const __bf16*)src; g.B = (const __bf16*)w;
  g.M = d->batch * d->h_out * d->w_out; g.N = d->n_out; g.K = d->taps * d->c_src;
  g.lda = d->c_src; g.ldb = g.K;
  g.Cb = (__bf16*)d->out_bf16; g.Cf = d->out_f32; g.ldcb = g.ldcf = d->ld_out;
  SFRON_CHECK_ARG((g.Cb != nullptr) != (g.Cf != nullptr) && d->ld_out % 4 == 0 && d->ld_out >= d->n_out);
  g.bias = d->bias; g.resid = d->resid; g.vec = d->sample_vec; g.ldvec = d->ld_vec; g.T = d->h_out * d->w_out;
  g.alpha = 1.0f; g.accumulate = d->accumulate;
  // few output pixels, deep contraction (the 16x16 / 8x8 levels of the LDM UNet: 512 x 1280 outputs over K = 11520..23040): split the
  // contraction over the chip into fp32 slabs, then one pass adds them and applies the epilogue
  const size_t a_rows = (size_t)d->batch * d->h_src * d->w_src;
  // workgroups the launch would have: 256 x 160 | 128 tiles when the pipelined kernel takes it, 128 x 128 otherwise
  const bool pipelined = g.M % 256 == 0 && g.K % BK == 0 && d->c_src % BK == 0;
  const int tiles = pipelined ? (g.M / 256) * ((g.N + (g.N % 160 == 0 ? 160 : 128) - 1) / (g.N % 160 == 0 ? 160 : 128))
                              : ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  if (d->split_ws && !d->accumulate && tiles < (pipelined ? 224 : 128) && g.K >= 2048) {
    int sp = 384 / tiles;
    if (sp > g.K / 512) sp = g.K / 512;
    if (pipelined) {
      // the time model of tt_splits: rounds of workgroups x K-tiles each walks + the slabs the finish kernel reads back
      int lim = g.K / 1024;
      if (lim > d->split_ws_slabs) lim = d->split_ws_slabs;
      double best_t = 1e30;
      sp = 1;
      for (int c = 1; c <= (lim < 1 ? 1 : lim); ++c) {
        const int rounds = (tiles * c + 255) / 256;
        const double t = rounds * ((double)g.K / c) * (1.3 / 64.0) + (c > 1 ? c + 1 : 0) * (double)g.M * g.N * 1e-6;
        if (t < best_t) { best_t = t; sp = c; }
      }
    }
    if (sp > d->split_ws_slabs) sp = d->split_ws_slabs;
    if (sp > 1) {
      BGemmArgs q = g;
      q.kchunk = ((g.K + sp - 1) / sp + BK - 1) / BK * BK;
      const int used = (g.K + q.kchunk - 1) / q.kchunk;
      q.Cb = nullptr; q.Cf = d->split_ws; q.ldcf = g.N; q.sC = (long)g.M * g.N;
      q.bias = nullptr; q.resid = nullptr; q.vec = nullptr;
      rc = try_cgemm(q, true, a_rows, used, (hipStream_t)stream);
      if (rc < 0) rc = launch_bgemm<false, false, EPI_RES, CONV_A>(q, used, (hipStream_t)stream);
      if (rc) return rc;
      const bool al = (((uintptr_t)d->split_ws | (uintptr_t)g.bias | (uintptr_t)g.vec | (uintptr_t)g.resid | (uintptr_t)g.Cf) & 15) == 0 &&
                      (((uintptr_t)g.Cb) & 7) == 0;
      const bool quads = g.N % 4 == 0 && d->ld_out % 4 == 0 && (!g.vec || g.ldvec % 4 == 0) && al;
      if (quads && d->split_pending && d->ld_out == g.N) {   // the caller's GroupNorm finishes the sum itself (sfron_groupnorm_*_src)
        *d->split_pending = used;
        return SFRON_OK;
      }
      if (quads) {
        const int cb = (g.N / 4 + TPB - 1) / TPB;
        int chunks = 2048 / cb; if (chunks > g.M) chunks = g.M; if (chunks < 1) chunks = 1;
        const int rpb = (g.M + chunks - 1) / chunks;
        hipLaunchKernelGGL(k_split_finish4, dim3(cb, (g.M + rpb - 1) / rpb), dim3(TPB), 0, (hipStream_t)stream, d->split_ws, used, (int64_t)g.M * g.N, g.M,
                           g.N, g.bias, g.vec, g.ldvec, g.T, g.resid, g.Cf, g.Cb, d->ld_out, rpb);
      } else
      hipLaunchKernelGGL(k_split_finish, dim3(grid_for((int64_t)g.M * g.N)), dim3(TPB), 0, (hipStream_t)stream, d->split_ws, used, (int64_t)g.M * g.N,
                         g.M, g.N, g.bias, g.vec, g.ldvec, g.T, g.resid, g.Cf, g.Cb, d->ld_out);
      SFRON_LAUNCH_STATUS();
      return SFRON_OK;
    }
  }
  rc = try_cgemm(g, true, a_rows, 1, (hipStream_t)stream);
  if (rc >= 0) return rc;
  return g.Cb ? launch_bgemm<false, false, EPI_BF16, CONV_A>(g, 1, (hipStream_t)stream)
              : launch_bgemm<false, false, EPI_RES, CONV_A>(g, 1, (hipStream_t)stream);
}

/* weight gradient: dw[n][tap][c] = sum_p dy[p][n] src[src(p, tap)][c]  (fp32 [n_out][taps * c_src], then k_conv_wgrad_scatter) */
static bool conv_wgrad_pipelined(const sfron_conv_desc* d) {
  if (cgemm_off() & 8) return false;
  const long K = (long)d->batch * d->h_out * d->w_out;
  return K % BK == 0 && K >= 256 && d->c_src % 8 == 0 && d->n_out % 8 == 0 && d->h_out >= 2 && d->w_out >= 2 && d->n_out >= 64 &&
         d->taps * d->c_src >= 64 && K < (1l << 21) && d->w_out <= 2048 && d->h_out <= 2048;
}
static int conv_wgrad_plan(const sfron_conv_desc* d) {
  const int K = d->batch * d->h_out * d->w_out;
  if (conv_wgrad_pipelined(d)) return tt_splits(tt_tiles(d->taps * d->c_src, d->n_out), K, 64, (int64_t)d->taps * d->c_src * d->n_out);
  return plan_splits(d->n_out, d->taps * d->c_src, K, 64);
}
// the number of slabs sfron_conv_wgrad WRITES (the planned split count, less the ones the rounding of the k range leaves without rows):
// the caller sizes dw_gemm for it and hands the same number to sfron_conv_wgrad_scatter -- nothing has to be zeroed
int sfron_conv_wgrad_splits(const sfron_conv_desc* d) {
  if (!d) return 0;
  const int sp = conv_wgrad_plan(d);
  if (sp <= 1) return sp;
  const int K = d->batch * d->h_out * d->w_out;
  const int kchunk = ((K + sp - 1) / sp + BK - 1) / BK * BK;
  return (K + kchunk - 1) / kchunk;
}
int sfron_conv_wgrad(const sfron_conv_desc* d, const uint16_t* dy, int ld_dy, const uint16_t* src, float* dw_gemm, void* stream) {
  SFRON_CHECK_ARG(d && dy && src && dw_gemm && d->n_out % 8 == 0 && ld_dy % 8 == 0);
  BGemmArgs g{};
  int rc = conv_geom(d, g.cg, d->c_src); if (rc) return rc;
  g.A = (const __bf16*)dy; g.B = (const __bf16*)src;
  g.M = d->n_out; g.N = d->taps * d->c_src; g.K = d->batch * d->h_out * d->w_out;
  g.lda = ld_dy; g.ldb = d->c_src;
  g.Cf = dw_gemm; g.ldcf = g.N; g.alpha = 1.0f; g.T = 1;
  // the contraction runs over every pixel of the batch and the output has few 128x128 tiles: split it; slab s of dw_gemm
  // ([splits][n_out][taps * c_src]) receives split s, sfron_conv_wgrad_scatter adds the slabs in order
  const int sp = conv_wgrad_plan(d);
  if (sp > 1) { g.kchunk = ((g.K + sp - 1) / sp + BK - 1) / BK * BK; g.sC = (long)g.M * g.N; }
  const int used = sp > 1 ? (g.K + g.kchunk - 1) / g.kchunk : 1;       // == sfron_conv_wgrad_splits(d)
  rc = -1;
  if (conv_wgrad_pipelined(d) && (ld_dy % 8) == 0 && (((uintptr_t)dy | (uintptr_t)src) & 15) == 0) {
    const size_t pb = (size_t)d->batch * d->h_src * d->w_src * d->c_src * 2, qb = ((size_t)(g.K - 1) * ld_dy + g.M) * 2;
    if (fits31(pb) && fits31(qb + (size_t)ld_dy * 2 * 64)) {
      const unsigned mw = (unsigned)(0x100000000ull / (unsigned)d->w_out) + 1u, mh = (unsigned)(0x100000000ull / (unsigned)d->h_out) + 1u;
      rc = launch_cgemm_tt<true, true, true, EPI_RES>(g, pb, qb, mw, mh, used, (hipStream_t)stream);
    }
  }
  if (rc < 0) rc = launch_bgemm<true, true, EPI_RES, CONV_B>(g, used, (hipStream_t)stream);
  return rc;
}

int sfron_conv_wprep(const float* w_oihw, int c_out, int c_in, int taps, int c_out_p, int c_in_p, uint16_t* w_fwd, uint16_t* w_dgrad,
                     void* stream) {
  SFRON_CHECK_ARG(w_oihw && w_fwd && c_out_p >= c_out && c_in_p >= c_in && (taps == 9 || taps == 1));
  if (taps == 9 && (int64_t)c_in_p * c_out_p >= 256 * 1024) {
    hipLaunchKernelGGL(k_conv_wprep9, dim3((c_in_p + 31) / 32, (c_out_p + 31) / 32), dim3(TPB), 0, (hipStream_t)stream, w_oihw, c_out, c_in, c_out_p,
                       c_in_p, (__bf16*)w_fwd, (__bf16*)w_dgrad);
    SFRON_LAUNCH_STATUS();
    return SFRON_OK;
  }
  const int64_t n = (int64_t)c_out_p * taps * c_in_p + (w_dgrad ? (int64_t)c_in * taps * c_out_p : 0);
  hipLaunchKernelGGL(k_conv_wprep, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, w_oihw, c_out, c_in, taps, c_out_p, c_in_p,
                     (__bf16*)w_fwd, (__bf16*)w_dgrad);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_conv_wprep_tiles(int c_out_p, int c_in_p) { return ((c_out_p + 31) / 32) * ((c_in_p + 31) / 32); }
int sfron_conv_wprep_batch(const sfron_wprep_item* items_dev, int n_items, int n_tiles, void* stream) {
  SFRON_CHECK_ARG(items_dev && n_items > 0 && n_tiles > 0);
  hipLaunchKernelGGL(k_conv_wprep9_batch, dim3(n_tiles), dim3(TPB), 0, (hipStream_t)stream, items_dev, n_items);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_conv_wgrad_scatter(const float* dw_gemm, int c_out, int c_in, int taps, int c_in_p, int n_slabs, int64_t slab_stride,
                             float* dw_oihw, void* stream) {
  SFRON_CHECK_ARG(dw_gemm && dw_oihw && n_slabs >= 1);
  SFRON_CHECK_ARG(taps <= 9);
  hipLaunchKernelGGL(k_conv_wgrad_scatter, dim3((c_in + 63) / 64, c_out), dim3(TPB), 0, (hipStream_t)stream, dw_gemm,
                     c_out, c_in, taps, c_in_p, n_slabs, slab_stride, dw_oihw);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_conv_wgrad_scatter_batch(const sfron_wgrad_scatter_item* items, int n_items, void* stream) {
  SFRON_CHECK_ARG(items && n_items > 0);
  for (int i = 0; i < n_items; ++i)
    SFRON_CHECK_ARG(items[i].dw_gemm && items[i].dw_oihw && items[i].n_slabs >= 1 && items[i].taps >= 1 && items[i].taps <= 9 && items[i].c_out > 0 &&
                    items[i].c_in > 0 && items[i].c_in_p >= items[i].c_in);
  for (int i0 = 0; i0 < n_items; i0 += WS_ITEMS) {
    const int n = n_items - i0 < WS_ITEMS ? n_items - i0 : WS_ITEMS;
    ScatterPack pack;
    long gx = 1;
    for (int i = 0; i < n; ++i) {
      pack.it[i] = items[i0 + i];
      const long need = (long)((items[i0 + i].c_in + 63) / 64) * items[i0 + i].c_out;
      gx = need > gx ? need : gx;
    }
    for (int i = n; i < WS_ITEMS; ++i) pack.it[i] = pack.it[0];
    SFRON_CHECK_ARG(gx <= 0x7fffffffL);
    hipLaunchKernelGGL(k_conv_wgrad_scatter_batch, dim3((unsigned)gx, n), dim3(TPB), 0, (hipStream_t)stream, pack);
    SFRON_LAUNCH_STATUS();
  }
  return SFRON_OK;
}

int sfron_nchw_to_rows_bf16(const float* x, int B, int C, int HW, int c_pad, uint16_t* rows, void* stream) {
  SFRON_CHECK_ARG(x && rows && c_pad >= C);
  hipLaunchKernelGGL(k_nchw_to_rows, dim3(grid_for((int64_t)B * HW * c_pad)), dim3(TPB), 0, (hipStream_t)stream, x, B, C, HW, c_pad, (__bf16*)rows);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_nchw_to_rows_f32(const float* x, int B, int C, int HW, int ld, float* rows, void* stream) {
  SFRON_CHECK_ARG(x && rows && ld >= C);
  hipLaunchKernelGGL(k_nchw_to_rows_f32, dim3(grid_for((int64_t)B * HW * ld)), dim3(TPB), 0, (hipStream_t)stream, x, B, C, HW, ld, rows);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_rows_to_nchw(const float* rows, int ld, int B, int C, int HW, float* x, void* stream) {
  SFRON_CHECK_ARG(x && rows && ld >= C);
  hipLaunchKernelGGL(k_rows_to_nchw, dim3(grid_for((int64_t)B * C * HW)), dim3(TPB), 0, (hipStream_t)stream, rows, ld, B, C, HW, x);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

// debug build only (tools/ A-B runs): SFRON_GN_ONE_LAUNCH=0 in the environment keeps every GroupNorm on the two-phase pair
static bool gn3_off() {
#ifdef SFRON_DEBUG_KNOBS
  static const bool v = [] { const char* e = getenv("SFRON_GN_ONE_LAUNCH"); return e && e[0] == '0'; }();
  return v;
#else
  return false;
#endif
}
/* scratch (bytes) the row-coalesced GroupNorm needs for a [B][HW][C] activation: per-chunk partial sums (forward fp64 per group,
 * backward fp32 per channel) */
int64_t sfron_groupnorm_scratch_bytes(int B, int HW, int C, int groups) {
  if (B <= 0 || HW <= 0 || C <= 0 || groups <= 0) return 0;
  const int64_t n = gn_chunks(B, HW);
  const int64_t f = (int64_t)B * n * groups * 2 * sizeof(double), bw = (int64_t)B * n * C * 2 * sizeof(float);
  return f > bw ? f : bw;
}
static bool gn2_ok(int ldx, int ld2, int C, int groups, const void* scratch) {
  return scratch && C % 4 == 0 && ldx % 4 == 0 && ld2 % 4 == 0 && C <= GN_MAXQ * 4 * GNB && groups <= 64 && ((uintptr_t)scratch & 15) == 0;
}
// groups per workgroup of the one-launch form (k_gn3_*), 0 = the two-phase pair: whole groups whose channels form float4 quads and runs that are
// multiples of half a cache line (a 96-byte run straddles lines: measured 1.7x slower than the pair), at least 192 workgroups, and a slab of at
// most 128 KB per workgroup (its second pass is an L2 hit) -- or, with eight groups in whole-line runs, up to 512 KB (second pass from the
// Infinity Cache, as the pair's)
static int gn3_gpb(int B, int HW, int C, int groups) {
  if (gn3_off()) return 0;
  const int cg = C / groups;
  for (const int gpb : {8, 4, 2, 1}) {
    if (groups % gpb) continue;
    const int Cs = gpb * cg;
    if (Cs % 16 || Cs > 1024) continue;
    if ((int64_t)HW * Cs * 4 > (128 << 10)) continue;
    if ((int64_t)B * (groups / gpb) < 192) continue;
    return gpb;
  }
  if (groups % 8 == 0) {
    const int Cs = 8 * cg;
    if (Cs % 32 == 0 && Cs <= 1024 && (int64_t)HW * Cs * 4 <= (512 << 10) && (int64_t)B * (groups / 8) >= 192) return 8;
  }
  return 0;
}
static size_t gn2_lds(int C, int elem) {
  const int qpr = C / 4, qw = qpr < GNB ? qpr : GNB, rpp = GNB / qw;
  return (size_t)rpp * C * 2 * elem;
}
int sfron_groupnorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, int B, int HW, int C, int groups, float eps,
                        int swish, const uint8_t* drop_mask, float drop_scale, uint16_t* y, float* mean, float* rstd, void* scratch,
                        void* stream) {
  SFRON_CHECK_ARG(x && gamma && beta && y && mean && rstd && groups > 0 && C % groups == 0 && ldx >= C && C / groups <= TPB);
  if (gn2_ok(ldx, C, C, groups, scratch) && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0 && (!drop_mask || ((uintptr_t)drop_mask & 3) == 0)) {
    if (const int gpb = gn3_gpb(B, HW, C, groups)) {
      hipLaunchKernelGGL(k_gn3_fwd<false>, dim3(B * (groups / gpb)), dim3(GN3_T), 0, (hipStream_t)stream, x, ldx, gamma, beta, HW, C, groups, gpb, eps, swish,
                         drop_mask, drop_scale, (__bf16*)y, mean, rstd, Gn3Src{});
      SFRON_LAUNCH_STATUS();
      return SFRON_OK;
    }
    const int nchunk = gn_chunks(B, HW);
    hipLaunchKernelGGL(k_gn2_stats, dim3(B * nchunk), dim3(GNB), gn2_lds(C, sizeof(double)), (hipStream_t)stream, x, ldx, HW, C, groups, nchunk,
                       (double*)scratch);
    SFRON_LAUNCH_STATUS();
    hipLaunchKernelGGL(k_gn2_apply, dim3(B * nchunk), dim3(GNB), 0, (hipStream_t)stream, x, ldx, gamma, beta, HW, C, groups, eps, swish, drop_mask,
                       drop_scale, nchunk, (const double*)scratch, (__bf16*)y, mean, rstd);
    SFRON_LAUNCH_STATUS();
    return SFRON_OK;
  }
  hipLaunchKernelGGL(k_gn_fwd, dim3(B * groups), dim3(TPB), 0, (hipStream_t)stream, x, ldx, gamma, beta, HW, C, groups, eps, swish, drop_mask,
                     drop_scale, (__bf16*)y, mean, rstd);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_groupnorm_bwd(const float* dy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                        const float* rstd, int B, int HW, int C, int groups, int swish, const uint8_t* drop_mask, float drop_scale,
                        float* dx, int lddx, int accumulate, float* part_gamma, float* part_beta, void* scratch, void* stream) {
  return sfron_groupnorm_bwd_res(dy, x, ldx, gamma, beta, mean, rstd, B, HW, C, groups, swish, drop_mask, drop_scale, dx, lddx, accumulate, nullptr, 0,
                                 part_gamma, part_beta, scratch, stream);
}
int sfron_groupnorm_bwd_res(const float* dy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                            const float* rstd, int B, int HW, int C, int groups, int swish, const uint8_t* drop_mask, float drop_scale,
                            float* dx, int lddx, int accumulate, const float* extra, int ld_extra, float* part_gamma, float* part_beta,
                            void* scratch, void* stream) {
  SFRON_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && part_gamma && part_beta && groups > 0 && C % groups == 0);
  const int cg = C / groups;
  SFRON_CHECK_ARG(cg <= TPB);
  SFRON_CHECK_ARG(!extra || (ld_extra >= C && extra != dx));
  const bool fused = gn2_ok(ldx, lddx, C, groups, scratch) && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0 &&
                     (!drop_mask || ((uintptr_t)drop_mask & 3) == 0);
  if (extra && !(fused && ld_extra % 4 == 0 && ((uintptr_t)extra & 15) == 0)) {       // the same sum as two passes
    const int rc = sfron_copy_cols(extra, ld_extra, (int64_t)B * HW, C, dx, lddx, accumulate, stream);
    if (rc) return rc;
    extra = nullptr; accumulate = 1;
  }
  if (fused) {
    if (const int gpb = gn3_gpb(B, HW, C, groups)) {
      hipLaunchKernelGGL(k_gn3_bwd<false>, dim3(B * (groups / gpb)), dim3(GN3_T), 0, (hipStream_t)stream, dy, x, ldx, gamma, beta, mean, rstd, HW, C, groups, gpb,
                         swish, drop_mask, drop_scale, dx, lddx, accumulate, part_gamma, part_beta, extra, ld_extra, (__bf16*)nullptr, (float*)nullptr, 1,
                         Gn3Src{});
      SFRON_LAUNCH_STATUS();
      return SFRON_OK;
    }
    const int nchunk = gn_chunks(B, HW);
    hipLaunchKernelGGL(k_gn2_bwd_stats, dim3(B * nchunk), dim3(GNB), gn2_lds(C, sizeof(float)), (hipStream_t)stream, dy, x, ldx, gamma, beta, mean, rstd,
                       HW, C, groups, swish, drop_mask, drop_scale, nchunk, (float*)scratch);
    SFRON_LAUNCH_STATUS();
    hipLaunchKernelGGL(k_gn2_bwd_apply, dim3(B * nchunk), dim3(GNB), (size_t)(3 * C + 2 * groups) * sizeof(float), (hipStream_t)stream, dy, x, ldx, gamma,
                       beta, mean, rstd, HW, C, groups, swish, drop_mask, drop_scale, nchunk, (const float*)scratch, dx, lddx, accumulate,
                       part_gamma, part_beta, extra, ld_extra, (__bf16*)nullptr, (float*)nullptr);
    SFRON_LAUNCH_STATUS();
    return SFRON_OK;
  }
  const size_t lds = (2 * (TPB / 64) + 2 * cg + 2 * TPB) * sizeof(float);
  hipLaunchKernelGGL(k_gn_bwd, dim3(B * groups), dim3(TPB), lds, (hipStream_t)stream, dy, x, ldx, gamma, beta, mean, rstd, HW, C, groups, swish,
                     drop_mask, drop_scale, dx, lddx, accumulate, part_gamma, part_beta);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_groupnorm_chunks(int B, int HW) { return (B > 0 && HW > 0) ? gn_chunks(B, HW) : 0; }
int sfron_groupnorm_bwd_cast_ok(int ldx, int C, int groups) {
  return C > 0 && groups > 0 && C % groups == 0 && C % 4 == 0 && ldx % 4 == 0 && C <= 2048 && C <= GN_MAXQ * 4 * GNB && groups <= 64;
}
int sfron_groupnorm_bwd_cast(const float* dy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                             const float* rstd, int B, int HW, int C, int groups, int swish, const uint8_t* drop_mask, float drop_scale,
                             uint16_t* dx_bf16, float* col_partials, float* part_gamma, float* part_beta, void* scratch, void* stream) {
  SFRON_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx_bf16 && col_partials && part_gamma && part_beta && scratch);
  SFRON_CHECK_ARG(sfron_groupnorm_bwd_cast_ok(ldx, C, groups) && B > 0 && HW > 0);
  SFRON_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)scratch) & 15) == 0 && ((uintptr_t)dx_bf16 & 7) == 0 &&
                  (!drop_mask || ((uintptr_t)drop_mask & 3) == 0));
  const int nchunk = gn_chunks(B, HW);
  if (const int gpb = gn3_gpb(B, HW, C, groups)) {
    hipLaunchKernelGGL(k_gn3_bwd<false>, dim3(B * (groups / gpb)), dim3(GN3_T), 0, (hipStream_t)stream, dy, x, ldx, gamma, beta, mean, rstd, HW, C, groups, gpb,
                       swish, drop_mask, drop_scale, (float*)nullptr, C, 0, part_gamma, part_beta, (const float*)nullptr, 0, (__bf16*)dx_bf16,
                       col_partials, nchunk, Gn3Src{});
    SFRON_LAUNCH_STATUS();
    return SFRON_OK;
  }
  hipLaunchKernelGGL(k_gn2_bwd_stats, dim3(B * nchunk), dim3(GNB), gn2_lds(C, sizeof(float)), (hipStream_t)stream, dy, x, ldx, gamma, beta, mean, rstd,
                     HW, C, groups, swish, drop_mask, drop_scale, nchunk, (float*)scratch);
  SFRON_LAUNCH_STATUS();
  const size_t lds = (size_t)(3 * C + 2 * groups + 4 * GNB) * sizeof(float);          // + [rows per pass][C] for the column sums
  hipLaunchKernelGGL(k_gn2_bwd_apply, dim3(B * nchunk), dim3(GNB), lds, (hipStream_t)stream, dy, x, ldx, gamma, beta, mean, rstd, HW, C, groups, swish,
                     drop_mask, drop_scale, nchunk, (const float*)scratch, (float*)nullptr, C, 0, part_gamma, part_beta, (const float*)nullptr, 0,
                     (__bf16*)dx_bf16, col_partials);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

// ---- the GroupNorm input as an unfinished split-K result (sfron_split_src)
static bool split_src_ok(const sfron_split_src* s, int C) {
  return s && s->slabs && s->n_slabs >= 1 && s->slab_stride > 0 && C % 4 == 0 && (((uintptr_t)s->slabs | (uintptr_t)s->bias | (uintptr_t)s->sample_vec |
         (uintptr_t)s->resid) & 15) == 0 && s->slab_stride % 4 == 0 && (!s->sample_vec || s->ld_vec % 4 == 0) && (!s->resid || s->ld_resid % 4 == 0);
}
static Gn3Src gn3_src(const sfron_split_src* s) {
  return Gn3Src{s->slabs, s->n_slabs, s->slab_stride, s->bias, s->sample_vec, s->ld_vec, s->resid, s->ld_resid};
}
int sfron_split_finish(const sfron_split_src* src, int64_t rows, int C, int rows_per_sample, float* out_f32, uint16_t* out_bf16, int ld_out,
                       void* stream) {
  SFRON_CHECK_ARG(split_src_ok(src, C) && rows > 0 && rows <= 0x7fffffff && rows_per_sample > 0 && ((out_f32 != nullptr) != (out_bf16 != nullptr)) &&
                  ld_out >= C && ld_out % 4 == 0);
  SFRON_CHECK_ARG(!src->resid || src->ld_resid == ld_out);
  SFRON_CHECK_ARG((((uintptr_t)out_f32) & 15) == 0 && (((uintptr_t)out_bf16) & 7) == 0);
  const int cb = (C / 4 + TPB - 1) / TPB;
  int chunks = 2048 / cb; if (chunks > rows) chunks = (int)rows; if (chunks < 1) chunks = 1;
  const int rpb = (int)((rows + chunks - 1) / chunks);
  hipLaunchKernelGGL(k_split_finish4, dim3(cb, (unsigned)((rows + rpb - 1) / rpb)), dim3(TPB), 0, (hipStream_t)stream, src->slabs, src->n_slabs,
                     src->slab_stride, (int)rows, C, src->bias, src->sample_vec, src->ld_vec, rows_per_sample, src->resid, out_f32, (__bf16*)out_bf16,
                     ld_out, rpb);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_groupnorm_one_launch(int B, int HW, int C, int groups) {
  return B > 0 && HW > 0 && C > 0 && groups > 0 && C % groups == 0 && C % 4 == 0 && groups <= 64 && gn3_gpb(B, HW, C, groups) > 0;
}
int sfron_groupnorm_fwd_src(const sfron_split_src* src, float* x, const float* gamma, const float* beta, int B, int HW, int C, int groups, float eps,
                            int swish, const uint8_t* drop_mask, float drop_scale, uint16_t* y, float* mean, float* rstd, void* stream) {
  SFRON_CHECK_ARG(x && gamma && beta && y && mean && rstd && groups > 0 && C % groups == 0 && split_src_ok(src, C));
  SFRON_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0 && (!drop_mask || ((uintptr_t)drop_mask & 3) == 0) && groups <= 64);
  SFRON_CHECK_ARG(!src->resid || (const float*)x != src->resid);
  const int gpb = gn3_gpb(B, HW, C, groups);
  if (!gpb) return SFRON_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_gn3_fwd<true>, dim3(B * (groups / gpb)), dim3(GN3_T), 0, (hipStream_t)stream, (const float*)x, C, gamma, beta, HW, C, groups, gpb,
                     eps, swish, drop_mask, drop_scale, (__bf16*)y, mean, rstd, gn3_src(src));
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_groupnorm_bwd_res_src(const sfron_split_src* src, float* dy, const float* x, int ldx, const float* gamma, const float* beta,
                                const float* mean, const float* rstd, int B, int HW, int C, int groups, int swish, const uint8_t* drop_mask,
                                float drop_scale, float* dx, int lddx, int accumulate, const float* extra, int ld_extra, float* part_gamma,
                                float* part_beta, void* stream) {
  SFRON_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && part_gamma && part_beta && groups > 0 && C % groups == 0 && split_src_ok(src, C));
  SFRON_CHECK_ARG(ldx % 4 == 0 && lddx % 4 == 0 && groups <= 64 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0 &&
                  (!drop_mask || ((uintptr_t)drop_mask & 3) == 0) && dy != dx);
  SFRON_CHECK_ARG(!extra || (ld_extra >= C && extra != dx && ld_extra % 4 == 0 && ((uintptr_t)extra & 15) == 0));
  const int gpb = gn3_gpb(B, HW, C, groups);
  if (!gpb) return SFRON_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_gn3_bwd<true>, dim3(B * (groups / gpb)), dim3(GN3_T), 0, (hipStream_t)stream, (const float*)dy, x, ldx, gamma, beta, mean, rstd, HW,
                     C, groups, gpb, swish, drop_mask, drop_scale, dx, lddx, accumulate, part_gamma, part_beta, extra, ld_extra, (__bf16*)nullptr,
                     (float*)nullptr, 1, gn3_src(src));
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_groupnorm_bwd_cast_src(const sfron_split_src* src, float* dy, const float* x, int ldx, const float* gamma, const float* beta,
                                 const float* mean, const float* rstd, int B, int HW, int C, int groups, int swish, const uint8_t* drop_mask,
                                 float drop_scale, uint16_t* dx_bf16, float* col_partials, float* part_gamma, float* part_beta, void* stream) {
  SFRON_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx_bf16 && col_partials && part_gamma && part_beta && split_src_ok(src, C));
  SFRON_CHECK_ARG(sfron_groupnorm_bwd_cast_ok(ldx, C, groups) && B > 0 && HW > 0);
  SFRON_CHECK_ARG((((uintptr_t)x | (uintptr_t)dy) & 15) == 0 && ((uintptr_t)dx_bf16 & 7) == 0 && (!drop_mask || ((uintptr_t)drop_mask & 3) == 0));
  const int gpb = gn3_gpb(B, HW, C, groups);
  if (!gpb) return SFRON_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_gn3_bwd<true>, dim3(B * (groups / gpb)), dim3(GN3_T), 0, (hipStream_t)stream, (const float*)dy, x, ldx, gamma, beta, mean, rstd, HW,
                     C, groups, gpb, swish, drop_mask, drop_scale, (float*)nullptr, C, 0, part_gamma, part_beta, (const float*)nullptr, 0,
                     (__bf16*)dx_bf16, col_partials, gn_chunks(B, HW), gn3_src(src));
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_softmax_fwd(const float* s, int64_t rows, int n, int n_valid, float scale, uint16_t* p, void* stream) {
  SFRON_CHECK_ARG(s && p && rows > 0 && n > 0 && n_valid > 0 && n_valid <= n);
  hipLaunchKernelGGL(k_softmax_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(TPB), 0, (hipStream_t)stream, s, rows, n, n_valid, scale, (__bf16*)p);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_layernorm_fwd(const float* x, const float* gamma, const float* beta, int64_t rows, int D, float eps, uint16_t* y, float* mean,
                        float* rstd, void* stream) {
  SFRON_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows > 0 && D > 0);
  const dim3 grid((unsigned)((rows + 3) / 4));
  const bool v4 = D % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0 && (((uintptr_t)y) & 7) == 0;
  if (v4 && D <= 512)       hipLaunchKernelGGL(k_layernorm_fwd_r<2>, grid, dim3(TPB), 0, (hipStream_t)stream, x, gamma, beta, rows, D, eps, (__bf16*)y, mean, rstd);
  else if (v4 && D <= 768)  hipLaunchKernelGGL(k_layernorm_fwd_r<3>, grid, dim3(TPB), 0, (hipStream_t)stream, x, gamma, beta, rows, D, eps, (__bf16*)y, mean, rstd);
  else if (v4 && D <= 1280) hipLaunchKernelGGL(k_layernorm_fwd_r<5>, grid, dim3(TPB), 0, (hipStream_t)stream, x, gamma, beta, rows, D, eps, (__bf16*)y, mean, rstd);
  else hipLaunchKernelGGL(k_layernorm_fwd, grid, dim3(TPB), 0, (hipStream_t)stream, x, gamma, beta, rows, D, eps, (__bf16*)y, mean, rstd);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
// rows per workgroup of the backward pass (= rows per partial parameter-gradient row): about 512 workgroups, 16 .. 64 rows each
int sfron_layernorm_rows_per_block(int64_t rows) {
  int64_t r = ((rows + 511) / 512 + 3) / 4 * 4;
  return (int)(r < 16 ? 16 : (r > 64 ? 64 : r));
}
int sfron_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, int64_t rows, int D, float* dx,
                        int accumulate, float* part_gamma, float* part_beta, void* stream) {
  return sfron_layernorm_bwd_res(dy, x, gamma, mean, rstd, rows, D, dx, accumulate, nullptr, part_gamma, part_beta, stream);
}
int sfron_layernorm_bwd_res(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, int64_t rows, int D, float* dx,
                            int accumulate, const float* extra, float* part_gamma, float* part_beta, void* stream) {
  SFRON_CHECK_ARG(dy && x && gamma && mean && rstd && dx && part_gamma && part_beta && rows > 0 && D > 0 && D <= 4096);
  SFRON_CHECK_ARG(extra != dx);
  const int rpb = sfron_layernorm_rows_per_block(rows);
  const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
  const size_t lds = (size_t)(TPB / 64) * 2 * D * sizeof(float);
  const bool v4 = D % 4 == 0 && ((((uintptr_t)dy) | ((uintptr_t)x) | ((uintptr_t)dx) | ((uintptr_t)gamma) | ((uintptr_t)extra)) & 15) == 0;
  if (v4 && D <= 512)       hipLaunchKernelGGL(k_layernorm_bwd_r<2>, grid, dim3(TPB), lds, (hipStream_t)stream, dy, x, gamma, mean, rstd, rows, D, dx, accumulate, part_gamma, part_beta, rpb, extra);
  else if (v4 && D <= 768)  hipLaunchKernelGGL(k_layernorm_bwd_r<3>, grid, dim3(TPB), lds, (hipStream_t)stream, dy, x, gamma, mean, rstd, rows, D, dx, accumulate, part_gamma, part_beta, rpb, extra);
  else if (v4 && D <= 1280) hipLaunchKernelGGL(k_layernorm_bwd_r<5>, grid, dim3(TPB), lds, (hipStream_t)stream, dy, x, gamma, mean, rstd, rows, D, dx, accumulate, part_gamma, part_beta, rpb, extra);
  else hipLaunchKernelGGL(k_layernorm_bwd, grid, dim3(TPB), lds, (hipStream_t)stream, dy, x, gamma, mean, rstd, rows, D, dx, accumulate, part_gamma, part_beta, rpb, extra);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_geglu_fwd(const float* h, int64_t rows, int F, uint16_t* out, void* stream) {
  SFRON_CHECK_ARG(h && out && rows > 0 && F > 0);
  if (F % 4 == 0 && ((((uintptr_t)h)) & 15) == 0 && (((uintptr_t)out) & 7) == 0) {
    const int cb = (F / 4 + TPB - 1) / TPB;
    int64_t chunks = 4096 / cb; if (chunks > rows) chunks = rows; if (chunks < 1) chunks = 1;
    const int rpb = (int)((rows + chunks - 1) / chunks);
    hipLaunchKernelGGL(k_geglu_fwd4, dim3(cb, (unsigned)((rows + rpb - 1) / rpb)), dim3(TPB), 0, (hipStream_t)stream, h, rows, F, rpb, (__bf16*)out);
    SFRON_LAUNCH_STATUS();
    return SFRON_OK;
  }
  hipLaunchKernelGGL(k_geglu_fwd, dim3(grid_for(rows * F)), dim3(TPB), 0, (hipStream_t)stream, h, rows, F, (__bf16*)out);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_geglu_bwd(const float* d_out, const float* h, int64_t rows, int F, uint16_t* dh, void* stream) {
  SFRON_CHECK_ARG(d_out && h && dh && rows > 0 && F > 0);
  if (F % 4 == 0 && ((((uintptr_t)h) | ((uintptr_t)d_out)) & 15) == 0 && (((uintptr_t)dh) & 7) == 0) {
    const int cb = (F / 4 + TPB - 1) / TPB;
    int64_t chunks = 4096 / cb; if (chunks > rows) chunks = rows; if (chunks < 1) chunks = 1;
    const int rpb = (int)((rows + chunks - 1) / chunks);
    hipLaunchKernelGGL(k_geglu_bwd4, dim3(cb, (unsigned)((rows + rpb - 1) / rpb)), dim3(TPB), 0, (hipStream_t)stream, d_out, h, rows, F, rpb, (__bf16*)dh);
    SFRON_LAUNCH_STATUS();
    return SFRON_OK;
  }
  hipLaunchKernelGGL(k_geglu_bwd, dim3(grid_for(rows * F)), dim3(TPB), 0, (hipStream_t)stream, d_out, h, rows, F, (__bf16*)dh);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_softmax_bwd(const uint16_t* p, const float* dp, int64_t rows, int n, float scale, uint16_t* ds, void* stream) {
  SFRON_CHECK_ARG(p && dp && ds && rows > 0 && n > 0);
  hipLaunchKernelGGL(k_softmax_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(TPB), 0, (hipStream_t)stream, (const __bf16*)p, dp, rows, n, scale, (__bf16*)ds);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

int sfron_axpby(const float* a, const float* b, float alpha, float beta, int64_t n, float* out, void* stream) {
  SFRON_CHECK_ARG(a && b && out);
  hipLaunchKernelGGL(k_axpby, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, a, b, alpha, beta, n, out);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_sample_colsum(const float* x, int ld, int B, int HW, int C, float* out, int ld_out, float* scratch, int64_t scratch_floats,
                        void* stream) {
  SFRON_CHECK_ARG(x && out && ld_out >= C && B > 0 && HW > 0 && C > 0);
  // few (column block, sample) pairs and many rows: chunk the rows, partial sums [B][nz][C] in the caller's scratch, fixed-order add
  int nz = 512 / (((C + 63) / 64) * B);
  if (nz > 32) nz = 32;
  if (nz > HW / 16) nz = HW / 16;
  if (scratch && nz > 1 && (int64_t)B * nz * C <= scratch_floats) {
    hipLaunchKernelGGL(k_sample_colsum, dim3((C + 63) / 64, B, nz), dim3(TPB), 0, (hipStream_t)stream, x, ld, HW, C, scratch, C);
    SFRON_LAUNCH_STATUS();
    return sfron_reduce_chunks(scratch, B, nz, C, out, ld_out, 0, stream);
  }
  hipLaunchKernelGGL(k_sample_colsum, dim3((C + 63) / 64, B, 1), dim3(TPB), 0, (hipStream_t)stream, x, ld, HW, C, out, ld_out);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_pool2_sum(const float* dy, int B, int H, int W, int C, float* dx, int accumulate, void* stream) {
  SFRON_CHECK_ARG(dy && dx);
  hipLaunchKernelGGL(k_pool2_sum, dim3(grid_for((int64_t)B * H * W * C)), dim3(TPB), 0, (hipStream_t)stream, dy, B, H, W, C, dx, accumulate);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_cast_rows_bf16(const float* x, int ldx, int64_t rows, int C, uint16_t* y, void* stream) {
  SFRON_CHECK_ARG(x && y && ldx >= C);
  if (C % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 7) == 0) {
    const int rpb = rows_chunk(rows, C);
    hipLaunchKernelGGL((k_cast_rows4<false>), dim3((C + 255) / 256, (unsigned)((rows + rpb - 1) / rpb)), dim3(TPB), 0, (hipStream_t)stream, x, ldx, rows, C,
                       rpb, (__bf16*)y, (float*)nullptr);
    SFRON_LAUNCH_STATUS();
    return SFRON_OK;
  }
  hipLaunchKernelGGL(k_cast_rows, dim3(grid_for(rows * C)), dim3(TPB), 0, (hipStream_t)stream, x, ldx, rows, C, (__bf16*)y);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
/* y = bf16(x) and colsum[c] = sum_r x[r][c] in one pass over x (the output gradient of a layer: its bf16 GEMM operand and its bias
 * gradient); partials: scratch of at least max_partials * C floats */
int sfron_cast_rows_colsum(const float* x, int ldx, int64_t rows, int C, uint16_t* y, float* partials, int max_partials, float* colsum,
                           void* stream) {
  SFRON_CHECK_ARG(colsum);
  int chunks = 0;
  const int rc = sfron_cast_rows_colsum_partials(x, ldx, rows, C, y, partials, max_partials, &chunks, stream);
  if (rc) return rc;
  return sfron_reduce_chunks(partials, 1, chunks, C, colsum, C, 0, stream);
}
int sfron_cast_rows_colsum_partials(const float* x, int ldx, int64_t rows, int C, uint16_t* y, float* partials, int max_partials,
                                    int* chunks_out, void* stream) {
  SFRON_CHECK_ARG(x && y && partials && chunks_out && rows > 0 && C > 0 && ldx >= C && max_partials > 0);
  SFRON_CHECK_ARG(C % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0 && (((uintptr_t)y) & 7) == 0);
  // about 512 workgroups over (column blocks x row chunks), at least 32 rows per chunk, within the caller's scratch
  int64_t chunks = 512 / ((C + 255) / 256);
  if (chunks > (rows + 31) / 32) chunks = (rows + 31) / 32;
  if (chunks > max_partials) chunks = max_partials;
  if (chunks < 1) chunks = 1;
  const int rpb = (int)((rows + chunks - 1) / chunks);
  chunks = (rows + rpb - 1) / rpb;
  hipLaunchKernelGGL((k_cast_rows4<true>), dim3((C + 255) / 256, (unsigned)chunks), dim3(TPB), 0, (hipStream_t)stream, x, ldx, rows, C, rpb, (__bf16*)y,
                     partials);
  SFRON_LAUNCH_STATUS();
  *chunks_out = (int)chunks;
  return SFRON_OK;
}
int sfron_dropout_mask(uint64_t seed, const int64_t* counter, int64_t salt, int64_t n, float p, uint8_t* mask, void* stream) {
  SFRON_CHECK_ARG(counter && mask && n > 0 && p >= 0.f && p < 1.f && (((uintptr_t)mask) & 3) == 0);
  const unsigned thresh = (unsigned)(p * 65536.0f + 0.5f);
  hipLaunchKernelGGL(k_dropout_mask, dim3(grid_for((n + 3) / 4)), dim3(TPB), 0, (hipStream_t)stream, seed, counter, salt, n, thresh, mask);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_dropout_mask_batch(uint64_t seed, const int64_t* counter, const int64_t* items_dev, int n_items, int64_t max_n, float p, uint8_t* mask_base,
                             void* stream) {
  SFRON_CHECK_ARG(counter && items_dev && mask_base && n_items > 0 && n_items <= 65535 && max_n > 0 && p >= 0.f && p < 1.f &&
                  (((uintptr_t)mask_base) & 3) == 0);
  const unsigned thresh = (unsigned)(p * 65536.0f + 0.5f);
  int gx = grid_for((max_n + 3) / 4);
  if (gx > 64) gx = 64;                         // n_items rows of workgroups fill the chip
  hipLaunchKernelGGL(k_dropout_mask_batch, dim3(gx, n_items), dim3(TPB), 0, (hipStream_t)stream, seed, counter, items_dev, thresh, mask_base);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_copy_cols(const float* x, int ldx, int64_t rows, int C, float* y, int ldy, int accumulate, void* stream) {
  SFRON_CHECK_ARG(x && y && ldx >= C && ldy >= C);
  if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {
    const int rpb = rows_chunk(rows, C);
    hipLaunchKernelGGL(k_copy_cols4, dim3((C + 255) / 256, (unsigned)((rows + rpb - 1) / rpb)), dim3(TPB), 0, (hipStream_t)stream, x, ldx, rows, C, y, ldy,
                       accumulate, rpb);
    SFRON_LAUNCH_STATUS();
    return SFRON_OK;
  }
  hipLaunchKernelGGL(k_copy_cols, dim3(grid_for(rows * C)), dim3(TPB), 0, (hipStream_t)stream, x, ldx, rows, C, y, ldy, accumulate);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_copy_cols2(const float* x0, int ldx0, int C0, float* y0, int ldy0, int acc0, const float* x1, int ldx1, int C1, float* y1, int ldy1,
                     int acc1, int64_t rows, void* stream) {
  SFRON_CHECK_ARG(x0 && y0 && x1 && y1 && rows > 0 && C0 > 0 && C1 > 0 && ldx0 >= C0 && ldy0 >= C0 && ldx1 >= C1 && ldy1 >= C1);
  const bool wide = ((C0 | C1 | ldx0 | ldx1 | ldy0 | ldy1) & 3) == 0 &&
                    ((((uintptr_t)x0) | ((uintptr_t)y0) | ((uintptr_t)x1) | ((uintptr_t)y1)) & 15) == 0;
  if (!wide) {                                   // unaligned shapes: the two plain launches
    const int rc = sfron_copy_cols(x0, ldx0, rows, C0, y0, ldy0, acc0, stream);
    return rc != SFRON_OK ? rc : sfron_copy_cols(x1, ldx1, rows, C1, y1, ldy1, acc1, stream);
  }
  const int Cm = C0 > C1 ? C0 : C1;
  const int rpb = rows_chunk(rows, Cm);
  hipLaunchKernelGGL(k_copy_cols4x2, dim3((Cm + 255) / 256, (unsigned)((rows + rpb - 1) / rpb), 2), dim3(TPB), 0, (hipStream_t)stream,
                     CopyPiece{x0, ldx0, y0, ldy0, C0, acc0}, CopyPiece{x1, ldx1, y1, ldy1, C1, acc1}, rows, rpb);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_ddpm_timestep_embed(const float* t, int n, int dim, uint16_t* out, void* stream) {
  SFRON_CHECK_ARG(t && out && dim >= 4);
  hipLaunchKernelGGL(k_ddpm_temb, dim3(grid_for((int64_t)n * dim)), dim3(TPB), 0, (hipStream_t)stream, t, n, dim, (__bf16*)out);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_class_embed_fwd(const float* table, const float* null_emb, const int64_t* c, const uint8_t* keep, int n_classes, int n, int D,
                          uint16_t* out, void* stream) {
  SFRON_CHECK_ARG(table && null_emb && c && out);
  hipLaunchKernelGGL(k_class_embed, dim3(grid_for((int64_t)n * D)), dim3(TPB), 0, (hipStream_t)stream, table, null_emb, c, keep, n_classes, n, D,
                     (__bf16*)out);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_class_embed_bwd(const float* d, const int64_t* c, const uint8_t* keep, int n_classes, int n, int D, float* d_table, float* d_null,
                          void* stream) {
  SFRON_CHECK_ARG(d && c && d_table && d_null);
  hipLaunchKernelGGL(k_class_embed_bwd, dim3((D + TPB - 1) / TPB), dim3(TPB), 0, (hipStream_t)stream, d, c, keep, n_classes, n, D, d_table, d_null);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}

}  // extern "C"
