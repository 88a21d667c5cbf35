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
  const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const long bz = blockIdx.y, bi = blockIdx.z;
  int kbeg = 0, kend = g.K;
  if (g.kchunk > 0) { kbeg = (int)bz * g.kchunk; kend = min(g.K, kbeg + g.kchunk); }
  else { g.A += bz * g.sA; g.B += bz * g.sB; }
  g.A += bi * g.sA2; g.B += bi * g.sB2;
  const long co = bz * g.sC + bi * g.sC2;
  if (g.Cb) g.Cb += co;
  if (g.Cf) g.Cf += co;
  if (g.resid) g.resid += co;

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

namespace {

constexpr int TPB = 256;
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
// weight gradient fp32 [nslab][Co_p][taps][Ci_p] (GEMM output, split-K slabs) -> OIHW gradient (overwrite; slabs added in order)
__global__ __launch_bounds__(TPB) void k_conv_wgrad_scatter(const float* __restrict__ g, int Co, int Ci, int taps, int Ci_p, int nslab,
                                                            int64_t slab_stride, float* __restrict__ dw) {
  const int64_t n = (int64_t)Co * Ci * taps;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const int t = (int)(i % taps); const int ci = (int)((i / taps) % Ci); const int co = (int)(i / ((int64_t)taps * Ci));
    const float* p = g + ((int64_t)co * taps + t) * Ci_p + ci;
    float a = 0.f;
    for (int sl = 0; sl < nslab; ++sl) a += p[sl * slab_stride];
    dw[i] = a;
  }
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
  double s = 0.0, ss = 0.0;
  for (int i = threadIdx.x; i < n; i += TPB) {
    const float v = xb[(size_t)(i / cg) * ldx + (i % cg)];
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
  for (int i = threadIdx.x; i < n; i += TPB) {
    const int p = i / cg, c = i % cg;
    float z = (xb[(size_t)p * ldx + c] - m) * r * gamma[gi * cg + c] + beta[gi * cg + c];
    if (swish) z = silu(z);
    if (mb) z = mb[(size_t)p * C + c] ? z * drop_scale : 0.f;
    yb[(size_t)p * C + c] = f2bf(z);
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
  for (int i = threadIdx.x; i < 2 * cg; i += TPB) chs[i] = 0.f;
  __syncthreads();
  const float m = mean[blockIdx.x], r = rstd[blockIdx.x];
  const float* xb = x + (size_t)b * HW * ldx + gi * cg;
  const float* dyb = dy + (size_t)b * HW * C + gi * cg;
  const uint8_t* mb = mask ? mask + (size_t)b * HW * C + gi * cg : nullptr;
  const int n = HW * cg;
  // pass 1: dz = dy * act'(z) (* dropout); group sums of dz * gamma and dz * gamma * xhat; per-channel sums
  // A thread visits indices i = tid + k * TPB: its channel (i % cg) is fixed when TPB % cg == 0 (cg is a power of two here)
  float s1 = 0.f, s2 = 0.f;
  const bool fixed_c = (TPB % cg) == 0;
  float ca = 0.f, cb = 0.f;
  for (int i = threadIdx.x; i < n; i += TPB) {
    const int p = i / cg, c = i % cg;
    const float xh = (xb[(size_t)p * ldx + c] - m) * r;
    const float ga = gamma[gi * cg + c];
    float d = dyb[(size_t)p * C + c];
    if (mb) d = mb[(size_t)p * C + c] ? d * drop_scale : 0.f;
    if (swish) d *= silu_grad(xh * ga + beta[gi * cg + c]);
    s1 += d * ga; s2 += d * ga * xh;
    if (fixed_c) { ca += d * xh; cb += d; }
    else { atomicAdd(&chs[c], d * xh); atomicAdd(&chs[cg + c], d); }      // LDS, small odd shapes only
  }
  if (fixed_c) {
    // threads l, l + cg, l + 2cg ... of the workgroup share a channel: reduce through LDS in a fixed order
    float* tmp = sh + 2 * (TPB / 64) + 2 * cg;   // [2][TPB]  (threads beyond the group's element count hold zeros)
    tmp[threadIdx.x] = ca; tmp[TPB + threadIdx.x] = cb;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s1; sh[TPB / 64 + (threadIdx.x >> 6)] = s2; }
  __syncthreads();
  if (fixed_c && threadIdx.x < cg) {
    const float* tmp = sh + 2 * (TPB / 64) + 2 * cg;
    float a = 0.f, q = 0.f;
    for (int j = threadIdx.x; j < TPB; j += cg) { a += tmp[j]; q += tmp[TPB + j]; }
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
  for (int i = threadIdx.x; i < n; i += TPB) {
    const int p = i / cg, c = i % cg;
    const float xh = (xb[(size_t)p * ldx + c] - m) * r;
    const float ga = gamma[gi * cg + c];
    float d = dyb[(size_t)p * C + c];
    if (mb) d = mb[(size_t)p * C + c] ? d * drop_scale : 0.f;
    if (swish) d *= silu_grad(xh * ga + beta[gi * cg + c]);
    const float v = r * (d * ga - k1 - xh * k2);
    float* o = dxb + (size_t)p * lddx + c;
    *o = accumulate ? *o + v : v;
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
// dx (+)= d LayerNorm; pg / pb [nblk][D]: per-workgroup (4 rows) partial sums of dy * xhat and dy, written with plain stores
__global__ __launch_bounds__(TPB) void k_layernorm_bwd(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, int64_t rows, int D,
                                                       float* __restrict__ dx, int accumulate, float* __restrict__ pg, float* __restrict__ pb,
                                                       int rows_per_block) {
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
      *o = accumulate ? *o + v : v;
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

// ---- small pieces
// out[b][c] = sum_{p < HW} x[(b * HW + p) * ld + c]  (per-sample column sums: gradient of a per-sample broadcast vector)
// one workgroup per (64-column slice, sample): the 4 waves stride over the rows (coalesced 256-B row reads), partial sums meet in LDS
// in a fixed order
__global__ __launch_bounds__(TPB) void k_sample_colsum(const float* __restrict__ x, int ld, int HW, int C, float* __restrict__ out, int ldo) {
  __shared__ float sh[TPB / 64][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < C) {
    const float* xb = x + (size_t)b * HW * ld + c;
    for (int p = wave; p < HW; p += TPB / 64) s += xb[(size_t)p * ld];
  }
  sh[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < C) {
    float a = 0.f;
    for (int w = 0; w < TPB / 64; ++w) a += sh[w][lane];
    out[(size_t)b * ldo + c] = a;
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
  hipLaunchKernelGGL((k_bgemm<A_TR, B_TR, EPI, CONV>), dim3(ntm * ntn, nbatch, ninner), dim3(NT), lds, s, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFRON_OK : (int)e;
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
  if (!d->a_transposed && !d->b_transposed)
    return bf ? launch_bgemm<false, false, EPI_BF16, CONV_NONE>(g, d->batch, s, ni) : launch_bgemm<false, false, EPI_RES, CONV_NONE>(g, d->batch, s, ni);
  if (!d->a_transposed && d->b_transposed)
    return bf ? launch_bgemm<false, true, EPI_BF16, CONV_NONE>(g, d->batch, s, ni) : launch_bgemm<false, true, EPI_RES, CONV_NONE>(g, d->batch, s, ni);
  if (d->a_transposed && d->b_transposed) {
    // a plain weight gradient (fp32 [M][N] contiguous, contraction over all rows): split-K through the caller's slab scratch
    if (!bf && d->split_ws && d->batch == 1 && ni == 1 && d->ldc == d->N && !d->bias && !d->resid && !d->sample_vec && !d->accumulate) {
      const int sp = plan_splits(d->M, d->N, d->K, d->split_ws_slabs);
      if (sp > 1) {
        g.kchunk = ((g.K + sp - 1) / sp + BK - 1) / BK * BK;
        const int used = (g.K + g.kchunk - 1) / g.kchunk;
        g.Cf = d->split_ws; g.sC = (long)d->M * d->N;
        const int rc = launch_bgemm<true, true, EPI_RES, CONV_NONE>(g, used, s, 1);
        if (rc) return rc;
        return sfron_reduce_chunks(d->split_ws, 1, used, d->M * d->N, d->c_f32, d->M * d->N, 0, stream);
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
  const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
  if (d->split_ws && !d->accumulate && tiles < 128 && g.K >= 2048) {
    int sp = 384 / tiles;
    if (sp > g.K / 512) sp = g.K / 512;
    if (sp > d->split_ws_slabs) sp = d->split_ws_slabs;
    if (sp > 1) {
      BGemmArgs q = g;
      q.kchunk = ((g.K + sp - 1) / sp + BK - 1) / BK * BK;
      const int used = (g.K + q.kchunk - 1) / q.kchunk;
      q.Cb = nullptr; q.Cf = d->split_ws; q.ldcf = g.N; q.sC = (long)g.M * g.N;
      q.bias = nullptr; q.resid = nullptr; q.vec = nullptr;
      rc = launch_bgemm<false, false, EPI_RES, CONV_A>(q, used, (hipStream_t)stream);
      if (rc) return rc;
      hipLaunchKernelGGL(k_split_finish, dim3(grid_for((int64_t)g.M * g.N)), dim3(TPB), 0, (hipStream_t)stream, d->split_ws, used, (int64_t)g.M * g.N,
                         g.M, g.N, g.bias, g.vec, g.ldvec, g.T, g.resid, g.Cf, g.Cb, d->ld_out);
      SFRON_LAUNCH_STATUS();
      return SFRON_OK;
    }
  }
  return g.Cb ? launch_bgemm<false, false, EPI_BF16, CONV_A>(g, 1, (hipStream_t)stream)
              : launch_bgemm<false, false, EPI_RES, CONV_A>(g, 1, (hipStream_t)stream);
}

/* weight gradient: dw[n][tap][c] = sum_p dy[p][n] src[src(p, tap)][c]  (fp32 [n_out][taps * c_src], then k_conv_wgrad_scatter) */
int sfron_conv_wgrad_splits(const sfron_conv_desc* d) {
  if (!d) return 0;
  return plan_splits(d->n_out, d->taps * d->c_src, d->batch * d->h_out * d->w_out, 64);
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
  const int sp = sfron_conv_wgrad_splits(d);
  if (sp > 1) { g.kchunk = ((g.K + sp - 1) / sp + BK - 1) / BK * BK; g.sC = (long)g.M * g.N; }
  const int used = sp > 1 ? (g.K + g.kchunk - 1) / g.kchunk : 1;
  rc = launch_bgemm<true, true, EPI_RES, CONV_B>(g, used, (hipStream_t)stream);
  if (rc) return rc;
  // slabs the rounding left without rows are zeroed so that the scatter can always add `sfron_conv_wgrad_splits` of them
  for (int sidx = used; sidx < sp; ++sidx)
    if (hipMemsetAsync(dw_gemm + (size_t)sidx * g.M * g.N, 0, (size_t)g.M * g.N * sizeof(float), (hipStream_t)stream) != hipSuccess) return (int)hipGetLastError();
  return SFRON_OK;
}

int sfron_conv_wprep(const float* w_oihw, int c_out, int c_in, int taps, int c_out_p, int c_in_p, uint16_t* w_fwd, uint16_t* w_dgrad,
                     void* stream) {
  SFRON_CHECK_ARG(w_oihw && w_fwd && c_out_p >= c_out && c_in_p >= c_in && (taps == 9 || taps == 1));
  const int64_t n = (int64_t)c_out_p * taps * c_in_p + (w_dgrad ? (int64_t)c_in * taps * c_out_p : 0);
  hipLaunchKernelGGL(k_conv_wprep, dim3(grid_for(n)), dim3(TPB), 0, (hipStream_t)stream, w_oihw, c_out, c_in, taps, c_out_p, c_in_p,
                     (__bf16*)w_fwd, (__bf16*)w_dgrad);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_conv_wgrad_scatter(const float* dw_gemm, int c_out, int c_in, int taps, int c_in_p, int n_slabs, int64_t slab_stride,
                             float* dw_oihw, void* stream) {
  SFRON_CHECK_ARG(dw_gemm && dw_oihw && n_slabs >= 1);
  hipLaunchKernelGGL(k_conv_wgrad_scatter, dim3(grid_for((int64_t)c_out * c_in * taps)), dim3(TPB), 0, (hipStream_t)stream, dw_gemm,
                     c_out, c_in, taps, c_in_p, n_slabs, slab_stride, dw_oihw);
  SFRON_LAUNCH_STATUS();
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

int sfron_groupnorm_fwd(const float* x, int ldx, const float* gamma, const float* beta, int B, int HW, int C, int groups, float eps,
                        int swish, const uint8_t* drop_mask, float drop_scale, uint16_t* y, float* mean, float* rstd, void* stream) {
  SFRON_CHECK_ARG(x && gamma && beta && y && mean && rstd && groups > 0 && C % groups == 0 && ldx >= C);
  hipLaunchKernelGGL(k_gn_fwd, dim3(B * groups), dim3(TPB), 0, (hipStream_t)stream, x, ldx, gamma, beta, HW, C, groups, eps, swish, drop_mask,
                     drop_scale, (__bf16*)y, mean, rstd);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_groupnorm_bwd(const float* dy, const float* x, int ldx, const float* gamma, const float* beta, const float* mean,
                        const float* rstd, int B, int HW, int C, int groups, int swish, const uint8_t* drop_mask, float drop_scale,
                        float* dx, int lddx, int accumulate, float* part_gamma, float* part_beta, void* stream) {
  SFRON_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && part_gamma && part_beta && groups > 0 && C % groups == 0);
  const int cg = C / groups;
  const size_t lds = (2 * (TPB / 64) + 2 * cg + 2 * TPB) * sizeof(float);
  hipLaunchKernelGGL(k_gn_bwd, dim3(B * groups), dim3(TPB), lds, (hipStream_t)stream, dy, x, ldx, gamma, beta, mean, rstd, HW, C, groups, swish,
                     drop_mask, drop_scale, dx, lddx, accumulate, part_gamma, part_beta);
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
  hipLaunchKernelGGL(k_layernorm_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(TPB), 0, (hipStream_t)stream, x, gamma, beta, rows, D, eps, (__bf16*)y,
                     mean, rstd);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_layernorm_rows_per_block(void) { return 64; }
int sfron_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, int64_t rows, int D, float* dx,
                        int accumulate, float* part_gamma, float* part_beta, void* stream) {
  SFRON_CHECK_ARG(dy && x && gamma && mean && rstd && dx && part_gamma && part_beta && rows > 0 && D > 0 && D <= 4096);
  const int rpb = 64;
  hipLaunchKernelGGL(k_layernorm_bwd, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(TPB), (size_t)(TPB / 64) * 2 * D * sizeof(float),
                     (hipStream_t)stream, dy, x, gamma, mean, rstd, rows, D, dx, accumulate, part_gamma, part_beta, rpb);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_geglu_fwd(const float* h, int64_t rows, int F, uint16_t* out, void* stream) {
  SFRON_CHECK_ARG(h && out && rows > 0 && F > 0);
  hipLaunchKernelGGL(k_geglu_fwd, dim3(grid_for(rows * F)), dim3(TPB), 0, (hipStream_t)stream, h, rows, F, (__bf16*)out);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_geglu_bwd(const float* d_out, const float* h, int64_t rows, int F, uint16_t* dh, void* stream) {
  SFRON_CHECK_ARG(d_out && h && dh && rows > 0 && F > 0);
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
int sfron_sample_colsum(const float* x, int ld, int B, int HW, int C, float* out, int ld_out, void* stream) {
  SFRON_CHECK_ARG(x && out && ld_out >= C);
  hipLaunchKernelGGL(k_sample_colsum, dim3((C + 63) / 64, B), dim3(TPB), 0, (hipStream_t)stream, x, ld, HW, C, out, ld_out);
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
  hipLaunchKernelGGL(k_cast_rows, dim3(grid_for(rows * C)), dim3(TPB), 0, (hipStream_t)stream, x, ldx, rows, C, (__bf16*)y);
  SFRON_LAUNCH_STATUS();
  return SFRON_OK;
}
int sfron_copy_cols(const float* x, int ldx, int64_t rows, int C, float* y, int ldy, int accumulate, void* stream) {
  SFRON_CHECK_ARG(x && y && ldx >= C && ldy >= C);
  hipLaunchKernelGGL(k_copy_cols, dim3(grid_for(rows * C)), dim3(TPB), 0, (hipStream_t)stream, x, ldx, rows, C, y, ldy, accumulate);
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
