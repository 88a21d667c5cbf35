// bf16 MFMA GEMM for the DiT token GEMMs (forward, dgrad, wgrad) with fused epilogues.
//
// Replaces the cuBLAS calls behind nn.Linear / Conv2d(k=s=p) forward and backward on the
// reference hot path (DiT/models.py:108-121,138-142,169 via timm Attention/Mlp/PatchEmbed):
//   forward  Y[M,N]  = X[M,K]  · W[N,K]^T      A direct, B direct
//   dgrad    dX[M,K] = dY[M,N] · W[N,K]        A direct, B transposed-read (contraction = rows of W)
//   wgrad    dW[N,K] = dY[M,N]^T · X[M,K]      A and B transposed-read (contraction = token rows)
// "transposed-read" operands stay row-major in HBM and in LDS; the MFMA fragment is gathered
// with ds_read_b64_tr_b16, so no transposed copies of weights or activations exist anywhere.
//
// Tile 128x128x64, 4 waves (2x2), each wave 4x4 MFMA 16x16x32 bf16 tiles, fp32 accumulate.
// The MFMA is issued as D^T = B^T·A^T so each lane ends up with 4 CONSECUTIVE output columns
// (8-byte bf16 / 16-byte fp32 stores).  LDS: double-buffered, XOR-swizzled so that both the
// ds_read_b128 row fragments and the transposed reads are bank-conflict free.
#include "common.h"
#include "../../include/sfron.h"

extern int g_conv_loader_waves;     // conv.hip: k_cgemm
extern int g_fp8_loader_waves;      // fp8.hip: the fp8 tiles switch form together with the bf16 ones (sfron_gemm_loader_waves)
#include <atomic>

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per function AND per device: remember per device whether it was set
static inline bool need_attr(std::atomic<uint64_t>& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  return (mask.fetch_or(bit) & bit) == 0;
}

struct GemmArgs {
  const __bf16* A; const __bf16* B;
  int M, N, K;              // output M x N, contraction K
  int lda, ldb;
  __bf16* Cb; int ldcb;     // bf16 output
  float* Cf; int ldcf;      // fp32 output / residual stream (EPI_GATE_RES)
  const float* bias;        // [N] fp32 or null
  __bf16* aux; int ldaux;   // EPI_GELU: pre-activation out; EPI_GATE_RES: branch out; EPI_DGELU: pre-activation in
  const float* gate; int ldgate;   // EPI_GATE_RES: gate[(row / T) * ldgate + col]
  const float* pos;         // EPI_POS: pos[(row % T) * N + col]
  int T;
  float alpha;
  int accumulate;           // EPI_F32: Cf += result
  const float* resid;       // EPI_GATE_RES: Cf = resid + gate * result (resid may alias Cf)
  int kchunk;               // split-K: contraction range per split (multiple of BK); K if no split
  long split_stride;        // split-K: fp32 slab stride (elements) between splits
  int nt_out;               // streaming (nontemporal) stores: bit 0 = the epilogue's saved-for-backward bf16 output (aux),
                            // bit 1 = fp32 weight-gradient output
  int ntm, ntn;
  int group_m;              // fast path: tile-rows per group of the grouped tile order
  float* bsum;              // weight-gradient layout: bsum[m] = sum_k op(A)[m][k] (bias gradient), or null
#ifdef SFRON_DEBUG_KNOBS
  int dbg_same;             // timing experiments only (see k_gemm_pipe)
  long long* dbg_clk;       // in-kernel clock stamps: [layout class 3][1024 workgroups][2] = (delta s_memtime, delta s_memrealtime) of the K-loop
#endif
  const uint8_t* sq_mask;   // EPI_SUMSQ: byte mask [M][N] (ld = N) or null; sq_out[workgroup] = sum over the tile of (mask ? c : 0)^2
  double* sq_out;
  float* colpart;           // EPI_DGELU on the 256-row pipelined tiles: colpart[tm * N + n] = sum over the tile's 256 rows of the
                            // fp32 output (before bf16 rounding) -- per-tile-row partials of the fc1 bias gradient, or null
};

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NT = 256;
constexpr int TILE_ELEMS = 128 * 64;   // both image kinds hold 8192 bf16 = 16 KiB

enum { EPI_BF16 = 0, EPI_F32 = 1, EPI_GELU = 2, EPI_GATE_RES = 3, EPI_DGELU = 4, EPI_POS = 5, EPI_SUMSQ = 6,
       // round 6, the 256 x 192 pipelined tile only: EPI_GELU / EPI_DGELU whose `aux` is GELU'(pre-activation) as ONE byte per element
       // (common.h geluq_pack4) instead of the bf16 pre-activation: fc1 writes 113 instead of 151 MB, the fc2 dgrad reads half and has no exp
       EPI_GELUQ = 7, EPI_DGELUQ = 8 };



// ---- LDS images -------------------------------------------------------------------------------
// direct image: [128 rows][64 k], 128-B rows, 16-B chunk c of row r stored at chunk c ^ (r & 7)
__device__ __forceinline__ int off_direct(int row, int ch) { return row * 64 + ((ch ^ (row & 7)) << 3); }
// transposed-read image: [64 k-rows][128 cols], 256-B rows, chunk c of row r at c ^ (((r&3)<<2)|((r>>2)&3))
__device__ __forceinline__ int swz_tr(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ int off_tr(int krow, int ch) { return krow * 128 + ((ch ^ swz_tr(krow)) << 3); }

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x8 frag_direct(const __bf16* img, int row, int kchunk) {
  return *reinterpret_cast<const bf16x8*>(img + off_direct(row, kchunk));
}
// fragment for 16 consecutive columns starting at col0 (multiple of 16), k rows kr0 + 8*(lane>>4) + 0..7
__device__ __forceinline__ bf16x8 frag_tr(const __bf16* img, int col0, int kr0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int ch = (col0 >> 3) + (p >> 1);
  const int r0 = kr0 + 8 * g + q;
  const __bf16* a0 = img + off_tr(r0, ch) + 4 * (p & 1);
  const __bf16* a1 = img + off_tr(r0 + 4, ch) + 4 * (p & 1);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a0);
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)a1);
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

// ---- global -> register staging --------------------------------------------------------------
// direct operand: rows = output index (clamped), k contiguous
template <bool TR>
struct Stager {
  const __bf16* base[4];
  bool valid[4];           // static validity (column range for TR; always true for direct)
  int kidx[4];             // direct: k offset of the chunk inside the tile; TR: k-row inside the tile
  int lds_off[4];
  int ld;
  __device__ __forceinline__ void init(const __bf16* P, int ld_, int dim, int d0, int tid) {
    ld = ld_;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + NT * i;
      if (!TR) {
        const int row = c >> 3, ch = c & 7;
        int gr = d0 + row;
        gr = gr < dim ? gr : dim - 1;
        base[i] = P + (size_t)gr * ld + ch * 8;
        kidx[i] = ch * 8;
        valid[i] = true;
        lds_off[i] = off_direct(row, ch);
      } else {
        const int krow = c >> 4, ch = c & 15;
        const int col = d0 + ch * 8;
        valid[i] = col < dim;
        base[i] = P + (size_t)krow * ld + (valid[i] ? col : 0);
        kidx[i] = krow;
        lds_off[i] = off_tr(krow, ch);
      }
    }
  }
  template <bool FULL>
  __device__ __forceinline__ void load(uint4 (&r)[4], int k0, int K) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __bf16* p = TR ? base[i] + (size_t)k0 * ld : base[i] + k0;
      if (FULL) {               // whole tile in range: no predication (each predicated load costs an exec-mask branch)
        r[i] = *reinterpret_cast<const uint4*>(p);
      } else {
        const bool ok = valid[i] && (k0 + kidx[i] < K);
        r[i] = ok ? *reinterpret_cast<const uint4*>(p) : make_uint4(0, 0, 0, 0);
      }
    }
  }
  __device__ __forceinline__ void store(__bf16* img, const uint4 (&r)[4]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(img + lds_off[i]) = r[i];
  }
};


// ---- fused epilogue for 4 consecutive output columns (row, col..col+3) -----------------------------
// streaming store for tensors that only the backward pass reads again (pre-activations, branch outputs): keeps them from
// displacing the operands of the next kernels in L2 / Infinity Cache
__device__ __forceinline__ void nt_store(bf16x4* p, bf16x4 v, int nt) {
  if (nt) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <int EPI>
__device__ __forceinline__ void epilogue_store(const GemmArgs& g, int row, int col, f32x4 v, const float4* pre_bias = nullptr) {
  v = v * g.alpha;
  if (pre_bias) {                                     // bias fetched before the main loop (its latency is not on the epilogue's path)
    v[0] += pre_bias->x; v[1] += pre_bias->y; v[2] += pre_bias->z; v[3] += pre_bias->w;
  } else if (g.bias) {
    const float4 b = *reinterpret_cast<const float4*>(g.bias + col);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (EPI == EPI_BF16) {
    bf16x4 o = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    *reinterpret_cast<bf16x4*>(g.Cb + (size_t)row * g.ldcb + col) = o;
  } else if (EPI == EPI_F32) {
    float4* dst = reinterpret_cast<float4*>(g.Cf + (size_t)row * g.ldcf + col);
    float4 o = make_float4(v[0], v[1], v[2], v[3]);
    if (g.accumulate) { const float4 c = *dst; o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
    if (g.nt_out & 2) __builtin_nontemporal_store(f32x4{o.x, o.y, o.z, o.w}, reinterpret_cast<f32x4*>(dst));
    else *dst = o;
  } else if (EPI == EPI_GELU) {
    bf16x4 h = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    nt_store(reinterpret_cast<bf16x4*>(g.aux + (size_t)row * g.ldaux + col), h, g.nt_out & 1);   // read again only by the backward pass
    const bf16x4 o = f2bf4(gelu_tanh4(v));
    *reinterpret_cast<bf16x4*>(g.Cb + (size_t)row * g.ldcb + col) = o;
  } else if (EPI == EPI_GATE_RES) {
    bf16x4 a = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    nt_store(reinterpret_cast<bf16x4*>(g.aux + (size_t)row * g.ldaux + col), a, g.nt_out & 1);
    const float4 gt = *reinterpret_cast<const float4*>(g.gate + (size_t)(row / g.T) * g.ldgate + col);
    float4 x = *reinterpret_cast<const float4*>(g.resid + (size_t)row * g.ldcf + col);
    x.x += gt.x * v[0]; x.y += gt.y * v[1]; x.z += gt.z * v[2]; x.w += gt.w * v[3];
    *reinterpret_cast<float4*>(g.Cf + (size_t)row * g.ldcf + col) = x;
  } else if (EPI == EPI_DGELU) {
    const bf16x4 h = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)row * g.ldaux + col);
    const bf16x4 o = f2bf4(v * gelu_tanh_grad4(bf2f4(h)));
    *reinterpret_cast<bf16x4*>(g.Cb + (size_t)row * g.ldcb + col) = o;
  } else if (EPI == EPI_POS) {
    const float4 pe = *reinterpret_cast<const float4*>(g.pos + (size_t)(row % g.T) * g.N + col);
    *reinterpret_cast<float4*>(g.Cf + (size_t)row * g.ldcf + col) =
        make_float4(v[0] + pe.x, v[1] + pe.y, v[2] + pe.z, v[3] + pe.w);
  }
}

template <bool A_TR, bool B_TR, int EPI, bool FULL>
__global__ __launch_bounds__(NT) void k_gemm(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  __bf16* sA = smem;                       // [2][TILE_ELEMS]
  __bf16* sB = smem + 2 * TILE_ELEMS;      // [2][TILE_ELEMS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware tile order: blocks b and b+8 share an XCD (speed only); give each XCD a contiguous
  // run of tiles so neighbouring tiles (same A row-panel) hit the same L2.
  const int nblk = gridDim.x;
  int id;
  {
    const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, x = b & 7, y = b >> 3;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  const int ntiles = g.ntm * g.ntn;
  const int split = id / ntiles;
  id -= split * ntiles;
  const int tm = id / g.ntn, tn = id % g.ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = split * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  if (EPI == EPI_F32) g.Cf += (size_t)split * g.split_stride;

  Stager<A_TR> stA;
  Stager<B_TR> stB;
  stA.init(g.A, g.lda, g.M, m0, tid);
  stB.init(g.B, g.ldb, g.N, n0, tid);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (kend - kbeg + BK - 1) / BK;
  // EPI_SUMSQ (a rank-R product: ONE K-tile): the tile's mask words are requested in front of the operand loads -- behind the K-loop they were a
  // second exposed memory latency in a workgroup that lives for two (188 us for the adaLN matrix of DiT-XL/2, on the critical stream).  Branch-free:
  // out-of-range positions read a clamped address and are zeroed below.
  [[maybe_unused]] uint32_t mkw[4][4];
  if constexpr (EPI == EPI_SUMSQ) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int row = m0 + wm * 64 + mt * 16 + (lane & 15), col = n0 + wn * 64 + nt * 16 + 4 * (lane >> 4);
        const int rc = row < g.M ? row : g.M - 1, cc = col < g.N ? col : g.N - 4;
        mkw[mt][nt] = g.sq_mask ? *reinterpret_cast<const uint32_t*>(g.sq_mask + (size_t)rc * g.N + cc) : 0x01010101u;
      }
  }
  uint4 ra[4], rb[4];
  stA.template load<FULL>(ra, kbeg, kend);
  stB.template load<FULL>(rb, kbeg, kend);
  stA.store(sA, ra);
  stB.store(sB, rb);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      stA.template load<FULL>(ra, kbeg + (kt + 1) * BK, kend);
      stB.template load<FULL>(rb, kbeg + (kt + 1) * BK, kend);
    }
    const __bf16* iA = sA + cur * TILE_ELEMS;
    const __bf16* iB = sB + cur * TILE_ELEMS;
    if (FULL) __builtin_amdgcn_sched_barrier(0);     // keep the staging loads ahead of this tile's MFMAs
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
    if (FULL) __builtin_amdgcn_sched_barrier(0);     // ... and the ds_writes of the next tile behind them
    if (more) {
      stA.store(sA + (cur ^ 1) * TILE_ELEMS, ra);
      stB.store(sB + (cur ^ 1) * TILE_ELEMS, rb);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = mt*16 + (lane&15)][n = nt*16 + 4*(lane>>4) + 0..3]
  if constexpr (EPI == EPI_SUMSQ) {
    // the product is never stored: masked sum of squares of the tile (the clip-norm pre-pass over a gradient that exists only as its
    // two factors: sweep.hip sfron_sumsq_lowrank).  All 16 mask words of a lane are fetched before the first use; lanes, then waves,
    // are summed in a fixed order (double from the wave sums on): bitwise reproducible.
    float sq = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int row = m0 + wm * 64 + mt * 16 + (lane & 15), col = n0 + wn * 64 + nt * 16 + 4 * (lane >> 4);
        const uint32_t m4 = (row < g.M && col < g.N) ? mkw[mt][nt] : 0u;
        const f32x4 v = acc[mt][nt] * g.alpha;
        sq += ((m4 & 0xffu) ? v[0] * v[0] : 0.f) + ((m4 & 0xff00u) ? v[1] * v[1] : 0.f) + ((m4 & 0xff0000u) ? v[2] * v[2] : 0.f) +
              ((m4 & 0xff000000u) ? v[3] * v[3] : 0.f);
      }
    double d = wave_sum_d((double)sq);
    double* red = reinterpret_cast<double*>(smem);          // (the main loop ended with a barrier: the staging images are free)
    if (lane == 0) red[wave] = d;
    __syncthreads();
    if (tid == 0) g.sq_out[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
  } else {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int row = m0 + wm * 64 + mt * 16 + (lane & 15);
      if (row >= g.M) continue;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int col = n0 + wn * 64 + nt * 16 + 4 * (lane >> 4);
        if (col >= g.N) continue;
        epilogue_store<EPI>(g, row, col, acc[mt][nt]);
      }
    }
  }
}

// =================================================================================================
// Skinny product: C[M <= 32][N] = alpha * A[M][K] W[N][K]^T + bias, fp32 out -- the conditioning path of the DiT forward pass: the adaLN
// modulation of every block at once ([32 x 1152] x [1152 x 195 840]: 446 MB of bf16 weight read once) and the two Linears of the timestep
// MLP.  On the 128 x 128 generic tile three quarters of the matrix work was padding and the two-deep staging kept 2.5 TB/s of weight
// rows in flight (179 us, on the critical stream in front of block 0: profiles/r04_stage_boundary.txt).  Here the rows of W are the MFMA's
// ROW operand, read straight from global memory as fragments (lane = row n0 + (lane & 15), 16 bytes of k-step ks at 8 (lane >> 4)): no
// staging, twelve 1 KB loads in flight per wave.  The activations (32 x K bf16, zero rows beyond M) sit in LDS once per workgroup, rows
// padded by 16 bytes (row stride = 4 banks mod 64: a 16-lane group's ds_read_b128 covers the 64 banks exactly).  A lane ends up with
// C[m = lane & 15 (+ 16)][n0 + 4 g .. + 3]: 16-byte stores.  Each wave walks `ntw` consecutive 16-row tiles of W.
// Round 6: EIGHT waves per workgroup over the same activation image (was four): a wave's loop is issue twelve 1 KB loads -> wait for all of
// them -> 24 MFMAs -> next group, i.e. bursts of 12 KB separated by a full memory latency; the activations' 74 KB of LDS allow two workgroups
// per CU either way, so the wave count per workgroup is what sets the bytes in flight per CU (96 -> 192 KB).  (A software-pipelined stream
// of groups in ONE wave was built first: hipcc drains vmcnt(0) at the loop header of a loop that carries loads in flight and the second
// group waited for the first.)  Same products in the same order per output element: bit-identical.
constexpr int SK_KB = 12, SK_NW = 8;
__global__ __launch_bounds__(SK_NW * 64) void k_gemm_skinny(GemmArgs g, int ntw) {
  extern __shared__ __attribute__((aligned(16))) __bf16 sk_a[];          // [32][K + 8]
  const int K = g.K, ldl = K + 8, kc = K >> 3;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int e = tid; e < 32 * kc; e += SK_NW * 64) {
    const int m = e / kc, c = e - m * kc;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m < g.M) v = *reinterpret_cast<const uint4*>(g.A + (size_t)m * g.lda + c * 8);
    *reinterpret_cast<uint4*>(sk_a + m * ldl + c * 8) = v;
  }
  __syncthreads();
  const int nks = K >> 5, ntiles = g.N >> 4;
  const __bf16* const a0p = sk_a + r * ldl + 8 * q;
  const __bf16* const a1p = sk_a + (16 + r) * ldl + 8 * q;
  for (int t = 0; t < ntw; ++t) {
    const int nt = (blockIdx.x * SK_NW + wave) * ntw + t;                  // wave-uniform
    if (nt >= ntiles) break;
    const __bf16* const wrow = g.B + (size_t)(nt * 16 + r) * g.ldb + 8 * q;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < nks; k0 += SK_KB) {
      bf16x8 wf[SK_KB];
#pragma unroll
      for (int u = 0; u < SK_KB; ++u)
        if (k0 + u < nks) wf[u] = *reinterpret_cast<const bf16x8*>(wrow + 32 * (k0 + u));
#pragma unroll
      for (int u = 0; u < SK_KB; ++u)
        if (k0 + u < nks) {
          const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(a0p + 32 * (k0 + u));
          const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(a1p + 32 * (k0 + u));
          acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], a0, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], a1, acc1, 0, 0, 0);
        }
    }
    const int col = nt * 16 + 4 * q;
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) b4 = as4(*reinterpret_cast<const float4*>(g.bias + col));
    if (r < g.M) *reinterpret_cast<f32x4*>(g.Cf + (size_t)r * g.ldcf + col) = acc0 * g.alpha + b4;
    if (16 + r < g.M) *reinterpret_cast<f32x4*>(g.Cf + (size_t)(16 + r) * g.ldcf + col) = acc1 * g.alpha + b4;
  }
}
inline bool skinny_ok(const GemmArgs& g) {
  return g.M <= 32 && g.N % 16 == 0 && g.K % 32 == 0 && g.K <= 2048 && g.ldb % 8 == 0 && g.lda % 8 == 0 && g.ldcf % 4 == 0 && g.kchunk == g.K &&
         !g.accumulate && (((uintptr_t)g.Cf | (uintptr_t)g.bias) & 15) == 0;
}
inline int launch_skinny(const GemmArgs& g, hipStream_t s) {
  const int ntiles = g.N / 16;
  int ntw = cdiv(ntiles, SK_NW * 512);                // one round of two workgroups per CU where the problem is that large
  if (ntw < 1) ntw = 1;
  const size_t lds = (size_t)32 * (g.K + 8) * sizeof(__bf16);
  // once per process, for the largest K this kernel takes (skinny_ok): the attribute call in front of EVERY launch left the stream idle for
  // ~60 us (profiles/r04_stage_boundary.txt's successor trace)
  static const int lds_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                32 * (2048 + 8) * (int)sizeof(__bf16)) == hipSuccess ? SFRON_OK : (int)hipGetLastError();
  if (lds_rc != SFRON_OK) return lds_rc;
  hipLaunchKernelGGL(k_gemm_skinny, dim3(cdiv(ntiles, SK_NW * ntw)), dim3(SK_NW * 64), lds, s, g, ntw);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFRON_OK : (int)e;
}

// Short contraction: C[M][N] = alpha * A[M][K <= 32] W[N][K]^T + bias + pos[(row % T)][col], fp32 out -- the patch embedding of the DiT
// forward pass (K = patch^2 * channels = 16: 37.7 MB of output from 0.3 MB of operands; 68 us on the generic tile, which stages 64-deep
// K-tiles of mostly padding).  ONE MFMA k-step per 16 x 16 output tile (k beyond K zero in both fragments -- the same products and the
// same accumulation as the generic kernel's zero-padded tiles: bit-identical), W as the row operand so that a lane holds four consecutive
// columns of one output row: the kernel is its 16-byte stores.  A wave = 16 rows x `cols_per_wave` columns.
__global__ __launch_bounds__(256) void k_gemm_shortk(GemmArgs g, int nt_per_wave) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wid = blockIdx.x * 4 + wave;
  const int ntiles = g.N >> 4, parts = (ntiles + nt_per_wave - 1) / nt_per_wave;
  const int mt = wid / parts, part = wid - mt * parts;
  if (mt * 16 >= g.M) return;
  bf16x8 z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.0f;
  const int row = mt * 16 + r;
  const bool kin = 8 * q < g.K;
  bf16x8 af = z;
  if (kin && row < g.M) af = *reinterpret_cast<const bf16x8*>(g.A + (size_t)row * g.lda + 8 * q);
  const float* const prow = g.pos ? g.pos + (size_t)(row % g.T) * g.N : nullptr;
  float* const crow = g.Cf + (size_t)row * g.ldcf;
  const int nt1 = min(ntiles, (part + 1) * nt_per_wave);
  for (int nt0 = part * nt_per_wave; nt0 < nt1; nt0 += 4) {             // four tiles per trip: their loads go out together
    bf16x8 wf[4];
    f32x4 add[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int nt = nt0 + u < nt1 ? nt0 + u : nt1 - 1;
      wf[u] = kin ? *reinterpret_cast<const bf16x8*>(g.B + (size_t)(nt * 16 + r) * g.ldb + 8 * q) : z;
      const int col = nt * 16 + 4 * q;
      add[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (g.bias) add[u] = as4(*reinterpret_cast<const float4*>(g.bias + col));
    }
    f32x4 pe[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int nt = nt0 + u < nt1 ? nt0 + u : nt1 - 1;
      pe[u] = (prow && row < g.M) ? as4(*reinterpret_cast<const float4*>(prow + nt * 16 + 4 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (nt0 + u >= nt1) break;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], af, acc, 0, 0, 0);
      // lane: C[row = mt * 16 + r][(nt0 + u) * 16 + 4 q + j]; the generic epilogue's order: (acc * alpha + bias) + pos
      const f32x4 v = (acc * g.alpha + add[u]) + pe[u];
      if (row < g.M) *reinterpret_cast<f32x4*>(crow + (nt0 + u) * 16 + 4 * q) = v;
    }
  }
}
inline bool shortk_ok(const GemmArgs& g) {
  return g.K <= 32 && g.K % 8 == 0 && g.N % 16 == 0 && g.lda % 8 == 0 && g.ldb % 8 == 0 && g.ldcf % 4 == 0 && g.kchunk == g.K && !g.accumulate &&
         (((uintptr_t)g.Cf | (uintptr_t)g.bias | (uintptr_t)g.pos) & 15) == 0;
}
inline int launch_shortk(const GemmArgs& g, hipStream_t s) {
  const int ntiles = g.N / 16, mtiles = cdiv(g.M, 16);
  int parts = 1;
  while (mtiles * parts < 2048 && parts * 2 <= ntiles / 4) parts *= 2;     // >= 2 048 waves where the output is that large
  const int per = cdiv(cdiv(ntiles, parts), 4) * 4;
  const int waves = mtiles * cdiv(ntiles, per);
  hipLaunchKernelGGL(k_gemm_shortk, dim3(cdiv(waves, 4)), dim3(256), 0, s, g, per);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFRON_OK : (int)e;
}

template <bool A_TR, bool B_TR, int EPI>
int launch(const GemmArgs& g, hipStream_t s) {
  const size_t lds = 4 * TILE_ELEMS * sizeof(__bf16);
  const int splits = (g.K + g.kchunk - 1) / g.kchunk;
  // FULL = true (unpredicated staging loads) measured 2.7x SLOWER on gfx950/ROCm 7.2: hipcc then sinks the loads next
  // to their ds_write and serialises load -> wait -> store -> compute; the predicated form keeps them a tile ahead.
  // FULL = true (unpredicated staging loads) measured 2.7x SLOWER on gfx950 / ROCm 7.2: without the predication
  // branches hipcc parks the staged tile in SCRATCH (scratch_store behind a vmcnt wait per load); kept off.
  const bool full = false;
  if (full) hipLaunchKernelGGL((k_gemm<A_TR, B_TR, EPI, true>), dim3(g.ntm * g.ntn * splits), dim3(NT), lds, s, g);
  else      hipLaunchKernelGGL((k_gemm<A_TR, B_TR, EPI, false>), dim3(g.ntm * g.ntn * splits), dim3(NT), lds, s, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFRON_OK : (int)e;
}


// =================================================================================================
// Fast path: full tiles only (M % BM == N % BN == K % 64 == 0).  LDS-DMA staging
// (global_load_lds_dwordx4: no VGPR round trip, no ds_write, swizzle applied on the per-lane SOURCE
// address), 2-deep LDS ring, one raw s_barrier per K-tile: tile t+1 streams in while tile t feeds
// the matrix cores.  WM x WN waves, each MT x NT MFMA tiles.
// =================================================================================================
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// transposed-read image [64 k-rows][COLS]: XOR applied to the 16-B chunk index of row r.  Derived from the
// ds_read_b64_tr_b16 bank rule (64 banks x 4 B, conflicts counted per 32-lane half = rows r..r+3 and r+8..r+11):
//   COLS 128/256/384 (row stride = 0 mod 256 B): 4-bit XOR ((r&3)<<2)|((r>>2)&3) inside each 16-chunk group
//   COLS 192 (row stride = 128 mod 256 B): 2-bit XOR on the chunk-pair index inside each 8-chunk group
//   COLS 144: the image row is padded to 160 columns (320 B: rows r and r+8 start on the same bank, rows r..r+3 are 16
//             banks apart) and rows with bit 3 set are shifted by two chunks: the 8 rows x 32 B of one half-wave read tile
//             the 64 banks exactly.  The two pad chunks of a row are never read; the DMA sends their lanes out of range.
template <int COLS> constexpr int tr_cols() { return COLS == 144 ? 160 : COLS; }   // physical columns of a transposed-read image row
template <int COLS> __device__ __forceinline__ int swz_chunk(int r, int ch) {
  if (COLS == 144) return ch + 2 * ((r >> 3) & 1);
  if (COLS == 192) return ch ^ ((((r >> 1) & 1) | (((r >> 3) & 1) << 1)) << 1);
  return (ch & ~15) | ((ch & 15) ^ swz_tr(r));
}

// Per-lane source offsets (elements, 32-bit) of the wave's LDS-DMA instructions for one operand tile; the
// per-K-tile advance is wave-uniform, so the loads use the saddr + voffset form (no 64-bit VGPR addresses).
// LDS image is lane-linear (1 KiB per wave-instruction); the swizzle lives in the SOURCE address.
template <int EXT, bool TR, int NW>
struct GldsPlan {
  static constexpr int CPR = TR ? tr_cols<EXT>() / 8 : 8;   // 16-B chunks per (physical) image row
  static constexpr int NINSTR = (TR ? 64 : EXT) * CPR / 64; // wave-instructions per tile
  static constexpr int PER_WAVE = (NINSTR + NW - 1) / NW;   // uneven splits: the waves without a last share issue a no-op
  static constexpr bool EVEN = NINSTR % NW == 0;
  int off[PER_WAVE];        // byte offsets (voffset of the buffer load)
  __device__ __forceinline__ void init(int ld, int d0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
      const int e = (wave + i * NW) * 64 + lane;            // linear chunk index inside the image
      const int row = e / CPR, p = e % CPR;
      if (!TR) off[i] = 2 * ((d0 + row) * ld + ((p ^ (row & 7)) << 3));
      else if (EXT == 144) {                               // physical chunk p holds logical chunk p - shift; pad chunks go out of range
        const int lc = p - 2 * ((row >> 3) & 1);
        off[i] = (lc >= 0 && lc < EXT / 8) ? 2 * (row * ld + d0 + (lc << 3)) : 0x7ffffff0;
      }
      else     off[i] = 2 * (row * ld + d0 + (swz_chunk<EXT>(row, p) << 3));
    }
  }
  __device__ __forceinline__ void issue_one(__amdgpu_buffer_rsrc_t rsrc, int soff_bytes, __bf16* img, int wave, int i) const {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t*)(img + (wave + i * NW) * 512), 16, off[i], soff_bytes, 0, 0);
  }
  // guarded form for uneven splits: a wave whose i-th share does not exist sends the request through a zero-length
  // descriptor (fetches nothing) towards a dummy 1 KB LDS region, so every wave issues the same number of vector-memory
  // operations per tile (the counted vmcnt waits stay uniform) without a branch
  __device__ __forceinline__ void issue_one_g(__amdgpu_buffer_rsrc_t rsrc, __amdgpu_buffer_rsrc_t rs_null, int soff_bytes, __bf16* img,
                                              __bf16* dummy, int wave, int i) const {
    if constexpr (EVEN) {
      issue_one(rsrc, soff_bytes, img, wave, i);
    } else {
      const bool ok = wave + i * NW < NINSTR;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ok ? rsrc : rs_null, (lptr_t*)(ok ? img + (wave + i * NW) * 512 : dummy), 16, off[i],
                                               soff_bytes, 0, 0);
    }
  }
  // buffer_load_dwordx4 ... offen lds: SRD (uniform) + per-lane voffset (loop invariant) + uniform soffset (K advance)
  __device__ __forceinline__ void issue(__amdgpu_buffer_rsrc_t rsrc, int soff_bytes, __bf16* img, int wave) const {
    static_assert(EVEN, "use issue_one_g for uneven splits");
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t*)(img + (wave + i * NW) * 512), 16, off[i], soff_bytes, 0, 0);
  }
};

template <int COLS>
__device__ __forceinline__ bf16x8 frag_tr_w(const __bf16* img, int col0, int kr0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const int ch = (col0 >> 3) + (p >> 1);
  const int r0 = kr0 + 8 * g + q;
  const int o0 = r0 * COLS + (swz_chunk<COLS>(r0, ch) << 3) + 4 * (p & 1);
  const int o1 = (r0 + 4) * COLS + (swz_chunk<COLS>(r0 + 4, ch) << 3) + 4 * (p & 1);
  bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + o0));
  bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + o1));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

}  // namespace  (kernel templates get external linkage: hipcc does not emit the kernel handle of an
   //             internal-linkage template instantiation that is only referenced through a launch)

template <int WM, int WN, int MT, int NT, bool A_TR, bool B_TR, int EPI, int DBG = 0>   // DBG: 1 = no MFMA, 2 = no DMA in the loop (timing ablations)
__global__ __launch_bounds__(WM * WN * 64) void k_gemm_fast(GemmArgs g) {
  constexpr int FBM = WM * MT * 16, FBN = WN * NT * 16, NW = WM * WN;
  constexpr int A_ELEMS = FBM * 64, B_ELEMS = FBN * 64;
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  __bf16* sA0 = smem;
  __bf16* sA1 = smem + A_ELEMS;
  __bf16* sB0 = smem + 2 * A_ELEMS;
  __bf16* sB1 = smem + 2 * A_ELEMS + B_ELEMS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nblk = gridDim.x;
  int id;
  {
    const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, x = b & 7, y = b >> 3;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  // grouped order: ids sweep GROUP_M tile-rows x all tile-columns, so the ~n/8 consecutive tiles one XCD owns
  // form a 2-D block whose A row-panels and B column-panels are both re-used out of that XCD's L2
  const int per_group = g.group_m * g.ntn;
  const int first_m = (id / per_group) * g.group_m;
  const int gsz = min(g.ntm - first_m, g.group_m);
  const int tm = first_m + (id % per_group) % gsz, tn = (id % per_group) / gsz;
  const int m0 = tm * FBM, n0 = tn * FBN;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  GldsPlan<FBM, A_TR, NW> planA;
  GldsPlan<FBN, B_TR, NW> planB;
  planA.init(g.lda, m0, wave, lane);
  planB.init(g.ldb, n0, wave, lane);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, 0x7fffffff, 0x00020000);
  auto stage = [&](__bf16* iA, __bf16* iB, int k0) {
    if (DBG == 2 && k0 > 0) return;
    planA.issue(rsA, 2 * (A_TR ? k0 * g.lda : k0), iA, wave);
    planB.issue(rsB, 2 * (B_TR ? k0 * g.ldb : k0), iB, wave);
  };
  auto load_a = [&](const __bf16* iA, int mt, int ks) -> bf16x8 {
    if (!A_TR) return frag_direct(iA, wm * MT * 16 + mt * 16 + (lane & 15), ks * 4 + (lane >> 4));
    return frag_tr_w<FBM>(iA, wm * MT * 16 + mt * 16, ks * 32, lane);
  };
  auto load_b = [&](const __bf16* iB, int nt, int ks) -> bf16x8 {
    if (!B_TR) return frag_direct(iB, wn * NT * 16 + nt * 16 + (lane & 15), ks * 4 + (lane >> 4));
    return frag_tr_w<FBN>(iB, wn * NT * 16 + nt * 16, ks * 32, lane);
  };
  // One K-tile of work, software pipelined by hand:
  //  * the operand with fewer fragments per k-step is HELD (loaded one k-step ahead), the other is STREAMED one
  //    fragment ahead of the MFMAs that consume it, so an LDS read is always in flight under the matrix pipe;
  //  * the LDS-DMA instructions of the NEXT tile are issued one per MFMA group instead of as a burst at the
  //    start of the tile (smoother L2 -> LDS traffic).
  constexpr int HELD = (NT <= MT) ? NT : MT, STRM = (NT <= MT) ? MT : NT;
  constexpr bool HOLD_B = NT <= MT;
  auto load_held = [&](const __bf16* iA, const __bf16* iB, int i, int ks) { return HOLD_B ? load_b(iB, i, ks) : load_a(iA, i, ks); };
  auto load_strm = [&](const __bf16* iA, const __bf16* iB, int i, int ks) { return HOLD_B ? load_a(iA, i, ks) : load_b(iB, i, ks); };
  constexpr int NDMA_A = decltype(planA)::PER_WAVE, NDMA = NDMA_A + decltype(planB)::PER_WAVE;
  constexpr bool PIPE = !A_TR && !B_TR;     // hand pipelining pays for row-fragment (ds_read_b128) operands only
  auto compute = [&](const __bf16* iA, const __bf16* iB, bool prefetch, __bf16* nA, __bf16* nB, int k_next) {
    if (DBG == 1) { if (prefetch) stage(nA, nB, k_next); return; }
    if constexpr (!PIPE) {
      if (prefetch) stage(nA, nB, k_next);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 hd[HELD];
#pragma unroll
        for (int i = 0; i < HELD; ++i) hd[i] = load_held(iA, iB, i, ks);
#pragma unroll
        for (int j = 0; j < STRM; ++j) {
          const bf16x8 f = load_strm(iA, iB, j, ks);
#pragma unroll
          for (int i = 0; i < HELD; ++i) {
            if (HOLD_B) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hd[i], f, acc[j][i], 0, 0, 0);
            else        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f, hd[i], acc[i][j], 0, 0, 0);
          }
        }
      }
      return;
    }
    const int soffA = 2 * (A_TR ? k_next * g.lda : k_next), soffB = 2 * (B_TR ? k_next * g.ldb : k_next);
    bf16x8 held[2][HELD];
#pragma unroll
    for (int i = 0; i < HELD; ++i) held[0][i] = load_held(iA, iB, i, 0);
    bf16x8 cur = load_strm(iA, iB, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int j = 0; j < STRM; ++j) {
        const int step = ks * STRM + j;
        // next streamed fragment (possibly of the next k-step) goes out before this group's MFMAs
        bf16x8 nxt = cur;
        if (j + 1 < STRM) nxt = load_strm(iA, iB, j + 1, ks);
        else if (ks == 0) nxt = load_strm(iA, iB, 0, 1);
        if (ks == 0 && j == STRM - 1) {
#pragma unroll
          for (int i = 0; i < HELD; ++i) held[1][i] = load_held(iA, iB, i, 1);
        }
        if (DBG != 2 && prefetch && step < NDMA) {
          if (step < NDMA_A) planA.issue_one(rsA, soffA, nA, wave, step);
          else               planB.issue_one(rsB, soffB, nB, wave, step - NDMA_A);
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the reads above ahead of this group's MFMAs (hipcc sinks them otherwise)
#pragma unroll
        for (int i = 0; i < HELD; ++i) {
          if (HOLD_B) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(held[ks][i], cur, acc[j][i], 0, 0, 0);
          else        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur, held[ks][i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
      }
    }
  };

  const int nk = g.K / BK;
  stage(sA0, sB0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int kt = 0;
  for (; kt + 2 <= nk - 1; kt += 2) {
    compute(sA0, sB0, true, sA1, sB1, (kt + 1) * BK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    compute(sA1, sB1, true, sA0, sB0, (kt + 2) * BK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (kt + 1 <= nk - 1) {          // two tiles left: kt (in buffer 0) and kt+1
    compute(sA0, sB0, true, sA1, sB1, (kt + 1) * BK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    compute(sA1, sB1, false, sA0, sB0, 0);
  } else {                          // one tile left, in buffer 0
    compute(sA0, sB0, false, sA1, sB1, 0);
  }

  // epilogues that READ a tensor (pre-activation / residual stream) issue all their loads first, so the 24..36
  // dependent load->math->store chains of one lane overlap instead of serialising on memory latency
  const int row_b = m0 + wm * MT * 16 + (lane & 15), col_b = n0 + wn * NT * 16 + 4 * (lane >> 4);
  if constexpr (EPI == EPI_DGELU) {
    bf16x4 hx[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        hx[mt][nt] = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)(row_b + mt * 16) * g.ldaux + col_b + nt * 16);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 v = acc[mt][nt] * g.alpha;
        const bf16x4 h = hx[mt][nt];
        const bf16x4 o = f2bf4(v * gelu_tanh_grad4(bf2f4(h)));
        *reinterpret_cast<bf16x4*>(g.Cb + (size_t)(row_b + mt * 16) * g.ldcb + col_b + nt * 16) = o;
      }
  } else if constexpr (EPI == EPI_GATE_RES) {
    float4 gt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)      // all rows of a tile belong to one sample when tokens % tile rows == 0; else per row
      gt[nt] = *reinterpret_cast<const float4*>(g.gate + (size_t)(row_b / g.T) * g.ldgate + col_b + nt * 16);
    const bool uniform_sample = (g.T % FBM) == 0;
    // the bias words and (token counts below the tile's rows: DiT-B/4 has 64) the row's gate words go out with the residual loads, before
    // the first use: inside the column loop each was a load-use pair -- MT x NT exposed latencies per tile (44 us for a 96-tile launch)
    float4 bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = g.bias ? *reinterpret_cast<const float4*>(g.bias + col_b + nt * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = row_b + mt * 16;
      float4 xr[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) xr[nt] = *reinterpret_cast<const float4*>(g.resid + (size_t)row * g.ldcf + col_b + nt * 16);
      if (!uniform_sample) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) gt[nt] = *reinterpret_cast<const float4*>(g.gate + (size_t)(row / g.T) * g.ldgate + col_b + nt * 16);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = col_b + nt * 16;
        f32x4 v = acc[mt][nt] * g.alpha;
        if (g.bias) { const float4 b = bv[nt]; v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
        bf16x4 a = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        nt_store(reinterpret_cast<bf16x4*>(g.aux + (size_t)row * g.ldaux + col), a, g.nt_out & 1);
        const float4 gg = gt[nt];
        float4 x = xr[nt];
        x.x += gg.x * v[0]; x.y += gg.y * v[1]; x.z += gg.z * v[2]; x.w += gg.w * v[3];
        *reinterpret_cast<float4*>(g.Cf + (size_t)row * g.ldcf + col) = x;
      }
    }
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        epilogue_store<EPI>(g, row_b + mt * 16, col_b + nt * 16, acc[mt][nt]);
  }
}


// explicit instantiations (hipcc / ROCm 7.2 does not reliably instantiate a kernel template that is only reached
// through a nested host template; without these the host stubs of some tiles are missing at dlopen time)
#define SFRON_INST_TILE(WM, WN, MT, NT)                                              \
  template __global__ void k_gemm_fast<WM, WN, MT, NT, false, false, 0>(GemmArgs);  \
  template __global__ void k_gemm_fast<WM, WN, MT, NT, false, false, 1>(GemmArgs);  \
  template __global__ void k_gemm_fast<WM, WN, MT, NT, false, false, 2>(GemmArgs);  \
  template __global__ void k_gemm_fast<WM, WN, MT, NT, false, false, 3>(GemmArgs);  \
  template __global__ void k_gemm_fast<WM, WN, MT, NT, false, false, 5>(GemmArgs);  \
  template __global__ void k_gemm_fast<WM, WN, MT, NT, false, true, 0>(GemmArgs);   \
  template __global__ void k_gemm_fast<WM, WN, MT, NT, false, true, 1>(GemmArgs);   \
  template __global__ void k_gemm_fast<WM, WN, MT, NT, false, true, 4>(GemmArgs);   \
  template __global__ void k_gemm_fast<WM, WN, MT, NT, true, true, 0>(GemmArgs);    \
  template __global__ void k_gemm_fast<WM, WN, MT, NT, true, true, 1>(GemmArgs);
SFRON_INST_TILE(2, 2, 4, 4)
SFRON_INST_TILE(4, 2, 4, 6)
SFRON_INST_TILE(2, 4, 8, 4)
SFRON_INST_TILE(4, 2, 6, 6)
SFRON_INST_TILE(4, 2, 3, 6)
SFRON_INST_TILE(2, 2, 8, 6)
SFRON_INST_TILE(2, 2, 6, 6)
#undef SFRON_INST_TILE
template __global__ void k_gemm_fast<4, 2, 4, 6, false, false, 0, 1>(GemmArgs);
template __global__ void k_gemm_fast<4, 2, 4, 6, false, false, 0, 2>(GemmArgs);
template __global__ void k_gemm_fast<4, 2, 4, 6, true, true, 1, 1>(GemmArgs);
template __global__ void k_gemm_fast<4, 2, 4, 6, true, true, 1, 2>(GemmArgs);

namespace {

template <int WM, int WN, int MT, int NT, bool A_TR, bool B_TR, int EPI, int DBG = 0>
int launch_fast(GemmArgs g, hipStream_t s) {
  constexpr int FBM = WM * MT * 16, FBN = WN * NT * 16;
  g.ntm = g.M / FBM; g.ntn = g.N / FBN;
  {
    // an XCD owns ~tiles/8 consecutive ids: make that run a near-square (in bytes) block of gm x gn tiles
    const double per_xcd = (double)g.ntm * g.ntn / 8.0;
    int gm = 1;
    while (gm * 2 <= g.ntm && (double)(gm * 2) * (gm * 2) * FBM <= per_xcd * FBN * 1.5) gm *= 2;
    g.group_m = gm;
  }
  const size_t lds = 2 * (FBM + FBN) * 64 * sizeof(__bf16);
  if (lds > 65536) {
    static std::atomic<uint64_t> done{0};      // per instantiation, one bit per device
    if (need_attr(done)) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_fast<WM, WN, MT, NT, A_TR, B_TR, EPI, DBG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return (int)hipGetLastError();
    }
  }
  hipLaunchKernelGGL((k_gemm_fast<WM, WN, MT, NT, A_TR, B_TR, EPI, DBG>), dim3(g.ntm * g.ntn), dim3(WM * WN * 64), lds, s, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFRON_OK : (int)e;
}

}  // namespace

// =================================================================================================
// Pipelined path ("k_gemm_pipe"): same tiles / LDS images / DMA as k_gemm_fast, but the fragment reads are
// software pipelined BY HAND one k-step ahead of the MFMAs that consume them, and the transposed reads are
// issued through inline asm.  Reason (measured): while an LDS-DMA is in flight hipcc puts `s_waitcnt vmcnt(0)`
// in front of every __builtin ds_read_tr16_b64 (it cannot prove the DMA does not alias the read), which
// serialises load and compute for the dgrad / wgrad layouts (wgrad fc1: 181 us = 112 us DMA-only + 125 us
// compute-only with almost no overlap).  An asm read is invisible to that bookkeeping; its completion is waited
// for explicitly (`s_waitcnt lgkmcnt(0)` + sched_barrier, cdna_hip_programming.md section 5.7 rule 18).
// =================================================================================================
__device__ __forceinline__ unsigned lds_addr(const __bf16* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(const void*)p;
}
__device__ __forceinline__ bf16x4 asm_read_tr(unsigned addr) {
  bf16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
__device__ __forceinline__ bf16x8 asm_read_b128(unsigned addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
template <int OFF>
__device__ __forceinline__ bf16x4 asm_read_tr_off(unsigned addr) {
  bf16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ bf16x8 asm_read_b128_off(unsigned addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
// Where the memory operations of a segment sit among its NM MFMAs.  They used to be spread evenly up to the LAST MFMA, so the
// `s_waitcnt lgkmcnt(0)` that closes the segment exposed one full LDS latency per k-step on BOTH waves of a SIMD at once (the two
// run the same code from the same barrier: neither has MFMAs left to cover the other).  Now the operations are packed into the
// first NM - SFRON_RD_TAIL MFMA gaps (at least ceil(NOPS / 2) gaps: two per gap), which leaves RD_TAIL MFMAs = RD_TAIL x 32 cycles at
// two waves per SIMD between the last read's issue and the wait.  op_cut(i) = operations issued before MFMA i's gap closes.
#ifndef SFRON_RD_TAIL
#define SFRON_RD_TAIL 4
#endif
constexpr int op_span(int nops, int nm) {
  int span = nm - SFRON_RD_TAIL;
  const int min_span = (nops + 1) / 2;
  if (span < min_span) span = min_span;
  if (span > nm || SFRON_RD_TAIL <= 0) span = nm;
  return span < 1 ? 1 : span;
}
constexpr int op_cut(int i, int nops, int nm) { return i >= op_span(nops, nm) ? nops : i * nops / op_span(nops, nm); }
__device__ __forceinline__ void lds_reads_done() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

// SCHED 1 ("interleaved"): the same pipeline, but the LDS-DMA issues and fragment reads of a segment are pinned BETWEEN
// the MFMAs they overlap with (one MFMA, then 1-2 memory instructions, sched_barrier), so their issue slots hide under
// the 16-cycle MFMA occupancy instead of forming an MFMA-free bubble after every barrier; buffer select and k-step are
// immediate ds offsets (K loop unrolled by two).
// BSUM (weight-gradient layout, three-slot schedule): the kernel also forms bsum[m] = sum_k op(A)[m][k] -- the bias gradient of
// the same Linear -- with MT extra MFMAs against a ones fragment (waves wn == 0 only): the dY tile is already in LDS /
// registers, so the separate column-sum kernel (a second 75 MB read of dY per launch) disappears.  The ntn workgroups of a
// tile row all stream the same dY tiles; they SHARE the extra work: workgroup (tm, tn) sums the k-tiles kt = tn mod ntn and
// writes partial sums bsum[tn][m] (the host adds the ntn partial rows in a fixed order).  (Giving all of it to the tn = 0
// column made those 1/ntn of the workgroups 17 % longer, and with one round of tiles the launch ends with its slowest
// workgroup: measured +20 us per launch.)  fp32 accumulation of bf16 values in a fixed order: deterministic.
// NL > 0 (three-slot schedule only): NL extra LOADER waves issue every LDS-DMA piece of the ring and nothing else; the WM * WN waves
// that multiply never issue one.  MI355X_MICROARCH.md prices a 1-KiB piece at 60-185 cycles of the issuing wave, 6-7 pieces per wave
// and K-step in the shared form, against 576 cycles of MFMA per wave and K-step; tools/probes/persist_gemm_probe.hip measured the
// split 13-22 % faster on the plain loop.  One s_barrier per K-tile for all waves: a loader arrives once its share of tile kt has
// landed (counted vmcnt), a consumer once the fragments of tile kt - 1 are in its registers; behind it the loaders refill the slot of
// tile kt - 1 with tile kt + 2.
template <int WM, int WN, int MT, int NT, bool A_TR, bool B_TR, int EPI, int SCHED = 0, int PRO = 2, bool BSUM = false, int NL = 0>
__global__ __launch_bounds__((WM * WN + NL) * 64) void k_gemm_pipe(GemmArgs g) {
  static_assert(!BSUM || (A_TR && B_TR && EPI == EPI_F32 && SCHED == 2), "row sums ride on the three-slot weight-gradient kernel");
  static_assert(NL == 0 || SCHED == 2, "loader waves feed the three-slot ring");
  static_assert(NL == 0 || !BSUM, "no row sums in the loader form");
  constexpr bool NL_SIMPLE = false;     // consumers of the loader form: false = the hand-interleaved body minus its DMA operations; true = a compiler-
                                        // scheduled loop (measured 5-10 % SLOWER than the default form on every shape: hipcc keeps two B fragments in flight)
  constexpr int FBM = WM * MT * 16, FBN = WN * NT * 16, NW = WM * WN;
  constexpr int FBM_P = A_TR ? tr_cols<FBM>() : FBM, FBN_P = B_TR ? tr_cols<FBN>() : FBN;   // physical row length of a transposed-read image
  constexpr int A_ELEMS = FBM_P * 64, B_ELEMS = FBN_P * 64;
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  constexpr int NS = SCHED == 2 ? 3 : 2;                  // LDS slots (64-deep tiles); SCHED 2 = interleaved schedule, 3 slots
  // ds offset field is 16 bits: the third slot of a 256-row direct A image starts at 65,536 -> second base register
  static_assert(NS == 2 || ((A_TR ? 2 * 2 * A_ELEMS + 64 * FBM_P < 65536 : 2 * A_ELEMS < 65536) && 2 * 2 * B_ELEMS + 64 * FBN_P < 65536),
                "ds offset field");
  auto sAp = [&](int b) { return smem + b * A_ELEMS; };
  auto sBp = [&](int b) { return smem + NS * A_ELEMS + b * B_ELEMS; };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SFRON_DEBUG_KNOBS
  const long long clk_e = g.dbg_clk ? (long long)__builtin_amdgcn_s_memrealtime() : 0;      // workgroup entry (constant 100 MHz counter)
#endif
  const int wm = wave / WN, wn = wave % WN;
  const int nblk = gridDim.x;
  int id;
  {
    const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, x = b & 7, y = b >> 3;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  const int per_group = g.group_m * g.ntn;
  const int first_m = (id / per_group) * g.group_m;
  const int gsz = min(g.ntm - first_m, g.group_m);
  int tm = first_m + (id % per_group) % gsz, tn = (id % per_group) / gsz;
#ifdef SFRON_DEBUG_KNOBS
  // timing experiment (results are wrong): g.group_m < 0 sends every workgroup to tile (0, 0) / every tile-row to column 0 / every
  // tile-column to row 0 -- all operand lines are L2 hits after the first touch: is a K-step bound by the L2-miss latency?
  if (g.dbg_same & 1) tm = 0;
  if (g.dbg_same & 2) tn = 0;
#endif
  const int m0 = tm * FBM, n0 = tn * FBN;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (SCHED >= 1) {
    // pin the zero-initialisation HERE: hipcc otherwise sinks each v_mov next to the first (asm) MFMA that reads the
    // accumulator, and the VALU-write -> MFMA-SrcC-read wait states it would add for a builtin MFMA are then missing
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) { f32x4& c = acc[i][j]; asm volatile("" : "+v"(c)); }
  }
  f32x4 bacc[BSUM ? MT : 1];
  bf16x8 ones;
  const bool do_bs = BSUM && g.bsum != nullptr && wn == 0 && blockIdx.y == 0;     // wave-uniform
  int bs_c = g.ntn - 1;                      // (index of the previous k-tile) mod ntn; body() advances it
  if constexpr (BSUM) {
#pragma unroll
    for (int i = 0; i < MT; ++i) { bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+v"(bacc[i])); }
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    asm volatile("" : "+v"(ones));
  }

  // forward layouts (the only ones that carry a bias): this lane's bias columns, one float4 per n-tile, fetched before the
  // main loop so that their latency is not on the epilogue's path; the transposed layouts keep the registers
  constexpr bool PRE_BIAS = !A_TR && !B_TR && NL == 0;      // (the loader form has 168 registers per wave: its epilogue fetches the bias)
  float4 bias_v[PRE_BIAS ? NT : 1];
  if constexpr (PRE_BIAS) {
    const int cb = n0 + wn * NT * 16 + 4 * (lane >> 4);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      bias_v[nt] = g.bias ? *reinterpret_cast<const float4*>(g.bias + cb + nt * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  constexpr int NWD = NL > 0 ? NL : NW;                        // waves that issue LDS-DMA
  const int dwave = NL > 0 ? wave - NW : wave;                 // (consumers of the NL form never use their plan)
  GldsPlan<FBM, A_TR, NWD> planA;
  GldsPlan<FBN, B_TR, NWD> planB;
  planA.init(g.lda, m0, dwave, lane);
  planB.init(g.ldb, n0, dwave, lane);
  // num_records = the operand's exact extent: loads past it (the "next tile" request of the last iteration) fetch nothing
  const int bytesA = 2 * (((A_TR ? g.K : g.M) - 1) * g.lda + (A_TR ? g.M : g.K));
  const int bytesB = 2 * (((B_TR ? g.K : g.N) - 1) * g.ldb + (B_TR ? g.N : g.K));
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, SCHED >= 1 ? bytesA : 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, SCHED >= 1 ? bytesB : 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsNull = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, 0, 0x00020000);
  __bf16* const dma_dummy = smem + NS * (A_ELEMS + B_ELEMS);     // 1 KB behind the slots (allocated for uneven plans only)
  auto stage = [&](int buf, int k0) {
#pragma unroll
    for (int i = 0; i < GldsPlan<FBM, A_TR, NWD>::PER_WAVE; ++i)
      planA.issue_one_g(rsA, rsNull, 2 * (A_TR ? k0 * g.lda : k0), sAp(buf), dma_dummy, dwave, i);
#pragma unroll
    for (int i = 0; i < GldsPlan<FBN, B_TR, NWD>::PER_WAVE; ++i)
      planB.issue_one_g(rsB, rsNull, 2 * (B_TR ? k0 * g.ldb : k0), sBp(buf), dma_dummy, dwave, i);
  };
  if constexpr (NL > 0) {
    if (wave >= NW) {                                          // ---- loader wave
      const int kb = blockIdx.y * g.kchunk;
      const int nkl = (min(g.K, kb + g.kchunk) - kb) / BK;     // >= 2 (launcher)
      constexpr int NPT = GldsPlan<FBM, A_TR, NWD>::PER_WAVE + GldsPlan<FBN, B_TR, NWD>::PER_WAVE;   // requests per tile and wave, no-ops included
      stage(0, kb);
      stage(1, kb + BK);
      int slot = 2;
      for (int kt = 0; kt < nkl; ++kt) {
        if (kt + 1 < nkl) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkl) stage(slot, kb + (kt + 2) * BK);
        slot = slot == 2 ? 0 : slot + 1;
      }
      if constexpr (EPI == EPI_DGELU || EPI == EPI_DGELUQ) {    // the consumers' column-partial reduction meets at two more barriers
        if (g.colpart) { __syncthreads(); __syncthreads(); }
      }
      if constexpr (EPI == EPI_F32 && A_TR && B_TR) {          // ... and so does the masked sum of squares of a weight gradient
        if (g.sq_out) { __syncthreads(); __syncthreads(); }
      }
      return;
    }
  }

  // per-lane LDS byte addresses of the fragments of buffer 0, k-step 0 (other buffer / k-step: + constant)
  unsigned adA[MT][A_TR ? 2 : 1], adB[NT][B_TR ? 2 : 1];
  {
    const int gq = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (!A_TR) {
        const int row = wm * MT * 16 + mt * 16 + i;
        adA[mt][0] = lds_addr(sAp(0)) + 2 * (row * 64 + ((gq ^ (row & 7)) << 3));
      } else {
        const int ch = ((wm * MT * 16 + mt * 16) >> 3) + (pp >> 1), r0 = 8 * gq + q;
        adA[mt][0] = lds_addr(sAp(0)) + 2 * (r0 * FBM_P + (swz_chunk<FBM>(r0, ch) << 3) + 4 * (pp & 1));
        adA[mt][1] = lds_addr(sAp(0)) + 2 * ((r0 + 4) * FBM_P + (swz_chunk<FBM>(r0 + 4, ch) << 3) + 4 * (pp & 1));
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (!B_TR) {
        const int row = wn * NT * 16 + nt * 16 + i;
        adB[nt][0] = lds_addr(sBp(0)) + 2 * (row * 64 + ((gq ^ (row & 7)) << 3));
      } else {
        const int ch = ((wn * NT * 16 + nt * 16) >> 3) + (pp >> 1), r0 = 8 * gq + q;
        adB[nt][0] = lds_addr(sBp(0)) + 2 * (r0 * FBN_P + (swz_chunk<FBN>(r0, ch) << 3) + 4 * (pp & 1));
        adB[nt][1] = lds_addr(sBp(0)) + 2 * ((r0 + 4) * FBN_P + (swz_chunk<FBN>(r0 + 4, ch) << 3) + 4 * (pp & 1));
      }
    }
  }
  // k-step 1 of a direct image flips chunk bit 2 (XOR 4 chunks = 64 B); of a transposed image it is 32 rows further
  struct Frags { bf16x8 a[MT], b[NT]; };
  auto read_frags = [&](Frags& f, int buf, int ks) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (!A_TR) {
        f.a[mt] = asm_read_b128((adA[mt][0] ^ (ks ? 64u : 0u)) + buf * (2 * A_ELEMS));
      } else {
        const unsigned o = buf * (2 * A_ELEMS) + ks * (2 * 32 * FBM_P);
        const bf16x4 lo = asm_read_tr(adA[mt][0] + o), hi = asm_read_tr(adA[mt][1] + o);
        f.a[mt] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (!B_TR) {
        f.b[nt] = asm_read_b128((adB[nt][0] ^ (ks ? 64u : 0u)) + buf * (2 * B_ELEMS));
      } else {
        const unsigned o = buf * (2 * B_ELEMS) + ks * (2 * 32 * FBN_P);
        const bf16x4 lo = asm_read_tr(adB[nt][0] + o), hi = asm_read_tr(adB[nt][1] + o);
        f.b[nt] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    }
  };
  auto mfmas = [&](const Frags& f) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b[nt], f.a[mt], acc[mt][nt], 0, 0, 0);
  };

  // split-K: blockIdx.y owns k in [kbeg, kbeg + kchunk) and writes its own fp32 slab (summed later in a fixed order)
  const int kbeg = blockIdx.y * g.kchunk;
  const int nk = (min(g.K, kbeg + g.kchunk) - kbeg) / BK;
  if constexpr (EPI == EPI_F32) g.Cf += (size_t)blockIdx.y * g.split_stride;
  Frags f0, f1;
  if constexpr (NL == 0) {
    stage(0, kbeg);
    if constexpr (NS == 3) stage(1, kbeg + BK);            // three slots: two tiles in flight (nk >= 2 guaranteed by the launcher)
  }
  if constexpr (NL > 0 && NL_SIMPLE) {
    // ---- consumer of the loader form, compiler-scheduled: a consumer has no vector-memory operation in flight, so plain LDS loads
    // carry no vmcnt drain (the reason the other schedules read through inline asm) and hipcc places the waits; the two consumers
    // of a SIMD hide each other's LDS latency.  Same products in the same order as every other schedule.
    const int gq = lane >> 4, li = lane & 15, q4 = li >> 2, pp = li & 3;
    auto ld_frag = [&](const __bf16* img, auto trc, auto extc, int blk0, int ks) -> bf16x8 {
      constexpr bool TR = decltype(trc)::value;
      constexpr int EXT = decltype(extc)::value;
      if constexpr (!TR) {
        const int row = blk0 + li;
        return *reinterpret_cast<const bf16x8*>(img + off_direct(row, ks * 4 + gq));
      } else {
        constexpr int P = tr_cols<EXT>();
        const int ch = (blk0 >> 3) + (pp >> 1), r0 = 8 * gq + q4;
        const int o0 = r0 * P + (swz_chunk<EXT>(r0, ch) << 3) + 4 * (pp & 1) + ks * 32 * P;
        const int o1 = (r0 + 4) * P + (swz_chunk<EXT>(r0 + 4, ch) << 3) + 4 * (pp & 1) + ks * 32 * P;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + o0));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(img + o1));
        return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    };
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_barrier();                          // tile kt is in LDS; this wave's reads of tile kt - 1 fed MFMAs already issued
      const __bf16* iA = sAp(buf);
      const __bf16* iB = sBp(buf);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fa[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          fa[mt] = ld_frag(iA, std::integral_constant<bool, A_TR>{}, std::integral_constant<int, FBM>{}, wm * MT * 16 + mt * 16, ks);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const bf16x8 fb = ld_frag(iB, std::integral_constant<bool, B_TR>{}, std::integral_constant<int, FBN>{}, wn * NT * 16 + nt * 16, ks);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[mt], acc[mt][nt], 0, 0, 0);
        }
      }
      buf = buf == 2 ? 0 : buf + 1;
    }
  } else if constexpr (SCHED >= 1) {
    constexpr int NM = MT * NT;
    constexpr int NDA = NL > 0 ? 0 : GldsPlan<FBM, A_TR, NW>::PER_WAVE, NDB = NL > 0 ? 0 : GldsPlan<FBN, B_TR, NW>::PER_WAVE;
    constexpr int NRA = MT * (A_TR ? 2 : 1), NRB = NT * (B_TR ? 2 : 1);
    unsigned adA1[A_TR ? 1 : MT], adB1[B_TR ? 1 : NT];         // direct images: k-step 1 flips chunk bit 2
    if constexpr (!A_TR) static_for<MT>([&](auto i) { adA1[i] = adA[i][0] ^ 64u; });
    constexpr bool A_HI = !A_TR && (NS - 1) * 2 * A_ELEMS > 65535;          // last slot beyond the offset field
    unsigned adAh[A_HI ? MT : 1], adA1h[A_HI ? MT : 1];
    if constexpr (A_HI) static_for<MT>([&](auto i) { adAh[i] = adA[i][0] + 65536u; adA1h[i] = adA1[i] + 65536u; });
    if constexpr (!B_TR) static_for<NT>([&](auto i) { adB1[i] = adB[i][0] ^ 64u; });
    // (Tried and removed: an L2 prefetch stream, one dword per 128-B line per lane three tiles ahead of the LDS-DMA, for
    // operands that are cold in L2 / Infinity Cache.  64 distinct lines per wave-instruction cost the texture path as
    // much as eight 16-B-per-lane DMA instructions: hot GEMMs lost 12 %, cold ones 5 %, the step 6 %.)
    auto read = [&](auto bufc, auto ksc, Frags& f, auto rc) {
      constexpr int BUF = decltype(bufc)::value, KS = decltype(ksc)::value, r = decltype(rc)::value;
      if constexpr (r < NRA) {
        if constexpr (!A_TR) {
          // direct image: the 16-row blocks of a wave are 2,048 B apart and share the swizzle term ((row & 7) = (lane & 7)): ONE base
          // register per k-step + immediates (was: one per block -- 2 (MT + NT) registers the loader form does not have)
          if constexpr (BUF * 2 * A_ELEMS > 65535) f.a[r] = asm_read_b128_off<BUF * 2 * A_ELEMS - 65536 + r * 2048>(KS ? adA1h[0] : adAh[0]);
          else f.a[r] = asm_read_b128_off<BUF * 2 * A_ELEMS + r * 2048>(KS ? adA1[0] : adA[0][0]);
        } else {
          constexpr int mt = r >> 1, h = r & 1;
          const bf16x4 t = asm_read_tr_off<BUF * 2 * A_ELEMS + KS * 64 * FBM_P>(adA[mt][h]);
          f.a[mt][4 * h] = t[0]; f.a[mt][4 * h + 1] = t[1]; f.a[mt][4 * h + 2] = t[2]; f.a[mt][4 * h + 3] = t[3];
        }
      } else {
        constexpr int q = r - NRA;
        if constexpr (!B_TR) {
          f.b[q] = asm_read_b128_off<BUF * 2 * B_ELEMS + q * 2048>(KS ? adB1[0] : adB[0][0]);
        } else {
          constexpr int nt = q >> 1, h = q & 1;
          // 144-column image: the chunk shift is additive (swz_chunk), so n-tile nt is 32 B behind n-tile 0 -- two base registers, not 2 NT
          constexpr bool AFF = FBN == 144 && WN == 1;
          const bf16x4 t = asm_read_tr_off<BUF * 2 * B_ELEMS + KS * 64 * FBN_P + (AFF ? nt * 32 : 0)>(adB[AFF ? 0 : nt][h]);
          f.b[nt][4 * h] = t[0]; f.b[nt][4 * h + 1] = t[1]; f.b[nt][4 * h + 2] = t[2]; f.b[nt][4 * h + 3] = t[3];
        }
      }
    };
    auto mfma1 = [&](const Frags& f, auto ic) {
      // the operand with MORE fragments is the outer index: its registers die two (MT) MFMAs apart, so the reads of the next k-step --
      // which follow the same order -- can take them over one by one (front-loaded reads would otherwise need both fragment sets whole)
#ifdef SFRON_MFMA_MT_MAJOR      // (A-B builds: the round-3 order)
      constexpr int mt = decltype(ic)::value / NT, nt = decltype(ic)::value % NT;
#else
      constexpr int mt = NT >= MT ? decltype(ic)::value % MT : decltype(ic)::value / NT;
      constexpr int nt = NT >= MT ? decltype(ic)::value / MT : decltype(ic)::value % NT;
#endif
      // asm, accumulating in place: through the builtin hipcc renames every accumulator here (vdst != srcC) and spills ~150 VGPRs
      f32x4& c = acc[mt][nt];
      const bf16x8 fb = f.b[nt], fa = f.a[mt];
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(fb), "v"(fa));
    };
    // D[i][j] = sum_k 1 * A^T[k][j]: every accumulator row holds the row sums of this k-step's A fragment
    auto bsum_step = [&](const Frags& f, int tile_mod) {
      if constexpr (BSUM) {
        if (do_bs && tile_mod == tn) {
          static_for<MT>([&](auto ic) {
            constexpr int mt = decltype(ic)::value;
            f32x4& c = bacc[mt];
            const bf16x8 fa = f.a[mt], o1 = ones;
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(o1), "v"(fa));
          });
        }
      }
    };
    // one K tile held by BUF; FIRST = nothing to multiply yet in segment 1.  The next tile is always requested: past the
    // end of the k range the request goes through a descriptor with num_records = 0, which makes every lane out of
    // range (an out-of-range raw buffer load fetches nothing), so every iteration runs the same straight-line code.
    const int kend = kbeg + nk * BK;
    auto body = [&](auto bufc, auto firstc, int kt) {
      constexpr int BUF = decltype(bufc)::value;
      constexpr bool FIRST = decltype(firstc)::value;
      // this wave's share of tile kt has landed
      if constexpr (NL > 0) {}                                                 // (a consumer of the loader form has no vector-memory operation in flight)
      else if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDA + NDB) : "memory");   // the younger tile may still be in flight
                                                                               // (every wave issues NDA + NDB requests, no-ops included)
      __builtin_amdgcn_s_barrier();                           // everybody's has; the slot of tile kt-1 is free
      __builtin_amdgcn_sched_barrier(0);
      constexpr int TGT = (BUF + NS - 1) % NS;                // slot the request of this iteration goes to
      const int knext = kbeg + (kt + NS - 1) * BK;
      const bool more = knext < kend;
      const int soA = 2 * (A_TR ? knext * g.lda : knext), soB = 2 * (B_TR ? knext * g.ldb : knext);
      const __amdgpu_buffer_rsrc_t rqA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, more ? bytesA : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rqB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, more ? bytesB : 0, 0x00020000);
      constexpr int NDMA = NDA + NDB;
      constexpr int NOPS1 = NDMA + NRA + NRB;
      auto op1 = [&](auto kc) {                               // DMA of the next tile first, then k-step 0 of this one
        constexpr int k = decltype(kc)::value;
        if constexpr (k < NDMA) {
          if constexpr (k < NDA) planA.issue_one_g(rqA, rsNull, soA, sAp(TGT), dma_dummy, wave, k);
          else planB.issue_one_g(rqB, rsNull, soB, sBp(TGT), dma_dummy, wave, k - NDA);
        } else {
          read(bufc, std::integral_constant<int, 0>{}, f0, std::integral_constant<int, k - NDMA>{});
        }
      };
      if constexpr (FIRST) {
        static_for<NOPS1>(op1);
      } else {
        static_for<NM>([&](auto ic) {
          constexpr int i = decltype(ic)::value, lo = op_cut(i, NOPS1, NM), hi = op_cut(i + 1, NOPS1, NM);
          mfma1(f1, ic);                                       // k-step 1 of the previous tile
          static_for<hi - lo>([&](auto jc) { op1(std::integral_constant<int, lo + decltype(jc)::value>{}); });
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      lds_reads_done();
      if constexpr (!FIRST) bsum_step(f1, bs_c);               // behind the wait: no asynchronous register write is in flight at the
                                                               // branch; f1 is not refilled before segment 2's first read
      bs_c = bs_c + 1 == g.ntn ? 0 : bs_c + 1;                 // now: (this k-tile) mod ntn
      constexpr int NOPS2 = NRA + NRB;
      static_for<NM>([&](auto ic) {
        constexpr int i = decltype(ic)::value, lo = op_cut(i, NOPS2, NM), hi = op_cut(i + 1, NOPS2, NM);
        mfma1(f0, ic);
        static_for<hi - lo>([&](auto jc) {
          read(bufc, std::integral_constant<int, 1>{}, f1, std::integral_constant<int, lo + decltype(jc)::value>{});
        });
        __builtin_amdgcn_sched_barrier(0);
      });
      lds_reads_done();
      bsum_step(f0, bs_c);                                     // f0 is refilled only after the next barrier
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
#ifdef SFRON_DEBUG_KNOBS
    // diagnostic build only (MI355X guide, DVFS item 6): the clock this workgroup's K-loop ran at = delta s_memtime / delta s_memrealtime x 100 MHz
    long long clk_t0 = 0, clk_r0 = 0;
    if (g.dbg_clk) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    // nk is even and >= 2 here (the launcher routes other k ranges to the plain schedule): no conditional tail, whose
    // control-flow merge made hipcc spill fragment registers that an asynchronous ds_read had not filled yet
    body(I0{}, std::true_type{}, 0);
    body(I1{}, std::false_type{}, 1);
    if constexpr (NS == 2) {
      for (int kt = 2; kt < nk; kt += 2) { body(I0{}, std::false_type{}, kt); body(I1{}, std::false_type{}, kt + 1); }
    } else {                                                  // nk >= PRO and (nk - PRO) % 3 == 0 guaranteed by the launcher
      using I2 = std::integral_constant<int, 2>;
      static_assert(PRO == 2 || PRO == 3, "prologue length");
      if constexpr (PRO == 2) {
        for (int kt = 2; kt < nk; kt += 3) {
          body(I2{}, std::false_type{}, kt); body(I0{}, std::false_type{}, kt + 1); body(I1{}, std::false_type{}, kt + 2);
        }
      } else {
        body(I2{}, std::false_type{}, 2);
        for (int kt = 3; kt < nk; kt += 3) {
          body(I0{}, std::false_type{}, kt); body(I1{}, std::false_type{}, kt + 1); body(I2{}, std::false_type{}, kt + 2);
        }
      }
    }
    static_for<NM>([&](auto ic) { mfma1(f1, ic); });
    bsum_step(f1, bs_c);
    // the asm MFMAs are invisible to hipcc's hazard recogniser: let the last results land before the epilogue reads them.  The padding must
    // also be a SCHEDULING barrier: a volatile asm keeps its place among other volatile asm only, and hipcc is free to hoist the epilogue's
    // arithmetic on the accumulators (ordinary VALU code that merely depends on the MFMA statements' outputs) above it.  Round 6: in the
    // EPI_GELUQ instantiation it did -- ~300 instructions of GELU arithmetic in front of the two s_nop -- and the kernel returned a few
    // thousand wrong elements per launch whenever another stream's traffic delayed the last MFMAs (tools/diag_geluq.py; the bit-for-bit
    // soak test caught it).  The other instantiations happened to keep the order; all of them are pinned now.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#ifdef SFRON_DEBUG_KNOBS
    if (g.dbg_clk && tid == 0 && blockIdx.y == 0) {            // (memory nothing else reads: the stamps never reach an output)
      const long long dt = __builtin_amdgcn_s_memtime() - clk_t0, dr = __builtin_amdgcn_s_memrealtime() - clk_r0;
      long long* const o = g.dbg_clk + ((A_TR ? 2 : B_TR ? 1 : 0) * 1024 + (blockIdx.x & 1023)) * 6;
      o[0] = dt; o[1] = dr; o[2] = clk_e; o[3] = clk_r0; o[4] = clk_r0 + dr;
    }
#endif
  } else
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of tile kt has landed
    __builtin_amdgcn_s_barrier();                           // ... everybody's has; buffer buf^1 is free (its reads were waited for)
    if (kt + 1 < nk) stage(buf ^ 1, kbeg + (kt + 1) * BK);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(f0, buf, 0);                                 // k-step 0 of this tile goes out ...
    __builtin_amdgcn_sched_barrier(0);
    if (kt > 0) mfmas(f1);                                  // ... under the MFMAs of k-step 1 of the previous tile
    lds_reads_done();
    read_frags(f1, buf, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f0);
    lds_reads_done();
  }
  if constexpr (SCHED == 0) mfmas(f1);

#ifdef SFRON_DEBUG_KNOBS
  // timing experiment (results are not written): SFRON_GEMM_SAME_TILE bit 2 ends the kernel in front of its epilogue -- what a perfectly
  // overlapped epilogue would leave (one accumulator element is stored so that the main loop stays live)
  if (g.dbg_same & 4) { if (acc[0][0][0] == 12345.678f && g.Cf) g.Cf[0] = 1.0f; return; }
#endif
  const int row_b = m0 + wm * MT * 16 + (lane & 15), col_b = n0 + wn * NT * 16 + 4 * (lane >> 4);
  // loader form, forward layouts: the bias columns of this lane, all NT loads issued together now that the fragment registers are free
  // (a load inside each store's `if (bias)` would be waited for one by one: NT x MT exposed L2 latencies per tile)
  constexpr bool LATE_BIAS = !A_TR && !B_TR && NL > 0;
  float4 bias_l[LATE_BIAS ? NT : 1];
  if constexpr (LATE_BIAS) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      bias_l[nt] = g.bias ? *reinterpret_cast<const float4*>(g.bias + col_b + nt * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if constexpr (BSUM) {
    if (do_bs && lane < 16) {                 // all four accumulator rows of a lane hold the same sum
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) g.bsum[(size_t)tn * g.M + row_b + mt * 16] = bacc[mt][0] * g.alpha;
    }
  }
  // 16-byte epilogue accesses (common.h pair_pack / pair_unpack): two horizontally adjacent 16 x 16 tiles per access where the leading
  // dimension keeps the pieces 16-B aligned (kernel-uniform); an odd last tile and other leading dimensions take the 8-byte form
  const int lane_g = lane >> 4;
  const int col_p = n0 + wn * NT * 16 + pair_col(lane_g);          // this lane's column of a tile PAIR's 16-byte piece (+ nt * 16)
  constexpr int NP = NT / 2;                                       // tile pairs; tile NT - 1 is alone when NT is odd
  if constexpr (EPI == EPI_DGELU || EPI == EPI_DGELUQ) {
    const bool wide = ((g.ldaux | g.ldcb) & 7) == 0;              // (EPI_DGELUQ: guaranteed by the launcher)
    bf16x4 hx[EPI == EPI_DGELUQ ? 1 : MT][NT];
    unsigned hq[EPI == EPI_DGELUQ ? MT : 1][NT];                  // EPI_DGELUQ: this lane's four GELU' codes of tile (mt, nt)
    if constexpr (EPI == EPI_DGELUQ) {
      static_assert(!(NT & 1), "byte-coded GELU': tile pairs only");
      const uint8_t* const aq = reinterpret_cast<const uint8_t*>(g.aux);
      uint2 hp8[MT][NP];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int np = 0; np < NP; ++np)
          hp8[mt][np] = *reinterpret_cast<const uint2*>(aq + (size_t)(row_b + mt * 16) * g.ldaux + col_p + np * 32);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int np = 0; np < NP; ++np) pair_unpack8(hp8[mt][np], hq[mt][2 * np], hq[mt][2 * np + 1]);
    } else if (wide) {
      uint4 hp[MT][NP > 0 ? NP : 1];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int np = 0; np < NP; ++np)
          hp[mt][np] = *reinterpret_cast<const uint4*>(g.aux + (size_t)(row_b + mt * 16) * g.ldaux + col_p + np * 32);
        if constexpr (NT & 1)
          hx[mt][NT - 1] = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)(row_b + mt * 16) * g.ldaux + col_b + (NT - 1) * 16);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int np = 0; np < NP; ++np) pair_unpack(hp[mt][np], hx[mt][2 * np], hx[mt][2 * np + 1]);
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          hx[mt][nt] = *reinterpret_cast<const bf16x4*>(g.aux + (size_t)(row_b + mt * 16) * g.ldaux + col_b + nt * 16);
    }
    f32x4 cs[NT];                                  // this lane's column sums over its MT rows (fc1 bias gradient partials)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) cs[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      bf16x4 ob[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 v = acc[mt][nt] * g.alpha;
        f32x4 r;
        if constexpr (EPI == EPI_DGELUQ) r = v * geluq_unpack4(hq[mt][nt]);
        else r = v * gelu_tanh_grad4(bf2f4(hx[mt][nt]));
        cs[nt] += r;
        ob[nt] = f2bf4(r);
      }
      __bf16* const crow = g.Cb + (size_t)(row_b + mt * 16) * g.ldcb;
      if (wide) {
#pragma unroll
        for (int np = 0; np < NP; ++np) *reinterpret_cast<uint4*>(crow + col_p + np * 32) = pair_pack(ob[2 * np], ob[2 * np + 1]);
        if constexpr (NT & 1) *reinterpret_cast<bf16x4*>(crow + col_b + (NT - 1) * 16) = ob[NT - 1];
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<bf16x4*>(crow + col_b + nt * 16) = ob[nt];
      }
    }
    if (g.colpart) {                               // kernel-uniform
      // rows of a wave: the 16 lanes that share lane >> 4 hold the same 4 columns -> butterfly over lane & 15, then the WM
      // waves stacked over the rows meet in LDS (free now: every wave is past the main loop after the barrier); fixed order
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = cs[nt][j];
          x = row16_sum(x);
          cs[nt][j] = x;
        }
      __syncthreads();
      float* red = reinterpret_cast<float*>(smem);                         // [WM][FBN]
      if ((lane & 15) == 0) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          *reinterpret_cast<f32x4*>(red + wm * FBN + wn * NT * 16 + nt * 16 + 4 * (lane >> 4)) = cs[nt];
      }
      __syncthreads();
      for (int cidx = tid; cidx < FBN; cidx += NW * 64) {
        float x = red[cidx];
#pragma unroll
        for (int w2 = 1; w2 < WM; ++w2) x += red[w2 * FBN + cidx];
        g.colpart[(size_t)tm * g.N + n0 + cidx] = x;
      }
    }
  } else if constexpr (EPI == EPI_GATE_RES) {
    // all rows of a tile belong to one sample when the token count is a multiple of the tile's rows (DiT: 256 tokens, 256-row tiles): the
    // gate row is then fetched once per n-tile, not once per 16-row block (MT x fewer 16-byte loads in an issue-bound tail)
    const bool one_sample = (g.T % FBM) == 0;
    float4 gt[NT];
    if (one_sample) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) gt[nt] = *reinterpret_cast<const float4*>(g.gate + (size_t)(row_b / g.T) * g.ldgate + col_b + nt * 16);
    }
    // (The residual loads stay per 16-row block.  Issuing all MT x NT of the tile up front saves one exposed latency -- and takes the 256 x 144
    // kernel from 206 to 234 registers = 240 allocated: two of its waves then leave a SIMD 32 registers, the parameter sweep that runs
    // beside the forward pass needs 48 per wave and loses every CU this kernel is on: +2.7 ms per step, profiles/r04_ab_log.txt.)
    // Round 6: the residual rows of block mt + 1 are requested WHILE block mt is finished -- piece nt of the next block right behind the
    // point where piece nt of this block (accumulator + residual: eight registers) dies, so the register count does not grow -- instead of
    // after this block's stores: one exposed memory latency per tile instead of MT.  (one_sample only: with several samples per tile the gate
    // rows change per block and the loop keeps the round-5 order.)  Same arithmetic: the same bits.
    float4 xr[2][NT];
#ifdef SFRON_TUNE_GATE_RES_R5           // A-B build only (tools/build_variant.sh): the round-5 order, one exposed latency per 16-row block
    const bool pipe = false;
#else
    const bool pipe = one_sample;
#endif
    if (pipe) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) xr[0][nt] = *reinterpret_cast<const float4*>(g.resid + (size_t)row_b * g.ldcf + col_b + nt * 16);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = row_b + mt * 16;
      if (!pipe) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          xr[mt & 1][nt] = *reinterpret_cast<const float4*>(g.resid + (size_t)row * g.ldcf + col_b + nt * 16);
          if (!one_sample) gt[nt] = *reinterpret_cast<const float4*>(g.gate + (size_t)(row / g.T) * g.ldgate + col_b + nt * 16);
        }
      }
      const bool wide = (g.ldaux & 7) == 0 && !(g.nt_out & 1);
      bf16x4 ab[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int col = col_b + nt * 16;
        f32x4 v = acc[mt][nt] * g.alpha;
        if constexpr (PRE_BIAS) v += as4(bias_v[nt]);
        else if constexpr (LATE_BIAS) v += as4(bias_l[nt]);
        else if (g.bias) v += as4(*reinterpret_cast<const float4*>(g.bias + col));
        ab[nt] = f2bf4(v);
        if (!wide) nt_store(reinterpret_cast<bf16x4*>(g.aux + (size_t)row * g.ldaux + col), ab[nt], g.nt_out & 1);
        const f32x4 x = as4(xr[mt & 1][nt]) + as4(gt[nt]) * v;
        if (pipe && mt + 1 < MT)
          xr[(mt + 1) & 1][nt] = *reinterpret_cast<const float4*>(g.resid + (size_t)(row + 16) * g.ldcf + col);
        *reinterpret_cast<f32x4*>(g.Cf + (size_t)row * g.ldcf + col) = x;
      }
#ifndef SFRON_TUNE_NO_BRANCH                       // timing experiment only (wrong d gate): what does the saved branch output cost?
      if (wide) {                                  // the bf16 branch output (read again by the backward pass): 16-byte pieces
        __bf16* const arow = g.aux + (size_t)row * g.ldaux;
#pragma unroll
        for (int np = 0; np < NP; ++np) *reinterpret_cast<uint4*>(arow + col_p + np * 32) = pair_pack(ab[2 * np], ab[2 * np + 1]);
        if constexpr (NT & 1) *reinterpret_cast<bf16x4*>(arow + col_b + (NT - 1) * 16) = ab[NT - 1];
      }
#endif
    }
  } else if constexpr (EPI == EPI_BF16 || EPI == EPI_GELU || EPI == EPI_GELUQ) {
    const bool wide = EPI == EPI_GELUQ || ((g.ldcb & 7) == 0 && (EPI != EPI_GELU || ((g.ldaux & 7) == 0 && !(g.nt_out & 1))));
    if (!wide) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          epilogue_store<EPI>(g, row_b + mt * 16, col_b + nt * 16, acc[mt][nt],
                              PRE_BIAS ? &bias_v[PRE_BIAS ? nt : 0] : LATE_BIAS ? &bias_l[LATE_BIAS ? nt : 0] : nullptr);
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = row_b + mt * 16;
        bf16x4 ob[NT], hb[EPI == EPI_GELU ? NT : 1];
        unsigned qb[EPI == EPI_GELUQ ? NT : 1];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          f32x4 v = acc[mt][nt] * g.alpha;
          if constexpr (PRE_BIAS) v += as4(bias_v[nt]);
          else if constexpr (LATE_BIAS) v += as4(bias_l[nt]);
          else if (g.bias) v += as4(*reinterpret_cast<const float4*>(g.bias + col_b + nt * 16));
          if constexpr (EPI == EPI_GELU) {
            hb[nt] = f2bf4(v);                                  // pre-activation (the backward pass reads it)
            ob[nt] = f2bf4(gelu_tanh4(v));
          } else if constexpr (EPI == EPI_GELUQ) {
            f32x4 y, dy;
            gelu_tanh_both4(v, y, dy);                          // the backward pass reads GELU'(v), one byte each
            ob[nt] = f2bf4(y);
            qb[nt] = geluq_pack4(dy);
          } else {
            ob[nt] = f2bf4(v);
          }
        }
        __bf16* const crow = g.Cb + (size_t)row * g.ldcb;
#pragma unroll
        for (int np = 0; np < NP; ++np) *reinterpret_cast<uint4*>(crow + col_p + np * 32) = pair_pack(ob[2 * np], ob[2 * np + 1]);
        if constexpr (NT & 1) *reinterpret_cast<bf16x4*>(crow + col_b + (NT - 1) * 16) = ob[NT - 1];
        if constexpr (EPI == EPI_GELUQ) {
          static_assert(EPI != EPI_GELUQ || !(NT & 1), "byte-coded GELU': tile pairs only");
          uint8_t* const qrow = reinterpret_cast<uint8_t*>(g.aux) + (size_t)row * g.ldaux;
#pragma unroll
          for (int np = 0; np < NP; ++np) *reinterpret_cast<uint2*>(qrow + col_p + np * 32) = pair_pack8(qb[EPI == EPI_GELUQ ? 2 * np : 0], qb[EPI == EPI_GELUQ ? 2 * np + 1 : 0]);
        }
#ifndef SFRON_TUNE_NO_HPRE_STORE          // timing experiment only (tools/build_variant.sh; wrong gradients): what would fc1 cost without its second output?
        if constexpr (EPI == EPI_GELU) {
          __bf16* const arow = g.aux + (size_t)row * g.ldaux;
#pragma unroll
          for (int np = 0; np < NP; ++np) *reinterpret_cast<uint4*>(arow + col_p + np * 32) = pair_pack(hb[2 * np], hb[2 * np + 1]);
          if constexpr (NT & 1) *reinterpret_cast<bf16x4*>(arow + col_b + (NT - 1) * 16) = hb[NT - 1];
        }
#endif
      }
    }
  } else {
    // weight gradients (dW = dY^T X, fp32 into the gradient arena): the clip norm's masked sum of squares of this tile, taken from the
    // accumulators that are about to be stored -- sq_out[workgroup] = sum over the tile of (mask ? dW : 0)^2 (fp64 from the wave sums on,
    // fixed order: bitwise reproducible).  Replaces the forget stage's pass over the whole block range of the gradient arena
    // (sweep.hip k_sumsq_masked: 450 M floats + mask bytes, 350 us on the critical stream at DiT-XL/2); the mask bytes (same offsets as
    // dW: ld = ldcf) are all requested before the first use.  Kernel-uniform branch.
    [[maybe_unused]] float sq = 0.f;
    if constexpr (EPI == EPI_F32 && A_TR && B_TR && !BSUM) {
      if (g.sq_out) {
        uchar4 mk[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            mk[mt][nt] = g.sq_mask ? *reinterpret_cast<const uchar4*>(g.sq_mask + (size_t)(row_b + mt * 16) * g.ldcf + col_b + nt * 16)
                                   : make_uchar4(1, 1, 1, 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const f32x4 v = acc[mt][nt] * g.alpha;
            const uchar4 m4 = mk[mt][nt];
            sq += (m4.x ? v[0] * v[0] : 0.f) + (m4.y ? v[1] * v[1] : 0.f) + (m4.z ? v[2] * v[2] : 0.f) + (m4.w ? v[3] * v[3] : 0.f);
          }
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        epilogue_store<EPI>(g, row_b + mt * 16, col_b + nt * 16, acc[mt][nt],
                            PRE_BIAS ? &bias_v[PRE_BIAS ? nt : 0] : LATE_BIAS ? &bias_l[LATE_BIAS ? nt : 0] : nullptr);
    if constexpr (EPI == EPI_F32 && A_TR && B_TR && !BSUM) {
      if (g.sq_out) {
        const double d = wave_sum_d((double)sq);
        __syncthreads();                                       // every wave is past the main loop: the staging images are free
        double* red = reinterpret_cast<double*>(smem);
        if (lane == 0) red[wave] = d;
        __syncthreads();
        if (tid == 0) {
          double t = red[0];
#pragma unroll
          for (int w2 = 1; w2 < NW; ++w2) t += red[w2];
          g.sq_out[blockIdx.x] = t;
        }
      }
    }
  }
#ifdef SFRON_DEBUG_KNOBS
  if (g.dbg_clk && tid == 0 && blockIdx.y == 0)               // every store of wave 0 issued (not necessarily written)
    g.dbg_clk[((A_TR ? 2 : B_TR ? 1 : 0) * 1024 + (blockIdx.x & 1023)) * 6 + 5] = (long long)__builtin_amdgcn_s_memrealtime();
#endif
}

#define SFRON_INST_PIPE(WM, WN, MT, NT)                                              \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, false, 0>(GemmArgs);  \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, false, 1>(GemmArgs);  \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, false, 2>(GemmArgs);  \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, false, 3>(GemmArgs);  \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, false, 5>(GemmArgs);  \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, true, 0>(GemmArgs);   \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, true, 1>(GemmArgs);   \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, true, 4>(GemmArgs);   \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, true, true, 0>(GemmArgs);    \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, true, true, 1>(GemmArgs);
SFRON_INST_PIPE(4, 2, 4, 6)
SFRON_INST_PIPE(4, 2, 3, 6)
SFRON_INST_PIPE(2, 2, 8, 6)
#undef SFRON_INST_PIPE
#define SFRON_INST_PIPE1(WM, WN, MT, NT)                                                \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, false, 0, 1>(GemmArgs);  \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, false, 1, 1>(GemmArgs);  \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, false, 2, 1>(GemmArgs);  \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, false, 3, 1>(GemmArgs);  \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, false, 5, 1>(GemmArgs);  \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, true, 0, 1>(GemmArgs);   \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, true, 1, 1>(GemmArgs);   \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, false, true, 4, 1>(GemmArgs);   \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, true, true, 0, 1>(GemmArgs);    \
  template __global__ void k_gemm_pipe<WM, WN, MT, NT, true, true, 1, 1>(GemmArgs);
SFRON_INST_PIPE1(4, 2, 4, 6)
SFRON_INST_PIPE1(4, 2, 3, 6)
#undef SFRON_INST_PIPE1
template __global__ void k_gemm_pipe<4, 2, 4, 6, false, false, EPI_GELUQ, 1>(GemmArgs);   // fc1 + GELU, GELU' as bytes (round 6)
template __global__ void k_gemm_pipe<4, 2, 4, 6, false, true, EPI_DGELUQ, 1>(GemmArgs);   // fc2 dgrad * GELU' from the bytes
template __global__ void k_gemm_pipe<4, 2, 3, 6, true, true, 0, 2>(GemmArgs);    // 192x192, three slots: weight gradients
template __global__ void k_gemm_pipe<4, 2, 3, 6, true, true, 1, 2>(GemmArgs);
template __global__ void k_gemm_pipe<4, 2, 3, 6, true, true, 1, 2, 2, true>(GemmArgs);   // ... + bias row sums
template __global__ void k_gemm_pipe<4, 2, 3, 6, false, false, 0, 2>(GemmArgs);  // (experiment: forward layout)
// 256x144 tile, 8 x 1 waves of 32x144, three slots, 3-tile prologue: the forward GEMMs.  36,864 outputs per tile = exactly
// 256 tiles for a [8192 x 1152] output (one per CU, where 256x192 gives 192), 768 for qkv, 1024 for fc1.
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, false, 0, 2, 3>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, false, 1, 2, 3>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, false, 2, 2, 3>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, false, 3, 2, 3>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, false, 5, 2, 3>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, true, 0, 2, 3>(GemmArgs);   // dgrad into a 1152-wide input gradient
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, true, 1, 2, 3>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, true, 4, 2, 3>(GemmArgs);
// ... and with four loader waves
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, false, 0, 2, 3, false, 4>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, false, 1, 2, 3, false, 4>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, false, 2, 2, 3, false, 4>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, false, 3, 2, 3, false, 4>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, false, 5, 2, 3, false, 4>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, true, 0, 2, 3, false, 4>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, true, 1, 2, 3, false, 4>(GemmArgs);
template __global__ void k_gemm_pipe<8, 1, 2, 9, false, true, 4, 2, 3, false, 4>(GemmArgs);
template __global__ void k_gemm_pipe<4, 2, 3, 6, true, true, 0, 2, 2, false, 4>(GemmArgs);
template __global__ void k_gemm_pipe<4, 2, 3, 6, false, false, 0, 2, 2, false, 4>(GemmArgs);   // (experiment: forward layout on the 4 x 2 tile)
template __global__ void k_gemm_pipe<4, 2, 3, 6, true, true, 1, 2, 2, false, 4>(GemmArgs);

#ifdef SFRON_DEBUG_KNOBS
// diagnostic build: SFRON_GEMM_CLK=1 makes every interleaved-schedule GEMM leave the clock stamps of its K-loop (last launch of each layout
// class wins a slot); sfron_dbg_gemm_clock() returns the median clock in MHz per class {forward, dgrad, weight gradient}
static long long* dbg_clk_buffer() {
  static long long* buf = [] {
    long long* p = nullptr;
    if (getenv("SFRON_GEMM_CLK") && hipMalloc(&p, 3 * 1024 * 6 * sizeof(long long)) == hipSuccess) (void)hipMemset(p, 0, 3 * 1024 * 6 * sizeof(long long));
    return p;
  }();
  return buf;
}
extern "C" int sfron_dbg_gemm_clock(double* mhz3, int* n3) {
  long long* d = dbg_clk_buffer();
  if (!d || !mhz3 || !n3) return SFRON_ERR_ARG;
  static long long host[3 * 1024 * 6];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host, d, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return (int)hipGetLastError();
  for (int c = 0; c < 3; ++c) {
    double v[1024]; int n = 0;
    for (int i = 0; i < 1024; ++i) { const long long dt = host[(c * 1024 + i) * 6], dr = host[(c * 1024 + i) * 6 + 1]; if (dr > 0) v[n++] = 100.0 * (double)dt / (double)dr; }
    for (int i = 1; i < n; ++i) { double x = v[i]; int j = i - 1; while (j >= 0 && v[j] > x) { v[j + 1] = v[j]; --j; } v[j + 1] = x; }
    mhz3[c] = n ? v[n / 2] : 0.0; n3[c] = n;
  }
  (void)hipMemset(d, 0, sizeof(host));
  return SFRON_OK;
}
// the raw stamps of one layout class: [1024][6] = (K-loop delta s_memtime, delta s_memrealtime, entry, loop start, loop end, exit; 100 MHz ticks)
extern "C" int sfron_dbg_gemm_phases(int cls, long long* out) {
  long long* d = dbg_clk_buffer();
  if (!d || !out || cls < 0 || cls > 2) return SFRON_ERR_ARG;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, d + (size_t)cls * 1024 * 6, 1024 * 6 * sizeof(long long), hipMemcpyDeviceToHost) != hipSuccess)
    return (int)hipGetLastError();
  (void)hipMemset(d + (size_t)cls * 1024 * 6, 0, 1024 * 6 * sizeof(long long));
  return SFRON_OK;
}
#endif

namespace {

template <int WM, int WN, int MT, int NT, bool A_TR, bool B_TR, int EPI, int SCHED = 0, int PRO = 2, bool BSUM = false, int NL = 0>
int launch_pipe(GemmArgs g, hipStream_t s) {
  constexpr int FBM = WM * MT * 16, FBN = WN * NT * 16;
  g.ntm = g.M / FBM; g.ntn = g.N / FBN;
#ifdef SFRON_DEBUG_KNOBS
  { static const int same = getenv("SFRON_GEMM_SAME_TILE") ? atoi(getenv("SFRON_GEMM_SAME_TILE")) : 0; g.dbg_same = same; }
  g.dbg_clk = dbg_clk_buffer();
#endif
  {
    const double per_xcd = (double)g.ntm * g.ntn / 8.0;
    int gm = 1;
    while (gm * 2 <= g.ntm && (double)(gm * 2) * (gm * 2) * FBM <= per_xcd * FBN * 1.5) gm *= 2;
    g.group_m = gm;
  }
  constexpr int NWD = NL > 0 ? NL : WM * WN;
  constexpr bool uneven = !GldsPlan<FBM, A_TR, NWD>::EVEN || !GldsPlan<FBN, B_TR, NWD>::EVEN;
  constexpr int FBM_P = A_TR ? tr_cols<FBM>() : FBM, FBN_P = B_TR ? tr_cols<FBN>() : FBN;
  const size_t lds = (SCHED == 2 ? 3 : 2) * (FBM_P + FBN_P) * 64 * sizeof(__bf16) + (uneven ? 1024 : 0);
  static std::atomic<uint64_t> done{0};        // per instantiation, one bit per device
  if (need_attr(done)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_pipe<WM, WN, MT, NT, A_TR, B_TR, EPI, SCHED, PRO, BSUM, NL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return (int)hipGetLastError();
  }
  SFRON_LAUNCH_EV((k_gemm_pipe<WM, WN, MT, NT, A_TR, B_TR, EPI, SCHED, PRO, BSUM, NL>), dim3(g.ntm * g.ntn, cdiv(g.K, g.kchunk)), dim3((WM * WN + NL) * 64), lds, s, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SFRON_OK : (int)e;
}

}  // namespace

namespace {

// fast tiles: 1 = 128x128 (4 waves), 2 = 256x192, 3 = 256x256, 4 = 384x192 (8 waves); 0 = generic kernel.
// 8-wave tiles halve the L1->LDS bytes per FLOP of the 128x128 tile (which is vector-memory bound at ~600 TF).
constexpr int N_TILES = 9;
static const int TILE_BM[N_TILES] = {0, 128, 256, 256, 384, 192, 256, 192, 256};
static const int TILE_BN[N_TILES] = {0, 128, 192, 256, 192, 192, 192, 192, 144};
// shapes whose weight gradient runs on the three-slot 192x192 tile (nk = 2 + 3j k-tiles of 64): the kernel that can also
// produce the bias row sums
inline bool rowsum_ok(int M, int N, int K) {
  return M % 192 == 0 && N % 192 == 0 && K % 64 == 0 && K / 64 >= 2 && (K / 64 - 2) % 3 == 0;
}
inline bool tile_fits(const GemmArgs& g, int t) {
  return t >= 1 && t < N_TILES && g.M % TILE_BM[t] == 0 && g.N % TILE_BN[t] == 0;
}
inline int pick_fast_tile(const GemmArgs& g, int force, int transposed_operands) {
  if (g.K % 64 || force < 0) return 0;
  if (g.kchunk != g.K) {       // split-K: only the pipelined tiles take a k range
    if (force == 32 || force == 35) return tile_fits(g, force - 30) ? force : 0;
    if (force != 0) return 0;
    if (transposed_operands == 2 && tile_fits(g, 5)) return 55;      // 55 -> 45 -> 35 and 42 -> 32 fall back by k-tile count
    if (transposed_operands <= 1 && tile_fits(g, 2)) return 42;      // (round 5: the forward layout too -- few-tile products of a small batch)
    return 0;
  }
  if (force == 21 || force == 22) return tile_fits(g, 2) ? force : 0;            // timing ablations of tile 2
  if (force == 32 || force == 35 || force == 36) return tile_fits(g, force - 30) ? force : 0;   // hand-pipelined variants of tiles 2, 5, 6
  if (force == 42 || force == 45) return tile_fits(g, force - 40) ? force : 0;                  // ... with the interleaved schedule
  if (force == 55) return tile_fits(g, 5) ? force : 0;                                          // ... and three LDS slots
  if (force == 62) return tile_fits(g, 8) ? force : (tile_fits(g, 2) ? 42 : 0);                 // 256x144, three slots (forward layouts)
  if (force > 10) return 0;
  if (force > 0) return tile_fits(g, force) ? force : 0;
  // Measured policy (tools/bench_gemm.py, DiT-XL/2 B=32 shapes, random data):
  //   both operands k-contiguous (forward)   -> 256x192 k_gemm_pipe, interleaved schedule (800-980 TF; k_gemm_fast 760-950)
  //   B transposed-read (dgrad)              -> 256x192 k_gemm_pipe, interleaved schedule (735-1040 TF; plain schedule 630-885)
  //   both transposed-read (wgrad)           -> 192x192 k_gemm_pipe, interleaved schedule, THREE LDS slots: two tiles
  //                                             (96 KB) in flight per CU hide the HBM latency of operands that are not
  //                                             in the 256 MB Infinity Cache -- saved activations, as in the real step:
  //                                             fc1 93 us hot / 113 us cold, against 104 / 155 us with two slots
  //                                             (tools/bench_cold.py, tools/bench_cold2.py)
  // a tile is used when it fills at least half of the last round of 256 CUs (wgrad runs beside the dgrad chain
  // on a side stream, so its own tile count does not have to fill the chip).
  auto eff = [&](int t) {
    const long tiles = (long)(g.M / TILE_BM[t]) * (g.N / TILE_BN[t]);
    return (double)tiles / (double)(((tiles + 255) / 256) * 256);
  };
  const bool even_nk = g.K % 128 == 0;       // the interleaved schedule's requirement
  if (transposed_operands == 2) return tile_fits(g, 5) ? 55 : 0;
  // forward layouts: 256x144 three-slot tile when it fills the CUs better than 256x192 (a [8192 x 1152] output is exactly 256
  // tiles instead of 192; measured cold: proj 35.7 -> 28.1 us, qkv 102.7 -> 78.3, fc2 103.9 -> 83.5; fc1 (both 100 %) stays)
#ifdef SFRON_DEBUG_KNOBS
  static const bool t62_all = getenv("SFRON_GEMM_T62_ALL") != nullptr;      // A-B knob (debug builds only): the 256x144 tile wherever it fits
#else
  constexpr bool t62_all = false;
#endif
  if (transposed_operands <= 1 && tile_fits(g, 8) && g.K % 192 == 0 && (t62_all || !tile_fits(g, 2) || eff(8) > eff(2) + 0.05)) return 62;
  if (tile_fits(g, 2) && eff(2) >= 0.5) return (transposed_operands == 1 || even_nk) ? 42 : 2;
  return (transposed_operands == 0 && tile_fits(g, 1)) ? 1 : 0;   // with transposed reads the generic kernel beats the 4-wave tile
}

int g_loader_waves = 4;      // process-wide: 4 = the three-slot tiles with a transposed operand run with four loader waves (k_gemm_pipe NL); sfron_gemm_loader_waves()

template <bool A_TR, bool B_TR, int EPI>
int launch_any(const GemmArgs& g, hipStream_t s, int force) {
  switch (pick_fast_tile(g, force, (A_TR ? 1 : 0) + (B_TR ? 1 : 0))) {
    case 1: return launch_fast<2, 2, 4, 4, A_TR, B_TR, EPI>(g, s);
    case 2: return launch_fast<4, 2, 4, 6, A_TR, B_TR, EPI>(g, s);
    case 3: return launch_fast<2, 4, 8, 4, A_TR, B_TR, EPI>(g, s);
    case 4: return launch_fast<4, 2, 6, 6, A_TR, B_TR, EPI>(g, s);
    case 5: return launch_fast<4, 2, 3, 6, A_TR, B_TR, EPI>(g, s);
    case 6: return launch_fast<2, 2, 8, 6, A_TR, B_TR, EPI>(g, s);
    case 7: return launch_fast<2, 2, 6, 6, A_TR, B_TR, EPI>(g, s);
    case 21: if constexpr (!A_TR && !B_TR && EPI == EPI_BF16) return launch_fast<4, 2, 4, 6, false, false, 0, 1>(g, s);
             else if constexpr (A_TR && B_TR && EPI == EPI_F32) return launch_fast<4, 2, 4, 6, true, true, 1, 1>(g, s); else return SFRON_ERR_UNSUPPORTED;
    case 22: if constexpr (!A_TR && !B_TR && EPI == EPI_BF16) return launch_fast<4, 2, 4, 6, false, false, 0, 2>(g, s);
             else if constexpr (A_TR && B_TR && EPI == EPI_F32) return launch_fast<4, 2, 4, 6, true, true, 1, 2>(g, s); else return SFRON_ERR_UNSUPPORTED;
    case 32: return launch_pipe<4, 2, 4, 6, A_TR, B_TR, EPI>(g, s);
    case 35: return launch_pipe<4, 2, 3, 6, A_TR, B_TR, EPI>(g, s);
    case 36: return launch_pipe<2, 2, 8, 6, A_TR, B_TR, EPI>(g, s);
    // interleaved schedule: needs an even number (>= 2) of 64-deep k tiles per split, else the plain schedule
    case 42: return (g.kchunk % 128 == 0 && g.K % g.kchunk == 0) ? launch_pipe<4, 2, 4, 6, A_TR, B_TR, EPI, 1>(g, s)
                                                                  : launch_pipe<4, 2, 4, 6, A_TR, B_TR, EPI>(g, s);
    case 45: return (g.kchunk % 128 == 0 && g.K % g.kchunk == 0) ? launch_pipe<4, 2, 3, 6, A_TR, B_TR, EPI, 1>(g, s)
                                                                  : launch_pipe<4, 2, 3, 6, A_TR, B_TR, EPI>(g, s);
    case 62:   // 256x144 with three LDS slots, forward layouts: needs nk = 3 + 3j tiles (K a multiple of 192), no split
      if constexpr ((!A_TR && !B_TR && (EPI == EPI_BF16 || EPI == EPI_F32 || EPI == EPI_GELU || EPI == EPI_GATE_RES || EPI == EPI_POS)) ||
                    (!A_TR && B_TR && (EPI == EPI_BF16 || EPI == EPI_F32 || EPI == EPI_DGELU))) {
        if (g.kchunk == g.K && g.K % 192 == 0)
          // loader waves: measured faster where an operand is read transposed (dgrad: qkv 68 -> 57 us, proj 28.4 -> 26.1, fc1 76.9 -> 72.6), slower
          // or equal on the forward layouts (qkv 76 -> 83)
          return (((g_loader_waves == 4 || g_loader_waves == 6) && B_TR) || g_loader_waves == 8) ? launch_pipe<8, 1, 2, 9, A_TR, B_TR, EPI, 2, 3, false, 4>(g, s) : launch_pipe<8, 1, 2, 9, A_TR, B_TR, EPI, 2, 3>(g, s);
      }
      if (!tile_fits(g, 2)) return launch<A_TR, B_TR, EPI>(g, s);
      return g.K % 128 == 0 && g.kchunk == g.K ? launch_pipe<4, 2, 4, 6, A_TR, B_TR, EPI, 1>(g, s) : launch_pipe<4, 2, 4, 6, A_TR, B_TR, EPI>(g, s);
    case 55:   // 192x192 with three LDS slots (weight-gradient layouts only): needs nk = 2 + 3j tiles per split
      if constexpr (A_TR && B_TR && EPI == EPI_F32) {
        if (g.bsum) return rowsum_ok(g.M, g.N, g.K) && g.kchunk == g.K ? launch_pipe<4, 2, 3, 6, true, true, EPI_F32, 2, 2, true>(g, s) : SFRON_ERR_UNSUPPORTED;
      }
      if constexpr ((A_TR && B_TR && (EPI == EPI_BF16 || EPI == EPI_F32)) || (!A_TR && !B_TR && EPI == EPI_BF16)) {
        if (g.K % g.kchunk == 0 && (g.kchunk / 64) >= 2 && ((g.kchunk / 64) - 2) % 3 == 0) {
          if constexpr (A_TR && B_TR) { if (g_loader_waves == 4 || g_loader_waves == 5 || g_loader_waves == 8) return launch_pipe<4, 2, 3, 6, A_TR, B_TR, EPI, 2, 2, false, 4>(g, s); }
          if constexpr (!A_TR && !B_TR && EPI == EPI_BF16) { if (g_loader_waves == 7) return launch_pipe<4, 2, 3, 6, false, false, EPI_BF16, 2, 2, false, 4>(g, s); }   // (experiment)
          return launch_pipe<4, 2, 3, 6, A_TR, B_TR, EPI, 2>(g, s);
        }
      }
      return (g.kchunk % 128 == 0 && g.K % g.kchunk == 0) ? launch_pipe<4, 2, 3, 6, A_TR, B_TR, EPI, 1>(g, s)
                                                           : launch_pipe<4, 2, 3, 6, A_TR, B_TR, EPI>(g, s);
    default: return launch<A_TR, B_TR, EPI>(g, s);
  }
}

template <int EPI>
int dispatch_layout(int a_tr, int b_tr, const GemmArgs& g, hipStream_t s, int force) {
  if (!a_tr && !b_tr) return launch_any<false, false, EPI>(g, s, force);
  if (!a_tr && b_tr) return launch_any<false, true, EPI>(g, s, force);
  if (a_tr && b_tr) return launch_any<true, true, EPI>(g, s, force);
  return SFRON_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int sfron_gemm_rowsum_supported(int M, int N, int K) { return rowsum_ok(M, N, K) ? 1 : 0; }
// SFRON_EPI_GELU_Q / SFRON_EPI_DGELU_Q run on the 256 x 192 pipelined tile with the interleaved schedule only (an even number of 64-deep k-tiles)
extern "C" int sfron_gemm_gelu_q_supported(int M, int N, int K) { return M > 0 && N > 0 && K > 0 && M % 256 == 0 && N % 192 == 0 && K % 128 == 0 ? 1 : 0; }

// Weight gradients dW[M][N] = dY[K][M]^T X[K][N] whose automatic tile is the 192 x 192 pipelined one can also leave the masked sum of squares of
// their output, one fp64 partial per tile: returns the number of partials (M / 192 * N / 192), or 0 when the shape goes another way.
extern "C" int sfron_gemm_sumsq_partials(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0 || M % 192 || N % 192 || K % 64) return 0;
  return (M / 192) * (N / 192);
}

// EPI_DGELU products dX[M][N] = dY[M][K] W[K][N] (B read transposed) whose automatic tile is one of the 256-row pipelined tiles can
// also write the per-tile-row column sums of their output: returns the number of partial rows (M / 256), or 0 when the shape goes
// to another kernel (the caller then uses sfron_colsum)
extern "C" int sfron_gemm_dgelu_colpart_rows(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  GemmArgs g{};
  g.M = M; g.N = N; g.K = K; g.kchunk = K;
  const int t = pick_fast_tile(g, 0, 1);
  if (t == 42) return M / 256;
  if (t == 62) return (K % 192 == 0 || tile_fits(g, 2)) ? M / 256 : 0;
  return 0;
}

// Masked sum of squares of W' = A^T B (A bf16 [R][NM], B bf16 [R][D], contraction over the R rows) without storing W': one fp64 partial per
// 128 x 128 tile, partials[0 .. *nblk).  C++ linkage: called by sweep.hip (sfron_sumsq_lowrank), not part of the C ABI.
int gemm_sumsq_lowrank(const uint16_t* a, const uint16_t* b, int R, int NM, int D, const uint8_t* mask, double* partials, int* nblk, void* stream) {
  GemmArgs g{};
  g.A = (const __bf16*)a; g.B = (const __bf16*)b;
  g.M = NM; g.N = D; g.K = R; g.lda = NM; g.ldb = D;
  g.alpha = 1.0f; g.kchunk = R; g.T = 1;
  g.ntm = cdiv(NM, BM); g.ntn = cdiv(D, BN);
  g.sq_mask = mask; g.sq_out = partials;
  *nblk = g.ntm * g.ntn;
  return launch<true, true, EPI_SUMSQ>(g, (hipStream_t)stream);
}

// Schedule switch (include/sfron.h): 4 = loader-wave form of the three-slot tiles (default), 0 = every wave issues its share of the LDS-DMA, 9 = as 4 and
// the fp8 tiles in their loader form too, 10 = as 4 with the convolution tiles in the shared form.  Results never depend on it (bit-identical
// schedules: tests/test_gpu_baseline_shapes.py, test_gpu_unet.py, test_gpu_fp8.py compare the forms).  Any other value is refused by the product build
// (the setting stays); the A-B values 5..8, 11, 12 of tools/ exist in the SFRON_DEBUG_KNOBS build (libsfron_dbg.so) only.  Returns the old value.
extern "C" int sfron_gemm_loader_waves(int n) {
  const int old = g_loader_waves;
#ifndef SFRON_DEBUG_KNOBS
  if (n != 0 && n != 4 && n != 9 && n != 10) return old;
#endif
  g_loader_waves = (n >= 9 && n <= 12) ? 4 : (n >= 4 && n <= 8) ? n : 0;      // (5 / 6: weight gradients / dgrad only; 8: as 4 + the forward layouts of the 256 x 144 tile)
  g_fp8_loader_waves = n == 9 ? 4 : 0;                                        // (9: measured no faster)
  g_conv_loader_waves = (n == 0 || n == 10) ? 0 : n == 11 ? 1 : n == 12 ? 2 : 3;   // (11 / 12: only k_cgemm / only k_cgemm_t in the loader form)
  return old;
}

extern "C"
int sfron_gemm_bf16(const sfron_gemm_desc* d, void* stream) {
  SFRON_CHECK_ARG(d && d->A && d->B && d->M > 0 && d->N > 0 && d->K > 0);
  SFRON_CHECK_ARG(d->N % 4 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0);
  if (!d->a_transposed || !d->b_transposed) SFRON_CHECK_ARG(d->K % 8 == 0);   // k-contiguous operands are staged in 8-element chunks
  SFRON_CHECK_ARG((((uintptr_t)d->A | (uintptr_t)d->B) & 15) == 0);
  if (d->a_transposed) SFRON_CHECK_ARG(d->M % 8 == 0);
  if (d->b_transposed) SFRON_CHECK_ARG(d->N % 8 == 0);
  GemmArgs g{};
  g.A = (const __bf16*)d->A; g.B = (const __bf16*)d->B;
  g.M = d->M; g.N = d->N; g.K = d->K; g.lda = d->lda; g.ldb = d->ldb;
  g.Cb = (__bf16*)d->c_bf16; g.ldcb = d->ldc_bf16;
  g.Cf = d->c_f32; g.ldcf = d->ldc_f32;
  g.bias = d->bias;
  g.aux = (__bf16*)d->aux; g.ldaux = d->ldaux;
  g.gate = d->gate; g.ldgate = d->ldgate;
  g.pos = d->pos; g.T = d->tokens > 0 ? d->tokens : 1;
  g.alpha = d->alpha; g.accumulate = d->accumulate;
  g.resid = d->resid ? d->resid : d->c_f32;
  g.kchunk = d->K; g.split_stride = 0;
  if (d->split_k > 1) {
    SFRON_CHECK_ARG(d->epilogue == SFRON_EPI_F32 && !d->bias && !d->accumulate && d->split_stride >= (long)d->M * d->ldc_f32);
    g.kchunk = cdiv(cdiv(d->K, d->split_k), BK) * BK;
    g.split_stride = d->split_stride;
  }
  // SFRON_GEMM_NT (A-B knob, default 0): bit 0 = saved-for-backward epilogue outputs (pre-activation, branch output) are
  // stored nontemporally, bit 1 = weight gradients too.  Same-box A-B runs: bit 0 +-0.3 ms/step (noise), bit 1 +0.3 ms: off.
#ifdef SFRON_DEBUG_KNOBS
  static const int nt_mask = [] { const char* e = getenv("SFRON_GEMM_NT"); return e ? atoi(e) : 0; }();
#else
  constexpr int nt_mask = 0;
#endif
  g.nt_out = (nt_mask & 1) | ((nt_mask & 2) && d->a_transposed && d->b_transposed && d->split_k <= 1 && !d->accumulate ? 2 : 0);
  g.ntm = cdiv(d->M, BM); g.ntn = cdiv(d->N, BN);
  g.bsum = nullptr;
  g.colpart = nullptr;
  if (d->col_partials) {      // only the auto-dispatched 256-row pipelined tiles form it (sfron_gemm_dgelu_colpart_rows)
    SFRON_CHECK_ARG((d->epilogue == SFRON_EPI_DGELU || d->epilogue == SFRON_EPI_DGELU_Q) && d->split_k <= 1 && d->tile_hint == 0);
    if (sfron_gemm_dgelu_colpart_rows(d->M, d->N, d->K) == 0) return SFRON_ERR_UNSUPPORTED;
    g.colpart = d->col_partials;
  }
  g.sq_mask = nullptr; g.sq_out = nullptr;
  if (d->sumsq_partials) {  // only the auto-dispatched 192 x 192 weight-gradient tiles form it (sfron_gemm_sumsq_partials)
    SFRON_CHECK_ARG(d->a_transposed && d->b_transposed && d->epilogue == SFRON_EPI_F32 && d->split_k <= 1 && d->tile_hint == 0 && !d->a_rowsum &&
                    !d->accumulate);
    if (sfron_gemm_sumsq_partials(d->M, d->N, d->K) == 0) return SFRON_ERR_UNSUPPORTED;
    g.sq_mask = d->sumsq_mask; g.sq_out = d->sumsq_partials;
  }
  if (d->a_rowsum) {        // only the auto-dispatched three-slot weight-gradient kernel forms it
    SFRON_CHECK_ARG(d->rowsum_ws && d->a_transposed && d->b_transposed && d->epilogue == SFRON_EPI_F32 && d->split_k <= 1 &&
                    d->tile_hint == 0);
    if (!rowsum_ok(d->M, d->N, d->K)) return SFRON_ERR_UNSUPPORTED;
    g.bsum = d->rowsum_ws;                                         // [N / 192][M] partial rows
    const int rc = dispatch_layout<EPI_F32>(1, 1, g, (hipStream_t)stream, 0);
    if (rc != SFRON_OK) return rc;
    return sfron_reduce_chunks(d->rowsum_ws, 1, d->N / 192, d->M, d->a_rowsum, d->M, 0, stream);
  }
  hipStream_t s = (hipStream_t)stream;
  const int force = d->tile_hint;   // 0 auto, -1 generic kernel, 1/2/3 force a fast tile (tests, tuning)
  switch (d->epilogue) {
    case SFRON_EPI_BF16:
      SFRON_CHECK_ARG(g.Cb && g.ldcb % 4 == 0);
      return dispatch_layout<EPI_BF16>(d->a_transposed, d->b_transposed, g, s, force);
    case SFRON_EPI_F32:
      SFRON_CHECK_ARG(g.Cf && g.ldcf % 4 == 0);
      if (!d->a_transposed && !d->b_transposed && force == 0 && skinny_ok(g)) return launch_skinny(g, s);      // at most 32 output rows
      return dispatch_layout<EPI_F32>(d->a_transposed, d->b_transposed, g, s, force);
    case SFRON_EPI_GELU:
      SFRON_CHECK_ARG(g.Cb && g.aux && g.ldcb % 4 == 0 && g.ldaux % 4 == 0 && !d->a_transposed && !d->b_transposed);
      return launch_any<false, false, EPI_GELU>(g, s, force);
    case SFRON_EPI_GATE_RES:
      SFRON_CHECK_ARG(g.Cf && g.aux && g.gate && g.ldcf % 4 == 0 && g.ldaux % 4 == 0 && g.ldgate % 4 == 0 &&
                      !d->a_transposed && !d->b_transposed);
      return launch_any<false, false, EPI_GATE_RES>(g, s, force);
    case SFRON_EPI_DGELU:
      SFRON_CHECK_ARG(g.Cb && g.aux && g.ldcb % 4 == 0 && g.ldaux % 4 == 0 && !d->a_transposed && d->b_transposed);
      return launch_any<false, true, EPI_DGELU>(g, s, force);
    case SFRON_EPI_GELU_Q:
      SFRON_CHECK_ARG(g.Cb && g.aux && g.ldcb % 8 == 0 && g.ldaux % 8 == 0 && !d->a_transposed && !d->b_transposed && d->split_k <= 1 && force == 0);
      SFRON_CHECK_ARG((((uintptr_t)g.aux | (uintptr_t)g.Cb) & 15) == 0);
      if (!sfron_gemm_gelu_q_supported(d->M, d->N, d->K)) return SFRON_ERR_UNSUPPORTED;
      return launch_pipe<4, 2, 4, 6, false, false, EPI_GELUQ, 1>(g, s);
    case SFRON_EPI_DGELU_Q:
      SFRON_CHECK_ARG(g.Cb && g.aux && g.ldcb % 8 == 0 && g.ldaux % 8 == 0 && !d->a_transposed && d->b_transposed && d->split_k <= 1 && force == 0);
      SFRON_CHECK_ARG((((uintptr_t)g.aux | (uintptr_t)g.Cb) & 15) == 0);
      if (!sfron_gemm_gelu_q_supported(d->M, d->N, d->K)) return SFRON_ERR_UNSUPPORTED;
      return launch_pipe<4, 2, 4, 6, false, true, EPI_DGELUQ, 1>(g, s);
    case SFRON_EPI_POS:
      SFRON_CHECK_ARG(g.Cf && g.pos && g.ldcf % 4 == 0 && !d->a_transposed && !d->b_transposed);
      if (force == 0 && shortk_ok(g)) return launch_shortk(g, s);          // the patch embedding: K = 16
      return launch_any<false, false, EPI_POS>(g, s, force);
    default:
      return SFRON_ERR_UNSUPPORTED;
  }
}
