// Probe: how much of a K = 1152 block GEMM is per-tile fixed cost, and what a PERSISTENT grid recovers.  bf16 C[M][N] = A[M][K] B[N][K]^T
// on the pipelined 256 x 128 x 64 three-slot tile (as csrc/conv.hip k_cgemm), in two forms:
//   plain      : one workgroup per output tile (what the library launches)
//   persistent : one workgroup per CU walks its tiles as ONE stream of K-tiles -- the LDS ring never drains at a tile boundary (the next
//                tile's first two K-tiles are in flight during the last two of the current one), and the epilogue's stores are issued
//                without waiting: the counted vmcnt waits of the next two steps allow for them (stores and LDS-DMA loads retire in order)
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value tools/probes/persist_gemm_probe.hip -o tools/probes/persist_gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int FBM = 256, BK = 64, NSLOT = 3;
// WM x WN waves, each MT x NT tiles of 16 x 16: the wave tile (16 MT x 16 NT) sets the LDS bytes read per FLOP, (MT + NT) / (MT NT)
template <int WM, int WN, int MT, int NT_> struct Cfg {
  static_assert(WM * MT * 16 == FBM, "256 rows");
  static constexpr int NW = WM * WN, FBN = WN * NT_ * 16, A_EL = FBM * BK, B_EL = FBN * BK, SLOT = A_EL + B_EL;
  static constexpr int NA = FBM * 8 / 64 / NW, NB_TOT = FBN * 8 / 64, NB = (NB_TOT + NW - 1) / NW;
  static constexpr bool EVEN = NB_TOT % NW == 0;
  static constexpr int NDMA = NA + NB, NST = MT * NT_;
  static constexpr size_t LDS = (size_t)NSLOT * SLOT * 2 + (EVEN ? 0 : 1024);
};
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, __bf16* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t*)dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ bf16x8 frag(const __bf16* img, int row, int kchunk) {
  return *reinterpret_cast<const bf16x8*>(img + row * 64 + ((kchunk ^ (row & 7)) << 3));
}

template <int WM, int WN, int MT, int NT_, bool PERSIST, int GM = 0>
__global__ __launch_bounds__(WM * WN * 64) void k_gemm(const __bf16* __restrict__ A, const __bf16* __restrict__ B, __bf16* __restrict__ C, int M, int N, int K) {
  using CT = Cfg<WM, WN, MT, NT_>;
  constexpr int NW = CT::NW;
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = N / CT::FBN, ntiles = (M / FBM) * ntn, nk = K / BK;
  int id;
  {
    const int nblk = gridDim.x, b = blockIdx.x, q = nblk >> 3, r = nblk & 7, x = b & 7, y = b >> 3;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  const int n_my = PERSIST ? (ntiles - id + (int)gridDim.x - 1) / (int)gridDim.x : 1;      // tiles id, id + G, id + 2G, ...
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, M * K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, N * K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0, 0x00020000);
  __bf16* dummy = smem + NSLOT * CT::SLOT;
  const int lc8 = ((lane & 7) ^ ((lane >> 3) & 7)) << 3;
  int a_off[CT::NA], b_off[CT::NB];                       // per-lane constants; the tile and the K step travel in the scalar offset
#pragma unroll
  for (int i = 0; i < CT::NA; ++i) a_off[i] = 2 * (((wave + i * NW) * 8 + (lane >> 3)) * K + lc8);
#pragma unroll
  for (int i = 0; i < CT::NB; ++i) b_off[i] = 2 * (((wave + i * NW) * 8 + (lane >> 3)) * K + lc8);
  auto tile_of = [&](int ti, int& m0, int& n0) {
    const int t = id + ti * (int)gridDim.x;
    int tm = t / ntn, tn = t - tm * ntn;
    if (GM > 0) {            // grouped order: ids sweep GM tile-rows x all columns, so the tiles an XCD runs at once form a 2-D block
      const int ntm = M / FBM, per = GM * ntn, grp = t / per, rem = t - grp * per, gsz = min(GM, ntm - grp * GM);
      tm = grp * GM + rem % gsz; tn = rem / gsz;
    }
    m0 = tm * FBM; n0 = tn * CT::FBN;
  };
  auto issue = [&](int slot, int ti, int kt) {
    int m0, n0; tile_of(ti, m0, n0);
    __bf16* iA = smem + slot * CT::SLOT;
    __bf16* iB = iA + CT::A_EL;
    const int sa = 2 * (m0 * K + kt * BK), sb = 2 * (n0 * K + kt * BK);
#pragma unroll
    for (int i = 0; i < CT::NA; ++i) dma16(rsA, iA + (wave + i * NW) * 512, a_off[i], sa);
#pragma unroll
    for (int i = 0; i < CT::NB; ++i) {
      if (CT::EVEN || i + 1 < CT::NB) dma16(rsB, iB + (wave + i * NW) * 512, b_off[i], sb);
      else { const bool ok = wave + i * NW < CT::NB_TOT; dma16(ok ? rsB : rs0, ok ? iB + (wave + i * NW) * 512 : dummy, b_off[i], sb); }
    }
  };
  f32x4 acc[MT][NT_];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](int slot) {
    const __bf16* iA = smem + slot * CT::SLOT;
    const __bf16* iB = iA + CT::A_EL;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = frag(iA, wm * MT * 16 + mt * 16 + fr, ks * 4 + fg);
#pragma unroll
      for (int nt = 0; nt < NT_; ++nt) {
        const bf16x8 fb = frag(iB, wn * NT_ * 16 + nt * 16 + fr, ks * 4 + fg);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[mt], acc[mt][nt], 0, 0, 0);
      }
    }
  };
  const int S = n_my * nk;
  // step s = (tile s / nk, K-tile s % nk); (ti, kt) of the step being issued run two steps ahead
  int i_ti = 0, i_kt = 0;
  auto issue_next = [&](int slot) { issue(slot, i_ti, i_kt); if (++i_kt == nk) { i_kt = 0; ++i_ti; } };
  if (S > 0) issue_next(0);
  if (S > 1) issue_next(1);
  int slot = 0, ti = 0, kt = 0;
  for (int s = 0; s < S; ++s) {
    // DMA(s) has landed: younger operations are DMA(s + 1) and the stores of a tile that ended one or two steps ago
    const bool st = PERSIST && ti > 0 && kt < 2;
    if (s + 1 < S) { if (st) wait_vmcnt<CT::NDMA + CT::NST>(); else wait_vmcnt<CT::NDMA>(); }
    else           { if (st) wait_vmcnt<CT::NST>(); else wait_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();
    if (s + 2 < S) issue_next(slot >= 1 ? slot - 1 : 2);
    compute(slot);
    slot = slot == 2 ? 0 : slot + 1;
    if (++kt == nk) {
      int m0, n0; tile_of(ti, m0, n0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = m0 + wm * MT * 16 + mt * 16 + fr;
#pragma unroll
        for (int nt = 0; nt < NT_; ++nt) {
          const f32x4 v = acc[mt][nt];
          *reinterpret_cast<bf16x4*>(C + (size_t)row * N + n0 + wn * NT_ * 16 + nt * 16 + 4 * fg) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
          acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      kt = 0; ++ti;
    }
  }
}

// ---- four waves, one per SIMD, 64 x (NT_*16) each: half the LDS reads of the 8-wave forms.  A single wave per SIMD has nobody to hide
// its LDS latency, so the loop is software-pipelined by hand: the fragments of the NEXT K-step are read while the MFMAs of the
// current one issue, and the tile hand-over (vmcnt wait, barrier, next DMA) sits between the two K-steps of a tile.
template <int NT_, int NS>
__global__ __launch_bounds__(256) void k_gemm4(const __bf16* __restrict__ A, const __bf16* __restrict__ B, __bf16* __restrict__ C, int M, int N, int K) {
  constexpr int NW = 4, MT = 4, FBN = NT_ * 16, A_EL = FBM * BK, B_EL = FBN * BK, SLOT = A_EL + B_EL;
  constexpr int NA = FBM * 8 / 64 / NW, NB_TOT = FBN * 8 / 64, NB = (NB_TOT + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N / FBN, nk = K / BK;
  int id;
  {
    const int nblk = gridDim.x, b = blockIdx.x, q = nblk >> 3, r = nblk & 7, x = b & 7, y = b >> 3;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  const int tm = id / ntn, m0 = tm * FBM, n0 = (id - tm * ntn) * FBN;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, M * K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, N * K * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0, 0x00020000);
  __bf16* dummy = smem + NS * SLOT;
  const int lc8 = ((lane & 7) ^ ((lane >> 3) & 7)) << 3;
  int a_off[NA], b_off[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) a_off[i] = 2 * ((m0 + (wave + i * NW) * 8 + (lane >> 3)) * K + lc8);
#pragma unroll
  for (int i = 0; i < NB; ++i) b_off[i] = 2 * ((n0 + (wave + i * NW) * 8 + (lane >> 3)) * K + lc8);
  auto issue = [&](int slot, int kt) {
    __bf16* iA = smem + slot * SLOT;
    __bf16* iB = iA + A_EL;
#pragma unroll
    for (int i = 0; i < NA; ++i) dma16(rsA, iA + (wave + i * NW) * 512, a_off[i], 2 * kt * BK);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const bool ok = wave + i * NW < NB_TOT;
      dma16(ok ? rsB : rs0, ok ? iB + (wave + i * NW) * 512 : dummy, b_off[i], 2 * kt * BK);
    }
  };
  f32x4 acc[MT][NT_];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  struct Frags { bf16x8 a[MT], b[NT_]; };
  auto read = [&](Frags& f, int slot, int ks) {
    const __bf16* iA = smem + slot * SLOT;
    const __bf16* iB = iA + A_EL;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) f.a[mt] = frag(iA, wave * 64 + mt * 16 + fr, ks * 4 + fg);
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt) f.b[nt] = frag(iB, nt * 16 + fr, ks * 4 + fg);
  };
  auto mfmas = [&](const Frags& f) {
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        // in place, accumulators pinned to AGPRs: through the builtin hipcc renames every accumulator of this straight-line body
        // (four v_accvgpr_mov + s_nop per MFMA)
        f32x4& c = acc[mt][nt];
        const bf16x8 fb = f.b[nt], fa = f.a[mt];
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(fb), "v"(fa));
      }
  };
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  if (nk > 1) wait_vmcnt<NA + NB>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  Frags f0, f1;
  read(f0, 0, 0);
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    read(f1, slot, 1);                                   // K-step 1 of this tile under the MFMAs of K-step 0
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f0);
    __builtin_amdgcn_sched_barrier(0);
    const int nslot = slot == NS - 1 ? 0 : slot + 1;
    if (kt + 1 < nk) {
      wait_vmcnt<0>();                                    // tile kt + 1 (the only DMA in flight) has landed
      // two slots: tile kt + 2 goes into THIS tile's slot -- legal once every wave holds both K-steps of tile kt in registers
      if (NS == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                       // ... for every wave, and every wave is done with tile kt - 1's slot
      if (kt + 2 < nk) issue(NS == 2 ? slot : (slot >= 1 ? slot - 1 : 2), kt + 2);
      read(f0, nslot, 0);                                 // K-step 0 of the next tile under the MFMAs of K-step 1
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f1);
    __builtin_amdgcn_sched_barrier(0);
    slot = nslot;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = m0 + wave * 64 + mt * 16 + fr;
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt) {
      const f32x4 v = acc[mt][nt];
      *reinterpret_cast<bf16x4*>(C + (size_t)row * N + n0 + nt * 16 + 4 * fg) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
  }
}


// ---- four CONSUMER waves (one per SIMD, 64 x (NT_*16) each, as k_gemm4) + four LOADER waves that issue every LDS-DMA piece of the ring:
// MI355X_MICROARCH.md prices one 1-KiB LDS-DMA piece at 60-185 cycles of the ISSUING wave (6-7 pieces per wave and K-step in the
// kernels above, against 576 cycles of MFMA per wave and K-step) -- here the waves that multiply never issue one.  One s_barrier per
// K-tile for all eight waves: a loader arrives once its share of tile j has landed (counted vmcnt), a consumer once tile j - 1's
// fragments are in its registers; after it the loaders refill tile j - 1's slot with tile j + 2.
template <int NT_, int MODE = 0>
__global__ __launch_bounds__(512) void k_gemm4l(const __bf16* __restrict__ A, const __bf16* __restrict__ B, __bf16* __restrict__ C, int M, int N, int K) {
  constexpr int NW = 4, MT = 4, NS = 3, FBN = NT_ * 16, A_EL = FBM * BK, B_EL = FBN * BK, SLOT = A_EL + B_EL;
  constexpr int NA = FBM * 8 / 64 / NW, NB_TOT = FBN * 8 / 64, NB = (NB_TOT + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave8 >= 4;
  const int wave = wave8 & 3;
  const int ntn = N / FBN, nk = K / BK;
  int id;
  {
    const int nblk = gridDim.x, b = blockIdx.x, q = nblk >> 3, r = nblk & 7, x = b & 7, y = b >> 3;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  const int tm = id / ntn, m0 = tm * FBM, n0 = (id - tm * ntn) * FBN;
  if (loader) {
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, M * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, N * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0, 0x00020000);
    __bf16* dummy = smem + NS * SLOT;
    const int lc8 = ((lane & 7) ^ ((lane >> 3) & 7)) << 3;
    int a_off[NA], b_off[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) a_off[i] = 2 * ((m0 + (wave + i * NW) * 8 + (lane >> 3)) * K + lc8);
#pragma unroll
    for (int i = 0; i < NB; ++i) b_off[i] = 2 * ((n0 + (wave + i * NW) * 8 + (lane >> 3)) * K + lc8);
    auto issue = [&](int slot, int kt) {
      __bf16* iA = smem + slot * SLOT;
      __bf16* iB = iA + A_EL;
#pragma unroll
      for (int i = 0; i < NA; ++i) dma16(rsA, iA + (wave + i * NW) * 512, a_off[i], 2 * kt * BK);
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const bool ok = wave + i * NW < NB_TOT;
        dma16(ok ? rsB : rs0, ok ? iB + (wave + i * NW) * 512 : dummy, b_off[i], 2 * kt * BK);
      }
    };
    if (MODE != 2) { issue(0, 0); if (nk > 1) issue(1, 1); }
    int slot2 = 2;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vmcnt<NA + NB>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (MODE != 2 && kt + 2 < nk) issue(slot2, kt + 2);
      slot2 = slot2 == 2 ? 0 : slot2 + 1;
    }
    return;
  }
  f32x4 acc[MT][NT_];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  struct Frags { bf16x8 a[MT], b[NT_]; };
  auto read = [&](Frags& f, int slot, int ks) {
    const __bf16* iA = smem + slot * SLOT;
    const __bf16* iB = iA + A_EL;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) f.a[mt] = frag(iA, wave * 64 + mt * 16 + fr, ks * 4 + fg);
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt) f.b[nt] = frag(iB, nt * 16 + fr, ks * 4 + fg);
  };
  auto mfmas = [&](const Frags& f) {
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4& c = acc[mt][nt];
        const bf16x8 fb = f.b[nt], fa = f.a[mt];
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(fb), "v"(fa));
      }
  };
  __builtin_amdgcn_s_barrier();                            // tile 0 is in LDS
  Frags f0, f1;
  if (MODE == 1) {                                         // DMA only: the consumers just keep the barrier count
    for (int kt = 1; kt < nk; ++kt) __builtin_amdgcn_s_barrier();
  } else {
  read(f0, 0, 0);
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    read(f1, slot, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f0);
    __builtin_amdgcn_sched_barrier(0);
    const int nslot = slot == NS - 1 ? 0 : slot + 1;
    if (kt + 1 < nk) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // both K-steps of tile kt are in registers: its slot may be refilled
      __builtin_amdgcn_s_barrier();                            // tile kt + 1 is in LDS
      read(f0, nslot, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfmas(f1);
    __builtin_amdgcn_sched_barrier(0);
    slot = nslot;
  }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = m0 + wave * 64 + mt * 16 + fr;
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt) {
      const f32x4 v = acc[mt][nt];
      *reinterpret_cast<bf16x4*>(C + (size_t)row * N + n0 + nt * 16 + 4 * fg) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
  }
}

// ---- eight CONSUMER waves (32 x (NT_*16) each, compiler-scheduled as the plain k_gemm<8,1,2,NT_>) + four LOADER waves: three waves per SIMD
// (<= 168 registers), the two consumers of a SIMD hide each other's LDS latency, the loader of the SIMD issues a quarter of the DMA pieces
template <int NT_, int MODE = 0, int NWL = 4, int WN = 1>
__global__ __launch_bounds__(512 + 64 * NWL) void k_gemm8l(const __bf16* __restrict__ A, const __bf16* __restrict__ B, __bf16* __restrict__ C, int M, int N, int K) {
  constexpr int WM = 8 / WN, MT = FBM / 16 / WM, NS = 3, FBN = WN * NT_ * 16, A_EL = FBM * BK, B_EL = FBN * BK, SLOT = A_EL + B_EL;   // NT_ = n-tiles per wave
  constexpr int NA = FBM * 8 / 64 / NWL, NB_TOT = FBN * 8 / 64, NB = (NB_TOT + NWL - 1) / NWL;
  extern __shared__ __attribute__((aligned(16))) __bf16 smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N / FBN, nk = K / BK;
  int id;
  {
    const int nblk = gridDim.x, b = blockIdx.x, q = nblk >> 3, r = nblk & 7, x = b & 7, y = b >> 3;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  const int tm = id / ntn, m0 = tm * FBM, n0 = (id - tm * ntn) * FBN;
  if (wave12 >= 8) {
    const int wave = wave12 - 8;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, M * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, N * K * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, 0, 0x00020000);
    __bf16* dummy = smem + NS * SLOT;
    const int lc8 = ((lane & 7) ^ ((lane >> 3) & 7)) << 3;
    int a_off[NA], b_off[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) a_off[i] = 2 * ((m0 + (wave + i * NWL) * 8 + (lane >> 3)) * K + lc8);
#pragma unroll
    for (int i = 0; i < NB; ++i) b_off[i] = 2 * ((n0 + (wave + i * NWL) * 8 + (lane >> 3)) * K + lc8);
    auto issue = [&](int slot, int kt) {
      __bf16* iA = smem + slot * SLOT;
      __bf16* iB = iA + A_EL;
#pragma unroll
      for (int i = 0; i < NA; ++i) dma16(rsA, iA + (wave + i * NWL) * 512, a_off[i], 2 * kt * BK);
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const bool ok = wave + i * NWL < NB_TOT;
        dma16(ok ? rsB : rs0, ok ? iB + (wave + i * NWL) * 512 : dummy, b_off[i], 2 * kt * BK);
      }
    };
    if (MODE != 2) { issue(0, 0); if (nk > 1) issue(1, 1); }
    int slot2 = 2;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vmcnt<NA + NB>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (MODE != 2 && kt + 2 < nk) issue(slot2, kt + 2);
      slot2 = slot2 == 2 ? 0 : slot2 + 1;
    }
    return;
  }
  const int wm = wave12 / WN, wn = wave12 % WN;
  f32x4 acc[MT][NT_];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_barrier();                          // tile kt is in LDS (and this wave's reads of tile kt - 1 were waited for by its MFMAs)
    const __bf16* iA = smem + slot * SLOT;
    const __bf16* iB = iA + A_EL;
    if (MODE != 1)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = frag(iA, wm * MT * 16 + mt * 16 + fr, ks * 4 + fg);
#pragma unroll
      for (int nt = 0; nt < NT_; ++nt) {
        const bf16x8 fb = frag(iB, wn * NT_ * 16 + nt * 16 + fr, ks * 4 + fg);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa[mt], acc[mt][nt], 0, 0, 0);
      }
    }
    slot = slot == NS - 1 ? 0 : slot + 1;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = m0 + wm * MT * 16 + mt * 16 + fr;
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt) {
      const f32x4 v = acc[mt][nt];
      *reinterpret_cast<bf16x4*>(C + (size_t)row * N + n0 + wn * NT_ * 16 + nt * 16 + 4 * fg) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
  }
}

static uint16_t f2bf(float f) { uint32_t u; std::memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

template <int WM, int WN, int MT, int NT_>
void run(const char* name, int M, int N, int K) {
  using CT = Cfg<WM, WN, MT, NT_>;
  if (N % CT::FBN || CT::LDS > 160 * 1024) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<WM, WN, MT, NT_, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT::LDS);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<WM, WN, MT, NT_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT::LDS);
  std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
  uint32_t st = 777u + N + K;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  for (auto& v : hA) v = f2bf((float)((int)(rnd() % 5) - 2));             // small integers: exact products and sums
  for (auto& v : hB) v = f2bf((float)((int)(rnd() % 5) - 2));
  __bf16 *dA, *dB, *dC;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  const int ntiles = (M / FBM) * (N / CT::FBN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<uint16_t> hC((size_t)M * N);
  for (int persist = 0; persist < 2; ++persist) {
    const dim3 grid(persist ? (ntiles < 256 ? ntiles : 256) : ntiles);
    auto launch = [&]() {
      if (persist) hipLaunchKernelGGL((k_gemm<WM, WN, MT, NT_, true>), grid, dim3(CT::NW * 64), CT::LDS, 0, dA, dB, dC, M, N, K);
      else         hipLaunchKernelGGL((k_gemm<WM, WN, MT, NT_, false>), grid, dim3(CT::NW * 64), CT::LDS, 0, dA, dB, dC, M, N, K);
    };
    hipMemset(dC, 0xff, (size_t)M * N * 2);
    launch(); hipDeviceSynchronize();
    hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 6000; ++t) {
      const int r = rnd() % M, c = rnd() % N;
      float ref = 0.f;
      for (int k = 0; k < K; ++k) ref += bf2f(hA[(size_t)r * K + k]) * bf2f(hB[(size_t)c * K + k]);
      if (hC[(size_t)r * N + c] != f2bf(ref)) ++bad;
    }
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0);
    const int reps = 30;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("%-16s tile 256x%-3d waves %dx%d (%3dx%-3d each) %-10s grid %4d: %7.1f us  %7.1f TFLOP/s   exact-integer check: %d / 6000 wrong\n", name, CT::FBN, WM, WN, MT * 16, NT_ * 16,
           persist ? "persistent" : "plain", (int)grid.x, ms * 1e3, 2.0 * M * N * K / ms / 1e9, bad);
  }
  hipFree(dA); hipFree(dB); hipFree(dC);
}

template <int GM>
void run_grouped(const char* name, int M, int N, int K) {
  using CT = Cfg<8, 1, 2, 9>;
  if (N % CT::FBN) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm<8, 1, 2, 9, false, GM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT::LDS);
  std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
  uint32_t st = 99u + N + K;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  for (auto& v : hA) v = f2bf((float)((int)(rnd() % 5) - 2));
  for (auto& v : hB) v = f2bf((float)((int)(rnd() % 5) - 2));
  __bf16 *dA, *dB, *dC;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  const dim3 grid((M / FBM) * (N / CT::FBN));
  auto launch = [&]() { hipLaunchKernelGGL((k_gemm<8, 1, 2, 9, false, GM>), grid, dim3(512), CT::LDS, 0, dA, dB, dC, M, N, K); };
  launch(); hipDeviceSynchronize();
  std::vector<uint16_t> hC((size_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 6000; ++t) {
    const int r = rnd() % M, c = rnd() % N;
    float ref = 0.f;
    for (int k = 0; k < K; ++k) ref += bf2f(hA[(size_t)r * K + k]) * bf2f(hB[(size_t)c * K + k]);
    if (hC[(size_t)r * N + c] != f2bf(ref)) ++bad;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e0);
  for (int i = 0; i < 30; ++i) launch();
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 30;
  printf("%-16s tile 256x144 waves 8x1 grouped order GM=%d grid %4d: %7.1f us  %7.1f TFLOP/s   exact-integer check: %d / 6000 wrong\n", name, GM,
         (int)grid.x, ms * 1e3, 2.0 * M * N * K / ms / 1e9, bad);
  hipFree(dA); hipFree(dB); hipFree(dC);
}

template <int NT_, int NS>
void run4(const char* name, int M, int N, int K) {
  constexpr int FBN = NT_ * 16;
  const size_t lds = (size_t)NS * (FBM + FBN) * BK * 2 + 1024;
  if (N % FBN || lds > 160 * 1024) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm4<NT_, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
  uint32_t st = 4242u + N + K;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  for (auto& v : hA) v = f2bf((float)((int)(rnd() % 5) - 2));
  for (auto& v : hB) v = f2bf((float)((int)(rnd() % 5) - 2));
  __bf16 *dA, *dB, *dC;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  const dim3 grid((M / FBM) * (N / FBN));
  auto launch = [&]() { hipLaunchKernelGGL((k_gemm4<NT_, NS>), grid, dim3(256), lds, 0, dA, dB, dC, M, N, K); };
  hipMemset(dC, 0xff, (size_t)M * N * 2);
  launch(); hipDeviceSynchronize();
  std::vector<uint16_t> hC((size_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 6000; ++t) {
    const int r = rnd() % M, c = rnd() % N;
    float ref = 0.f;
    for (int k = 0; k < K; ++k) ref += bf2f(hA[(size_t)r * K + k]) * bf2f(hB[(size_t)c * K + k]);
    if (hC[(size_t)r * N + c] != f2bf(ref)) ++bad;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e0);
  const int reps = 30;
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("%-16s tile 256x%-3d waves 4x1 ( 64x%-3d each) hand-piped %d slots grid %4d: %7.1f us  %7.1f TFLOP/s   exact-integer check: %d / 6000 wrong\n", name, FBN,
         FBN, NS, (int)grid.x, ms * 1e3, 2.0 * M * N * K / ms / 1e9, bad);
  hipFree(dA); hipFree(dB); hipFree(dC);
}

template <int NT_, int FORM, int MODE = 0, int NWL = 4, int WN = 1>
void run4l(const char* name, int M, int N, int K) {
  constexpr int FBN = WN * NT_ * 16;
  const size_t lds = (size_t)3 * (FBM + FBN) * BK * 2 + 1024;
  if (N % FBN || lds > 160 * 1024) return;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm4l<NT_, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm8l<NT_, MODE, NWL, WN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
  uint32_t st = 31337u + N + K;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
  for (auto& v : hA) v = f2bf((float)((int)(rnd() % 5) - 2));
  for (auto& v : hB) v = f2bf((float)((int)(rnd() % 5) - 2));
  __bf16 *dA, *dB, *dC;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, (size_t)M * N * 2);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  const dim3 grid((M / FBM) * (N / FBN));
  auto launch = [&]() {
    if (FORM == 4) hipLaunchKernelGGL((k_gemm4l<NT_, MODE>), grid, dim3(512), lds, 0, dA, dB, dC, M, N, K);
    else hipLaunchKernelGGL((k_gemm8l<NT_, MODE, NWL, WN>), grid, dim3(512 + 64 * NWL), lds, 0, dA, dB, dC, M, N, K);
  };
  hipMemset(dC, 0xff, (size_t)M * N * 2);
  launch(); hipDeviceSynchronize();
  std::vector<uint16_t> hC((size_t)M * N);
  hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 6000; ++t) {
    const int r = rnd() % M, c = rnd() % N;
    float ref = 0.f;
    for (int k = 0; k < K; ++k) ref += bf2f(hA[(size_t)r * K + k]) * bf2f(hB[(size_t)c * K + k]);
    if (hC[(size_t)r * N + c] != f2bf(ref)) ++bad;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e0);
  const int reps = 30;
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("%-16s tile 256x%-3d %d consumer waves (%d x %d) + %d loader waves (mode %d: 1 = DMA only, 2 = compute only), 3 slots grid %4d: %7.1f us  %7.1f TFLOP/s   exact-integer check: %d / 6000 wrong\n", name, FBN,
         FORM, 8 / WN, WN, NWL, MODE, (int)grid.x, ms * 1e3, 2.0 * M * N * K / ms / 1e9, bad);
  hipFree(dA); hipFree(dB); hipFree(dC);
}

int main() {
  const int M = 8192;
  // (A) the block GEMMs of DiT-XL/2: shared-wave forms (plain / persistent), loader-wave forms and their floors (mode 1 = loaders only,
  //     mode 2 = consumers only), 2 / 4 / 8 loader waves, four hand-pipelined consumer waves at 256 x 128
  struct Shape { const char* name; int N, K; } shapes[] = {{"qkv  1152->3456", 3456, 1152}, {"proj 1152->1152", 1152, 1152},
                                                           {"fc1  1152->4608", 4608, 1152}, {"fc2  4608->1152", 1152, 4608}};
  for (const Shape& s : shapes) {
    run<8, 1, 2, 9>(s.name, M, s.N, s.K);
    run4<9, 3>(s.name, M, s.N, s.K);
    run4l<9, 8>(s.name, M, s.N, s.K); run4l<9, 8, 1>(s.name, M, s.N, s.K); run4l<9, 8, 2>(s.name, M, s.N, s.K);
    run4l<9, 8, 0, 2>(s.name, M, s.N, s.K); run4l<9, 8, 0, 8>(s.name, M, s.N, s.K);
    run4l<8, 4>(s.name, M, s.N, s.K); run4l<8, 4, 2>(s.name, M, s.N, s.K);
  }
  // (B) wave layout of the consumers, 256 x 160 (N a multiple of 160): 8 x 1 waves of 32 x 160 against 4 x 2 of 64 x 80 (3/4 of the LDS reads)
  Shape shapes160[] = {{"qkv-like 1152->3520", 3520, 1152}, {"proj-like 1152->1280", 1280, 1152}, {"fc2-like 4608->1280", 1280, 4608}};
  for (const Shape& s : shapes160) {
    run<8, 1, 2, 10>(s.name, M, s.N, s.K); run<4, 2, 4, 5>(s.name, M, s.N, s.K);
    run4l<10, 8, 0, 4, 1>(s.name, M, s.N, s.K); run4l<5, 8, 0, 4, 2>(s.name, M, s.N, s.K);
    run4l<10, 8, 2, 4, 1>(s.name, M, s.N, s.K); run4l<5, 8, 2, 4, 2>(s.name, M, s.N, s.K);
  }
  return 0;
}
