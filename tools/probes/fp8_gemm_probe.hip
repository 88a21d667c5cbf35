// Probe for BASELINE config 5 (fp8 operands on the CDNA4 fp8 MFMA): what does the pipelined tile of csrc/conv.hip (k_cgemm: 256 x 128,
// three LDS slots, LDS-DMA staging with the XOR swizzle on the source address) reach on the DiT-XL/2 block shapes when BOTH operands are
// fp8 e4m3 (1 byte: a 128-byte LDS row holds 128 k-values instead of 64) and the products run on v_mfma_scale_f32_16x16x128_f8f6f4
// (unit block scales)?  C[M][N] (bf16) = A[M][K] . B[N][K]^T, fp32 accumulation.  Not part of the library: there is no quantising
// producer behind it -- it answers "how fast would the GEMMs be", not "how accurate is the step".
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/fp8_gemm_probe.hip -o tools/probes/fp8_gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((address_space(3))) void lptr_t;

constexpr int FBM = 256, FBN = 128, BKB = 128, NW = 8, NSLOT = 3, NT_ = FBN / 16;
constexpr int A_BYTES = FBM * BKB, B_BYTES = FBN * BKB, SLOT = A_BYTES + B_BYTES;
constexpr int NA = FBM * 8 / 64 / NW, NB = FBN * 8 / 64 / NW, NDMA = NA + NB;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, uint8_t* dst, int voff, int soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t*)dst, 16, voff, soff, 0, 0);
}
// 32 consecutive k-bytes of row `row`, k group g (k = 32 g .. 32 g + 31): chunks 2g and 2g + 1 of the 128-byte row, chunk ^= row & 7
__device__ __forceinline__ i32x8 frag32(const uint8_t* img, int row, int g) {
  const i32x4 lo = *reinterpret_cast<const i32x4*>(img + row * 128 + (((2 * g) ^ (row & 7)) << 4));
  const i32x4 hi = *reinterpret_cast<const i32x4*>(img + row * 128 + (((2 * g + 1) ^ (row & 7)) << 4));
  return i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int DBG, bool BF16>
__global__ __launch_bounds__(512) void k_gemm8(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B, __bf16* __restrict__ C, int M, int N,
                                               int K) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = N / FBN;
  int id;
  {
    const int nblk = gridDim.x, b = blockIdx.x, q = nblk >> 3, r = nblk & 7, x = b & 7, y = b >> 3;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
  }
  const int tm = id / ntn, tn = id - tm * ntn, m0 = tm * FBM, n0 = tn * FBN;
  const int nk = K / BKB;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, M * K, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)B, 0, N * K, 0x00020000);
  const int lc16 = ((lane & 7) ^ ((lane >> 3) & 7)) << 4;
  int a_off[NA], b_off[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) a_off[i] = (m0 + (wave + i * NW) * 8 + (lane >> 3)) * K + lc16;
#pragma unroll
  for (int i = 0; i < NB; ++i) b_off[i] = (n0 + (wave + i * NW) * 8 + (lane >> 3)) * K + lc16;
  auto issue = [&](int slot, int k0) {
    uint8_t* iA = smem + slot * SLOT;
    uint8_t* iB = iA + A_BYTES;
#pragma unroll
    for (int i = 0; i < NA; ++i) dma16(rsA, iA + (wave + i * NW) * 1024, a_off[i], k0);
#pragma unroll
    for (int i = 0; i < NB; ++i) dma16(rsB, iB + (wave + i * NW) * 1024, b_off[i], k0);
  };
  f32x4 acc[2][NT_];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT_; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](int slot) {
    const uint8_t* iA = smem + slot * SLOT;
    const uint8_t* iB = iA + A_BYTES;
    const i32x8 fa0 = frag32(iA, wave * 32 + fr, fg), fa1 = frag32(iA, wave * 32 + 16 + fr, fg);
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt) {
      const i32x8 fb = frag32(iB, nt * 16 + fr, fg);
      if (DBG == 2) { acc[0][nt][0] += (float)(fb[0] ^ fa0[1]); acc[1][nt][0] += (float)(fb[2] ^ fa1[3]); continue; }
      if constexpr (BF16) {
        // the same 128-byte rows read as 64 bf16: two k-steps of 32 (bytes 16 g' .. of each half), i.e. the bf16 kernel's work per tile
        typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
        const i32x4 b0 = {fb[0], fb[1], fb[2], fb[3]}, b1 = {fb[4], fb[5], fb[6], fb[7]};
        const i32x4 a00 = {fa0[0], fa0[1], fa0[2], fa0[3]}, a01 = {fa0[4], fa0[5], fa0[6], fa0[7]};
        const i32x4 a10 = {fa1[0], fa1[1], fa1[2], fa1[3]}, a11 = {fa1[4], fa1[5], fa1[6], fa1[7]};
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((bf16x8)b0, (bf16x8)a00, acc[0][nt], 0, 0, 0);
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((bf16x8)b1, (bf16x8)a01, acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((bf16x8)b0, (bf16x8)a10, acc[1][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((bf16x8)b1, (bf16x8)a11, acc[1][nt], 0, 0, 0);
      } else {
        // cbsz = blgp = 0: both operands e4m3; block scales 2^(127 - 127) = 1
        acc[0][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa0, acc[0][nt], 0, 0, 0, 127, 0, 127);
        acc[1][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb, fa1, acc[1][nt], 0, 0, 0, 127, 0, 127);
      }
    }
  };
  if (nk > 0) issue(0, 0);
  if (nk > 1) issue(1, BKB);
  if (DBG == 3 || DBG == 4) { wait_vmcnt<0>(); }
  int slot = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (DBG == 3 || DBG == 4) {} else if (kt + 1 < nk) wait_vmcnt<NDMA>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk && DBG != 3 && DBG != 4) issue(slot >= 1 ? slot - 1 : 2, (kt + 2) * BKB);
    if (DBG != 4) compute(slot);
    slot = slot == 2 ? 0 : slot + 1;
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = m0 + wave * 32 + mt * 16 + fr;
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt) {
      const int col = n0 + nt * 16 + 4 * fg;
      const f32x4 v = acc[mt][nt];
      if (DBG == 1 && v[0] != 12345.678f) continue;
      *reinterpret_cast<bf16x4*>(C + (size_t)row * N + col) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
  }
}

static float e4m3_to_f(uint8_t b) {            // OCP e4m3fn
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? m / 8.0f * 0.015625f : (1.0f + m / 8.0f) * (float)(1 << e) / 128.0f;
  return s ? -v : v;
}
static float bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; std::memcpy(&f, &u, 4); return f; }

int main() {
  const int M = 8192;
  struct Shape { const char* name; int N, K; } shapes[] = {{"qkv  1152->3456", 3456, 1152}, {"proj 1152->1152", 1152, 1152},
                                                           {"fc1  1152->4608", 4608, 1152}, {"fc2  4608->1152", 1152, 4608}, {"8192^3", 8192, 8192}};
  const size_t lds = (size_t)NSLOT * SLOT;
  auto attr = [&](const void* f) { (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); };
  attr(reinterpret_cast<const void*>(&k_gemm8<0, false>)); attr(reinterpret_cast<const void*>(&k_gemm8<1, false>));
  attr(reinterpret_cast<const void*>(&k_gemm8<2, false>)); attr(reinterpret_cast<const void*>(&k_gemm8<3, false>));
  attr(reinterpret_cast<const void*>(&k_gemm8<4, false>)); attr(reinterpret_cast<const void*>(&k_gemm8<0, true>));
  attr(reinterpret_cast<const void*>(&k_gemm8<1, true>)); attr(reinterpret_cast<const void*>(&k_gemm8<3, true>));
  // small-integer fp8 codes: exact products, exact sums in fp32 -> the fragment maps are checked bit for bit
  const uint8_t codes[5] = {0x00, 0x38, 0x40, 0xB8, 0xC0};       // 0, 1, 2, -1, -2
  for (const Shape& s : shapes) {
    const int N = s.N, K = s.K;
    std::vector<uint8_t> hA((size_t)M * K), hB((size_t)N * K);
    uint32_t st = 12345u + N + K;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
    for (auto& v : hA) v = codes[rnd() % 5];
    for (auto& v : hB) v = codes[rnd() % 5];
    uint8_t *dA, *dB; __bf16* dC;
    hipMalloc(&dA, hA.size()); hipMalloc(&dB, hB.size()); hipMalloc(&dC, (size_t)M * N * 2);
    hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size(), hipMemcpyHostToDevice);
    const dim3 grid((M / FBM) * (N / FBN));
    hipLaunchKernelGGL((k_gemm8<0, false>), grid, dim3(512), lds, 0, dA, dB, dC, M, N, K);
    hipDeviceSynchronize();
    std::vector<uint16_t> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 4000; ++t) {
      const int r = rnd() % M, c = rnd() % N;
      float ref = 0.f;
      for (int k = 0; k < K; ++k) ref += e4m3_to_f(hA[(size_t)r * K + k]) * e4m3_to_f(hB[(size_t)c * K + k]);
      // bf16 output: |ref| <= 4 K is an integer; compare after the same rounding
      uint32_t u; std::memcpy(&u, &ref, 4); u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; float rr; std::memcpy(&rr, &u, 4);
      if (bf16_to_f(hC[(size_t)r * N + c]) != rr) ++bad;
    }
    // timing on arbitrary finite codes (NaN encodings 0x7f / 0xff avoided)
    for (auto& v : hA) { v = rnd() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
    for (auto& v : hB) { v = rnd() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }
    hipMemcpy(dA, hA.data(), hA.size(), hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto kern, int Kk) {
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, dA, dB, dC, M, N, Kk);
      hipEventRecord(e0);
      const int reps = 30;
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, dA, dB, dC, M, N, Kk);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps * 1e3f;
    };
    const float t0 = timeit(k_gemm8<0, false>, K), t1 = timeit(k_gemm8<1, false>, K), t2 = timeit(k_gemm8<2, false>, K),
                t3 = timeit(k_gemm8<3, false>, K), t4 = timeit(k_gemm8<4, false>, K);
    // bf16 on the same structure: the byte matrices read as [M][K/2] bf16 (same LDS rows, same DMA; half the k per tile, so the
    // same k-values need twice the bytes: run it on the first K bytes = K/2 values and on 2K bytes where they exist)
    const float b0 = timeit(k_gemm8<0, true>, K), b1 = timeit(k_gemm8<1, true>, K), b3 = timeit(k_gemm8<3, true>, K);
    printf("fp8 x fp8 -> bf16  %-16s M=%d N=%d K=%d: %7.1f us  %7.1f TFLOP/s   exact-integer check: %d / 4000 wrong\n", s.name, M, N, K, t0,
           2.0 * M * N * K / t0 / 1e6, bad);
    printf("    ablations (us): no stores %.1f | no MFMA %.1f | no DMA after the first two tiles %.1f | prologue + epilogue only %.1f\n", t1, t2, t3, t4);
    printf("    same kernel on bf16 MFMA over the same BYTES (= K/2 values, half the FLOPs): %.1f us (%.1f TFLOP/s); no stores %.1f; no DMA %.1f\n", b0,
           1.0 * M * N * K / b0 / 1e6, b1, b3);
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
