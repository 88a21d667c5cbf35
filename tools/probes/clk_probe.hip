// Probe: sustained MFMA rate and core clock under an all-CU bf16 MFMA load (no memory traffic).
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/clk_probe.hip -o tools/probes/clk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NACC>
__global__ __launch_bounds__(512) void k_mfma(int iters, float* out, long long* clk) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0, 0, 0, 0};
  long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
  const int iters = 20000;
  float* out; long long* clk;
  hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&clk, 4096 * 16);
  for (int blocks : {64, 256, 512}) {
    for (int threads : {256, 512}) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(threads), 0, 0, 100, out, clk);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mfma<16>, dim3(blocks), dim3(threads), 0, 0, iters, out, clk);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(2 * blocks);
      hipMemcpy(h.data(), clk, 16 * blocks, hipMemcpyDeviceToHost);
      double cyc = 0, wall = 0;
      for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
      cyc /= blocks; wall /= blocks;
      const double flops = (double)blocks * (threads / 64) * iters * 16 * 16384.0;
      printf("blocks %4d threads %3d: %.3f ms  %.1f TFLOP/s  core cycles %.0f, wall ticks(100MHz) %.0f -> %.0f MHz; cycles per MFMA per wave %.2f\n",
             blocks, threads, ms, flops / ms / 1e9, cyc, wall, cyc / (wall / 100.0), cyc / (iters * 16.0));
    }
  }
  return 0;
}
