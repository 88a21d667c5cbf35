// Stand-in for the compute-unit and HBM footprint of a gradient exchange (RCCL ring kernels) on ONE GPU -- tools/bench_dp_footprint.py.
// k workgroups of 512 threads (one per CU, like RCCL's channels) stream `bytes` from src to dst `passes` times, summing into the destination
// on the later passes (a ring reduce-scatter + all-gather reads and writes each element about twice).  No product code: built by the tool
// into tools/probes/libdpfoot.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_dp_footprint(const f4* __restrict__ src, f4* __restrict__ dst, int64_t n4, int passes) {
  const int64_t stride = (int64_t)gridDim.x * 512;
  for (int p = 0; p < passes; ++p)
    for (int64_t i = (int64_t)blockIdx.x * 512 + threadIdx.x; i < n4; i += stride) {
      f4 v = __builtin_nontemporal_load(src + i);
      if (p) v += __builtin_nontemporal_load(dst + i);
      __builtin_nontemporal_store(v, dst + i);
    }
}

extern "C" int dp_footprint(const void* src, void* dst, int64_t bytes, int passes, int workgroups, void* stream) {
  if (!src || !dst || bytes <= 0 || passes <= 0 || workgroups <= 0) return 1;
  hipLaunchKernelGGL(k_dp_footprint, dim3(workgroups), dim3(512), 0, (hipStream_t)stream, (const f4*)src, (f4*)dst, bytes / 16, passes);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
