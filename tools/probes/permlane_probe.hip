// Probe: what v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950) leave in their two registers, against the xor-16 / xor-32 shuffles.
// dst = {x.row0, y.row0, x.row2, y.row2}, src = {x.row1, y.row1, x.row3, y.row3} (rows of 16 lanes); halves likewise for the 32 form.
// Under hipcc 7.2 three other ways of getting at the pair returned the FIRST register twice: the __builtin_amdgcn_permlane*_swap result,
// an asm statement with two "+v" operands, and an ext_vector built from the two asm outputs -- hence separate tied outputs into scalars.
// x = lane, y = 100 + lane.  build: hipcc -O3 --offload-arch=gfx950 tools/probes/permlane_probe.hip -o tools/probes/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void sw16(unsigned& x, unsigned& y) {
  unsigned xo, yo;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "=v"(xo), "=v"(yo) : "0"(x), "1"(y));
  x = xo; y = yo;
}
__device__ __forceinline__ void sw32(unsigned& x, unsigned& y) {
  unsigned xo, yo;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "=v"(xo), "=v"(yo) : "0"(x), "1"(y));
  x = xo; y = yo;
}
__global__ void k(unsigned* o) {
  const unsigned l = threadIdx.x;
  unsigned x = l, y = 100 + l, w = l, z = 100 + l;
  sw16(x, y);
  sw32(w, z);
  o[l] = x; o[64 + l] = y; o[128 + l] = w; o[192 + l] = z;
  o[256 + l] = __shfl_xor((int)l, 16, 64); o[320 + l] = __shfl_xor((int)l, 32, 64);
}
int main() {
  unsigned r[384]; unsigned* o; hipMalloc(&o, 384 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o); hipMemcpy(r, o, 384 * 4, hipMemcpyDeviceToHost);
  const char* nm[6] = {"swap16 dst", "swap16 src", "swap32 dst", "swap32 src", "xor16", "xor32"};
  for (int j = 0; j < 6; ++j) { printf("%-11s", nm[j]); for (int i = 0; i < 64; i += 4) printf(" %3u", r[64 * j + i]); printf("\n"); }
  return 0;
}
