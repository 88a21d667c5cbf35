// Shared device helpers for the SFR-on HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SFRON_OK 0
#define SFRON_ERR_ARG 1001      // bad argument (shape / alignment / null pointer)
#define SFRON_ERR_UNSUPPORTED 1002

#define SFRON_CHECK_ARG(cond) do { if (!(cond)) return SFRON_ERR_ARG; } while (0)
#define SFRON_LAUNCH_STATUS() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned short bf16_raw;

static constexpr int WAVE = 64;

__device__ __forceinline__ float bf2f(__bf16 x) { return (float)x; }
__device__ __forceinline__ __bf16 f2bf(float x) { return (__bf16)x; }   // v_cvt_pk_bf16_f32: RNE, NaN-preserving

// Sum over the 64 lanes, every lane gets it.  Data-parallel-primitive form: four adds inside the 16-lane rows (quad swaps, half-row and
// row mirrors), two row broadcasts, one readlane -- seven short vector operations in a fixed order (bitwise reproducible).  The
// `__shfl_xor` butterfly compiles to six DEPENDENT ds_bpermute_b32 (an LDS-crossbar round trip each, ~700 cycles per sum): the row
// kernels take two sums per token row.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_get(float v) {       // lanes of rows outside ROW_MASK get 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the 16 lanes of a row (lanes that share lane >> 4), every lane of the row gets it: the same pairing -- the same bits -- as the
// xor 1 / 2 / 4 / 8 butterfly
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_get<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xf>(v);     // row_half_mirror
  return v + dpp_get<0x140, 0xf>(v);   // row_mirror
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_get<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
  v += dpp_get<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
  v += dpp_get<0x141, 0xf>(v);     // row_half_mirror
  v += dpp_get<0x140, 0xf>(v);     // row_mirror: every lane of a row holds the row's sum
  v += dpp_get<0x142, 0xa>(v);     // row_bcast15 into rows 1 and 3
  v += dpp_get<0x143, 0xc>(v);     // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// tanh-approximate GELU of torch.nn.GELU(approximate="tanh"): 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715*x^3).
// Written through the identity 0.5*(1+tanh(u)) = sigmoid(2u): one v_exp + one v_rcp instead of a tanhf call
// (the epilogue of the fc1 / fc2-dgrad GEMMs evaluates this 4 times per accumulator register).
__device__ __forceinline__ float gelu_sig(float x, float& du) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float x2 = x * x;
  const float u = k0 * (x + k1 * x * x2);
  du = k0 * (1.0f + 3.0f * k1 * x2);
  return __frcp_rn(1.0f + __expf(-2.0f * u));
}
// Epilogue forms (results are rounded to bf16 right after): the GEMM epilogues are VALU-bound -- six extra FMAs per value
// cost the fc1 kernel 14 % -- so the sigmoid argument is built with two FMAs on folded constants, exp / rcp are the raw
// hardware transcendentals (1 ulp), and 1 - s is taken as e * s (s = 1/(1+e), e = exp(-2u)).
__device__ __forceinline__ float gelu_tanh(float x) {
  const float c0 = -2.0f * 1.4426950408889634f * 0.7978845608028654f, c1 = c0 * 0.044715f;   // -2*log2(e)*k0*(1 + k1 x^2)
  const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, c1, c0));
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_tanh_grad(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float c0 = -2.0f * 1.4426950408889634f * k0, c1 = c0 * k1;
  const float t = x * x;
  const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(t, c1, c0));
  const float s = __builtin_amdgcn_rcpf(1.0f + e);                       // 0.5*(1+tanh u)
  const float du2 = __builtin_fmaf(t, 6.0f * k0 * k1, 2.0f * k0);        // 2 du/dx
  return __builtin_fmaf((x * s) * (e * s), du2, s);                      // s + x s (1-s) 2 du
}
// The same two functions on the 4 accumulator values of a lane, written with vector operations so that hipcc emits the PACKED forms
// (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two values per VALU slot): the fused GELU epilogues are VALU-bound -- 5.8 of the 30 us an
// fc1 workgroup lives, 6.0 of 25.8 in the fc2 dgrad (tools/bench_gemm.py phases, profiles/r04_gemm_phases.txt).  Operation for
// operation the scalar sequence above: bit-identical results.
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }
__device__ __forceinline__ f32x4 gelu_tanh4(f32x4 x) {
  const float c0 = -2.0f * 1.4426950408889634f * 0.7978845608028654f, c1 = c0 * 0.044715f;
  const f32x4 a = x * __builtin_elementwise_fma(x * x, splat4(c1), splat4(c0));
  const f32x4 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1]), __builtin_amdgcn_exp2f(a[2]), __builtin_amdgcn_exp2f(a[3])};
  const f32x4 d = splat4(1.0f) + e;
  const f32x4 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
  return x * r;
}
__device__ __forceinline__ f32x4 gelu_tanh_grad4(f32x4 x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float c0 = -2.0f * 1.4426950408889634f * k0, c1 = c0 * k1;
  const f32x4 t = x * x;
  const f32x4 a = x * __builtin_elementwise_fma(t, splat4(c1), splat4(c0));
  const f32x4 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1]), __builtin_amdgcn_exp2f(a[2]), __builtin_amdgcn_exp2f(a[3])};
  const f32x4 d = splat4(1.0f) + e;
  const f32x4 s = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
  const f32x4 du2 = __builtin_elementwise_fma(t, splat4(6.0f * k0 * k1), splat4(2.0f * k0));
  return __builtin_elementwise_fma((x * s) * (e * s), du2, s);
}
// GELU and its derivative from ONE exp / rcp pair (operation for operation gelu_tanh4 and gelu_tanh_grad4: the same bits as either alone).
__device__ __forceinline__ void gelu_tanh_both4(f32x4 x, f32x4& y, f32x4& dy) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float c0 = -2.0f * 1.4426950408889634f * k0, c1 = c0 * k1;
  const f32x4 t = x * x;
  const f32x4 a = x * __builtin_elementwise_fma(t, splat4(c1), splat4(c0));
  const f32x4 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1]), __builtin_amdgcn_exp2f(a[2]), __builtin_amdgcn_exp2f(a[3])};
  const f32x4 d = splat4(1.0f) + e;
  const f32x4 s = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
  const f32x4 du2 = __builtin_elementwise_fma(t, splat4(6.0f * k0 * k1), splat4(2.0f * k0));
  y = x * s;
  dy = __builtin_elementwise_fma(y * (e * s), du2, s);
}
// GELU'(x) as ONE byte (round 6: the second output of fc1 + GELU, sfron_gemm_desc SFRON_EPI_GELU_Q / SFRON_EPI_DGELU_Q).  GELU'_tanh takes
// values in [-0.1290, 1.1290]; code = round((g' + 0.15) * 196) in 0 .. 251, g' = code / 196 - 0.15: a step of 0.0051, i.e. at most 0.0026 of
// absolute error -- the size of what the bf16 rounding of x (2^-9 relative) does to GELU'(x) through GELU'' (<= 0.5 |x| 2^-9 ~ 0.001 |x|).
// v_cvt_pk_u8_f32 rounds to nearest and saturates to 0 .. 255.
constexpr float GELUQ_SCALE = 196.0f, GELUQ_OFF = 0.15f;
__device__ __forceinline__ unsigned geluq_pack4(f32x4 dy) {
  const f32x4 c = __builtin_elementwise_fma(dy, splat4(GELUQ_SCALE), splat4(GELUQ_OFF * GELUQ_SCALE));
  unsigned w = 0;
  w = __builtin_amdgcn_cvt_pk_u8_f32(c[0], 0, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(c[1], 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(c[2], 2, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(c[3], 3, w);
  return w;
}
__device__ __forceinline__ f32x4 geluq_unpack4(unsigned w) {
  const f32x4 c = {(float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24)};     // v_cvt_f32_ubyte0..3
  return __builtin_elementwise_fma(c, splat4(1.0f / GELUQ_SCALE), splat4(-GELUQ_OFF));
}
__device__ __forceinline__ f32x4 bf2f4(bf16x4 h) { return f32x4{bf2f(h[0]), bf2f(h[1]), bf2f(h[2]), bf2f(h[3])}; }
__device__ __forceinline__ bf16x4 f2bf4(f32x4 v) { return bf16x4{f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])}; }
__device__ __forceinline__ f32x4 as4(float4 b) { return f32x4{b.x, b.y, b.z, b.w}; }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_grad(float x) {
  float s = 1.0f / (1.0f + __expf(-x));
  return s * (1.0f + x * (1.0f - s));
}


// ---- 16-byte epilogue accesses for the D^T accumulator layout (MI355X guide T21) ------------------------------------------------
// After an MFMA 16x16 issued as D^T = B^T A^T a lane holds row (lane & 15), columns 4 g .. 4 g + 3 (g = lane >> 4) of a 16 x 16 tile:
// 8 bytes of bf16, so the natural store covers 16 rows x 32 B per wave-instruction and a tile row needs one instruction per tile.  The
// store tail of a GEMM / attention workgroup is ISSUE-bound (instruction count, not bytes).  v_permlane16_swap_b32 x, y leaves
// x = {x.row0, y.row0, x.row2, y.row2}, y = {x.row1, y.row1, x.row3, y.row3} (rows = 16-lane groups; tools/probes/permlane_probe.hip),
// so with x / y the packed dwords of two horizontally adjacent tiles k, k + 1 a lane ends up with 16 CONTIGUOUS bytes:
//   g = 0: tile k cols 0..7   g = 1: tile k+1 cols 0..7   g = 2: tile k cols 8..15   g = 3: tile k+1 cols 8..15
// i.e. element offset pair_col(g) = 16 (g & 1) + 8 (g >> 1) from tile k's first column: ONE 16-byte access per lane and tile pair.
// The swap is an involution: the same two swaps turn a 16-byte load at that offset back into the two tiles' native fragments.
// (asm with separate tied outputs: under hipcc 7.2 the builtin and the "+v" forms returned the first register twice; s_nop 1 pads the
// VALU-write -> permlane-read hazard the compiler cannot see inside asm.)
__device__ __forceinline__ void permlane16_swap(unsigned& x, unsigned& y) {
  unsigned xo, yo;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "=v"(xo), "=v"(yo) : "0"(x), "1"(y));
  x = xo; y = yo;
}
__device__ __forceinline__ int pair_col(int g) { return 16 * (g & 1) + 8 * (g >> 1); }
__device__ __forceinline__ uint4 pair_pack(bf16x4 a, bf16x4 b) {          // a = tile k, b = tile k + 1 (this lane's 4 columns of each)
  uint2 ua = __builtin_bit_cast(uint2, a), ub = __builtin_bit_cast(uint2, b);
  permlane16_swap(ua.x, ub.x);
  permlane16_swap(ua.y, ub.y);
  return make_uint4(ua.x, ua.y, ub.x, ub.y);
}
__device__ __forceinline__ void pair_unpack(uint4 v, bf16x4& a, bf16x4& b) {
  uint2 ua = make_uint2(v.x, v.y), ub = make_uint2(v.z, v.w);
  permlane16_swap(ua.x, ub.x);
  permlane16_swap(ua.y, ub.y);
  a = __builtin_bit_cast(bf16x4, ua); b = __builtin_bit_cast(bf16x4, ub);
}

// the same piece map for ONE-byte elements: a lane's 4 bytes of tile k and of tile k + 1 -> its 8 contiguous bytes at pair_col(g)
__device__ __forceinline__ uint2 pair_pack8(unsigned a, unsigned b) { permlane16_swap(a, b); return make_uint2(a, b); }
__device__ __forceinline__ void pair_unpack8(uint2 v, unsigned& a, unsigned& b) { a = v.x; b = v.y; permlane16_swap(a, b); }

// ---- a completion event carried by the producing kernel's own dispatch ----------------------------------------------------------
// hipEventRecord puts a marker packet into the stream behind the kernel it follows: ~5 us of the main stream per hand-off to the
// weight-gradient stream, four per DiT block (csrc/dit_engine.hip produced()).  hipExtLaunchKernelGGL attaches the event to the kernel's own
// dispatch packet instead: nothing extra in the producing stream.  The engine ARMS an event (sfron_arm_stop_event) right before it calls the
// library entry point whose ONE kernel produces the tensor; SFRON_LAUNCH_EV in that entry point's launcher takes it.  An entry point that
// launches through another path leaves it armed: the engine then records it the old way (sfron_take_stop_event() != null).  Per host thread.
#include <hip/hip_ext.h>
void sfron_arm_stop_event(hipEvent_t ev);
hipEvent_t sfron_take_stop_event();
#define SFRON_LAUNCH_EV(kernel, grid, block, lds, stream, ...)                                                         \
  do {                                                                                                                 \
    hipEvent_t ev__ = sfron_take_stop_event();                                                                         \
    if (ev__) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, ev__, 0, __VA_ARGS__);                  \
    else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                            \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
