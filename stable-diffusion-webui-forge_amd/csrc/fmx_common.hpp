// Shared helpers for the gfx950 (CDNA4, wave64) kernels of libfmx.  HIP only; no CUDA paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The 16-bit kernels are written once against the element type `f16` of their translation unit.  The default build is IEEE half
// (the SD / SDXL path).  The files the Flux transformer needs (GEMMs, attention, LayerNorm, q/k-norm + RoPE) are compiled a second
// time with -DFMX_ELEM_BF16: same code with bfloat16 storage and bf16 MFMA operands (the reference computes Flux in bf16), every
// entry point exported under a _bf16 name (fmx_bf16_names.hpp).
#ifdef FMX_ELEM_BF16
#include "fmx_bf16_names.hpp"
#endif

#include "fmx.h"

#ifdef FMX_ELEM_BF16
typedef __bf16 f16;
typedef __bf16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 f16x8 __attribute__((ext_vector_type(8)));
#define FMX_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define FMX_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#else
typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define FMX_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define FMX_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FMX_WAVE 64

int fmx_set_error(int code, const char* fmt, ...);

// Development A/B knobs (FMX_GEMM_MFMA, FMX_ATTN_SPLIT, ...): an environment variable changes which kernel a launch takes ONLY when the process
// also carries FMX_ALLOW_KNOBS=1 -- a stray variable must not change what a production process runs.  Returns the value of `name` when knobs are
// allowed and the variable is set, else nullptr; every knob that took effect is listed by fmx_active_knobs() (bench.py prints them into its line).
const char* fmx_knob(const char* name);

#define FMX_REQUIRE(cond, ...)                         \
  do {                                                 \
    if (!(cond)) return fmx_set_error(FMX_E_BADARG, __VA_ARGS__); \
  } while (0)

#define FMX_LAUNCH_CHECK(name)                                                        \
  do {                                                                                \
    hipError_t _e = hipGetLastError();                                                \
    if (_e != hipSuccess) return fmx_set_error((int)_e, "%s: %s", name, hipGetErrorString(_e)); \
  } while (0)

static inline bool fmx_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// LDS-DMA: 16 bytes per lane from a per-lane global address into LDS at (wave-uniform base + lane*16).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// XCD-aware, bijective remap of a 1-D workgroup id: consecutive logical ids land on the same XCD
// (hardware places workgroup b on XCD b % 8), so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = orig & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (orig >> 3);
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp): the IEEE division of `x / (1 + e^-x)` is a 10-instruction sequence per element, which made
// the HBM-bound GroupNorm+SiLU pass VALU-bound (profiles/r04a_pmc_*: VALU issue 50 % of the kernel's cycles)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact-erf GELU (backend/nn/unet.py:111 F.gelu default):  0.5 x (1 + erf(x / sqrt 2)) = x/2 + |x/2| (1 - p(t) t e^{-z^2}),  z = |x| / sqrt 2,
// erf by Abramowitz-Stegun 7.1.26 (t = 1 / (1 + 0.3275911 z); |abs err| of the GELU <= 5e-7, far below the fp16 rounding of the result).
// 13 VALU instructions, two of them transcendental: z is carried pre-multiplied by sqrt(log2 e) so that e^{-z^2} is one v_exp_f32 of -u^2,
// the reciprocal is the 1-ulp v_rcp_f32 (`__frcp_rn` is the correctly rounded one: a 10-instruction IEEE division sequence), and the sign
// is folded into the last FMA instead of a copysign.  The GEGLU epilogue of the ff1 GEMM evaluates this 80 times per lane with both waves
// of a SIMD in the epilogue at the same time: it was VALU-bound at ~27 instructions per value (profiles/r04k_gemm_microbench.jsonl: the
// same GEMM without the activation ran 16 % faster).
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float u = fabsf(x) * 0.8493218f;                                      // |x| / sqrt 2 * sqrt(log2 e)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.27273747f, u, 1.0f));          // 0.3275911 / sqrt(log2 e)
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float y = fmaf(-(pl * t), __builtin_amdgcn_exp2f(-(u * u)), 1.0f);    // erf(|x| / sqrt 2)
  const float hx = 0.5f * x;
  return fmaf(fabsf(hx), y, hx);
}

// nn.GELU(approximate="tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))), tanh(u) = 1 - 2 / (1 + e^{2u})
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float u = 0.7978845608028654f * fmaf(0.044715f * x * x, x, x);
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));
  return 0.5f * x * (1.0f + t);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
