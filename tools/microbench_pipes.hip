// Which pipes of a gfx950 SIMD run beside each other?  A micro-benchmark behind DESIGN.md's account of what bounds the attention kernels:
// issue cost of v_exp_f32 / plain VALU / v_mfma_f32_32x32x16_f16 from one wave, and whether a second wave on the SAME SIMD hides its
// transcendental / vector work behind the first wave's MFMAs.  One 8-wave workgroup per CU (waves w and w + 4 share a SIMD); each wave runs
// `iters` blocks of one role; the kernel reports s_memtime cycles per block for wave 0 (role A) and wave 4 (role B).
//   build: hipcc --offload-arch=gfx950 -O3 tools/microbench_pipes.hip -o tools/_build/microbench_pipes      run: tools/_build/microbench_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Role { IDLE = 0, EXP = 1, VALU = 2, MFMA = 3, MFMA_EXP = 4, MFMA2 = 5, MFMA_VALU = 6 };

// one block = 32 v_exp_f32 on 8 independent registers
__device__ __forceinline__ void block_exp(float (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
}
// one block = 32 v_fma_f32 on 8 independent registers
__device__ __forceinline__ void block_valu(float (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
}
// one block = 16 MFMAs over `CH` independent accumulator chains
template <int CH>
__device__ __forceinline__ void block_mfma(f32x16 (&acc)[4], const f16x8& a, const f16x8& b) {
#pragma unroll
  for (int r = 0; r < 16; ++r) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[r % CH]) : "v"(a), "v"(b));
}
// one block = 16 x (1 MFMA + 2 filler instructions) from ONE wave, 4 chains
template <bool TRANS>
__device__ __forceinline__ void block_mixed(f32x16 (&acc)[4], const f16x8& a, const f16x8& b, float (&v)[8]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[r % 4]) : "v"(a), "v"(b));
    if (TRANS) {
      asm volatile("v_exp_f32 %0, %0" : "+v"(v[(2 * r) % 8]));
      asm volatile("v_exp_f32 %0, %0" : "+v"(v[(2 * r + 1) % 8]));
    } else {
      asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[(2 * r) % 8]));
      asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[(2 * r + 1) % 8]));
    }
  }
}

__global__ __launch_bounds__(512) void pipes_kernel(int role_a, int role_b, int iters, long long* out, float* sink) {
  __shared__ char pad[100 * 1024];   // one workgroup per CU
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) pad[blockIdx.x & 1023] = 1;
  const int role = wave < 4 ? role_a : role_b;
  float v[8];
  f32x16 acc[4];
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = -0.001f * (float)(lane + i + 1);
    a[i] = (_Float16)0.01f;
    b[i] = (_Float16)0.02f;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    switch (role) {
      case EXP: block_exp(v); break;
      case VALU: block_valu(v); break;
      case MFMA: block_mfma<4>(acc, a, b); break;
      case MFMA2: block_mfma<2>(acc, a, b); break;
      case MFMA_EXP: block_mixed<true>(acc, a, b, v); break;
      case MFMA_VALU: block_mixed<false>(acc, a, b, v); break;
      default: break;
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int c = 0; c < 4; ++c) s += acc[c][0];
  if (s == 12345.678f) sink[0] = s + pad[lane];
  if (blockIdx.x == 8 && lane == 0 && (wave == 0 || wave == 4)) out[wave >> 2] = t1 - t0;
}

static const char* name(int r) {
  static const char* n[] = {"idle", "32 v_exp_f32", "32 v_fma_f32", "16 MFMA (4 chains)", "16 x (MFMA + 2 v_exp)", "16 MFMA (2 chains)", "16 x (MFMA + 2 v_fma)"};
  return n[r];
}

int main() {
  long long* out;
  float* sink;
  hipMalloc(&out, 16);
  hipMalloc(&sink, 16);
  const int iters = 2000;
  const int cases[][2] = {{EXP, IDLE},  {VALU, IDLE}, {MFMA, IDLE},     {MFMA2, IDLE},     {MFMA_EXP, IDLE}, {MFMA_VALU, IDLE}, {MFMA, EXP},
                          {MFMA, VALU}, {EXP, VALU},  {EXP, EXP},       {MFMA, MFMA},      {MFMA2, MFMA2},   {MFMA2, EXP},      {MFMA_EXP, MFMA_EXP},
                          {MFMA_VALU, MFMA_VALU}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("{\"note\": \"cycles per block (s_memtime) for the wave in role A / role B; both waves on the same SIMD; 256 workgroups x 8 waves\", \"rows\": [\n");
  for (unsigned c = 0; c < sizeof(cases) / sizeof(cases[0]); ++c) {
    hipMemset(out, 0, 16);
    hipLaunchKernelGGL(pipes_kernel, dim3(256), dim3(512), 0, 0, cases[c][0], cases[c][1], 200, out, sink);   // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(pipes_kernel, dim3(256), dim3(512), 0, 0, cases[c][0], cases[c][1], iters, out, sink);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    long long h[2];
    float ms = 0.f;
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    hipEventElapsedTime(&ms, e0, e1);
    printf("  {\"A\": \"%s\", \"B\": \"%s\", \"cycles_per_block_A\": %.1f, \"cycles_per_block_B\": %.1f, \"kernel_us\": %.1f}%s\n", name(cases[c][0]),
           name(cases[c][1]), (double)h[0] / iters, (double)h[1] / iters, ms * 1e3, c + 1 < sizeof(cases) / sizeof(cases[0]) ? "," : "");
  }
  printf("]}\n");
  return 0;
}
