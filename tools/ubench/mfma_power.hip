// What does the chip sustain on bare MFMA streams (no LDS, no memory) with RANDOM operands, all CUs busy?  The GEMM K loops sit at matrix-pipe-busy x
// clock ~ 1.1 GHz: is that the silicon's power line for this instruction, or the kernel's?  Two waves per SIMD, 8 independent accumulator
// chains per wave, operands from registers; v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16; random vs zero operands.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k_mfma(const f16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  f16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 0xffff]; b[i] = src[(tid * 8 + 4 + i) & 0xffff]; }
  float acc_sum = 0.f;
  if (SHAPE == 32) {
    f32x16 c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j & 3], b[(j + u) & 3], c[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc_sum += c[j][0] + c[j][15];
  } else {
    f32x4 c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[j][r] = 0.f;
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j & 3], b[(j + u) & 3], c[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc_sum += c[j][0] + c[j][3];
  }
  if (acc_sum == 12345.678f) out[tid] = acc_sum;
}

template <int SHAPE>
static void run(const char* name, const f16x8* src, float* out, int grid) {
  const int iters = 400000;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_mfma<SHAPE>, dim3(grid), dim3(512), 0, 0, src, out, 2000);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a, 0));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_mfma<SHAPE>, dim3(grid), dim3(512), 0, 0, src, out, iters);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
  const double flop_per_mfma = SHAPE == 32 ? 32.0 * 32 * 16 * 2 : 16.0 * 16 * 32 * 2;
  const double flops = 3.0 * grid * 8.0 /*waves*/ * iters * 8.0 * flop_per_mfma;
  const double cyc_per_mfma_at_peak = SHAPE == 32 ? 32.0 : 16.0;   // per SIMD
  const double mfma_per_simd = 3.0 * (grid / 256.0) * 2.0 * iters * 8.0;
  printf("{\"case\": \"%s\", \"grid\": %d, \"ms\": %.2f, \"tflops\": %.0f, \"busy_x_clock_ghz\": %.3f}\n", name, grid, ms, flops / (ms * 1e-3) / 1e12,
         mfma_per_simd * cyc_per_mfma_at_peak / (ms * 1e-3) / 1e9);
}


// The GEMM's own MFMA sequence without its LDS / DMA traffic: a wave tile of 64 x 160 fp32 accumulators (160 registers), one k-step = every weight
// fragment against every activation fragment, weight-major (consecutive MFMAs share the A operand).  32x32x16: 5 x 2 blocks, 10 MFMAs per 16 of K;
// 16x16x32: 10 x 4 blocks, 40 MFMAs per 32 of K.  Same FLOPs per k, same register footprint.
template <int SHAPE>
__global__ __launch_bounds__(512, 2) void k_tile(const f16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  float acc_sum = 0.f;
  if (SHAPE == 32) {
    f16x8 w[5], a[2];
#pragma unroll
    for (int i = 0; i < 5; ++i) w[i] = src[(tid * 8 + i) & 0xffff];
#pragma unroll
    for (int i = 0; i < 2; ++i) a[i] = src[(tid * 8 + 5 + i) & 0xffff];
    f32x16 c[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)   // 2 x 16 of K
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[j], a[i], c[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc_sum += c[i][j][0] + c[i][j][15];
  } else {
    f16x8 w[10], a[4];
#pragma unroll
    for (int i = 0; i < 10; ++i) w[i] = src[(tid * 16 + i) & 0xffff];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = src[(tid * 16 + 10 + i) & 0xffff];
    f32x4 c[4][10];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 10; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 10; ++j)     // 32 of K
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[j], a[i], c[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 10; ++j) acc_sum += c[i][j][0] + c[i][j][3];
  }
  if (acc_sum == 12345.678f) out[tid] = acc_sum;
}

template <int SHAPE>
static void run_tile(const char* name, const f16x8* src, float* out, int grid) {
  const int iters = 100000;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_tile<SHAPE>, dim3(grid), dim3(512), 0, 0, src, out, 2000);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a, 0));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_tile<SHAPE>, dim3(grid), dim3(512), 0, 0, src, out, iters);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
  const double flops = 3.0 * grid * 8.0 * iters * (2.0 * 64 * 160 * 32);   // per wave and iteration: 64 x 160 x 32 of K
  printf("{\"case\": \"%s\", \"grid\": %d, \"ms\": %.2f, \"tflops\": %.0f, \"busy_x_clock_ghz\": %.3f}\n", name, grid, ms, flops / (ms * 1e-3) / 1e12,
         flops / (ms * 1e-3) / (256.0 * 4 * 1024.0) / 1e9 * (256.0 / grid) * (grid / 256.0));
}

int main() {
  const int n = 65536;
  f16x8* h = (f16x8*)malloc(n * sizeof(f16x8));
  f16x8 *rnd, *zero; float* out;
  CHECK(hipMalloc(&rnd, n * sizeof(f16x8))); CHECK(hipMalloc(&zero, n * sizeof(f16x8))); CHECK(hipMalloc(&out, 1 << 22));
  srand(1);
  for (int i = 0; i < n; ++i) for (int e = 0; e < 8; ++e) {
    float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
    h[i][e] = (_Float16)(sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * 0.05f);
  }
  CHECK(hipMemcpy(rnd, h, n * sizeof(f16x8), hipMemcpyHostToDevice));
  CHECK(hipMemset(zero, 0, n * sizeof(f16x8)));
  for (int rep = 0; rep < 2; ++rep) {
    run<32>("32x32x16 f16, N(0, 0.05) operands, all CUs", rnd, out, 256);
    run<16>("16x16x32 f16, N(0, 0.05) operands, all CUs", rnd, out, 256);
    run<32>("32x32x16 f16, zero operands, all CUs", zero, out, 256);
    run<32>("32x32x16 f16, N(0, 0.05) operands, half the CUs", rnd, out, 128);
    run_tile<32>("GEMM wave-tile sequence (64 x 160), 32x32x16, random", rnd, out, 256);
    run_tile<16>("GEMM wave-tile sequence (64 x 160), 16x16x32, random", rnd, out, 256);
  }
  return 0;
}
