// Micro-benchmark: throughput ceiling of one GEMM k-step's instruction mix with NO barriers and NO data dependencies
// between loads and MFMAs: per iteration a wave issues NR ds_read_b128 (the GEMM's swizzled fragment pattern), ND
// buffer_load_dwordx4...lds pieces (L2-resident source) and NM v_mfma_f32_32x32x16_f16 on NM different accumulators.
// 8 waves per CU (2 per SIMD), 1 workgroup per CU, 256 workgroups.  Prints shader cycles per iteration per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 512;

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int NR, int ND, int NM, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const char* __restrict__ src, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, li = lane & 31;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long)blockIdx.x * 262144), 0, 262144, 0x00020000);
  const unsigned voff = (unsigned)((wave * 8 + (lane >> 3)) * 2560 + (((lane & 7) ^ ((lane >> 4) & 7)) * 16));
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 fr[2][8];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 8; ++i) fr[b][i] = f16x8{1, 1, 1, 1, 1, 1, 1, 1};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned soff = 0;
  for (int it = 0; it < ITERS; ++it) {
    const int fb = it & 1;
    const int ks = it & 3;
    const char* base = smem + ((it >> 2) & 1) * 65536 + (wave & 3) * 16384;
#pragma unroll
    for (int i = 0; i < NR; ++i) fr[fb ^ 1][i] = *reinterpret_cast<const f16x8*>(base + lds_off(i * 32 + li, ks * 2 + hi));
#pragma unroll
    for (int i = 0; i < ND; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (((it + 1) >> 2) & 1) * 65536 + (i * 8 + wave) * 1024), 16, voff, soff + i * 20480, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[fb][i & 1], fr[fb][2 + (i >> 1)], acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    soff = (soff + 128) & 2047;
    if (ND && (it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)fr[0][i][0] + (float)fr[1][i][0];
  if (sink) sink[threadIdx.x] = s;
}

template <int NR, int ND, int NM, int WAVES>
void run(const char* src, unsigned long long* dout) {
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NR, ND, NM, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<NR, ND, NM, WAVES>), dim3(256), dim3(WAVES * 64), 131072, 0, src, dout, (float*)nullptr);
  CHECK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(256 * 8);
  CHECK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double mx = 0, sum = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < WAVES; ++w) { mx = mx > (double)h[b * 8 + w] ? mx : (double)h[b * 8 + w]; sum += (double)h[b * 8 + w]; }
  const double per_it = sum / (256.0 * WAVES) / ITERS;
  printf("{\"waves\": %d, \"ds_read_b128\": %d, \"dma_pieces\": %d, \"mfma\": %d, \"cycles_per_iter_per_wave\": %.1f, \"max\": %.1f, \"mfma_cycles_needed_per_simd\": %d, \"lds_read_B_per_clk_cu\": %.1f}\n",
         WAVES, NR, ND, NM, per_it, mx / ITERS, NM * 32 * (WAVES / 4), per_it > 0 ? NR * 1024.0 * WAVES / per_it : 0.0);
  fflush(stdout);
}

int main() {
  char* src; unsigned long long* dout;
  CHECK(hipMalloc(&src, 262144L * 256 + (1 << 20)));
  CHECK(hipMemset(src, 0, 262144L * 256 + (1 << 20)));
  CHECK(hipMalloc(&dout, 256 * 8 * 8));
  run<0, 0, 8, 8>(src, dout);
  run<6, 0, 8, 8>(src, dout);
  run<0, 1, 8, 8>(src, dout);
  run<0, 2, 8, 8>(src, dout);
  run<6, 1, 8, 8>(src, dout);
  run<6, 2, 8, 8>(src, dout);
  run<8, 2, 8, 8>(src, dout);
  run<4, 2, 8, 8>(src, dout);
  run<6, 2, 0, 8>(src, dout);
  run<6, 0, 0, 8>(src, dout);
  run<8, 0, 0, 8>(src, dout);
  run<0, 2, 0, 8>(src, dout);
  run<0, 0, 8, 4>(src, dout);
  run<4, 2, 8, 4>(src, dout);
  run<6, 2, 8, 4>(src, dout);
  run<8, 4, 8, 4>(src, dout);
  return 0;
}
