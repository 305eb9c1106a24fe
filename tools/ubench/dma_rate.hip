// Micro-benchmark: per-CU issue/throughput cost of the ways to bring a 1-KiB piece (64 lanes x 16 B) on chip, alone and
// beside MFMA streams on the sibling waves.  One 512-thread workgroup per CU (like gemm256), data L2-resident.
//   FORM 0: global_load_lds_dwordx4, 64-bit vaddr            FORM 1: global_load_lds_dwordx4, saddr + 32-bit voffset
//   FORM 2: buffer_load_dwordx4 ... lds (raw buffer -> LDS)   FORM 3: global_load_dwordx4 -> VGPR (no LDS write)
//   FORM 4: global_load_dwordx4 -> VGPR -> ds_write_b128
//   PAT  0: 1 KiB contiguous per instruction   1: 8 rows x 128 B, row stride 2560 B   2: same, XOR-swizzled chunks
// usage: dma_rate  -> prints one line per (form, pattern, loader waves, mfma waves)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ITERS = 256;   // pieces per loader wave

template <int FORM, int PAT>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, long per_cu_bytes, int loaders, int mfmas, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (long)blockIdx.x * per_cu_bytes;
  // per-lane byte offset inside a piece
  unsigned loff;
  if (PAT == 0) loff = lane * 16;
  else if (PAT == 1) loff = (lane >> 3) * 2560 + (lane & 7) * 16;
  else loff = (lane >> 3) * 2560 + (((lane & 7) ^ ((lane >> 4) & 7)) * 16);
  const unsigned piece_stride = (PAT == 0) ? 1024 : 128;  // PAT 1/2: next K-tile = next 128-B column block of the same rows
  const bool is_loader = wave < loaders;
  const bool is_mfma = wave >= 8 - mfmas;
  f32x16 acc = {0};
  f16x8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  f32x4 vsum = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (is_loader) {
    char* lds = smem + wave * 16384;
    const unsigned wrap = (PAT == 0) ? (unsigned)(per_cu_bytes / 8) : 2048;  // stay L2-resident
    unsigned off = wave * ((PAT == 0) ? (unsigned)(per_cu_bytes / 8) : 8 * 2560 * 1);
    unsigned cur = 0;
#pragma unroll 4
    for (int i = 0; i < ITERS; ++i) {
      const char* g = base + off + cur + loff;
      char* l = lds + (i & 15) * 1024;
      if (FORM == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
      } else if (FORM == 1) {
        // saddr form: uniform base in SGPRs, 32-bit lane offset
        const char* ub = base + off;  // uniform
        unsigned vo = cur + loff;
        asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(ub), "s"((unsigned)(size_t)l) : "memory");
      } else if (FORM == 2) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + off), 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, cur + loff, 0, 0, 0);
      } else {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g);
        if (FORM == 4) *reinterpret_cast<f32x4*>(l + lane * 16) = v;
        else vsum += v;
      }
      cur += piece_stride;
      if (cur >= wrap) cur = 0;
      if ((i & 7) == 7) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (is_mfma && !is_loader) {
#pragma unroll 8
    for (int i = 0; i < ITERS * 2; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (sink) sink[threadIdx.x] = acc[0] + vsum[0] + smem[threadIdx.x];
}

template <int FORM, int PAT>
void run(const char* src, long per_cu, unsigned long long* dout, int loaders, int mfmas) {
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<FORM, PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<FORM, PAT>), dim3(256), dim3(512), 131072, 0, src, per_cu, loaders, mfmas, dout, (float*)nullptr);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k<FORM, PAT>), dim3(256), dim3(512), 131072, 0, src, per_cu, loaders, mfmas, dout, (float*)nullptr);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(256 * 8);
  CHECK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  double lmax = 0, mmax = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) {
      if (w < loaders) lmax = lmax > (double)h[b * 8 + w] ? lmax : (double)h[b * 8 + w];
      else if (w >= 8 - mfmas) mmax = mmax > (double)h[b * 8 + w] ? mmax : (double)h[b * 8 + w];
    }
  // s_memtime ticks at 100 MHz on this part? report both raw ticks and wall
  const double pieces_per_cu = (double)loaders * ITERS;
  printf("{\"form\": %d, \"pat\": %d, \"loaders\": %d, \"mfma_waves\": %d, \"wall_us\": %.1f, \"ns_per_piece_per_cu\": %.2f, \"loader_ticks\": %.0f, \"mfma_ticks\": %.0f, \"mfma_ns_each\": %.2f}\n",
         FORM, PAT, loaders, mfmas, ms * 1e3, ms * 1e6 / pieces_per_cu, lmax, mmax, mfmas ? ms * 1e6 / (ITERS * 2) : 0.0);
  fflush(stdout);
}

int main() {
  const long per_cu = 256 * 1024;  // 64 MB total: L2/MALL-resident after the warm-up launches
  char* src; unsigned long long* dout;
  CHECK(hipMalloc(&src, per_cu * 256 + (1 << 20)));
  CHECK(hipMemset(src, 0, per_cu * 256 + (1 << 20)));
  CHECK(hipMalloc(&dout, 256 * 8 * 8));
  const int cfgs[][2] = {{8, 0}, {4, 0}, {2, 0}, {1, 0}, {4, 4}, {2, 4}, {0, 4}};
  for (auto& c : cfgs) {
    run<0, 0>(src, per_cu, dout, c[0], c[1]);
    run<0, 1>(src, per_cu, dout, c[0], c[1]);
    run<0, 2>(src, per_cu, dout, c[0], c[1]);
    run<1, 2>(src, per_cu, dout, c[0], c[1]);
    run<2, 2>(src, per_cu, dout, c[0], c[1]);
    run<3, 2>(src, per_cu, dout, c[0], c[1]);
    run<4, 2>(src, per_cu, dout, c[0], c[1]);
    run<3, 0>(src, per_cu, dout, c[0], c[1]);
  }
  return 0;
}
