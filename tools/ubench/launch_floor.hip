// What does ONE more kernel launch cost on this part when the kernels are dependent (one stream / one graph chain), as the UNet forward's
// ~1.6 k launches are?  Chains of N launches, HIP-event time / N, for kernels that do (almost) nothing:
//   a: 1 workgroup of 64 threads            b: 256 workgroups x 512 threads, no LDS         c: 256 x 512 threads with 160 KB of dynamic LDS
//   d: c with a 256-byte by-value argument  e: c that also stores 160 KB per workgroup (42 MB per launch: the output of a K = 1280 GEMM tile round)
// each as eager launches and inside a hipGraph.  (round 3: is the ~9 us between a GEMM's wall time and its workgroup's own life the launch floor?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Big { unsigned w[64]; };

__global__ void k_small(unsigned* out) { if (threadIdx.x == 0 && out[0] == 0xffffffffu) out[1] = 1; }
__global__ __launch_bounds__(512) void k_wide(unsigned* out) { if (threadIdx.x == 0 && out[0] == 0xffffffffu) out[1] = blockIdx.x; }
__global__ __launch_bounds__(512) void k_lds(unsigned* out) {
  extern __shared__ unsigned lds[];
  if (out[0] == 0xffffffffu) { lds[threadIdx.x] = 1; __syncthreads(); out[1] = lds[0]; }
}
__global__ __launch_bounds__(512) void k_arg(unsigned* out, Big b) {
  extern __shared__ unsigned lds[];
  if (out[0] == 0xffffffffu) { lds[threadIdx.x] = b.w[threadIdx.x & 63]; __syncthreads(); out[1] = lds[0]; }
}
__global__ __launch_bounds__(512) void k_store(unsigned* out, uint4* dst) {
  extern __shared__ unsigned lds[];
  if (out[0] == 0xffffffffu) { lds[threadIdx.x] = 1; __syncthreads(); out[1] = lds[0]; }
  uint4* p = dst + (size_t)blockIdx.x * (160 * 1024 / 16) + threadIdx.x;
  const uint4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
#pragma unroll
  for (int i = 0; i < 20; ++i) p[i * 512] = v;   // 20 x 512 x 16 B = 160 KB per workgroup, row-contiguous 16-byte stores
}

__global__ __launch_bounds__(512) void k_store_nt(unsigned* out, uint4* dst) {
  extern __shared__ unsigned lds[];
  if (out[0] == 0xffffffffu) { lds[threadIdx.x] = 1; __syncthreads(); out[1] = lds[0]; }
  uint4* p = dst + (size_t)blockIdx.x * (160 * 1024 / 16) + threadIdx.x;
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
#pragma unroll
  for (int i = 0; i < 20; ++i) __builtin_nontemporal_store(v, reinterpret_cast<u4*>(p + i * 512));
}
// the same bytes written by a quarter of the workgroups (64 CUs busy): is the store phase bandwidth- or per-CU-issue-bound?
__global__ __launch_bounds__(512) void k_store4(unsigned* out, uint4* dst) {
  extern __shared__ unsigned lds[];
  if (out[0] == 0xffffffffu) { lds[threadIdx.x] = 1; __syncthreads(); out[1] = lds[0]; }
  uint4* p = dst + (size_t)blockIdx.x * (4 * 160 * 1024 / 16) + threadIdx.x;
  const uint4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
#pragma unroll
  for (int i = 0; i < 80; ++i) p[i * 512] = v;
}

template <class F>
static void run(const char* name, F launch, hipStream_t st) {
  const int N = 400;
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int i = 0; i < 20; ++i) launch(st);
  CHECK(hipStreamSynchronize(st));
  CHECK(hipEventRecord(a, st));
  for (int i = 0; i < N; ++i) launch(st);
  CHECK(hipEventRecord(b, st));
  CHECK(hipEventSynchronize(b));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
  const float eager = ms * 1e3f / N;
  hipGraph_t g; hipGraphExec_t ge;
  CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < N; ++i) launch(st);
  CHECK(hipStreamEndCapture(st, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CHECK(hipGraphLaunch(ge, st)); CHECK(hipStreamSynchronize(st));
  CHECK(hipEventRecord(a, st));
  for (int r = 0; r < 5; ++r) CHECK(hipGraphLaunch(ge, st));
  CHECK(hipEventRecord(b, st));
  CHECK(hipEventSynchronize(b));
  CHECK(hipEventElapsedTime(&ms, a, b));
  printf("{\"kernel\": \"%s\", \"eager_us_per_launch\": %.2f, \"graph_us_per_launch\": %.2f}\n", name, eager, ms * 1e3f / (5 * N));
  CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
}

int main() {
  unsigned* out; uint4* dst;
  CHECK(hipMalloc(&out, 4096)); CHECK(hipMemset(out, 0, 4096));
  CHECK(hipMalloc(&dst, (size_t)256 * 160 * 1024));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  const int LDS = 160 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_arg), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_store), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  Big big; for (int i = 0; i < 64; ++i) big.w[i] = i;
  run("a: 1 workgroup x 64 threads", [&](hipStream_t s) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, out); }, st);
  run("b: 256 workgroups x 512 threads", [&](hipStream_t s) { hipLaunchKernelGGL(k_wide, dim3(256), dim3(512), 0, s, out); }, st);
  run("c: 256 x 512 threads, 160 KB LDS", [&](hipStream_t s) { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), LDS, s, out); }, st);
  run("d: c + 256-byte argument", [&](hipStream_t s) { hipLaunchKernelGGL(k_arg, dim3(256), dim3(512), LDS, s, out, big); }, st);
  run("e: c + 160 KB of stores per workgroup (42 MB per launch)", [&](hipStream_t s) { hipLaunchKernelGGL(k_store, dim3(256), dim3(512), LDS, s, out, dst); }, st);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_store_nt), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_store4), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  run("e-nt: e with non-temporal stores", [&](hipStream_t s) { hipLaunchKernelGGL(k_store_nt, dim3(256), dim3(512), LDS, s, out, dst); }, st);
  run("e/4: the same 42 MB from 64 workgroups", [&](hipStream_t s) { hipLaunchKernelGGL(k_store4, dim3(64), dim3(512), LDS, s, out, dst); }, st);
  run("f: 2048 workgroups x 512 threads, 160 KB LDS (8 rounds)", [&](hipStream_t s) { hipLaunchKernelGGL(k_lds, dim3(2048), dim3(512), LDS, s, out); }, st);
  return 0;
}
