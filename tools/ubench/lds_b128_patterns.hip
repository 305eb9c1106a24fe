// Which per-lane address patterns does a ds_read_b128 serve without bank conflicts on gfx950?  One wave issues 8 192 reads of one pattern; cycles per read
// from s_memtime.  Patterns: the fragment reads of the GEMM kernels' LDS images (128-byte rows of fmx_gemm256p.hip; 64-byte rows of fmx_gemm4w.hip with
// several candidate XOR keys) -- profiles/r10_pmc_gemm4w: the first 64-byte layout ran with SQ_LDS_BANK_CONFLICT = 0.46 of SQ_LDS_IDX_ACTIVE.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_b128_patterns.hip -o tools/ubench/lds_b128_patterns && tools/ubench/lds_b128_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <string>

__global__ void k(const unsigned* offs, unsigned long long* cyc, float* sink) {
  extern __shared__ char smem[];
  for (int i = threadIdx.x; i < 16384; i += 64) reinterpret_cast<float*>(smem)[i] = (float)i;
  __syncthreads();
  const unsigned a = offs[threadIdx.x];
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 v0, v1, v2, v3, acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 2048; ++it) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16384\n\tds_read_b128 %2, %4 offset:32768\n\tds_read_b128 %3, %4 offset:49152\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(a) : "memory");
    acc += v0 + v1 + v2 + v3;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  std::vector<std::pair<std::string, std::vector<unsigned>>> pats;
  auto add = [&](const char* name, auto f) {
    std::vector<unsigned> o(64);
    for (int l = 0; l < 64; ++l) o[l] = f(l);
    pats.push_back({name, o});
  };
  add("contiguous lane*16", [](int l) { return (unsigned)l * 16; });
  add("all lanes one address", [](int) { return 0u; });
  add("stride 256 B (worst)", [](int l) { return (unsigned)(l & 15) * 256 + (l >> 4) * 16; });
  // 128-byte rows (fmx_gemm256p.hip, 16x16x32 loop): lane (l16, kg) reads row l16, chunk kg ^ ((row >> 1) & 7)
  add("128B rows, key (row>>1)&7 [gemm256p]", [](int l) { int r = l & 15, kg = l >> 4; return (unsigned)(r * 128 + ((kg ^ ((r >> 1) & 7)) << 4)); });
  add("128B rows, no swizzle", [](int l) { int r = l & 15, kg = l >> 4; return (unsigned)(r * 128 + (kg << 4)); });
  // 64-byte rows: row l16, chunk kg ^ key(row)
  add("64B rows, key (row>>2)&3 [gemm4w v1]", [](int l) { int r = l & 15, kg = l >> 4; return (unsigned)(r * 64 + ((kg ^ ((r >> 2) & 3)) << 4)); });
  add("64B rows, key (row>>1)&3", [](int l) { int r = l & 15, kg = l >> 4; return (unsigned)(r * 64 + ((kg ^ ((r >> 1) & 3)) << 4)); });
  add("64B rows, key row&3", [](int l) { int r = l & 15, kg = l >> 4; return (unsigned)(r * 64 + ((kg ^ (r & 3)) << 4)); });
  add("64B rows, key (row>>3)&1 *2 ^ ((row>>2)&1)", [](int l) { int r = l & 15, kg = l >> 4; return (unsigned)(r * 64 + ((kg ^ (((r >> 3) & 1) * 2) ^ ((r >> 2) & 1)) << 4)); });
  add("64B rows, no swizzle", [](int l) { int r = l & 15, kg = l >> 4; return (unsigned)(r * 64 + (kg << 4)); });
  // two matrix rows per 128-byte LDS row (rows r and r + 8 side by side), the 128-byte key
  add("paired rows (r, r+8) in 128B, key (rho>>1)&7", [](int l) { int r = l & 15, kg = l >> 4; int rho = r & 7, half = r >> 3; return (unsigned)(rho * 128 + (((half * 4 + kg) ^ ((rho >> 1) & 7)) << 4)); });
  add("paired rows (r, r+1) in 128B, key (rho>>1)&7", [](int l) { int r = l & 15, kg = l >> 4; int rho = r >> 1, half = r & 1; return (unsigned)(rho * 128 + (((half * 4 + kg) ^ ((rho >> 1) & 7)) << 4)); });
  add("paired rows (r, r+1) in 128B, key rho&7", [](int l) { int r = l & 15, kg = l >> 4; int rho = r >> 1, half = r & 1; return (unsigned)(rho * 128 + (((half * 4 + kg) ^ (rho & 7)) << 4)); });
  add("paired rows (r, r+8) in 128B, key rho&7", [](int l) { int r = l & 15, kg = l >> 4; int rho = r & 7, half = r >> 3; return (unsigned)(rho * 128 + (((half * 4 + kg) ^ (rho & 7)) << 4)); });
  // MFMA 32x32x16 operand read of the 128-byte layout (lane (li, hi): row li, chunk 2 ks + hi) for reference
  add("128B rows, 32-row fragment, key (row>>1)&7", [](int l) { int r = l & 31, hi = l >> 5; return (unsigned)(r * 128 + ((hi ^ ((r >> 1) & 7)) << 4)); });
  unsigned* d_offs; unsigned long long* d_cyc; float* d_sink;
  hipMalloc(&d_offs, 256); hipMalloc(&d_cyc, 8); hipMalloc(&d_sink, 256);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (auto& p : pats) {
    hipMemcpy(d_offs, p.second.data(), 256, hipMemcpyHostToDevice);
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 65536, 0, d_offs, d_cyc, d_sink);
      unsigned long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
      if (c < best) best = c;
    }
    printf("{\"pattern\": \"%s\", \"memtime_ticks_per_ds_read_b128\": %.3f}\n", p.first.c_str(), (double)best / 8192.0);
  }
  return 0;
}
