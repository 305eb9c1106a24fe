// Probe: does buffer_load_dwordx4 ... lds write ZEROS into LDS for out-of-range lanes (raw buffer, stride 0)?
// Cases: (a) voffset >= num_records, (b) voffset in range but voffset + soffset >= num_records, (c) in range.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void probe(const unsigned* src, unsigned nbytes, unsigned voff_bad, unsigned soff, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 3];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 4 * 3; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  // case a: odd lanes out of range through voffset
  unsigned va = (lane & 1) ? voff_bad : lane * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds), 16, va, 0, 0, 0);
  // case b: all lanes in range by voffset, soffset pushes lanes >= 32 out (voffset + soffset >= nbytes)
  unsigned vb = lane * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 256), 16, vb, soff, 0, 0);
  // case c: everything in range with a soffset
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 512), 16, vb, 1024, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 64 * 4 * 3; i += 64) out[i] = lds[i];
}

int main() {
  const unsigned nbytes = 4096;
  unsigned* src; unsigned* out;
  CHECK(hipMalloc(&src, 65536)); CHECK(hipMalloc(&out, 64 * 4 * 3 * 4));
  unsigned h[16384];
  for (int i = 0; i < 16384; ++i) h[i] = 0x1000000u + i;
  CHECK(hipMemcpy(src, h, 65536, hipMemcpyHostToDevice));
  // soff = nbytes - 512: lanes with lane*16 + soff >= nbytes, i.e. lane >= 32, are out of range if soffset counts
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, nbytes, 0xC0000000u, nbytes - 512, out);
  CHECK(hipDeviceSynchronize());
  unsigned r[64 * 4 * 3];
  CHECK(hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost));
  int a_ok = 1, a_zero = 1;
  for (int l = 0; l < 64; ++l)
    for (int d = 0; d < 4; ++d) {
      unsigned v = r[l * 4 + d];
      if (l & 1) { if (v != 0) a_zero = 0; }
      else if (v != 0x1000000u + l * 4 + d) a_ok = 0;
    }
  printf("case a (voffset OOB): in-range lanes correct=%d, OOB lanes zero=%d (sample OOB lane1 = %08x)\n", a_ok, a_zero, r[4]);
  int b_lo = 1, b_hi_zero = 1, b_hi_data = 1;
  for (int l = 0; l < 64; ++l)
    for (int d = 0; d < 4; ++d) {
      unsigned v = r[256 + l * 4 + d], want = 0x1000000u + (nbytes - 512) / 4 + l * 4 + d;
      if (l < 32) { if (v != want) b_lo = 0; }
      else { if (v != 0) b_hi_zero = 0; if (v != want) b_hi_data = 0; }
    }
  printf("case b (soffset pushes OOB): low lanes correct=%d, high lanes zero=%d, high lanes read-through=%d (sample lane40 = %08x)\n", b_lo, b_hi_zero, b_hi_data, r[256 + 160]);
  int c_ok = 1;
  for (int l = 0; l < 64; ++l)
    for (int d = 0; d < 4; ++d)
      if (r[512 + l * 4 + d] != 0x1000000u + 256 + l * 4 + d) c_ok = 0;
  printf("case c (in range, soffset 1024): correct=%d\n", c_ok);
  return 0;
}
