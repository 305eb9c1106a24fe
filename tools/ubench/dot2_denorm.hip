// Does v_dot2c_f32_f16 (and the f16 MFMA) keep fp16 DENORMAL inputs?  The attention kernels sum the softmax's P (fp16, as fed to the P.V MFMA) with
// v_dot2: if the dot product flushed denormal P (below 2^-14) while the MFMA kept them, numerator and denominator of a spiked row would disagree.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dot2_denorm.hip -o tools/ubench/dot2_denorm && tools/ubench/dot2_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* o) {
  const _Float16 d = (_Float16)3.0e-6f;   // fp16 denormal (2^-24 * 50)
  h2 a = {d, d}, ones = {(_Float16)1.0f, (_Float16)1.0f};
  asm volatile("" : "+v"(a));
  o[0] = __builtin_amdgcn_fdot2(a, ones, 0.0f, false);
  o[1] = (float)a[0] + (float)a[1];
  h8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = d; y[e] = (_Float16)1.0f; }
  asm volatile("" : "+v"(x));
  f16v z;
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  z = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, z, 0, 0, 0);
  if (threadIdx.x == 0) o[2] = z[0];     // 16 products of denormal * 1
}
int main() {
  float* d; float h[3];
  hipMalloc(&d, 12);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
  printf("{\"dot2_of_two_fp16_denormals\": %g, \"their_fp32_sum\": %g, \"mfma_16_denormal_products\": %g, \"expected_mfma\": %g}\n", h[0], h[1], h[2], 8.0 * h[1]);
  return 0;
}
