// Flux (MMDiT) per-head q/k RMSNorm + rotary embedding + V transpose, gfx950 -- the ops between the fused qkv Linear and the
// joint txt||img attention of a DoubleStream / SingleStream block (backend/nn/flux.py:15-18 attention(), :43-49 apply_rope,
// :115-139 RMSNorm/QKNorm, :222-247 / :291-297).  The reference runs them as ~12 elementwise / reduction kernels with fp32
// upcasts and a [B,H,L,D] permute; here one HBM-bound pass reads the qkv GEMM output once and writes
//   q, k : [B][L_pad][H*D] fp16 (the layout fmx_attention_f16 reads), rows row_off.. of the joint sequence
//   v^T  : [H*D][B*L_pad]  fp16 (the attention kernel's V operand), through a 64-token LDS transpose.
// D = 128 (every Flux variant): one wave per (token, head), one rotary PAIR per lane.
#include "fmx_common.hpp"

namespace {

constexpr int D = 128;

struct RopeParams {
  const f16* qkv;     // [B*L][ld_qkv]: q at col 0, k at col H*D, v at col 2*H*D (each [H][D])
  long ld_qkv;
  const f16* q_scale; // [D]
  const f16* k_scale; // [D]
  const float* pe;    // [L_total][D/2][2] = (cos, sin) of the joint sequence
  f16* q_out;
  f16* k_out;         // [B][l_pad][H*D]
  f16* vt_out;        // [H*D][B*l_pad]
  int B, L, H, row_off, l_pad;
  float eps;
};

__global__ __launch_bounds__(256) void qk_norm_rope_kernel(const RopeParams p) {
  __shared__ f16 vtile[D][64 + 8];  // [d][token], +8 halfs padding (16 B) against bank conflicts on the transposed writes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int HD = p.H * D;
  const float qs0 = (float)p.q_scale[2 * lane], qs1 = (float)p.q_scale[2 * lane + 1];
  const float ks0 = (float)p.k_scale[2 * lane], ks1 = (float)p.k_scale[2 * lane + 1];
  for (int i = wave; i < 64; i += 4) {
    const int t = t0 + i;
    f16x2 vv = f16x2{(f16)0.f, (f16)0.f};
    if (t < p.L) {
      const f16* row = p.qkv + ((long)b * p.L + t) * p.ld_qkv + h * D + 2 * lane;
      const f16x2 q = *reinterpret_cast<const f16x2*>(row);
      const f16x2 k = *reinterpret_cast<const f16x2*>(row + HD);
      vv = *reinterpret_cast<const f16x2*>(row + 2 * HD);
      const float q0 = (float)q[0], q1 = (float)q[1], k0 = (float)k[0], k1 = (float)k[1];
      const float qr = rsqrtf(wave_sum(q0 * q0 + q1 * q1) * (1.0f / D) + p.eps);
      const float kr = rsqrtf(wave_sum(k0 * k0 + k1 * k1) * (1.0f / D) + p.eps);
      // fp32 from the RMSNorm through the rotary, one rounding at the store (the parity oracle is the fp32 CPU path)
      const float qa = q0 * qr * qs0, qb = q1 * qr * qs1;
      const float ka = k0 * kr * ks0, kb = k1 * kr * ks1;
      const f32x2 cs = *reinterpret_cast<const f32x2*>(p.pe + ((long)(p.row_off + t) * (D / 2) + lane) * 2);
      const long o = ((long)b * p.l_pad + p.row_off + t) * HD + h * D + 2 * lane;
      *reinterpret_cast<f16x2*>(p.q_out + o) = f16x2{(f16)(cs[0] * qa - cs[1] * qb), (f16)(cs[1] * qa + cs[0] * qb)};
      *reinterpret_cast<f16x2*>(p.k_out + o) = f16x2{(f16)(cs[0] * ka - cs[1] * kb), (f16)(cs[1] * ka + cs[0] * kb)};
    }
    vtile[2 * lane][i] = vv[0];
    vtile[2 * lane + 1][i] = vv[1];
  }
  __syncthreads();
  // v^T rows: thread -> (d = tid / 2, 32-token half); 64 B per thread, 128 B contiguous per row
  const int d = tid >> 1, half = tid & 1;
  const long col0 = (long)b * p.l_pad + p.row_off + t0 + half * 32;
  f16* dst = p.vt_out + ((long)h * D + d) * ((long)p.B * p.l_pad) + col0;
  const int valid = min(32, p.L - (t0 + half * 32));
  if (valid == 32 && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f16x8*>(dst + j * 8) = *reinterpret_cast<const f16x8*>(&vtile[d][half * 32 + j * 8]);
  } else {
    for (int j = 0; j < valid; ++j) dst[j] = vtile[d][half * 32 + j];
  }
}

}  // namespace

extern "C" int fmx_flux_qk_norm_rope_f16(const void* qkv, int64_t ld_qkv, const void* q_scale, const void* k_scale, const float* pe,
                                         void* q_out, void* k_out, void* vt_out, int32_t batch, int32_t tokens, int32_t heads,
                                         int32_t head_dim, int32_t row_off, int32_t l_pad, float eps, void* stream) {
  FMX_REQUIRE(qkv && q_scale && k_scale && pe && q_out && k_out && vt_out, "flux_qk_norm_rope: null pointer");
  FMX_REQUIRE(head_dim == D, "flux_qk_norm_rope: head_dim %d != 128", head_dim);
  FMX_REQUIRE(batch > 0 && tokens > 0 && heads > 0 && row_off >= 0 && l_pad >= row_off + tokens && (ld_qkv % 2) == 0, "flux_qk_norm_rope: bad dims");
  RopeParams p;
  p.qkv = (const f16*)qkv; p.ld_qkv = ld_qkv; p.q_scale = (const f16*)q_scale; p.k_scale = (const f16*)k_scale; p.pe = pe;
  p.q_out = (f16*)q_out; p.k_out = (f16*)k_out; p.vt_out = (f16*)vt_out;
  p.B = batch; p.L = tokens; p.H = heads; p.row_off = row_off; p.l_pad = l_pad; p.eps = eps;
  hipLaunchKernelGGL(qk_norm_rope_kernel, dim3((tokens + 63) / 64, heads, batch), dim3(256), 0, (hipStream_t)stream, p);
  FMX_LAUNCH_CHECK("fmx_flux_qk_norm_rope_f16");
  return FMX_OK;
}

#ifdef FMX_ELEM_BF16
// The two small elementwise ops on the Flux `vec` path (timestep / guidance embedding -> MLPEmbedder, SiLU; backend/nn/flux.py:52-73,
// 142-153).  Their fp16 forms live in fmx_elementwise.hip next to the UNet's other elementwise kernels; the bfloat16 build needs only
// these two, so they are here rather than in a second build of that whole file.
namespace {

__global__ void timestep_embedding_bf16_kernel(const float* __restrict__ t, f16* __restrict__ emb, int b, int dim, float log_period) {
  const int half = dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b * half) return;
  const int bi = i / half, k = i - bi * half;
  const float a = t[bi] * expf(-log_period * (float)k / (float)half);
  emb[(long)bi * dim + k] = (f16)cosf(a);
  emb[(long)bi * dim + half + k] = (f16)sinf(a);
}

__global__ void silu_bf16_kernel(const f16* __restrict__ x, f16* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = (f16)silu_f((float)x[i]);
}

}  // namespace

extern "C" int fmx_timestep_embedding_bf16(const float* t, void* emb, int32_t b, int32_t dim, float max_period, void* stream) {
  FMX_REQUIRE(t && emb && b > 0 && dim > 0 && (dim % 2) == 0 && max_period > 1.0f, "timestep_embedding_bf16: bad args");
  const int total = b * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, (f16*)emb, b, dim,
                     logf(max_period));
  FMX_LAUNCH_CHECK("fmx_timestep_embedding_bf16");
  return FMX_OK;
}

extern "C" int fmx_silu_bf16(const void* x, void* y, int64_t n, void* stream) {
  FMX_REQUIRE(x && y && n > 0, "silu_bf16: bad args");
  const long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(silu_bf16_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, (const f16*)x,
                     (f16*)y, (long)n);
  FMX_LAUNCH_CHECK("fmx_silu_bf16");
  return FMX_OK;
}
#endif
